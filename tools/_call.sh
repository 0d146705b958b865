mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_convtasnet_gpu.py tests/test_dpccn_gpu.py tests/test_ecapa_gpu.py tests/test_resnet_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200 | tee gpurun_out/t1.log
timeout 900 python -m pytest tests/test_bsrnn_gpu.py tests/test_tfgridnet_gpu.py -m gpu -x -q -k "fixture or resrnn_block or training_step" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200 | tee gpurun_out/t2.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['ms_per_step'], j['value'], j['kernel_ms_per_step'])
"
timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 3 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['ms_per_step'], j['value'], j['roofline']['kernel_ms_per_step'])
"
