mkdir -p gpurun_out
cd /root/repo
timeout 600 python -m pytest tests/test_dpccn_gpu.py -m gpu -x -q -k "implicit or kernels or fixture" 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/t1.log
timeout 600 python -m pytest tests/test_resnet_gpu.py tests/test_tfgridnet_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/t2.log
echo "== dpccn joint rows 32"; timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 2 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bsrnn joint"; timeout 300 python bench.py --joint --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], j['value'])
"
