mkdir -p gpurun_out
cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/t1.log
timeout 900 python -m pytest tests/test_bsrnn_gpu.py tests/test_tfgridnet_blocked_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/t2.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_bench1.log
