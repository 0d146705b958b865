mkdir -p gpurun_out
cd /root/repo
timeout 600 python -m pytest tests/test_dpccn_gpu.py -m gpu -x -q -k "implicit or fixture or kernels" 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/t2.log
timeout 600 python -m pytest tests/test_convtasnet_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/t3.log
echo "== dpccn joint rows 32"; timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 3 2>&1 | grep '^{' | tee gpurun_out/r02_dpccn_bench.json
