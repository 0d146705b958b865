mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_dpccn_gpu.py tests/test_tfgridnet_gpu.py -m gpu -x -q -k "fixture or unbuilt" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-260 | tee gpurun_out/t1.log
