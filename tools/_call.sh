cd /root/repo
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "b2p" 2>&1 | grep -v amdgpu.ids | tail -2
PROBE_R=32 timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | grep "b2p"
