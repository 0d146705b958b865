mkdir -p gpurun_out
cd /root/repo
timeout 900 python -X faulthandler -m pytest tests/test_cluster_robustness_gpu.py tests/test_convtasnet_gpu.py tests/test_dpccn_gpu.py tests/test_ecapa_gpu.py tests/test_engine_gpu.py tests/test_fbank_gpu.py tests/test_kernels_gpu.py tests/test_resnet_gpu.py tests/test_tfgridnet_blocked_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | tee gpurun_out/t1.log
