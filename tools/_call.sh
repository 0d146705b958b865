mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_ecapa_gpu.py tests/test_resnet_gpu.py -m gpu -x -q -k "ecapa or conv_bn or tstp or joint_training" 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/t1.log
