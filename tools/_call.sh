mkdir -p gpurun_out
timeout 200 python tools/kernel_race3.py 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kernel_race3b.log
