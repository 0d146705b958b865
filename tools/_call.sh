mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "narrow" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/t1.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dpccn -- python tools/bench_dpccn.py --rows 32 --joint --steps 2 > gpurun_out/prof_dpccn.log 2>&1
find gpurun_out/prof_dpccn -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r02_dpccn_kernel_stats.csv
find gpurun_out/prof_dpccn -name '*.csv' ! -name '*stats*' -size +4M -delete
