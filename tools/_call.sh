mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd /root/repo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_prof.log 2>&1
find gpurun_out/r02_prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_r32_kernel_stats.csv
find gpurun_out/r02_prof -name '*.csv' ! -name '*stats*' -size +4M -delete
