#!/bin/bash
# round 6, call 25: per-kernel times of the step with all three FP8 lo-term variants on, cluster2's off / on
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c8 in 0 1; do
  rm -rf /tmp/prof_c8$c8
  WESEP_FUSED_F8=1 WESEP_PAIR_RF=3 WESEP_CLUSTER2_F8=$c8 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c8$c8 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r06_c25_bench_c8$c8.json 2> /tmp/prof_c8$c8.err
  cp "$(find /tmp/prof_c8$c8 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r06_c25_kernel_stats_c8$c8.csv
  head -12 $R/gpurun_out/r06_c25_kernel_stats_c8$c8.csv | cut -c1-150
done
