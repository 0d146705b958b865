#!/bin/bash
# per-kernel time of the headline step (rocprofv3 kernel trace) -> gpurun_out/r03_kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_r03
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/r03_prof_bench.json 2> $ROOT/gpurun_out/r03_prof.err
echo "rocprof exit $?"
f=$(find /tmp/prof_r03 -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/r03_bench_r32_kernel_stats.csv
head -40 $ROOT/gpurun_out/r03_bench_r32_kernel_stats.csv | cut -c1-200
cut -c1-200 $ROOT/gpurun_out/r03_prof_bench.json
