#!/bin/bash
# round 6, call 32: the timeline of one step (which weight-gradient GEMMs sit on the main queue while the side queue is empty)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_c32
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c32 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/r06_c32_bench_rocprof.json 2> /tmp/prof_c32.err
T="$(find /tmp/prof_c32 -name '*kernel_trace.csv' | head -1)"
python $R/tools/r06_step_timeline.py "$T" --min-us 120 > $O/r06_c32_step_timeline.txt 2>&1; python $R/tools/r06_step_timeline.py "$T" --min-us 0 > $O/r06_c32_step_timeline_all.txt 2>&1
python $R/tools/trace_gaps.py "$T" --steps 2 > $O/r06_c32_trace_gaps.txt 2>&1
head -5 $O/r06_c32_step_timeline.txt; wc -l $O/r06_c32_step_timeline.txt
