#!/bin/bash
# round 6, call 37: one TF-GridNet step as a timeline (where the 18 ms of ATen copies / adds per step sit)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_c37
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c37 -- python $R/tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --warmup 1 > $O/r06_c37_tfg_rocprof.json 2> /tmp/prof_c37.err
T="$(find /tmp/prof_c37 -name '*kernel_trace.csv' | head -1)"
python $R/tools/r06_step_timeline.py "$T" --min-us 60 > $O/r06_c37_tfg_step_timeline.txt 2>&1
head -2 $O/r06_c37_tfg_step_timeline.txt; wc -l $O/r06_c37_tfg_step_timeline.txt
