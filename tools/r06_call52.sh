#!/bin/bash
# round 6, call 52: why do some bench.py runs take 344-850 ms per step with every timed kernel class at its usual duration?
# per-step wall-clock + allocator counters (headline and joint, WESEP_GEMM_NN=1/0), then a kernel trace of the slow joint run
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
WESEP_GEMM_NN=1 timeout 200 python tools/r06_diag_steps.py --tag head_nn1 --steps 6
WESEP_GEMM_NN=1 timeout 300 python tools/r06_diag_steps.py --tag joint_nn1 --joint --steps 5
WESEP_GEMM_NN=0 timeout 300 python tools/r06_diag_steps.py --tag joint_nn0 --joint --steps 5
WESEP_GEMM_NN=1 timeout 300 python tools/r06_diag_steps.py --tag joint_nn1_sync --joint --steps 5 --sync-each
WESEP_GEMM_NN=1 timeout 200 python tools/r06_diag_steps.py --tag head_nn1_again --steps 6
} > $O/r06_c52_diag.txt 2>&1
grep -E "steps:|step [0-9]:" $O/r06_c52_diag.txt | cut -c1-230
cd /tmp; rm -rf /tmp/prof_j
WESEP_GEMM_NN=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j -- python $R/bench.py --joint --steps 2 --warmup 2 --no-cpu-baseline > $O/r06_c52_joint_under_rocprof.json 2> /tmp/prof_j.err
T="$(find /tmp/prof_j -name '*kernel_trace.csv' | head -1)"
cp "$(find /tmp/prof_j -name '*kernel_stats.csv' | head -1)" $O/r06_c52_joint_kernel_stats.csv
python $R/tools/trace_gaps.py "$T" --steps 1 --top 25 > $O/r06_c52_joint_trace_gaps.txt 2>&1
python $R/tools/r06_step_timeline.py "$T" --min-us 400 > $O/r06_c52_joint_timeline.txt 2>&1
head -30 $O/r06_c52_joint_trace_gaps.txt
head -12 $O/r06_c52_joint_kernel_stats.csv | cut -c1-160
