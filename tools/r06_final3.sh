#!/bin/bash
# Round 6, second profile set (after the FP8 lo terms, the occupancy fixes of ws_gemm_b2p / gemm_nt_bf16 and the 16-byte FiLM backward):
# all four models' lines with counters, the recipe variants, the whole -m gpu suite, smoke(), and the R = 32 oracle comparison
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_final.sh
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/r06_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -16 $O/r06_full_gpu_suite.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1
echo "== smoke exit $?"; tail -2 $O/r06_smoke.log | cut -c1-300
WESEP_RUN_SLOW=1 timeout 1500 python -m pytest tests/test_bsrnn_gpu.py -m gpu -q -s -k "headline_batch_r32" > $O/r06_headline_r32_vs_oracle.log 2>&1
echo "== r32 oracle exit $?"; grep -i "r32\|rel\|passed\|failed" $O/r06_headline_r32_vs_oracle.log | tail -8 | cut -c1-250
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_tl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_tl.err
T="$(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1)"
python $R/tools/trace_gaps.py "$T" --steps 2 > $R/$O/r06_bsrnn_trace_gaps.txt 2>&1
python $R/tools/r06_step_timeline.py "$T" --min-us 100 > $R/$O/r06_bsrnn_step_timeline.txt 2>&1
head -3 $R/$O/r06_bsrnn_trace_gaps.txt
