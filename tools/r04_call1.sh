#!/bin/bash
# Round 4, GPU call 1: the 2-byte storage formats of the saved recurrence state (WS_GATES_H2 / H2S, ABI v15) on hardware:
#   1. the new kernel tests + the legacy-format kernel tests of the files that were touched
#   2. the assembled-model parity tests (default format H2) with their measured errors, then fixtures + trajectory with H2S
#   3. bench.py A/B on the same box: f32 (round 3's format), h2s, h2 (default)
#   4. rocprofv3 kernel stats of the default
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gates_h2_gpu.py -q --tb=short -x > $O/r04_c1_h2_kernels.log 2>&1
echo "== h2 kernel tests exit $?"; tail -15 $O/r04_c1_h2_kernels.log | cut -c1-220
timeout 900 python -m pytest tests/test_bptt_survival_gpu.py -q -s --tb=short > $O/r04_c1_survival.log 2>&1
echo "== survival tests exit $?"; grep -E "backward at|passed|failed|Error|assert" $O/r04_c1_survival.log | cut -c1-250 | tail -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "lstm or b2p or tnb or p2b or clip_adam" > $O/r04_c1_legacy_kernels.log 2>&1
echo "== legacy kernel tests exit $?"; tail -5 $O/r04_c1_legacy_kernels.log | cut -c1-220
timeout 1200 python -m pytest tests/test_bsrnn_gpu.py -q -s --tb=short > $O/r04_c1_bsrnn_h2.log 2>&1
echo "== bsrnn (h2) exit $?"; grep -E "rel|trajectory|passed|failed|Error|assert" $O/r04_c1_bsrnn_h2.log | cut -c1-300 | tail -30
WESEP_GATES=h2s timeout 900 python -m pytest tests/test_bsrnn_gpu.py -q -s --tb=short -k "fixture or trajectory or resrnn_block" > $O/r04_c1_bsrnn_h2s.log 2>&1
echo "== bsrnn (h2s) exit $?"; grep -E "rel|trajectory|passed|failed|Error|assert" $O/r04_c1_bsrnn_h2s.log | cut -c1-300 | tail -14
for f in f32 h2s h2; do
  WESEP_GATES=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c1_bench_$f.json 2> $O/r04_c1_bench_$f.err
  echo "== bench $f exit $?"; cut -c1-420 $O/r04_c1_bench_$f.json
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_r04
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_c1_prof_bench.json 2> $O/r04_c1_prof.err
echo "rocprof exit $?"
cp "$(find /tmp/prof_r04 -name '*kernel_stats.csv' | head -1)" $O/r04_c1_kernel_stats.csv
t=$(find /tmp/prof_r04 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_gaps.py "$t" --steps 4 > $O/r04_c1_trace_gaps.txt 2>&1
head -30 $O/r04_c1_kernel_stats.csv | cut -c1-170
tail -12 $O/r04_c1_trace_gaps.txt
