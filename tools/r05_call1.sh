#!/bin/bash
# Round 5, call 1: the pair BPTT's fp16 recurrence (ABI v17, rfmt = 1), un-clamped scaled-fp16 d(gates) with 2^7 headroom --
# kernel tests, isolated timing + cycle stamps, bench A/B on one box, the fast parity subset; then the SSA step's kernel stats.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x > $O/r05_c1_h2_kernels.log 2>&1
echo "== h2 kernel tests exit $?"; tail -3 $O/r05_c1_h2_kernels.log | cut -c1-300
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -x -k "pair or lstm" > $O/r05_c1_kernels.log 2>&1
echo "== lstm kernel tests exit $?"; tail -3 $O/r05_c1_kernels.log | cut -c1-300
timeout 200 python tools/r05_recur_probe.py > $O/r05_c1_recur_probe.txt 2>&1
echo "== probe exit $?"; cat $O/r05_c1_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c1_bench_$name.json 2> $O/r05_c1_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c1_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline'].get('kernel_ms_per_step'))" 2>&1)"; tail -1 $O/r05_c1_bench_$name.err | cut -c1-200
}
run rf1 WESEP_PAIR_RF=1
run rf0 WESEP_PAIR_RF=0
run rf1_b WESEP_PAIR_RF=1
timeout 400 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "resrnn_block or fixture or full_size_row or training_step_matches" > $O/r05_c1_bsrnn.log 2>&1
echo "== bsrnn parity subset exit $?"; grep -E "est rel|trajectory|passed|failed|Error|assert " $O/r05_c1_bsrnn.log | cut -c1-260
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ssa -- python $OLDPWD/tools/bench_ssa.py --what ssa --steps 3 --warmup 2 > $OLDPWD/$O/r05_ssa_under_rocprof.jsonl 2> /tmp/prof_ssa.err
cd $OLDPWD
cp "$(find /tmp/prof_ssa -name '*kernel_stats.csv' | head -1)" $O/r05_ssa_kernel_stats.csv 2>/dev/null
cp "$(find /tmp/prof_ssa -name '*kernel_trace.csv' | head -1)" /tmp/ssa_trace.csv 2>/dev/null
python tools/trace_gaps.py /tmp/ssa_trace.csv --steps 3 > $O/r05_ssa_trace_gaps.txt 2>&1
head -25 $O/r05_ssa_kernel_stats.csv | cut -c1-160
grep "^{" $O/r05_ssa_under_rocprof.jsonl | cut -c1-300
