#!/bin/bash
# Round 5, call 13: the pair BPTT with W_hh's lo plane as block-scaled FP8 (rfmt 2: all of W_hh resident on the CU) -- pack and
# kernel tests, launch times and stamps alone, bench A/B against rfmt 1 on one box, the parity tests that decide
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -s -k "fp8 or bptt or pair" > $O/r05_c13_pair.log 2>&1
echo "== pair tests exit $?"; grep -E "rfmt|fp8 lo|passed|failed|Error|assert " $O/r05_c13_pair.log | cut -c1-300 | tail -16
timeout 200 python tools/r05_recur_probe.py > $O/r05_c13_recur_probe.txt 2>&1
echo "== probe exit $?"; grep -A40 "pair BPTT, fp32" $O/r05_c13_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c13_bench_$name.json 2> $O/r05_c13_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c13_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r05_c13_bench_$name.err | cut -c1-200
}
run rf1 WESEP_PAIR_RF=1
run rf2 WESEP_PAIR_RF=2
run rf1_b WESEP_PAIR_RF=1
run rf2_b WESEP_PAIR_RF=2
WESEP_PAIR_RF=2 timeout 400 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "trajectory or full_size_row or training_step" > $O/r05_c13_parity_rf2.log 2>&1
echo "== rf2 parity subset exit $?"; grep -E "trajectory|full-size|passed|failed|worst" $O/r05_c13_parity_rf2.log | cut -c1-300
