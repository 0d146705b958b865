#!/bin/bash
# Round 5, call 17 (= call 16 with the register-resident fragments converted once, in front of the step loop): rfmt 2, where to touch the next step's cache lines: not at all / all twelve in front of the MFMA loop /
# three per chunk over the loop's first half (dbg bits 16 / 32 via WESEP_PAIR_TOUCH), one box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -s -k "bptt and pair" > $O/r05_c17_pair.log 2>&1
echo "== pair tests exit $?"; grep -E "rfmt|passed|failed|Error|assert " $O/r05_c17_pair.log | cut -c1-300 | tail -8
timeout 200 python tools/r05_recur_probe.py > $O/r05_c17_recur_probe.txt 2>&1
echo "== probe exit $?"; grep -E "^pair BPTT|status" $O/r05_c17_recur_probe.txt; tail -48 $O/r05_c17_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c17_bench_$name.json 2> $O/r05_c17_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c17_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r05_c17_bench_$name.err | cut -c1-200
}
run t0 WESEP_PAIR_RF=2 WESEP_PAIR_TOUCH=0
run t16 WESEP_PAIR_RF=2 WESEP_PAIR_TOUCH=16
run t32 WESEP_PAIR_RF=2 WESEP_PAIR_TOUCH=32
run t0_b WESEP_PAIR_RF=2 WESEP_PAIR_TOUCH=0
run rf1 WESEP_PAIR_RF=1
