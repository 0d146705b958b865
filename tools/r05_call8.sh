#!/bin/bash
# Round 5, call 8: cluster2 with the full split-bf16 x-projection (BLS input, three terms) -- the 60-step trajectory bound.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -s > $O/r05_c8_cluster2.log 2>&1
echo "== cluster2 tests exit $?"; grep -E "cluster2|passed|failed|Error|assert " $O/r05_c8_cluster2.log | cut -c1-300 | tail -12
timeout 200 python tools/r05_recur_probe.py > $O/r05_c8_recur_probe.txt 2>&1
echo "== probe exit $?"; grep -E "^cluster|^pair|status|stamps|done|arrived|written|barrier|loop top" $O/r05_c8_recur_probe.txt | cut -c1-200 | head -40
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c8_bench_$name.json 2> $O/r05_c8_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c8_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r05_c8_bench_$name.err | cut -c1-200
}
run head A=1
run head_b A=1
timeout 900 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "trajectory or full_size_row or training_step_matches or config2 or fixture" > $O/r05_c8_bsrnn.log 2>&1
echo "== bsrnn parity subset exit $?"; grep -E "est rel|full-size|trajectory\[|config 2|passed|failed|Error|assert " $O/r05_c8_bsrnn.log | cut -c1-300
