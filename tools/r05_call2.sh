#!/bin/bash
# Round 5, call 2: the second-generation cluster forward (ws_lstm_fwd_cluster2: fp16 h, fused x-projection, tagged hand-off).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -s > $O/r05_c2_cluster2.log 2>&1
echo "== cluster2 tests exit $?"; grep -E "cluster2|passed|failed|Error|assert " $O/r05_c2_cluster2.log | cut -c1-300 | tail -20
timeout 200 python tools/r05_recur_probe.py --no-stamps > $O/r05_c2_recur_probe.txt 2>&1
echo "== probe exit $?"; cat $O/r05_c2_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c2_bench_$name.json 2> $O/r05_c2_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c2_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline'].get('class_ms_per_step'))" 2>&1)"; tail -1 $O/r05_c2_bench_$name.err | cut -c1-200
}
run c2on WESEP_LSTM_CLUSTER2=1
run c2off WESEP_LSTM_CLUSTER2=0
run c2on_b WESEP_LSTM_CLUSTER2=1
timeout 400 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "full_size_row or training_step_matches or batch_rows or uninitialised" > $O/r05_c2_bsrnn.log 2>&1
echo "== bsrnn parity subset exit $?"; grep -E "est rel|full-size|passed|failed|Error|assert " $O/r05_c2_bsrnn.log | cut -c1-260
