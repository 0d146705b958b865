#!/bin/bash
# Round 5, call 4: cluster2's HBM traffic on the X-waves (A/B vs the M-waves), pair BPTT early prefetch (A/B), guard test.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py tests/test_bptt_survival_gpu.py -q -x -s > $O/r05_c4_kernels.log 2>&1
echo "== cluster2 / survival tests exit $?"; grep -E "cluster2|passed|failed|Error|assert " $O/r05_c4_kernels.log | cut -c1-300 | tail -12
WESEP_CLUSTER2_IO=m timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -k "on_vs_off" > $O/r05_c4_c2m.log 2>&1
echo "== cluster2 (I/O on M) composition test exit $?"; tail -2 $O/r05_c4_c2m.log | cut -c1-200
WESEP_PAIR_EARLY=1 timeout 300 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "full_size_row" > $O/r05_c4_early.log 2>&1
echo "== pair early-prefetch full-size parity exit $?"; grep -E "full-size|passed|failed" $O/r05_c4_early.log | cut -c1-200
timeout 200 python tools/r05_recur_probe.py > $O/r05_c4_recur_probe.txt 2>&1
echo "== probe exit $?"; cat $O/r05_c4_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c4_bench_$name.json 2> $O/r05_c4_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c4_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r05_c4_bench_$name.err | cut -c1-200
}
run iox A=1
run iom WESEP_CLUSTER2_IO=m
run early WESEP_PAIR_EARLY=1
run iox_b A=1
