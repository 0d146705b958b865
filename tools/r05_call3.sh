#!/bin/bash
# Round 5, call 3: cluster2 with the direct (stage-free) publish, the pair BPTT's data-tagged hand-off, the optimizer guard's
# device-side step lag; cycle stamps of both recurrences.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py tests/test_bptt_survival_gpu.py tests/test_cluster_robustness_gpu.py -q -x -s > $O/r05_c3_kernels.log 2>&1
echo "== cluster2 / survival tests exit $?"; grep -E "cluster2|passed|failed|Error|assert " $O/r05_c3_kernels.log | cut -c1-300 | tail -20
timeout 300 python -m pytest tests/test_gates_h2_gpu.py tests/test_kernels_gpu.py -q -x -k "pair or bptt or lstm" > $O/r05_c3_pair.log 2>&1
echo "== pair tests exit $?"; tail -3 $O/r05_c3_pair.log | cut -c1-300
timeout 200 python tools/r05_recur_probe.py > $O/r05_c3_recur_probe.txt 2>&1
echo "== probe exit $?"; cat $O/r05_c3_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c3_bench_$name.json 2> $O/r05_c3_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c3_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r05_c3_bench_$name.err | cut -c1-200
}
run head A=1
run head_b A=1
timeout 400 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "full_size_row or training_step_matches or side_stream" > $O/r05_c3_bsrnn.log 2>&1
echo "== bsrnn parity subset exit $?"; grep -E "est rel|full-size|passed|failed|Error|assert " $O/r05_c3_bsrnn.log | cut -c1-260
