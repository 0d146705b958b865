#!/bin/bash
# Round 5, call 5: cluster2 with scheduling fences at the phase boundaries (I/O on X vs M waves), gemm_tnb fp16 operands vs fp64 at
# the headline geometry, TF-GridNet on the round-5 recurrences (cluster2 + fp16 pair), SSA / joint variants.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py tests/test_gates_h2_gpu.py -q -x -s -k "cluster2 or headline" > $O/r05_c5_kernels.log 2>&1
echo "== cluster2 / tnb headline tests exit $?"; grep -E "cluster2 on|gemm_tnb fp16|passed|failed|Error|assert " $O/r05_c5_kernels.log | cut -c1-300 | tail -12
timeout 200 python tools/r05_recur_probe.py --no-stamps > $O/r05_c5_recur_probe.txt 2>&1
echo "== probe exit $?"; grep -E "^cluster|^pair|status" $O/r05_c5_recur_probe.txt | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c5_bench_$name.json 2> $O/r05_c5_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c5_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r05_c5_bench_$name.err | cut -c1-200
}
run iox A=1
run iom WESEP_CLUSTER2_IO=m
run iox_b A=1
run iom_b WESEP_CLUSTER2_IO=m
timeout 500 python -m pytest tests/test_tfgridnet_gpu.py tests/test_tfgridnet_blocked_gpu.py -q -x > $O/r05_c5_tfg_tests.log 2>&1
echo "== tfgridnet tests exit $?"; tail -3 $O/r05_c5_tfg_tests.log | cut -c1-300
for v in new old; do
  if [ $v = old ]; then export WESEP_LSTM_CLUSTER2=0 WESEP_PAIR_RF=0; fi
  timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 4 --warmup 2 > $O/r05_c5_tfg_$v.json 2> $O/r05_c5_tfg_$v.err
  echo "== tfgridnet $v exit $?: $(python -c "import json;d=json.loads(open('$O/r05_c5_tfg_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['peak_mem_GB'], d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
unset WESEP_LSTM_CLUSTER2 WESEP_PAIR_RF
timeout 400 python tools/bench_ssa.py --what joint,ssa,multi > $O/r05_c5_ssa_multi.jsonl 2> $O/r05_c5_ssa_multi.err
echo "== ssa / joint / multi exit $?"; grep "^{" $O/r05_c5_ssa_multi.jsonl | cut -c1-330
