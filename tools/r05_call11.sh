#!/bin/bash
# Round 5, call 11: cluster2 with one accumulator chain and the folded activation scale; TF-GridNet with cluster2 (now fp16 h only)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -s > $O/r05_c11_cluster2.log 2>&1
echo "== cluster2 tests exit $?"; grep -E "cluster2|passed|failed|Error|assert " $O/r05_c11_cluster2.log | cut -c1-300 | tail -12
timeout 200 python tools/r05_recur_probe.py > $O/r05_c11_recur_probe.txt 2>&1
echo "== probe exit $?"; sed -n 1,28p $O/r05_c11_recur_probe.txt | cut -c1-160
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c11_bench_$name.json 2> $O/r05_c11_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c11_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r05_c11_bench_$name.err | cut -c1-200
}
run head A=1
run head_b A=1
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "trajectory or full_size_row or training_step_matches" > $O/r05_c11_bsrnn.log 2>&1
echo "== bsrnn parity subset exit $?"; grep -E "full-size|trajectory\[|passed|failed|Error|assert " $O/r05_c11_bsrnn.log | cut -c1-300
timeout 700 python tools/r05_tfg_cfg5.py > $O/r05_tfg_cfg5_b.txt 2>&1
echo "== tfgridnet precision probe exit $?"; tail -16 $O/r05_tfg_cfg5_b.txt | cut -c1-200
for v in c2 def; do
  if [ $v = c2 ]; then export WESEP_TFG_CLUSTER2=1; else unset WESEP_TFG_CLUSTER2; fi
  timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 4 --warmup 2 > $O/r05_c11_tfg_$v.json 2> $O/r05_c11_tfg_$v.err
  echo "== tfgridnet $v exit $?: $(python -c "import json;d=json.loads(open('$O/r05_c11_tfg_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['peak_mem_GB'], d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
