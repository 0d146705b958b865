#!/bin/bash
# Round 4, call 11: (a) DPCCN with the InstanceNorm sums folded (ws_in_act_sums_fold) -- kernel + model tests, bench A/B;
# (b) MEASUREMENT of VERDICT round 3 item 1d: the band-view fused forward with ONE weight plane (WS_FUSED_W1=1: bf16
# weights, two MFMAs per product, half the weight stream) -- what it would buy and what it costs in parity.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 400 python -m pytest tests/test_dpccn_gpu.py -q -x > $O/r04_c11_dpccn_tests.log 2>&1
echo "== dpccn tests exit $?"; tail -3 $O/r04_c11_dpccn_tests.log
for v in 1 0; do
  WESEP_IN_FOLD=$v timeout 300 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r04_c11_dpccn_fold$v.json 2> $O/r04_c11_dpccn_fold$v.err
  echo "== dpccn fold=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c11_dpccn_fold$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"
done
for v in 0 1; do
  WS_FUSED_W1=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c11_bench_w1_$v.json 2> $O/r04_c11_bench_w1_$v.err
  echo "== bench W1=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c11_bench_w1_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline'].get('kernel_ms_per_step'))" 2>&1)"
done
WS_FUSED_W1=1 WS_FUSED_SEQS=64 timeout 500 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "full_model_vs_oracle or config2 or full_size" > $O/r04_c11_w1_parity.log 2>&1
echo "== W1 parity run exit $? (failures expected: measurement)"; grep -E "est rel|worst grad|passed|failed|Error|assert " $O/r04_c11_w1_parity.log | cut -c1-260 | head -30
