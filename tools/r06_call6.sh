#!/bin/bash
# Round 6, call 6 (DPCCN): the 16-output-channel halo weight gradient on v_mfma_f32_16x16x32_bf16 with the next tile's loads in
# flight (conv3x3_wgrad16_kernel): kernel test, A/B on one box, model tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_dpccn_gpu.py -q -x -k "halo_weight_gradient" > $O/r06_c6_wgrad.log 2>&1
echo "== wgrad kernel test exit $?"; tail -3 $O/r06_c6_wgrad.log | cut -c1-200
for v in 1 0 1; do
  WS_CONV3X3_WGRAD16=$v timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c6_dpccn_w16_$v.json 2> $O/r06_c6_dpccn.err
  echo "== dpccn bench wgrad16=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r06_c6_dpccn_w16_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
timeout 900 python -m pytest tests/test_dpccn_gpu.py -q -x > $O/r06_c6_dpccn_tests.log 2>&1
echo "== dpccn tests exit $?"; tail -3 $O/r06_c6_dpccn_tests.log
