#!/bin/bash
# Round 6, call 10 (DPCCN): the one-launch weight pack (ws_conv3x3_pack): kernel test, model tests, bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dpccn_gpu.py -q -x > $O/r06_c10_dpccn_tests.log 2>&1
echo "== dpccn tests exit $?"; tail -3 $O/r06_c10_dpccn_tests.log | cut -c1-200
for v in a b; do
  timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c10_dpccn_$v.json 2> $O/r06_c10_dpccn.err
  echo "== dpccn bench $v exit $?: $(python -c "import json;d=json.loads(open('$O/r06_c10_dpccn_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
timeout 600 python -m pytest tests/test_zzz_engine_separators_gpu.py -q -x -k "dpccn" > $O/r06_c10_engine_dpccn.log 2>&1
echo "== engine dpccn tests exit $?"; tail -2 $O/r06_c10_engine_dpccn.log | cut -c1-200
