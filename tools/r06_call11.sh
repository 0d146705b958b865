#!/bin/bash
# Round 6, call 11 (DPCCN): the halo weight gradients' staging items remapped (lane = channel quad x row quad): kernel tests, bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_dpccn_gpu.py tests/test_resnet_gpu.py -q -x > $O/r06_c11_tests.log 2>&1
echo "== dpccn + resnet tests exit $?"; tail -3 $O/r06_c11_tests.log | cut -c1-200
for v in a b; do
  timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c11_dpccn_$v.json 2> $O/r06_c11_dpccn.err
  echo "== dpccn bench $v exit $?: $(python -c "import json;d=json.loads(open('$O/r06_c11_dpccn_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
