#!/bin/bash
# Round 6, call 9: the band-view kernels' two directions interleaved across workgroups (WS_BAND_DIRMAP): alone and in the step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do
  WS_BAND_DIRMAP=$v timeout 300 python tools/r06_band_probe.py > $O/r06_c9_band_probe_dm$v.txt 2>&1
  echo "== band probe dirmap=$v exit $?"; grep -v amdgpu.ids $O/r06_c9_band_probe_dm$v.txt | grep -E "vmcnt\(48\)|two terms, 96|rel-L2"
done
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c9_bench_$name.json 2> $O/r06_c9_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c9_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r06_c9_bench_$name.err | cut -c1-200
}
run dm0 WS_BAND_DIRMAP=0
run dm1 WS_BAND_DIRMAP=1
run dm0_b WS_BAND_DIRMAP=0
run dm1_b WS_BAND_DIRMAP=1
WS_BAND_DIRMAP=1 timeout 300 python -m pytest tests/test_gates_h2_gpu.py tests/test_cluster2_gpu.py -q -x -k "bptt or fused_band or formats" > $O/r06_c9_tests_dm1.log 2>&1
echo "== kernel tests with dirmap exit $?"; tail -2 $O/r06_c9_tests_dm1.log | cut -c1-200
