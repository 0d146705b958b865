#!/bin/bash
# Round 6, call 14: ring fragments refilled as early as possible (fused forward: gate by gate; band BPTT: k-step by k-step)
# instead of slot-wise -- alone (WS_FUSED_ER / WS_BAND_ER = 0 / 1), in the step, kernel tests + quick parity
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do
  WS_FUSED_ER=$v WS_BAND_ER=$v timeout 300 python tools/r06_band_probe.py > $O/r06_c14_band_probe_er$v.txt 2>&1
  echo "== band probe ER=$v exit $?"; grep -v amdgpu.ids $O/r06_c14_band_probe_er$v.txt | grep -E "fp16 h|BPTT|rel-L2"
done
timeout 300 python -m pytest tests/test_gates_h2_gpu.py tests/test_cluster2_gpu.py -q -x > $O/r06_c14_tests.log 2>&1
echo "== kernel tests exit $?"; tail -2 $O/r06_c14_tests.log | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c14_bench_$name.json 2> $O/r06_c14_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c14_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r06_c14_bench_$name.err | cut -c1-200
}
run er0 WS_FUSED_ER=0 WS_BAND_ER=0
run er1
run er0_b WS_FUSED_ER=0 WS_BAND_ER=0
run er1_b
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "training_step_matches or trajectory or resrnn_block or fused_input" > $O/r06_c14_parity.log 2>&1
echo "== quick parity exit $?"; grep -E "trajectory|passed|failed|worst|Error" $O/r06_c14_parity.log | cut -c1-300 | tail -8
