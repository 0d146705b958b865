#!/bin/bash
# Round 6, call 13: the fused band forward with W_hh's lo plane as FP8 (hfmt 1 | 4, WESEP_FUSED_F8=1): kernel test, alone, in the
# step, trajectory / training-step parity with it on
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -s -k "fused_band" > $O/r06_c13_fused.log 2>&1
echo "== fused tests exit $?"; grep -E "hfmt 5|passed|failed|Error|assert " $O/r06_c13_fused.log | cut -c1-200 | tail -8
timeout 300 python tools/r06_band_probe.py > $O/r06_c13_band_probe.txt 2>&1
echo "== band probe exit $?"; grep -v amdgpu.ids $O/r06_c13_band_probe.txt | grep "fused band forward"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c13_bench_$name.json 2> $O/r06_c13_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c13_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r06_c13_bench_$name.err | cut -c1-200
}
run f8_0 WESEP_FUSED_F8=0
run f8_1 WESEP_FUSED_F8=1
run f8_0b WESEP_FUSED_F8=0
run f8_1b WESEP_FUSED_F8=1
WESEP_FUSED_F8=1 timeout 900 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "training_step_matches or trajectory or resrnn_block or full_size_row or fused_input" > $O/r06_c13_parity_f8.log 2>&1
echo "== parity with the FP8 lo plane exit $?"; grep -E "trajectory|full-size|passed|failed|worst|Error" $O/r06_c13_parity_f8.log | cut -c1-300 | tail -12
