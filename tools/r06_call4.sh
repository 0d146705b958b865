#!/bin/bash
# Round 6, call 4: after the ISA findings (a loop header's s_waitcnt is the minimum over the ways into it: the prologue order made
# every BPTT step drain vmcnt to 0; the fused forward drained its stores every step by design): kernel tests, the band kernels
# alone, A/B in the step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -s -k "bptt and blk32" > $O/r06_c4_blk.log 2>&1
echo "== streaming BPTT tests exit $?"; grep -E "passed|failed|Error|assert " $O/r06_c4_blk.log | cut -c1-300 | tail -4
timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -s -k "fused_band" > $O/r06_c4_fused.log 2>&1
echo "== fused h16 tests exit $?"; grep -E "passed|failed|Error|assert " $O/r06_c4_fused.log | cut -c1-300 | tail -4
timeout 300 python tools/r06_band_probe.py > $O/r06_c4_band_probe.txt 2>&1
echo "== band probe exit $?"; grep -v amdgpu.ids $O/r06_c4_band_probe.txt | tail -12
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c4_bench_$name.json 2> $O/r06_c4_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c4_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r06_c4_bench_$name.err | cut -c1-200
}
run new
run nodx WESEP_BAND_DX=0
run drain WESEP_FUSED_DRAIN=1

run new_b
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "training_step_matches or trajectory or resrnn_block or side_stream or fused_input" > $O/r06_c4_parity.log 2>&1
echo "== quick parity exit $?"; grep -E "trajectory|passed|failed|worst|step|Error" $O/r06_c4_parity.log | cut -c1-300 | tail -12
