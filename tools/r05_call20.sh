#!/bin/bash
# Round 5, call 20: the band-view (streaming) BPTT on fp16 + FP8 weights (ws_lstm_args.rfmt = 2, opt-in WESEP_BAND_RF=2): kernel
# test, launch time alone, bench A/B on one box, the quick parity tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gates_h2_gpu.py -q -x -s -k "bptt and blk32" > $O/r05_c20_blk.log 2>&1
echo "== streaming BPTT tests exit $?"; grep -E "rfmt|passed|failed|Error|assert " $O/r05_c20_blk.log | cut -c1-300 | tail -8
timeout 120 python tools/r05_band_probe.py > $O/r05_c20_band_probe.txt 2>&1
echo "== band probe exit $?"; grep -v amdgpu.ids $O/r05_c20_band_probe.txt | tail -4
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_c20_bench_$name.json 2> $O/r05_c20_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r05_c20_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r05_c20_bench_$name.err | cut -c1-200
}
run b0 WESEP_BAND_RF=0
run b2 WESEP_BAND_RF=2
run b0_b WESEP_BAND_RF=0
run b2_b WESEP_BAND_RF=2
WESEP_BAND_RF=2 timeout 200 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "trajectory or training_step" > $O/r05_c20_parity_b2.log 2>&1
echo "== band rfmt 2 quick parity exit $?"; grep -E "trajectory|passed|failed|worst|step" $O/r05_c20_parity_b2.log | cut -c1-300
