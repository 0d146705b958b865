#!/bin/bash
# Round 6, call 3: d(xn) inside the band BPTT on the SHARED B fragments (32x32x16 tiles, k-step parity per wave half), the fused
# band forward on fp16 h (hfmt 1), per-workgroup wall-clock stamps of the pair BPTT inside the step: kernel tests, A/B, parity
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -s -k "bptt and blk32" > $O/r06_c3_blk.log 2>&1
echo "== streaming BPTT tests exit $?"; grep -E "d\(xn\)|d\(gates\)|rfmt|passed|failed|Error|assert " $O/r06_c3_blk.log | cut -c1-300 | tail -14
timeout 300 python -m pytest tests/test_cluster2_gpu.py -q -x -s -k "fused_band" > $O/r06_c3_fused.log 2>&1
echo "== fused h16 tests exit $?"; grep -E "fused band|passed|failed|Error|assert " $O/r06_c3_fused.log | cut -c1-300 | tail -10
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c3_bench_$name.json 2> $O/r06_c3_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c3_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r06_c3_bench_$name.err | cut -c1-200
}
run new
run nodx WESEP_BAND_DX=0
run noh16 WESEP_FUSED_H16=0
run old WESEP_BAND_DX=0 WESEP_SIDE_GATE=0 WESEP_BAND_RF=0 WESEP_FUSED_H16=0
run new_b
timeout 300 python tools/r06_instep_stamps.py > $O/r06_c3_instep_pair_stamps.txt 2> $O/r06_c3_instep.err
echo "== in-step stamps exit $?"; grep -v amdgpu.ids $O/r06_c3_instep_pair_stamps.txt | grep -A8 "wall clock\|===" | cut -c1-260
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "training_step_matches or trajectory or resrnn_block or side_stream or fused_input" > $O/r06_c3_parity.log 2>&1
echo "== quick parity exit $?"; grep -E "trajectory|passed|failed|worst|step|Error" $O/r06_c3_parity.log | cut -c1-300 | tail -12
cd /tmp; rm -rf /tmp/prof_c3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_c3_bench_under_rocprof.json 2> /tmp/prof_c3.err
echo "== rocprof exit $?"
cp "$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1)" $O/r06_c3_kernel_stats.csv
head -12 $O/r06_c3_kernel_stats.csv | cut -c1-160
