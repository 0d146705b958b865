#!/bin/bash
# Round 6, call 2: d(xn) inside the band-view BPTT (ws_lstm_args.dxn, rfmt 2 default) and the residency gate in front of the
# side stream's jobs (ws_wait_word): kernel tests, A/B on one box, quick parity
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -s -k "bptt and blk32" > $O/r06_c2_blk.log 2>&1
echo "== streaming BPTT tests exit $?"; grep -E "d\(xn\)|rfmt|passed|failed|Error|assert " $O/r06_c2_blk.log | cut -c1-300 | tail -14
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c2_bench_$name.json 2> $O/r06_c2_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c2_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r06_c2_bench_$name.err | cut -c1-200
}
run new
run nodx WESEP_BAND_DX=0
run nogate WESEP_SIDE_GATE=0
run old WESEP_BAND_DX=0 WESEP_SIDE_GATE=0 WESEP_BAND_RF=0
run new_b
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "training_step_matches or trajectory or resrnn_block or side_stream" > $O/r06_c2_parity.log 2>&1
echo "== quick parity exit $?"; grep -E "trajectory|passed|failed|worst|step|Error" $O/r06_c2_parity.log | cut -c1-300 | tail -12
cd /tmp; rm -rf /tmp/prof_c2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_c2_bench_under_rocprof.json 2> /tmp/prof_c2.err
echo "== rocprof exit $?"
TR="$(find /tmp/prof_c2 -name '*kernel_trace.csv' | head -1)"
python $ROOT/tools/side_tax.py "$TR" --steps 4 > $O/r06_c2_side_tax.txt 2>&1
grep -A4 "lstm_bwd_pair_kernel\|lstm_bwd_bf16\|gemm_b2p_kernel<2>" $O/r06_c2_side_tax.txt | cut -c1-230
cp "$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1)" $O/r06_c2_kernel_stats.csv
head -14 $O/r06_c2_kernel_stats.csv | cut -c1-160
