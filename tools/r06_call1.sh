#!/bin/bash
# Round 6, call 1 (code = round 5's last commit + measurement plumbing): (a) is the band-view BPTT's rfmt 2 product? -- config 2's
# parity with it forced on; (b) the side-stream tax (item 2): bench with / without the deferred
# weight-gradient jobs on one box, the pair BPTT stamped inside the step, the kernel trace binned by company
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c1_bench_$name.json 2> $O/r06_c1_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c1_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1)"; tail -1 $O/r06_c1_bench_$name.err | cut -c1-200
}
run base
run skipwg WESEP_PROBE_SKIP_WGRAD=1
run brf2 WESEP_BAND_RF=2
run base_b
timeout 300 python tools/r06_instep_stamps.py > $O/r06_instep_pair_stamps.txt 2> $O/r06_instep.err
echo "== in-step stamps exit $?"; grep -v amdgpu.ids $O/r06_instep_pair_stamps.txt | cut -c1-220
cd /tmp; rm -rf /tmp/prof_c1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_c1_bench_under_rocprof.json 2> /tmp/prof_c1.err
echo "== rocprof exit $?"
TR="$(find /tmp/prof_c1 -name '*kernel_trace.csv' | head -1)"
python $ROOT/tools/side_tax.py "$TR" --steps 4 > $O/r06_c1_side_tax.txt 2>&1
cat $O/r06_c1_side_tax.txt | cut -c1-230
python $ROOT/tools/trace_gaps.py "$TR" --steps 4 > $O/r06_c1_trace_gaps.txt 2>&1
gzip -c "$TR" > $O/r06_c1_kernel_trace.csv.gz
cd $ROOT
WESEP_BAND_RF=2 timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "config2 or full_size_row or training_step_matches" > $O/r06_c1_parity_brf2.log 2>&1
echo "== band rfmt 2 parity exit $?"; grep -E "passed|failed|worst|rel|Error" $O/r06_c1_parity_brf2.log | cut -c1-300 | tail -12
