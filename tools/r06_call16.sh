#!/bin/bash
# Round 6, call 16: the last time-view layer's weight-gradient jobs released behind its own BPTT (WESEP_TAIL_FLUSH): A/B, trace gaps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c16_bench_$name.json 2> $O/r06_c16_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c16_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r06_c16_bench_$name.err | cut -c1-200
}
run tf0 WESEP_TAIL_FLUSH=0
run tf1
run tf0_b WESEP_TAIL_FLUSH=0
run tf1_b
timeout 300 python -m pytest tests/test_bsrnn_gpu.py -q -k "side_stream or training_step_matches or trajectory" > $O/r06_c16_tests.log 2>&1
echo "== tests exit $?"; tail -2 $O/r06_c16_tests.log | cut -c1-200
cd /tmp; rm -rf /tmp/prof_c16
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c16 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_c16.err
python $ROOT/tools/trace_gaps.py "$(find /tmp/prof_c16 -name '*kernel_trace.csv' | head -1)" --steps 4 > $O/r06_c16_trace_gaps.txt 2>&1
head -12 $O/r06_c16_trace_gaps.txt | cut -c1-150
