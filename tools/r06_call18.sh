#!/bin/bash
# Round 6, call 18: weight-gradient carriers created before the band split (its backward runs under the side stream's last jobs):
# bench x2, trace gaps, quick parity
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c18_bench_$name.json 2> $O/r06_c18_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c18_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r06_c18_bench_$name.err | cut -c1-200
}
run a
run notail WESEP_TAIL_FLUSH=0
run b
timeout 400 python -m pytest tests/test_bsrnn_gpu.py -q -k "side_stream or training_step_matches or trajectory or ddp or full_model" > $O/r06_c18_tests.log 2>&1
echo "== tests exit $?"; tail -2 $O/r06_c18_tests.log | cut -c1-200
cd /tmp; rm -rf /tmp/prof_c18
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c18 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_c18.err
python $ROOT/tools/trace_gaps.py "$(find /tmp/prof_c18 -name '*kernel_trace.csv' | head -1)" --steps 4 > $O/r06_c18_trace_gaps.txt 2>&1
head -10 $O/r06_c18_trace_gaps.txt | cut -c1-150
