#!/bin/bash
# Round 6, call 17: the optimizer's tensor table kept across steps (no per-step pageable upload = no per-step host wait): A/B, gaps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c17_bench_$name.json 2> $O/r06_c17_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r06_c17_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])" 2>&1)"; tail -1 $O/r06_c17_bench_$name.err | cut -c1-200
}
run tc0 WESEP_TABLE_CACHE=0
run tc1
run tc0_b WESEP_TABLE_CACHE=0
run tc1_b
timeout 300 python -m pytest tests/test_bsrnn_gpu.py tests/test_kernels_gpu.py -q -k "training_step_matches or trajectory or adam or clip" > $O/r06_c17_tests.log 2>&1
echo "== tests exit $?"; tail -2 $O/r06_c17_tests.log | cut -c1-200
cd /tmp; rm -rf /tmp/prof_c17
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c17 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_c17.err
python $ROOT/tools/trace_gaps.py "$(find /tmp/prof_c17 -name '*kernel_trace.csv' | head -1)" --steps 4 > $O/r06_c17_trace_gaps.txt 2>&1
head -12 $O/r06_c17_trace_gaps.txt | cut -c1-150
for t in dpccn tfgridnet; do
  if [ $t = dpccn ]; then CMD="tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2"; else CMD="tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1"; fi
  cd $ROOT
  for v in 0 1; do
    WESEP_TABLE_CACHE=$v timeout 500 python $CMD > $O/r06_c17_${t}_tc$v.json 2> /dev/null
    echo "== $t table cache $v: $(python -c "import json;d=json.loads(open('$O/r06_c17_${t}_tc$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"
  done
done
