#!/bin/bash
# Round 6, the round's LAST profile pass (after the run-ahead fix): all four models' profile sets at one commit, the recipe variants,
# the bench with the driver's default flags three times, the whole -m gpu suite, smoke(), the R = 32 oracle comparison, a timeline
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_prof.sh bsrnn 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-200
for i in 1 2 3; do
  timeout 300 python bench.py > $O/r06_bench_default_flags_run$i.json 2> /dev/null
  python -c "import json;d=json.load(open('$O/r06_bench_default_flags_run$i.json'));print('default flags run $i:', d['ms_per_step'], d['value'])"
done
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_joint.json 2> /dev/null
timeout 300 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r06_ssa_multi_bench.jsonl 2> $O/r06_ssa_multi.err
grep "^{" $O/r06_ssa_multi_bench.jsonl | cut -c1-170
bash tools/r06_prof.sh tfgridnet dpccn convtasnet 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q --durations=3 > $O/r06_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; grep -E "passed|failed" $O/r06_full_gpu_suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1
echo "== smoke exit $?"; tail -1 $O/r06_smoke.log | cut -c1-200
WESEP_RUN_SLOW=1 timeout 1200 python -m pytest tests/test_bsrnn_gpu.py -m gpu -q -s -k headline_batch_r32 > $O/r06_headline_r32_vs_oracle.log 2>&1
echo "== R = 32 vs oracle exit $?"; grep -E "passed|failed|worst|waveform" $O/r06_headline_r32_vs_oracle.log | tail -4 | cut -c1-200
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_tl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_tl.err
T="$(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1)"
python $R/tools/trace_gaps.py "$T" --steps 2 > $R/$O/r06_bsrnn_trace_gaps.txt 2>&1
python $R/tools/r06_step_timeline.py "$T" --min-us 100 > $R/$O/r06_bsrnn_step_timeline.txt 2>&1
head -4 $R/$O/r06_bsrnn_trace_gaps.txt
