#!/bin/bash
# round 6, call 34: the FiLM layers' Linear weight gradients on the side stream (LinearDeferFn): model tests, the line, the timeline
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_bsrnn_gpu.py -x -q -m gpu -k "not config2 and not full_size" 2>&1 | tail -4
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c34_bench_run$i.json 2> $O/r06_c34_bench.err
  python -c "import json;d=json.load(open('$O/r06_c34_bench_run$i.json'));print('bsrnn run $i:', d['ms_per_step'], d['value'], {k:round(v['ms_per_step'],2) for k,v in d['roofline_by_class'].items()})"
done
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_c34
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c34 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/r06_c34_bench_rocprof.json 2> /tmp/prof_c34.err
T="$(find /tmp/prof_c34 -name '*kernel_trace.csv' | head -1)"
python $R/tools/r06_step_timeline.py "$T" --min-us 0 > $O/r06_c34_step_timeline_all.txt 2>&1
python $R/tools/trace_gaps.py "$T" --steps 2 > $O/r06_c34_trace_gaps.txt 2>&1
head -1 $O/r06_c34_step_timeline_all.txt
