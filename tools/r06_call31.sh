#!/bin/bash
# round 6, call 31: ws_affine_bwd with 16-byte accesses: test + the headline line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "affine" 2>&1 | tail -3
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c31_bench_run$i.json 2> $O/r06_c31_bench.err
  python -c "import json;d=json.load(open('$O/r06_c31_bench_run$i.json'));print('bsrnn run $i:', d['ms_per_step'], d['value'], {k:round(v['ms_per_step'],2) for k,v in d['roofline_by_class'].items()})"
done
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_c31
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c31 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/r06_c31_bench_rocprof.json 2> /tmp/prof_c31.err
cp "$(find /tmp/prof_c31 -name '*kernel_stats.csv' | head -1)" $R/$O/r06_c31_kernel_stats.csv
python - <<P
import csv
rows=list(csv.DictReader(open("$R/$O/r06_c31_kernel_stats.csv")))
for r in rows[:26]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>5s} {int(r['TotalDurationNs'])/7e6:8.2f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us")
P
