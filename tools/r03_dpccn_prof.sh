#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_dp
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -- python $ROOT/tools/bench_dpccn.py --rows 32 --joint --steps 3 > $ROOT/gpurun_out/r03_dpccn_prof_bench.json 2> $ROOT/gpurun_out/r03_dpccn_prof.err
echo "exit $?"
cp "$(find /tmp/prof_dp -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03_dpccn_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/r03_dpccn_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=4
print('total kernel ms per step', tot/1e6/n)
for r in rows[:32]:
    print(f"{r['Name'][:96]:96s} n/step {int(r['Calls'])/n:7.1f} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6/n:6.1f} ms/step")
PY
