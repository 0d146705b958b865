#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_tfg
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tfg -- python $ROOT/tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 > $ROOT/gpurun_out/r03_tfgridnet_prof_bench.json 2> $ROOT/gpurun_out/r03_tfgridnet_prof.err
echo "exit $?"
cp "$(find /tmp/prof_tfg -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03_tfgridnet_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/r03_tfgridnet_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/3)
for r in rows[:34]:
    print(f"{r['Name'][:90]:90s} n/step {int(r['Calls'])/3:7.1f} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/1e6/3:7.1f} ms/step")
PY
