#!/bin/bash
# Round 3 profile collection for the headline step (one MI355X):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2`  -> r03_bench_r32_kernel_stats.csv,
#      the bench line of that run, the raw kernel trace and the main-stream idle analysis (tools/trace_gaps.py)
#   2. three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE) of
#      `bench.py --steps 1 --warmup 1`, summarised by tools/pmc_summary.py / pmc_mfma_summary.py
# Counter passes carry no trace options (gpurun refuses --pmc with sys/hip traces).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_r03
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/r03_prof_bench.json 2> $ROOT/gpurun_out/r03_prof.err
echo "rocprof exit $?"
f=$(find /tmp/prof_r03 -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/r03_bench_r32_kernel_stats.csv
t=$(find /tmp/prof_r03 -name "*kernel_trace.csv" | head -1)
cp "$t" $ROOT/gpurun_out/r03_kernel_trace.csv
python $ROOT/tools/trace_gaps.py $ROOT/gpurun_out/r03_kernel_trace.csv --steps 4 > $ROOT/gpurun_out/r03_trace_gaps.txt 2>&1
cat $ROOT/gpurun_out/r03_trace_gaps.txt
head -24 $ROOT/gpurun_out/r03_bench_r32_kernel_stats.csv | cut -c1-160
cut -c1-300 $ROOT/gpurun_out/r03_prof_bench.json
if [ "$1" = "pmc" ]; then
  COMMIT=$(cat $ROOT/.commit_for_profiles 2>/dev/null)
  SHA=$(python -c "import hashlib;print(hashlib.sha256(open('$ROOT/bench.py','rb').read()).hexdigest()[:16])")
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 500 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
    echo "pmc $c exit $?"
    cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
  done
  python $ROOT/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv $ROOT/gpurun_out/r03_pmc_traffic.json "$COMMIT" "$SHA" \
    "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
  rm -rf /tmp/pmc_mfma
  timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmc_mfma.log 2>&1
  echo "pmc mfma exit $?"
  python $ROOT/tools/pmc_mfma_summary.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $ROOT/gpurun_out/r03_pmc_mfma.json "$COMMIT" "$SHA" 2>&1 | tail -3
  python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/r03_pmc_traffic.json"))["kernels"]
tot=sum(v["hbm_bytes_per_launch_corrected"]*v["launches"] for v in d.values())
print("PMC traffic total over the collected steps: %.1f GB"%(tot/1e9))
for k,v in list(d.items())[:14]:
    print("%-60s n=%3d %.3f GB/launch"%(k[:60],v["launches"],v["hbm_bytes_per_launch_corrected"]/1e9))
PY
fi
