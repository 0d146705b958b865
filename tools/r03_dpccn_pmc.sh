#!/bin/bash
# Round 3: HBM traffic of the DPCCN step from the PMC counters (two separate passes, FETCH_SIZE and WRITE_SIZE; no trace
# options), summarised per kernel by tools/pmc_summary.py -> gpurun_out/r03_dpccn_pmc_traffic.json
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
COMMIT=$(cat $ROOT/.commit_for_profiles 2>/dev/null)
SHA=$(python -c "import hashlib;print(hashlib.sha256(open('$ROOT/tools/bench_dpccn.py','rb').read()).hexdigest()[:16])")
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcd_$c
  timeout 110 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcd_$c -- python $ROOT/tools/bench_dpccn.py --rows 32 --joint --steps 1 --warmup 1 > /tmp/pmcd_$c.log 2>&1
  echo "pmc $c exit $?"
  cp $(find /tmp/pmcd_$c -name "*counter_collection.csv" | head -1) /tmp/pmcd_$c.csv
done
python $ROOT/tools/pmc_summary.py /tmp/pmcd_FETCH_SIZE.csv /tmp/pmcd_WRITE_SIZE.csv $ROOT/gpurun_out/r03_dpccn_pmc_traffic.json "$COMMIT" "$SHA" \
  "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python tools/bench_dpccn.py --rows 32 --joint --steps 1 --warmup 1"
python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/r03_dpccn_pmc_traffic.json"))["kernels"]
tot=sum(v["hbm_bytes_per_launch_corrected"]*v["launches"] for v in d.values())
print("PMC traffic total over the collected steps (1 warm-up + 1 timed): %.1f GB"%(tot/1e9))
for k,v in list(d.items())[:16]:
    print("%-64s n=%4d %.4f GB/launch"%(k[:64],v["launches"],v["hbm_bytes_per_launch_corrected"]/1e9))
PY
