#!/bin/bash
# What round 3 left for the first GPU call of the next round (one MI355X, ~12 GPU-minutes):
#   1. the full `-m gpu` suite in ONE run -- round 3 ended with it split over three calls -- including
#      tests/test_zzz_engine_separators_gpu.py (six engine-vs-Python comparisons that have never run)
#   2. the Conv-TasNet / SpEx+ bench line with the `roofline` block the tool writes since the end of round 3, and cpu_baseline
#   3. PMC traffic (FETCH_SIZE, WRITE_SIZE: separate passes, no trace options) for the TF-GridNet step, so that its line gets a
#      non-null roofline.traffic like the DPCCN one (tools/r03_dpccn_pmc.sh is the model)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/ -q -m gpu -x --tb=short --durations=12 > gpurun_out/r04_full_gpu_suite.log 2>&1
echo "== suite exit $?"; tail -20 gpurun_out/r04_full_gpu_suite.log | cut -c1-200
timeout 300 python tools/bench_convtasnet.py --cpu > gpurun_out/r04_convtasnet_bench.json 2> gpurun_out/r04_convtasnet.err
echo "== convtasnet exit $?"; cut -c1-400 gpurun_out/r04_convtasnet_bench.json
export TMPDIR=/tmp
cd /tmp
SHA=$(python -c "import hashlib;print(hashlib.sha256(open('$ROOT/tools/bench_tfgridnet.py','rb').read()).hexdigest()[:16])")
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmct_$c -- python $ROOT/tools/bench_tfgridnet.py --rows 8 --recipe --steps 1 --warmup 1 > /tmp/pmct_$c.log 2>&1
  echo "pmc $c exit $?"
  cp $(find /tmp/pmct_$c -name "*counter_collection.csv" | head -1) /tmp/pmct_$c.csv
done
python $ROOT/tools/pmc_summary.py /tmp/pmct_FETCH_SIZE.csv /tmp/pmct_WRITE_SIZE.csv $ROOT/gpurun_out/r04_tfgridnet_pmc_traffic.json "$(cat $ROOT/.commit_for_profiles 2>/dev/null)" "$SHA" \
  "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python tools/bench_tfgridnet.py --rows 8 --recipe --steps 1 --warmup 1"
python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/r04_tfgridnet_pmc_traffic.json"))["kernels"]
tot=sum(v["hbm_bytes_per_launch_corrected"]*v["launches"] for v in d.values())
print("TF-GridNet PMC traffic over 1 warm-up + 1 timed step: %.1f GB"%(tot/1e9))
for k,v in list(d.items())[:12]:
    print("%-64s n=%4d %.4f GB/launch"%(k[:64],v["launches"],v["hbm_bytes_per_launch_corrected"]/1e9))
PY
