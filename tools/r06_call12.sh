#!/bin/bash
# Round 6, call 12: where do the band kernels' weight fragments come from?  L2 hit / miss and the L2's memory-side read requests
# of the band-view kernels alone (tools/r06_band_probe.py), one counter pass each (no trace options beside --pmc)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$tag -- python $ROOT/tools/r06_band_probe.py > /tmp/pmc_$tag.log 2>&1
  echo "[pmc $c] exit $?"
  f="$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)"
  python - "$f" >> $O/r06_c12_band_l2_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no counters:", e)
for k, cs in agg.items():
    if "lstm" in k or "gemm_b2p" in k:
        print(k, {c: f"{sum(v)/len(v):.4g} (n={len(v)})" for c, v in cs.items()})
PY
done
cat $O/r06_c12_band_l2_counters.txt | cut -c1-250
