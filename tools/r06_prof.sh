#!/bin/bash
# Round 6 (same recipe as rounds 4-5): every committed bench line reproducible at ONE commit, one script.  Per model, in one
# GPU call:  FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (separate runs, no trace options: gpurun refuses --pmc with
# sys / hip traces), rocprofv3 --kernel-trace --stats of the bench command, and the bench line itself with `roofline`
# (non-null `traffic`, `counters_from` with commit + command) and `cpu_baseline`.
#   tools/r06_prof.sh bsrnn | dpccn | tfgridnet | convtasnet [...]
# Output: gpurun_out/r06_<model>_{bench.json,kernel_stats.csv,pmc_traffic.json,pmc_mfma.json}; copy to profiles/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
COMMIT=$(cat $ROOT/.commit_for_profiles 2>/dev/null)
for M in "$@"; do
  case $M in
    bsrnn)      TOOL=bench.py;                  LONG="--steps 10 --warmup 3";                                  SHORT="--steps 1 --warmup 1 --no-cpu-baseline"; STATS="--steps 5 --warmup 2 --no-cpu-baseline";;
    dpccn)      TOOL=tools/bench_dpccn.py;      LONG="--rows 32 --joint --steps 5 --warmup 2 --cpu";         SHORT="--rows 32 --joint --steps 1 --warmup 1";  STATS="--rows 32 --joint --steps 3 --warmup 1";;
    tfgridnet)  TOOL=tools/bench_tfgridnet.py;  LONG="--rows 8 --recipe --steps 3 --warmup 1 --cpu";         SHORT="--rows 8 --recipe --steps 1 --warmup 1";  STATS="--rows 8 --recipe --steps 2 --warmup 1";;
    convtasnet) TOOL=tools/bench_convtasnet.py; LONG="--steps 20 --warmup 5 --cpu";                                                SHORT="--steps 1 --warmup 1";                    STATS="--steps 5 --warmup 2";;
    *) echo "unknown model $M"; continue;;
  esac
  SHA=$(python -c "import hashlib;print(hashlib.sha256(open('$ROOT/$TOOL','rb').read()).hexdigest()[:16])")
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${M}_$c
    timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${M}_$c -- python $ROOT/$TOOL $SHORT > /tmp/pmc_${M}_$c.log 2>&1
    echo "[$M] pmc $c exit $?"
    cp "$(find /tmp/pmc_${M}_$c -name '*counter_collection.csv' | head -1)" /tmp/pmc_${M}_$c.csv
  done
  python $ROOT/tools/pmc_summary.py /tmp/pmc_${M}_FETCH_SIZE.csv /tmp/pmc_${M}_WRITE_SIZE.csv $O/r06_${M}_pmc_traffic.json "$COMMIT" "$SHA" \
    "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python $TOOL $SHORT"
  rm -rf /tmp/pmc_${M}_mfma
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_${M}_mfma -- python $ROOT/$TOOL $SHORT > /tmp/pmc_${M}_mfma.log 2>&1
  echo "[$M] pmc mfma exit $?"
  python $ROOT/tools/pmc_mfma_summary.py "$(find /tmp/pmc_${M}_mfma -name '*counter_collection.csv' | head -1)" $O/r06_${M}_pmc_mfma.json "$COMMIT" "$SHA" \
    "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $TOOL $SHORT" 2>&1 | tail -2
  rm -rf /tmp/prof_$M
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$M -- python $ROOT/$TOOL $STATS > $O/r06_${M}_bench_under_rocprof.json 2> /tmp/prof_$M.err
  echo "[$M] rocprof stats exit $?"
  cp "$(find /tmp/prof_$M -name '*kernel_stats.csv' | head -1)" $O/r06_${M}_kernel_stats.csv
  if [ $M = bsrnn ]; then
    python $ROOT/tools/trace_gaps.py "$(find /tmp/prof_$M -name '*kernel_trace.csv' | head -1)" --steps 4 > $O/r06_bsrnn_trace_gaps.txt 2>&1
    # bench.py reads the counters of the SAME commit from profiles/ (on the box: this call's own passes)
    cp $O/r06_bsrnn_pmc_traffic.json $ROOT/profiles/r06_pmc_traffic.json
    cp $O/r06_bsrnn_pmc_mfma.json $ROOT/profiles/r06_pmc_mfma.json
  fi
  cd $ROOT
  timeout 500 python $TOOL $LONG > /tmp/line_$M.json 2> /tmp/line_$M.err
  echo "[$M] bench exit $?"
  if [ $M = bsrnn ]; then
    cp /tmp/line_$M.json $O/r06_bsrnn_bench.json
  else
    python tools/r04_attach_traffic.py /tmp/line_$M.json $O/r06_${M}_pmc_traffic.json $O/r06_${M}_pmc_mfma.json > $O/r06_${M}_bench.json
  fi
  cut -c1-600 $O/r06_${M}_bench.json
  python - <<PY
import json
d=json.load(open("$O/r06_${M}_pmc_traffic.json"))["kernels"]
tot=sum(v["hbm_bytes_per_launch_corrected"]*v["launches"] for v in d.values())
print("[$M] PMC traffic over the collected steps (1 warm-up + 1 timed): %.1f GB"%(tot/1e9))
for k,v in list(d.items())[:12]:
    print("  %-60s n=%4d %.3f GB/launch"%(k[:60],v["launches"],v["hbm_bytes_per_launch_corrected"]/1e9))
PY
  head -16 $O/r06_${M}_kernel_stats.csv | cut -c1-150
done
