#!/bin/bash
# Round 6: the whole -m gpu suite in ONE call + smoke(), per-file durations recorded; then the TF-GridNet line at config 5's
# per-GPU shape with the round's defaults (fp16 h in its fused intra-frame forward, rfmt 2 in its streaming BPTT).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/r06_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -25 $O/r06_full_gpu_suite.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1
echo "== smoke exit $?"; tail -3 $O/r06_smoke.log | cut -c1-300
for v in new old; do
  if [ $v = old ]; then export WESEP_FUSED_H16=0 WESEP_BAND_RF=0; fi
  timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_tfg_bench_$v.json 2> $O/r06_tfg_$v.err
  echo "== tfgridnet $v exit $?: $(python -c "import json;d=json.loads(open('$O/r06_tfg_bench_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d.get('peak_mem_GB'), d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
