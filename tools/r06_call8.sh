#!/bin/bash
# Round 6, call 8: the storage-format test on the round's defaults, the R = 32 oracle comparison with its figures printed,
# TF-GridNet with the one-term weight gradient (WESEP_TFG_TNB_A16) on / off, the new bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py tests/test_cluster2_gpu.py -q -x > $O/r06_c8_formats.log 2>&1
echo "== formats + cluster2 files exit $?"; tail -3 $O/r06_c8_formats.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c8_bench.json 2> $O/r06_c8_bench.err
echo "== bench exit $?"; python -c "
import json;d=json.loads(open('$O/r06_c8_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['critical_path_largest'], d['largest_any_stream'], d['step_roofline'].get('traffic_over_algorithmic'))
for k,v in d['roofline_by_class'].items(): print(' ', k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','launches_per_step','frac_hbm','frac_mfma_algorithmic','frac','bound','traffic_over_algorithmic','counted_launches','timed_launches')})
"; tail -2 $O/r06_c8_bench.err | cut -c1-300
for v in 0 1 0 1; do
  WESEP_TFG_TNB_A16=$v timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c8_tfg_a16_$v.json 2> $O/r06_c8_tfg.err
  echo "== tfgridnet A16=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r06_c8_tfg_a16_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d.get('peak_mem_GB'))" 2>&1)"
done
WESEP_RUN_SLOW=1 timeout 1200 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "headline_batch_r32" > $O/r06_headline_r32_vs_oracle.log 2>&1
echo "== R = 32 vs oracle exit $?"; grep -E "headline batch|passed|failed" $O/r06_headline_r32_vs_oracle.log | cut -c1-300
