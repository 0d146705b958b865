#!/bin/bash
# round 6, call 27: the whole -m gpu suite + smoke() with the FP8 lo term as the default in the band forward, the pair BPTT and the
# streaming BPTT; the headline line (two runs); TF-GridNet with the defaults and with the round's earlier arithmetic
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/r06_c27_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -18 $O/r06_c27_full_gpu_suite.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_c27_smoke.log 2>&1
echo "== smoke exit $?"; tail -2 $O/r06_c27_smoke.log | cut -c1-300
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c27_bench_run$i.json 2> $O/r06_c27_bench.err
  python -c "import json;d=json.load(open('$O/r06_c27_bench_run$i.json'));print('bench run $i:', d['ms_per_step'], d['value'], d['dtype'])"
done
for v in new old; do
  if [ $v = old ]; then export WESEP_PAIR_RF=2 WESEP_BAND_RF=2; fi
  timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c27_tfg_bench_$v.json 2> $O/r06_c27_tfg_$v.err
  echo "== tfgridnet $v exit $?: $(python -c "import json;d=json.loads(open('$O/r06_c27_tfg_bench_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d.get('peak_mem_GB'), d['roofline']['kernel_ms_per_step'])" 2>&1)"
done
