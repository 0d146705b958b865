#!/bin/bash
# Round 6, after ws_gemm_nt's transposed-W staging and the 32-sequence FP8 forward: TF-GridNet's profile set again, the whole -m gpu
# suite + smoke(), the headline line once more (no headline kernel changed: a check that it did not move)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_prof.sh tfgridnet 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/r06_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -10 $O/r06_full_gpu_suite.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1
echo "== smoke exit $?"; tail -1 $O/r06_smoke.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c40_bench_check.json 2> /dev/null
python -c "import json;d=json.load(open('$O/r06_c40_bench_check.json'));print('bsrnn check:', d['ms_per_step'], d['value'])"
