#!/bin/bash
# Round 6, with the pack prefetch opt-in (the default path of the round's last commit): the headline profile set, default-flags bench twice,
# the whole -m gpu suite and smoke()
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_prof.sh bsrnn 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-200
for i in 1 2; do
  timeout 300 python bench.py > $O/r06_bench_default_flags_run$i.json 2> /dev/null
  python -c "import json;d=json.load(open('$O/r06_bench_default_flags_run$i.json'));print('default flags run $i:', d['ms_per_step'], d['value'])"
done
timeout 2400 python -m pytest tests -m gpu -q --durations=3 > $O/r06_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; grep -E "passed|failed" $O/r06_full_gpu_suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1
echo "== smoke exit $?"; tail -1 $O/r06_smoke.log | cut -c1-200
