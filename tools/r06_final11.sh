#!/bin/bash
# Round 6: the headline profile set at the round's last code commit once more, on one of the pool's FASTER boxes (the lease of the
# slower one had expired): PMC passes, kernel stats, the bench line, the driver's default flags three times, 20 steps, joint
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_prof.sh bsrnn 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-200
for i in 1 2 3; do
  timeout 300 python bench.py > $O/r06_bench_default_flags_run$i.json 2> /dev/null
  python -c "import json;d=json.load(open('$O/r06_bench_default_flags_run$i.json'));print('default flags run $i:', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_20steps.json 2>/dev/null
python -c "import json;d=json.load(open('$O/r06_bench_20steps.json'));print('20 steps:', d['ms_per_step'], d['value'])"
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_joint.json 2> /dev/null
python -c "import json;d=json.load(open('$O/r06_bench_joint.json'));print('joint:', d['ms_per_step'], d['value'])"
