#!/bin/bash
# round 6, call 53: the step fence (dev.StepFence, WESEP_RUN_AHEAD): per-step wall-clock + allocator counters at depth 1 (default),
# 2, 0 and off; then the bench lines the way the driver runs them (with the CPU baseline), twice, and the joint variant
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
for ra in 1 2 0 -1; do
  WESEP_RUN_AHEAD=$ra timeout 200 python tools/r06_diag_steps.py --tag head_ra$ra --steps 8
done
WESEP_RUN_AHEAD=1 timeout 300 python tools/r06_diag_steps.py --tag joint_ra1 --joint --steps 6
} > $O/r06_c53_diag.txt 2>&1
grep -E "steps:|step [0-9]:" $O/r06_c53_diag.txt | cut -c1-200
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 > $O/r06_c53_bench_full_run$i.json 2> $O/r06_c53_bench.err
  python -c "import json;d=json.load(open('$O/r06_c53_bench_full_run$i.json'));print('full run $i:', d['ms_per_step'], d['value'])"
done
timeout 600 python bench.py > $O/r06_c53_bench_default.json 2> $O/r06_c53_bench.err
python -c "import json;d=json.load(open('$O/r06_c53_bench_default.json'));print('default flags:', d['ms_per_step'], d['value'], d['steps'], d['warmup'])"
timeout 300 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_c53_bench_joint.json 2>> $O/r06_c53_bench.err
python -c "import json;d=json.load(open('$O/r06_c53_bench_joint.json'));print('joint:', d['ms_per_step'], d['value'])"
