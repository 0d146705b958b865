#!/bin/bash
# round 6, call 54: operands of the deferred weight-gradient jobs held by the box (WESEP_WGRAD_HOLD=1, default) instead of
# record_stream: allocator counters with the fence on / off, the bench the way the driver runs it (default flags) x 3, joint,
# TF-GridNet / DPCCN lines, and the parity tests that cover the backward
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
WESEP_WGRAD_HOLD=1 WESEP_RUN_AHEAD=1  timeout 200 python tools/r06_diag_steps.py --tag hold1_ra1 --steps 8
WESEP_WGRAD_HOLD=1 WESEP_RUN_AHEAD=-1 timeout 200 python tools/r06_diag_steps.py --tag hold1_raoff --steps 8
WESEP_WGRAD_HOLD=0 WESEP_RUN_AHEAD=1  timeout 200 python tools/r06_diag_steps.py --tag hold0_ra1 --steps 8
WESEP_WGRAD_HOLD=1 WESEP_RUN_AHEAD=1  timeout 300 python tools/r06_diag_steps.py --tag joint_hold1_ra1 --joint --steps 6
} > $O/r06_c54_diag.txt 2>&1
grep -E "steps:|step [0-9]:" $O/r06_c54_diag.txt | cut -c1-200
for i in 1 2 3; do
  timeout 600 python bench.py > $O/r06_c54_bench_default_run$i.json 2> $O/r06_c54_bench.err
  python -c "import json;d=json.load(open('$O/r06_c54_bench_default_run$i.json'));print('default flags run $i:', d['ms_per_step'], d['value'], d['steps'], d['warmup'])"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_c54_bench_20.json 2>> $O/r06_c54_bench.err
python -c "import json;d=json.load(open('$O/r06_c54_bench_20.json'));print('20 steps:', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_c54_bench_joint.json 2>> $O/r06_c54_bench.err
python -c "import json;d=json.load(open('$O/r06_c54_bench_joint.json'));print('joint:', d['ms_per_step'], d['value'])"
timeout 400 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c54_tfg.json 2>> $O/r06_c54_bench.err
python -c "import json;d=json.load(open('$O/r06_c54_tfg.json'));print('tfgridnet:', d['ms_per_step'], d.get('peak_mem_GB'))"
timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c54_dpccn.json 2>> $O/r06_c54_bench.err
python -c "import json;d=json.load(open('$O/r06_c54_dpccn.json'));print('dpccn:', d['ms_per_step'], d.get('peak_mem_GB'))"
timeout 1500 python -m pytest tests/test_bsrnn_gpu.py tests/test_tfgridnet_gpu.py tests/test_tfgridnet_blocked_gpu.py tests/test_cross_stream_gpu.py -x -q -m gpu 2>&1 | tail -3
