#!/bin/bash
# round 6, call 41: ws_scale_bf_bwd with 16-byte accesses: kernel test + TF-GridNet line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_dpccn_gpu.py tests/test_tfgridnet_gpu.py -x -q -m gpu -k "pool or small or scale or elementwise or fixture or config5" 2>&1 | tail -3
for i in 1 2; do
timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c41_tfg_run$i.json 2> $O/r06_c41_tfg.err
python -c "import json;d=json.loads(open('$O/r06_c41_tfg_run$i.json').read().strip().splitlines()[-1]);print('tfgridnet run $i:', d['ms_per_step'], d['value'])"
done
