#!/bin/bash
# round 6, call 39: the FP8 lo term in the 32-sequence fused forward (TF-GridNet's intra-frame path) + a_fmt 3 in its d(xn) GEMM
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_cluster2_gpu.py -x -q -m gpu -k "fused_band_forward" -s 2>&1 | grep -i "hfmt 5\|passed\|failed" | tail -8
timeout 1500 python -m pytest tests/test_tfgridnet_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in new old; do
  if [ $v = old ]; then export WESEP_FUSED_F8=0 WESEP_DXN_F8=0; fi
  timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c39_tfg_$v.json 2> $O/r06_c39_tfg.err
  python -c "import json;d=json.loads(open('$O/r06_c39_tfg_$v.json').read().strip().splitlines()[-1]);print('tfgridnet $v:', d['ms_per_step'], d['value'], d.get('peak_mem_GB'), d['roofline']['kernel_ms_per_step'])"
done
