#!/bin/bash
# round 6, call 38: ws_gemm_nt with W as it lies (vec bit 3): the attention products of TF-GridNet without transposing copies
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_tfgridnet_gpu.py -x -q -m gpu 2>&1 | tail -3
for nn in 1 0; do
  WESEP_GEMM_NN=$nn timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c38_tfg_nn$nn.json 2> $O/r06_c38_tfg.err
  python -c "import json;d=json.loads(open('$O/r06_c38_tfg_nn$nn.json').read().strip().splitlines()[-1]);print('tfgridnet GEMM_NN=$nn:', d['ms_per_step'], d['value'], d.get('peak_mem_GB'), d['roofline']['kernel_ms_per_step'])"
done
