#!/bin/bash
# round 6, call 49: the mask MLP's data-gradient GEMMs on the weights as they lie (no per-step transposes): tests + A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_bsrnn_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "fixture or training_step or trajectory or mask or grouped" 2>&1 | tail -2
for i in 1 2; do for nn in 1 0; do
  WESEP_GEMM_NN=$nn timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c49_bench_nn${nn}_run$i.json 2> $O/r06_c49_bench.err
  python -c "import json;d=json.load(open('$O/r06_c49_bench_nn${nn}_run$i.json'));print('GEMM_NN=$nn run $i:', d['ms_per_step'], d['value'])"
done; done
