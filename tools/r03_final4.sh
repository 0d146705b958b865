#!/bin/bash
# Round 3: the `-m gpu` files the last kernel changes did not already run (r03_halo.sh: dpccn, resnet; r03_tfg2.sh: tfgridnet)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout ${LIMIT:-270} python -m pytest tests/test_convtasnet_gpu.py tests/test_kernels_gpu.py tests/test_bsrnn_gpu.py tests/test_ecapa_gpu.py tests/test_campplus_gpu.py tests/test_tfgridnet_blocked_gpu.py tests/test_engine_gpu.py tests/test_fbank_gpu.py tests/test_bsrnn_multi_gpu.py tests/test_cross_stream_gpu.py tests/test_cluster_robustness_gpu.py tests/test_zz_engine_encoders_gpu.py -q -m gpu -x --tb=short --durations=8 > gpurun_out/r03_subset_gpu_suite.log 2>&1
echo "== suite exit $?"; tail -16 gpurun_out/r03_subset_gpu_suite.log | cut -c1-200
