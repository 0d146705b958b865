#!/bin/bash
# Round 5, call 18: the GPU test files the rfmt 2 default touches (every pBSRNN / TF-GridNet / recurrence-kernel file) + smoke
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gates_h2_gpu.py tests/test_bptt_survival_gpu.py tests/test_cluster2_gpu.py tests/test_cluster_robustness_gpu.py tests/test_bsrnn_gpu.py tests/test_tfgridnet_gpu.py tests/test_tfgridnet_blocked_gpu.py tests/test_bsrnn_multi_gpu.py tests/test_campplus_gpu.py -q -s --durations=8 > $O/r05_rf2_gpu_files.log 2>&1
echo "== affected gpu files exit $?"; grep -E "trajectory|full-size|config 2|config 5|worst|passed|failed|Error|assert " $O/r05_rf2_gpu_files.log | cut -c1-260 | tail -30
timeout 300 python __graft_entry__.py smoke > $O/r05_smoke.log 2>&1
echo "== smoke exit $?"; tail -1 $O/r05_smoke.log
