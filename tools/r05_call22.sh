#!/bin/bash
# Round 5, call 22: every -m gpu file that calls 18 and 21 did not cover, at the round's last commit (ABI v18 library + runtime)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 150 python -m pytest tests -m gpu -q --deselect tests/test_bsrnn_gpu.py --deselect tests/test_tfgridnet_gpu.py --deselect tests/test_tfgridnet_blocked_gpu.py --deselect tests/test_bsrnn_multi_gpu.py --deselect tests/test_gates_h2_gpu.py --deselect tests/test_kernels_gpu.py --deselect tests/test_bptt_survival_gpu.py --deselect tests/test_cluster_robustness_gpu.py --deselect tests/test_cluster2_gpu.py --deselect tests/test_campplus_gpu.py > $O/r05_c22_other_gpu_files.log 2>&1
echo "== other gpu files exit $?"; tail -3 $O/r05_c22_other_gpu_files.log | cut -c1-200
