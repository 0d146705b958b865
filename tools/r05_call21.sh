#!/bin/bash
# Round 5, call 21: after the streaming BPTT's source was restructured for its opt-in variant (default path: the same instruction
# stream up to register names) -- the recurrence-kernel files, the quick pBSRNN tests and smoke once more
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gates_h2_gpu.py tests/test_kernels_gpu.py tests/test_bptt_survival_gpu.py tests/test_cluster_robustness_gpu.py -q > $O/r05_c21_kernels.log 2>&1
echo "== kernel files exit $?"; tail -2 $O/r05_c21_kernels.log | cut -c1-200
timeout 300 python -m pytest tests/test_bsrnn_gpu.py -q -s -k "trajectory or training_step or resrnn_block or reference_fixture" > $O/r05_c21_bsrnn_quick.log 2>&1
echo "== pBSRNN quick tests exit $?"; grep -E "trajectory\[bf|passed|failed" $O/r05_c21_bsrnn_quick.log | cut -c1-250
timeout 300 python __graft_entry__.py smoke > $O/r05_smoke.log 2>&1
echo "== smoke exit $?"; tail -1 $O/r05_smoke.log
