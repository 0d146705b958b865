#!/bin/bash
# Round 3, GPU call: the tightened gradient checks (per-tensor rel-L2 vs oracle autograd) -- first numbers
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dpccn_gpu.py tests/test_tfgridnet_gpu.py tests/test_resnet_gpu.py -q --tb=short -m gpu -s \
   -k "fixture or config3 or config5 or resnet18_matches" > gpurun_out/r03_gradchecks.log 2>&1
echo "exit $?"; grep -a "worst\|passed\|failed\|assert\|Error" gpurun_out/r03_gradchecks.log | cut -c1-260 | tail -40
