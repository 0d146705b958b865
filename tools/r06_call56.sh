#!/bin/bash
# round 6, call 56: the run-ahead regression test + the backward-path suites at the hold default
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_bsrnn_gpu.py tests/test_tfgridnet_gpu.py tests/test_tfgridnet_blocked_gpu.py tests/test_bsrnn_multi_gpu.py -q -m gpu --durations=5 > $O/r06_c56_tests.log 2>&1
echo "exit $?"; grep -E "passed|failed|Error|error" $O/r06_c56_tests.log | tail -5
