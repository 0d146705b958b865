#!/bin/bash
# Round 5, call 12: runtime tail on hardware (CAM++ plan, concat fusion for DPCCN / TF-GridNet) and TF-GridNet's tests with
# ws_lstm_fwd_cluster2 on by default
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_zz_engine_encoders_gpu.py tests/test_zzz_engine_separators_gpu.py -q -s -k "campplus or oracle" > $O/r05_c12_engine.log 2>&1
echo "== engine tests exit $?"; grep -E "engine CAM|passed|failed|Error|assert |rel " $O/r05_c12_engine.log | cut -c1-300 | tail -20
timeout 600 python -m pytest tests/test_tfgridnet_gpu.py tests/test_tfgridnet_blocked_gpu.py -q -s > $O/r05_c12_tfg.log 2>&1
echo "== tfgridnet tests exit $?"; grep -E "config 5|passed|failed|Error|assert " $O/r05_c12_tfg.log | cut -c1-300 | tail -10
