#!/bin/bash
# Round 3: engines overlapping on one GPU (per-device lock now opt-in) -- round 2's engine_race3 survey and the engine tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 500 python tools/engine_race3.py 40 > gpurun_out/r03_engine_race3.log 2>&1; echo "exit $?"; grep -v amdgpu.ids gpurun_out/r03_engine_race3.log | tail -14
timeout 500 python -m pytest tests/test_engine_gpu.py tests/test_cross_stream_gpu.py -q --tb=short -m gpu > gpurun_out/r03_engine_tests.log 2>&1; echo "exit $?"; tail -6 gpurun_out/r03_engine_tests.log
