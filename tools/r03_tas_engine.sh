#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -q --tb=short -m gpu -k "convtasnet" > gpurun_out/r03_tas_engine.log 2>&1; echo "exit $?"; tail -30 gpurun_out/r03_tas_engine.log
