#!/bin/bash
# round 3: halo-tile conv3x3 (forward / dx / weight gradient): DPCCN + ResNet tests, bench lines per variant, kernel profile
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dpccn_gpu.py tests/test_resnet_gpu.py -q --tb=short -m gpu > gpurun_out/r03_halo_tests.log 2>&1; echo "tests exit $?"; tail -15 gpurun_out/r03_halo_tests.log | cut -c1-200
for v in ${VARIANTS:-0}; do
  echo "WS_CONV3X3_VARIANT=$v"
  WS_CONV3X3_VARIANT=$v timeout 600 python tools/bench_dpccn.py --rows 32 --joint --steps 3 2>/dev/null | cut -c90-200
done
tools/r03_dpccn_prof.sh 2>&1 | head -${PROF_LINES:-24}
