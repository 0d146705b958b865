#!/bin/bash
# round 3: fused attention heads (heads.hip) + one projection GEMM: TF-GridNet tests, bench with / without, kernel profile
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tfgridnet_gpu.py -q --tb=short -m gpu > gpurun_out/r03_tfg2_tests.log 2>&1; echo "tests exit $?"; tail -12 gpurun_out/r03_tfg2_tests.log | cut -c1-220
for f in ${FUSED:-1 0}; do
  echo "WESEP_TFG_HEADS_FUSED=$f"
  WESEP_TFG_HEADS_FUSED=$f timeout 600 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 2>/dev/null | cut -c100-330
done
tools/r03_tfg_prof.sh 2>&1 | head -${PROF_LINES:-30}
