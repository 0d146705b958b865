#!/bin/bash
# Round 5: the whole -m gpu suite in ONE call + smoke(), per-file durations recorded.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $O/r05_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -25 $O/r05_full_gpu_suite.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1
echo "== smoke exit $?"; tail -3 $O/r05_smoke.log | cut -c1-300
