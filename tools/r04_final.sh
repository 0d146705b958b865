#!/bin/bash
# Round 4, final call: the profile set of every model at ONE commit (tools/r04_prof.sh), the whole -m gpu suite in one
# process, smoke().
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r04_prof.sh bsrnn tfgridnet dpccn convtasnet 2>&1 | grep -v '^"' | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -5 $O/r04_full_gpu_suite.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $O/r04_smoke.log 2>&1
echo "== smoke exit $?"; tail -2 $O/r04_smoke.log
