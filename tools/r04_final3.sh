#!/bin/bash
# Round 4, the last call: profile set of every model at the round's last code commit (fp16 A operand of gemm_tnb, ABI v16),
# the whole -m gpu suite in one process, smoke().
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r04_prof.sh bsrnn tfgridnet dpccn convtasnet 2>&1 | grep -v '^"' | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q -rs > $O/r04_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; grep -E "passed|failed|SKIPPED" $O/r04_full_gpu_suite.log | tail -8 | cut -c1-250
timeout 200 python __graft_entry__.py smoke > $O/r04_smoke.log 2>&1
echo "== smoke exit $?"; tail -1 $O/r04_smoke.log
