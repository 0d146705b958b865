#!/bin/bash
# Round 4, last call: the profile set of every model again at the round's last code commit (the `ready` gate of the deferred
# weight-gradient jobs restored), then the aggressor bisect of the packed-FP32 disturbance.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
bash tools/r04_prof.sh bsrnn tfgridnet dpccn convtasnet 2>&1 | grep -v '^"' | cut -c1-400
timeout 200 python -m pytest tests/test_bsrnn_gpu.py -q -x -k "side_stream or training_step_matches or resrnn_block" > gpurun_out/r04_f2_bsrnn_quick.log 2>&1
echo "== bsrnn quick exit $?"; tail -2 gpurun_out/r04_f2_bsrnn_quick.log
bash tools/r04_race_bisect.sh > /dev/null 2>&1
grep -E "^==|differ|mismatch|launches|exit" gpurun_out/r04_race_bisect.txt | cut -c1-200 | head -120
