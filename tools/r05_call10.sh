#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -k "formats_agree" > $O/r05_c10_formats.log 2>&1
echo "== formats test exit $?"; tail -2 $O/r05_c10_formats.log | cut -c1-200
timeout 200 python tools/r05_recur_probe.py > $O/r05_c10_recur_probe.txt 2>&1
echo "== probe exit $?"; sed -n 1,40p $O/r05_c10_recur_probe.txt | cut -c1-160
bash tools/r05_prof.sh bsrnn 2>&1 | tail -60
