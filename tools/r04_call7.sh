#!/bin/bash
# Round 4, GPU call 7: TF-GridNet with the strided inter-frame path (tests + bench with / without it).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_tfgridnet_blocked_gpu.py tests/test_tfgridnet_gpu.py -q --tb=short > $O/r04_c7_tfg_tests.log 2>&1
echo "== tfgridnet tests exit $?"; tail -6 $O/r04_c7_tfg_tests.log | cut -c1-250
timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --warmup 1 > $O/r04_c7_tfgridnet.json 2> $O/r04_c7_tfgridnet.err
echo "== tfgridnet exit $?"; cut -c1-420 $O/r04_c7_tfgridnet.json
WESEP_TFG_STRIDED=0 timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --warmup 1 > $O/r04_c7_tfgridnet_nostrided.json 2> $O/r04_c7_tfgridnet_nostrided.err
echo "== tfgridnet (transposed copies) exit $?"; cut -c1-420 $O/r04_c7_tfgridnet_nostrided.json
