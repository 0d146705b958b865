#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_tfgridnet_gpu.py tests/test_tfgridnet_blocked_gpu.py -q --tb=short -m gpu -k "rowln or tfgridnet" > gpurun_out/r03_rowln_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/r03_rowln_tests.log
timeout 600 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 > gpurun_out/r03_tfgridnet_rowln.json 2> gpurun_out/r03_tfgridnet_rowln.err; echo "bench exit $?"; cut -c1-420 gpurun_out/r03_tfgridnet_rowln.json
