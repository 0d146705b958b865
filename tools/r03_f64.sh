#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 200 tools/cbench/lstm_bench --view band --rows 32 --what fused --iters 5 2>&1 | tail -3
timeout 200 tools/cbench/lstm_bench --view band --rows 16 --what fused --iters 5 2>&1 | tail -3
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bsrnn_gpu.py -q --tb=short -m gpu -k "fused or lstm or resrnn or training or baseline or fixture" > gpurun_out/r03_f64_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r03_f64_tests.log
for i in 1 2; do
WS_FUSED_SEQS=32 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('32-seq', j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('auto  ', j['ms_per_step'], j['kernel_ms_per_step'])"
done
