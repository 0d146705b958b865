#!/bin/bash
# Round 3, GPU call 2: after removing every register soffset from the 16-byte buffer stores (store-data hazard):
# recurrence timings, kernel tests, the pBSRNN model tests, the trajectory test, bench lines.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
for view in time band; do
  timeout 180 tools/cbench/lstm_bench --view $view --rows 32 --what fwd,bwd,fused,cluster,cluster_bwd,pair --compare 1 --iters 5 \
    > gpurun_out/r03_lstm_bench_${view}.txt 2>&1
  echo "== lstm_bench $view: exit $?"; cat gpurun_out/r03_lstm_bench_${view}.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_cluster_robustness_gpu.py -q --tb=short -m gpu -x > gpurun_out/r03_kernel_tests.log 2>&1
echo "== kernel tests: exit $?"; tail -n 12 gpurun_out/r03_kernel_tests.log
timeout 1200 python -m pytest tests/test_bsrnn_gpu.py -q --tb=short -m gpu -s > gpurun_out/r03_bsrnn_tests.log 2>&1
echo "== bsrnn tests: exit $?"; grep -a "trajectory\|passed\|failed\|Error\|assert" gpurun_out/r03_bsrnn_tests.log | tail -n 25
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_pair.json 2> gpurun_out/r03_bench_pair.err
echo "== bench (pair): exit $?"; cut -c1-400 gpurun_out/r03_bench_pair.json; grep -o '"kernel_ms_per_step": {[^}]*}' gpurun_out/r03_bench_pair.json
