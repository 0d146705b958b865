#!/bin/bash
# Round 3, GPU call 1: the pair BPTT kernel (lstm_pair.hip) -- isolated timing + device-side parity without Python,
# then its pytest, then the headline bench with and without it.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 180 tools/cbench/lstm_bench --view time --rows 32 --what bwd,cluster_bwd,pair --compare 1 --iters 5 \
  > gpurun_out/r03_lstm_bench_time.txt 2>&1
echo "== lstm_bench time: exit $?"; cat gpurun_out/r03_lstm_bench_time.txt
timeout 120 tools/cbench/lstm_bench --view time --rows 16 --what bwd,pair --compare 1 --iters 5 \
  > gpurun_out/r03_lstm_bench_time_r16.txt 2>&1
echo "== lstm_bench time r16: exit $?"; cat gpurun_out/r03_lstm_bench_time_r16.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -m gpu -k "pair or lstm_cluster" > gpurun_out/r03_pair_tests.log 2>&1
echo "== pair tests: exit $?"; tail -n 25 gpurun_out/r03_pair_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_pair.json 2> gpurun_out/r03_bench_pair.err
echo "== bench (pair): exit $?"; cat gpurun_out/r03_bench_pair.json; tail -n 3 gpurun_out/r03_bench_pair.err
WESEP_LSTM_PAIR_BWD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_nopair.json 2> gpurun_out/r03_bench_nopair.err
echo "== bench (streaming BPTT): exit $?"; cat gpurun_out/r03_bench_nopair.json; tail -n 3 gpurun_out/r03_bench_nopair.err
timeout 900 python -m pytest tests/test_bsrnn_gpu.py -q --tb=short -m gpu -x > gpurun_out/r03_bsrnn_tests.log 2>&1
echo "== bsrnn tests: exit $?"; tail -n 15 gpurun_out/r03_bsrnn_tests.log
