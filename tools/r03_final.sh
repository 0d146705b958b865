#!/bin/bash
# Round 3 validation at HEAD on one MI355X: full `-m gpu` suite, smoke(), the default bench line (with cpu_baseline),
# and the non-headline separators' bench lines (no profiler attached).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -q -m gpu -x --tb=short > gpurun_out/r03_full_gpu_suite.log 2>&1
echo "== suite exit $?"; tail -6 gpurun_out/r03_full_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1; echo "== smoke exit $?"; tail -2 gpurun_out/r03_smoke.log
timeout 600 python bench.py > gpurun_out/r03_bench_r32.json 2> gpurun_out/r03_bench_r32.err; echo "== bench exit $?"; cut -c1-600 gpurun_out/r03_bench_r32.json
timeout 600 python tools/bench_dpccn.py --rows 32 --joint --steps 3 --cpu > gpurun_out/r03_dpccn_bench.json 2> gpurun_out/r03_dpccn.err; echo "== dpccn exit $?"; cut -c1-330 gpurun_out/r03_dpccn_bench.json
timeout 600 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --cpu > gpurun_out/r03_tfgridnet_bench.json 2> gpurun_out/r03_tfgridnet.err; echo "== tfgridnet exit $?"; cut -c1-420 gpurun_out/r03_tfgridnet_bench.json
timeout 300 python tools/bench_convtasnet.py > gpurun_out/r03_convtasnet_bench.json 2> gpurun_out/r03_convtasnet.err; echo "== convtasnet exit $?"; cut -c1-330 gpurun_out/r03_convtasnet_bench.json
