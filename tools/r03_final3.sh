#!/bin/bash
# Round 3, last validation at HEAD (the GPU budget no longer covers the full suite: the files of the changed paths ran in
# r03_halo.sh / r03_tfg2.sh): smoke(), the default bench line, the DPCCN and TF-GridNet bench lines with cpu_baseline.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1; echo "== smoke exit $?"; tail -2 gpurun_out/r03_smoke.log
timeout 300 python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "== bench exit $?"; cut -c1-700 gpurun_out/r03_bench_final.json
timeout 300 python tools/bench_dpccn.py --rows 32 --joint --steps 3 --cpu > gpurun_out/r03_dpccn_bench.json 2> gpurun_out/r03_dpccn.err; echo "== dpccn exit $?"; cut -c1-330 gpurun_out/r03_dpccn_bench.json
timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --cpu > gpurun_out/r03_tfgridnet_bench.json 2> gpurun_out/r03_tfgridnet.err; echo "== tfgridnet exit $?"; cut -c1-420 gpurun_out/r03_tfgridnet_bench.json
