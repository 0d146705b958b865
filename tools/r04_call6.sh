#!/bin/bash
# Round 4, GPU call 6: agent-scope single-word hand-off in the in-kernel reductions (no cache-wide fences); bench + stats.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "gn_ or group_stats or affine" > $O/r04_c6_kernels.log 2>&1
echo "== kernel tests exit $?"; tail -5 $O/r04_c6_kernels.log | cut -c1-250
timeout 300 python -m pytest tests/test_bsrnn_gpu.py -q -s --tb=short -k "fixture or side_stream" > $O/r04_c6_bsrnn.log 2>&1
echo "== bsrnn subset exit $?"; grep -E "rel|passed|failed|Error|assert" $O/r04_c6_bsrnn.log | cut -c1-300 | tail -6
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c6_bench$i.json 2> $O/r04_c6_bench$i.err
echo "== bench exit $?"; cut -c1-330 $O/r04_c6_bench$i.json
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_b
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_c6_prof_bench.json 2> /tmp/prof_b.err
cp "$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1)" $O/r04_c6_kernel_stats.csv
grep -E "affine_bwd|gn_bwd|reduce_slabs|tnb|pair|b2p" $O/r04_c6_kernel_stats.csv | cut -c1-150
