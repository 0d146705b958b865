#!/bin/bash
# Round 4, GPU call 5: the two-level in-kernel reductions (ws_tree_sum256, per-row counters in affine_bwd); bench + kernel
# stats of the headline step; TF-GridNet step with kernel stats (what got slower than round 3's 363.5 ms?).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "gn_ or group_stats or affine" > $O/r04_c5_kernels.log 2>&1
echo "== kernel tests exit $?"; tail -5 $O/r04_c5_kernels.log | cut -c1-250
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c5_bench.json 2> $O/r04_c5_bench.err
echo "== bench exit $?"; cut -c1-330 $O/r04_c5_bench.json
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_b /tmp/prof_t
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_c5_prof_bench.json 2> /tmp/prof_b.err
cp "$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1)" $O/r04_c5_kernel_stats.csv
head -28 $O/r04_c5_kernel_stats.csv | cut -c1-140
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- python $ROOT/tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --warmup 1 > $O/r04_c5_tfgridnet_under_rocprof.json 2> /tmp/prof_t.err
cp "$(find /tmp/prof_t -name '*kernel_stats.csv' | head -1)" $O/r04_c5_tfgridnet_kernel_stats.csv
cut -c1-400 $O/r04_c5_tfgridnet_under_rocprof.json
head -32 $O/r04_c5_tfgridnet_kernel_stats.csv | cut -c1-140
cd $ROOT
timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --warmup 1 > $O/r04_c5_tfgridnet.json 2> $O/r04_c5_tfgridnet.err
echo "== tfgridnet exit $?"; cut -c1-330 $O/r04_c5_tfgridnet.json
WESEP_GATES=f32 timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --warmup 1 > $O/r04_c5_tfgridnet_f32.json 2> $O/r04_c5_tfgridnet_f32.err
echo "== tfgridnet f32 format exit $?"; cut -c1-330 $O/r04_c5_tfgridnet_f32.json
