#!/bin/bash
# Round 4, GPU call 3: b2p on f16 MFMA (a_fmt 2), the deeper G prefetch of gemm_tnb16, isolated timings of the blocked GEMMs,
# bench lines (default, G depth 2, no weight gradients at all), the new engine-vs-oracle tests.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gates_h2_gpu.py -q --tb=short > $O/r04_c3_h2_kernels.log 2>&1
echo "== h2 kernel tests exit $?"; tail -4 $O/r04_c3_h2_kernels.log | cut -c1-250
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s --tb=short -k "fixture or resrnn_block or trajectory" > $O/r04_c3_bsrnn.log 2>&1
echo "== bsrnn subset exit $?"; grep -E "rel|trajectory|passed|failed|Error|assert" $O/r04_c3_bsrnn.log | cut -c1-300 | tail -10
for gd in 4 2; do
  WS_TNB_GDEPTH=$gd timeout 300 python tools/r04_blk_probe.py > $O/r04_c3_blk_probe_gd$gd.txt 2>&1
  echo "== probe gdepth $gd exit $?"; cat $O/r04_c3_blk_probe_gd$gd.txt | grep -v Warn | head -40
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c3_bench_h2.json 2> $O/r04_c3_bench_h2.err
echo "== bench h2 exit $?"; cut -c1-330 $O/r04_c3_bench_h2.json
WS_TNB_GDEPTH=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c3_bench_h2_gd2.json 2> $O/r04_c3_bench_h2_gd2.err
echo "== bench h2 gdepth 2 exit $?"; cut -c1-330 $O/r04_c3_bench_h2_gd2.json
WESEP_PROBE_SKIP_WGRAD=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c3_bench_nowgrad.json 2> $O/r04_c3_bench_nowgrad.err
echo "== bench without weight gradients exit $?"; cut -c1-330 $O/r04_c3_bench_nowgrad.json
WESEP_WGRAD_OVERLAP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c3_bench_serial.json 2> $O/r04_c3_bench_serial.err
echo "== bench single stream exit $?"; cut -c1-330 $O/r04_c3_bench_serial.json
timeout 600 python -m pytest tests/test_zzz_engine_separators_gpu.py -q --tb=short -k oracle > $O/r04_c3_engine_oracle.log 2>&1
echo "== engine vs oracle exit $?"; tail -4 $O/r04_c3_engine_oracle.log | cut -c1-250
