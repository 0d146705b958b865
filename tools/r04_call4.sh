#!/bin/bash
# Round 4, GPU call 4: last-workgroup reductions + scratch-free fused64 + scaled-unit BPTT on hardware; bench; the profile
# set of the headline step at this commit (tools/r04_prof.sh bsrnn); TF-GridNet step time / memory with the 2-byte formats.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gates_h2_gpu.py tests/test_bptt_survival_gpu.py -q --tb=short -k "gn_ or group_stats or affine or gates or bptt or pair or resrnn or fused or forward or timeout or nonfinite or inplace" > $O/r04_c4_kernels.log 2>&1
echo "== kernel tests exit $?"; tail -6 $O/r04_c4_kernels.log | cut -c1-250
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -s --tb=short -k "fixture or resrnn_block or trajectory or side_stream or uninitialised" > $O/r04_c4_bsrnn.log 2>&1
echo "== bsrnn subset exit $?"; grep -E "rel|trajectory|passed|failed|Error|assert" $O/r04_c4_bsrnn.log | cut -c1-300 | tail -10
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c4_bench.json 2> $O/r04_c4_bench.err
echo "== bench exit $?"; cut -c1-330 $O/r04_c4_bench.json
timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r04_c4_tfgridnet.json 2> $O/r04_c4_tfgridnet.err
echo "== tfgridnet exit $?"; cut -c1-500 $O/r04_c4_tfgridnet.json
bash tools/r04_prof.sh bsrnn
