#!/bin/bash
# Round 4, GPU call 2: WS_GATES_H2F (scaled-fp16 d(gates), the new default) on hardware -- kernel tests, the assembled
# pBSRNN parity tests with their measured errors, the TF-GridNet / BSRNN_Multi paths that share the kernels, the bench A/B
# (h2 = H2F against h2b = bf16 d(gates)) and the kernel stats of the default.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gates_h2_gpu.py tests/test_bptt_survival_gpu.py -q --tb=short -s > $O/r04_c2_h2_kernels.log 2>&1
echo "== h2 kernel + survival tests exit $?"; grep -E "backward at|passed|failed|^E  |FAILED" $O/r04_c2_h2_kernels.log | cut -c1-250 | tail -25
timeout 1200 python -m pytest tests/test_bsrnn_gpu.py -q -s --tb=short > $O/r04_c2_bsrnn_h2f.log 2>&1
echo "== bsrnn (h2 = H2F) exit $?"; grep -E "rel|trajectory|passed|failed|Error|assert" $O/r04_c2_bsrnn_h2f.log | cut -c1-300 | tail -24
timeout 900 python -m pytest tests/test_tfgridnet_blocked_gpu.py tests/test_bsrnn_multi_gpu.py tests/test_kernels_gpu.py -q --tb=short -k "not generic and not nt_ and not tn_" > $O/r04_c2_others.log 2>&1
echo "== tfgridnet blocked + multi + kernels exit $?"; tail -6 $O/r04_c2_others.log | cut -c1-250
for f in h2 h2b; do
  WESEP_GATES=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c2_bench_$f.json 2> $O/r04_c2_bench_$f.err
  echo "== bench $f exit $?"; cut -c1-330 $O/r04_c2_bench_$f.json
done
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_r04
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04 -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_c2_prof_bench.json 2> $O/r04_c2_prof.err
echo "rocprof exit $?"
cp "$(find /tmp/prof_r04 -name '*kernel_stats.csv' | head -1)" $O/r04_c2_kernel_stats.csv
python $ROOT/tools/trace_gaps.py "$(find /tmp/prof_r04 -name '*kernel_trace.csv' | head -1)" --steps 4 > $O/r04_c2_trace_gaps.txt 2>&1
head -24 $O/r04_c2_kernel_stats.csv | cut -c1-150
grep -E "^queue|busy" $O/r04_c2_trace_gaps.txt | head
