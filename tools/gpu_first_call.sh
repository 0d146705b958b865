#!/bin/bash
# First GPU call of a round: run the GPU tests that were written without hardware access (fbank / BSRNN_Multi / engine / TF-GridNet blocked path),
# time the native runtime, and re-take the headline bench line.  Everything lands under gpurun_out/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_first_call.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
# seconds, no Python: isolated recurrence timings + device-side parity between the kernel families
for view in time band; do
  timeout 120 tools/cbench/lstm_bench --view $view --rows 32 --what fwd,bwd,fused,cluster,cluster_bwd --compare 1 \
    > "gpurun_out/lstm_bench_${view}.txt" 2>&1
  echo "== lstm_bench ${view}: exit $?"; cat "gpurun_out/lstm_bench_${view}.txt"
done
for t in fbank bsrnn_multi engine tfgridnet_blocked; do
  timeout 600 python -m pytest "tests/test_${t}_gpu.py" -q --tb=short -m gpu > "gpurun_out/pending_${t}.log" 2>&1
  echo "== pending ${t}: exit $?"; tail -n 15 "gpurun_out/pending_${t}.log"
done
timeout 300 python tools/bench_engine.py > gpurun_out/engine_bench.json 2> gpurun_out/engine_bench.err
echo "== engine bench: exit $?"; cat gpurun_out/engine_bench.json; tail -n 5 gpurun_out/engine_bench.err
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "== bench: exit $?"; cat gpurun_out/bench.json
for flags in "--recipe --rowmajor" "--recipe"; do
  timeout 600 python tools/bench_tfgridnet.py --rows 8 $flags > "gpurun_out/tfgridnet_${flags// /}.json" 2> gpurun_out/tfgridnet.err
  echo "== tfgridnet ${flags}: exit $?"; cat "gpurun_out/tfgridnet_${flags// /}.json"
done
