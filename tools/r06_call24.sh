#!/bin/bash
# round 6, call 24: the time view's forward (ws_lstm_fwd_cluster2) with the lo term on the FP8 matrix instruction (rfmt 1)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cluster2_gpu.py -x -q -m gpu -k "cluster2" -s 2>&1 | tail -30 > gpurun_out/r06_c24_test.txt
timeout 600 python tools/r05_recur_probe.py --no-stamps > gpurun_out/r06_c24_recur_probe.txt 2>&1
cat gpurun_out/r06_c24_test.txt; head -12 gpurun_out/r06_c24_recur_probe.txt
for i in 1 2; do
  for c8 in 0 1; do
    WESEP_FUSED_F8=1 WESEP_PAIR_RF=3 WESEP_CLUSTER2_F8=$c8 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c24_bench_c8${c8}_run${i}.json 2> gpurun_out/r06_c24_err.txt
    python - <<P
import json
d=json.load(open("gpurun_out/r06_c24_bench_c8${c8}_run${i}.json"))
print("CLUSTER2_F8=${c8} (FUSED_F8=1 PAIR_RF=3) run ${i}:", d["ms_per_step"], d["value"], {k:round(v["ms_per_step"],2) for k,v in d["roofline_by_class"].items()})
P
  done
done
