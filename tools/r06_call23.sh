#!/bin/bash
# round 6, call 23: the pair BPTT with the lo term on the FP8 matrix instruction (rfmt 3): parity test, launch times alone, step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gates_h2_gpu.py -x -q -m gpu -k "pair" -s 2>&1 | tail -30 > gpurun_out/r06_c23_test.txt
timeout 600 python tools/r05_recur_probe.py --no-stamps > gpurun_out/r06_c23_recur_probe.txt 2>&1
cat gpurun_out/r06_c23_test.txt gpurun_out/r06_c23_recur_probe.txt
for i in 1 2; do
  for rf in 2 3; do
    WESEP_FUSED_F8=1 WESEP_PAIR_RF=$rf timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c23_bench_rf${rf}_run${i}.json 2> gpurun_out/r06_c23_err.txt
    python - <<P
import json
d=json.load(open("gpurun_out/r06_c23_bench_rf${rf}_run${i}.json"))
print("PAIR_RF=${rf} (FUSED_F8=1) run ${i}:", d["ms_per_step"], d["value"], {k:round(v["ms_per_step"],2) for k,v in d["roofline_by_class"].items()})
P
  done
done
