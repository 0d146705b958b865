#!/bin/bash
# round 6, call 22: headline step with the band forward's lo term on the FP8 MFMA off / on (two runs each, one box)
mkdir -p gpurun_out
for i in 1 2; do
  for f8 in 0 1; do
    WESEP_FUSED_F8=$f8 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c22_bench_f8_${f8}_run${i}.json 2> gpurun_out/r06_c22_err.txt
    python - <<P
import json
d=json.load(open("gpurun_out/r06_c22_bench_f8_${f8}_run${i}.json"))
print("F8=${f8} run ${i}:", d["ms_per_step"], d["value"], {k:round(v["ms_per_step"],2) for k,v in d["roofline_by_class"].items()})
P
  done
done
