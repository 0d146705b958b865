#!/bin/bash
# round 6, call 26: the band view's streaming BPTT with the lo term on the FP8 matrix instruction (rfmt 3): parity, alone, in the step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gates_h2_gpu.py -x -q -m gpu -k "blk32 or stream or h2f" -s 2>&1 | grep -i "rfmt\|passed\|failed\|error" | tail -20 > gpurun_out/r06_c26_test.txt
timeout 600 python tools/r06_band_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_c26_band_probe.txt
cat gpurun_out/r06_c26_test.txt gpurun_out/r06_c26_band_probe.txt
for i in 1 2; do
  for rf in 2 3; do
    WESEP_FUSED_F8=1 WESEP_PAIR_RF=3 WESEP_BAND_RF=$rf timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c26_bench_brf${rf}_run${i}.json 2> gpurun_out/r06_c26_err.txt
    python - <<P
import json
d=json.load(open("gpurun_out/r06_c26_bench_brf${rf}_run${i}.json"))
print("BAND_RF=${rf} (FUSED_F8=1 PAIR_RF=3) run ${i}:", d["ms_per_step"], d["value"], {k:round(v["ms_per_step"],2) for k,v in d["roofline_by_class"].items()})
P
  done
done
