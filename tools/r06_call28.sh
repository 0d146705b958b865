#!/bin/bash
# round 6, call 28: ws_gemm_b2p -- a_fmt 2 at four waves per SIMD (128 registers: two workgroups per CU instead of one) and a_fmt 3
# (the lo term on the FP8 matrix instruction): parity, alone, in the step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gates_h2_gpu.py -x -q -m gpu -k "gemm_b2p" -s 2>&1 | grep -i "a_fmt\|passed\|failed\|error" | tail -20 > gpurun_out/r06_c28_test.txt
timeout 600 python tools/r06_band_probe.py 2>&1 | grep -i "b2p\|d(xn)" > gpurun_out/r06_c28_band_probe.txt
cat gpurun_out/r06_c28_test.txt gpurun_out/r06_c28_band_probe.txt
for i in 1 2; do
  for f8 in 0 1; do
    WESEP_DXN_F8=$f8 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c28_bench_dxn${f8}_run${i}.json 2> gpurun_out/r06_c28_err.txt
    python - <<P
import json
d=json.load(open("gpurun_out/r06_c28_bench_dxn${f8}_run${i}.json"))
print("DXN_F8=${f8} run ${i}:", d["ms_per_step"], d["value"], {k:round(v["ms_per_step"],2) for k,v in d["roofline_by_class"].items()})
P
  done
done
