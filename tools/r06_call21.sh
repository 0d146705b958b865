#!/bin/bash
# round 6, call 21: the fused band forward with the lo term on the FP8 matrix instruction (hfmt 5): parity test + launch times
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cluster2_gpu.py -x -q -m gpu -k "fused_band_forward" -s 2>&1 | tail -30 > gpurun_out/r06_c21_test.txt
timeout 600 python tools/r06_band_probe.py > gpurun_out/r06_c21_band_probe.txt 2>&1
cat gpurun_out/r06_c21_test.txt gpurun_out/r06_c21_band_probe.txt
