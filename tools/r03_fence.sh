#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
WESEP_HIP_LIB=$PWD/wesep_amd/_pk/libwesep_hip.so timeout 300 python tools/r03_fence_dbg.py > gpurun_out/r03_fence_dbg_pk.txt 2>&1; cat gpurun_out/r03_fence_dbg_pk.txt | tail -14
timeout 300 python tools/r03_fence_dbg.py > gpurun_out/r03_fence_dbg_nopk.txt 2>&1; tail -13 gpurun_out/r03_fence_dbg_nopk.txt
WESEP_HIP_LIB=$PWD/wesep_amd/_pk/libwesep_hip.so timeout 300 python tools/kernel_race.py 20 2>&1 | grep -v "^aggressor\|^victim stage" | head -24
echo == fenced; timeout 300 python tools/kernel_race.py 20 2>&1 | grep -v "^aggressor\|^victim stage" | head -24
