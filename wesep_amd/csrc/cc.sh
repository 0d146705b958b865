#!/bin/bash
# compile one .hip for gfx950 with the LIBRARY's flags (wesep_amd/build.py: no packed FP32) and print its kernels' resource
# usage:  csrc/cc.sh lstm_bf16
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -c $1.hip -o _obj/$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|error|LDS Size" | sed 's/.*remark: *//'
