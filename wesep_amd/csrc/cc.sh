#!/bin/bash
# compile one .hip for gfx950 and print its kernels' resource usage:  csrc/cc.sh lstm_bf16
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $1.hip -o _obj/$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|error|LDS Size" | sed 's/.*remark: *//'
