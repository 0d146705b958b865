#!/bin/bash
# round 6, call 63: HIP runtime knob A/B on one box: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) -- the main queue
# spends 2.4 ms per step in ~6.6 us gaps between 366 dependent launches
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out; mkdir -p $O
echo "HIP_FORCE_DEV_KERNARG in the environment: ${HIP_FORCE_DEV_KERNARG:-unset}"
python -c "import os,torch;print('after import torch:', os.environ.get('HIP_FORCE_DEV_KERNARG'))"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c63_$tag.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/r06_c63_$tag.json'));print('$tag', round(d['ms_per_step'],2))"; }
run ka1_a HIP_FORCE_DEV_KERNARG=1
run ka0_a HIP_FORCE_DEV_KERNARG=0
run ka1_b HIP_FORCE_DEV_KERNARG=1
run ka0_b HIP_FORCE_DEV_KERNARG=0
run unset_a X=1
