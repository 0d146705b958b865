#!/bin/bash
# round 6, call 61: stream-priority A/B in the regime where the host no longer limits the step (run-ahead bounded): two baselines,
# main stream at high priority, side stream at high / low priority.  10 timed steps each, one box.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c61_$tag.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/r06_c61_$tag.json'));print('$tag', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
run base1 X=1
run main_hi WESEP_MAIN_PRIORITY=-1
run side_hi WESEP_SIDE_PRIORITY=-1
run side_lo WESEP_SIDE_PRIORITY=1
run main_hi_side_lo WESEP_MAIN_PRIORITY=-1 WESEP_SIDE_PRIORITY=1
run base2 X=1
python -c "import torch;print(torch.cuda.Stream.priority_range())"
