#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
echo "default:"; run env
echo "side priority 1:"; run env WESEP_SIDE_PRIORITY=1
echo "side priority -1:"; run env WESEP_SIDE_PRIORITY=-1
echo "main priority -1:"; run env WESEP_MAIN_PRIORITY=-1
echo "main -1, side 1:"; run env WESEP_MAIN_PRIORITY=-1 WESEP_SIDE_PRIORITY=1
echo "default:"; run env
