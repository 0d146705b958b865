#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
for w in 256 512 128 256 512; do
  WESEP_TNB_WGS=$w timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WESEP_TNB_WGS=$w', round(j['ms_per_step'],2), {k:round(v,1) for k,v in j['kernel_ms_per_step'].items()})"
done
