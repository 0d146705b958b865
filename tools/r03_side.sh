#!/bin/bash
# what the weight-gradient stream costs at HEAD: no weight gradients at all / everything on one stream / default
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
for cfg in "WESEP_PROBE_SKIP_WGRAD=1" "WESEP_WGRAD_OVERLAP=0" "X=1"; do
  env $cfg timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(j['ms_per_step'],2), {k:round(v,1) for k,v in j['kernel_ms_per_step'].items()})"
done
