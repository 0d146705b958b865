#!/bin/bash
# round 6, call 66: the round's HEAD against commit 2513944 (before bench.py's memory object and the pack prefetch) on ONE box,
# alternating.  Needs that commit checked out beside the tree with the built libraries copied in:
#   git worktree add -f _old 2513944 && cp wesep_amd/libwesep_hip.so _old/wesep_amd/ && cp runtime/libwesep_engine.so _old/runtime/
# (removed again after the call: git worktree remove --force _old)
cd "${GRAFT_REPO_ROOT}" || exit 1
for i in 1 2; do
  (cd _old && timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null) | python -c "import json,sys;d=json.loads(sys.stdin.read());print('old 2513944:', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('HEAD       :', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
done
