#!/bin/bash
# same-box A/B of the headline step: pair BPTT (default) vs the 16-sequence streaming BPTT, plus the isolated timings
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 120 tools/cbench/lstm_bench --view time --rows 32 --what bwd,pair --compare 1 --iters 5 2>&1 | grep -v "^compare blk32\|^compare cluster"
for rep in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pair   ', round(d['ms_per_step'],2), 'ms', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'roofline frac', round(d['roofline']['frac'],3), 'loss', d['config'].get('final_loss_dB'))"
  WESEP_LSTM_PAIR_BWD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream ', round(d['ms_per_step'],2), 'ms', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'roofline frac', round(d['roofline']['frac'],3), 'loss', d['config'].get('final_loss_dB'))"
done
