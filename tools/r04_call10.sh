#!/bin/bash
# Round 4, call 10: TF-GridNet with the BLSTMs' weight gradients deferred to the side stream (released under the inter-frame
# BPTTs, delivered by functional.WGradCarrierFn) -- parity tests, then the bench A/B on one box.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_tfgridnet_gpu.py -q -x > $O/r04_c10_tfg_tests.log 2>&1
echo "== tfgridnet tests exit $?"; tail -4 $O/r04_c10_tfg_tests.log
for v in 1 0; do
  WESEP_WGRAD_OVERLAP=$v timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r04_c10_tfgridnet_overlap$v.json 2> $O/r04_c10_tfgridnet_overlap$v.err
  echo "== tfgridnet overlap=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c10_tfgridnet_overlap$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['peak_mem_GB'], d['final_loss_dB'])" 2>&1)"
done
