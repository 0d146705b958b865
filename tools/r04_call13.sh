#!/bin/bash
# Round 4, call 13: TF-GridNet row streams (two halves of the batch on two HIP streams) -- parity, then bench A/B.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_tfgridnet_gpu.py -q -x -s -k "row_streams or config5" > $O/r04_c13_tfg_tests.log 2>&1
echo "== tfgridnet row-stream tests exit $?"; grep -E "row streams|config 5|passed|failed|Error|assert" $O/r04_c13_tfg_tests.log | cut -c1-250 | tail -12
for v in 2 1 4; do
  WESEP_TFG_ROW_STREAMS=$v timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 4 --warmup 2 > $O/r04_c13_tfgridnet_rs$v.json 2> $O/r04_c13_tfgridnet_rs$v.err
  echo "== tfgridnet row streams=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c13_tfgridnet_rs$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['peak_mem_GB'], d['final_loss_dB'])" 2>&1)"; tail -2 $O/r04_c13_tfgridnet_rs$v.err | cut -c1-300
done
