#!/bin/bash
# Round 3: XCD partition experiment -- pair BPTT on XCDs 0..3, gemm_tnb on XCDs 4..7
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 120 tools/cbench/lstm_bench --view time --rows 32 --what bwd,pair --iters 5 2>&1 | grep "pair/0\|pair/16 "
for cfg in "0 0" "1 0" "1 1" "0 1" "0 0" "1 1"; do
  set -- $cfg
  WESEP_PAIR_XCD4=$1 WESEP_TNB_XCD47=$2 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_xcd_$1$2.json 2> gpurun_out/r03_xcd.err
  python - "$1" "$2" <<'PY'
import json,sys
j=json.loads(open(f"gpurun_out/r03_xcd_{sys.argv[1]}{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("pair_xcd4",sys.argv[1],"tnb_xcd47",sys.argv[2],"ms/step %.2f"%j["ms_per_step"],"frac %.3f"%j["roofline"]["frac"],{k:round(v,2) for k,v in j["kernel_ms_per_step"].items()})
PY
done
