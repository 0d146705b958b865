#!/bin/bash
# round 6, call 64: weight packs prefetched on the side stream (WESEP_PACK_PREFETCH): parity tests, then A/B on one box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_bsrnn_gpu.py tests/test_bsrnn_multi_gpu.py -q -m gpu -k "prefetch or run_ahead or side_stream or training_step or trajectory or ddp or multi or fixture" > $O/r06_c64_tests.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|Error" $O/r06_c64_tests.log | tail -3
for i in 1 2; do for pf in 1 0; do
  WESEP_PACK_PREFETCH=$pf timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c64_pf${pf}_run$i.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/r06_c64_pf${pf}_run$i.json'));print('PACK_PREFETCH=$pf run $i:', round(d['ms_per_step'],2), d['memory']['device_allocs_in_timed_steps'])"
done; done
WESEP_PACK_PREFETCH=1 timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('joint pf1', round(d['ms_per_step'],2))"
WESEP_PACK_PREFETCH=0 timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('joint pf0', round(d['ms_per_step'],2))"
