#!/bin/bash
# Round 4, call 14: the recipe variants around the headline step re-measured with the round's code (joint ResNet34, SSA,
# BSRNN_Multi), and the cross-stream test file with the new documentary test of the aggressor pair.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 300 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_c14_bench_joint.json 2> $O/r04_c14_bench_joint.err
echo "== bench --joint exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c14_bench_joint.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"
timeout 400 python tools/bench_ssa.py > $O/r04_c14_ssa_multi.jsonl 2> $O/r04_c14_ssa_multi.err
echo "== bench_ssa exit $?"; python - <<PY
import json
for l in open("$O/r04_c14_ssa_multi.jsonl"):
    l=l.strip()
    if l.startswith("{"):
        d=json.loads(l); print({k:d[k] for k in d if k in ("what","variant","metric","ms_per_step","value","peak_mem_GB")})
PY
timeout 300 python -m pytest tests/test_cross_stream_gpu.py -q -rsx > $O/r04_c14_cross_stream.log 2>&1
echo "== cross-stream tests exit $?"; tail -6 $O/r04_c14_cross_stream.log | cut -c1-200
