#!/bin/bash
# Round 4, call 12: does a high-priority main stream get CUs from a multi-wave gemm_tnb grid sooner?  (Round 3 found stream
# priorities without effect -- with ONE wave of 256 whole-duration gemm_tnb workgroups there was nothing to arbitrate.)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c12_bench_$name.json 2> $O/r04_c12_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r04_c12_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"
}
run base X=1
run prio WESEP_MAIN_PRIORITY=-1
run prio_wgs512 WESEP_MAIN_PRIORITY=-1 WESEP_TNB_WGS=512
run prio_wgs1024 WESEP_MAIN_PRIORITY=-1 WESEP_TNB_WGS=1024 WESEP_TNB_MAXSPLIT=128
run wgs192 WESEP_TNB_WGS=192
run wgs128 WESEP_TNB_WGS=128
run base2 X=1
