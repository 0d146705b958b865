#!/bin/bash
# Round 5, final call: the whole -m gpu suite in one process + smoke(), then the profile set of every bench line at this commit
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
bash tools/r05_full_suite.sh
bash tools/r05_prof.sh bsrnn tfgridnet dpccn convtasnet 2>&1 | grep -vE "^\s+void|^\s+[a-z_]+_kernel|^\"" | tail -60
timeout 400 python tools/bench_ssa.py --what joint,ssa,multi > gpurun_out/r05_ssa_multi_bench.jsonl 2> gpurun_out/r05_ssa_multi.err
grep "^{" gpurun_out/r05_ssa_multi_bench.jsonl | cut -c1-200
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_joint.json 2> /dev/null
cut -c1-160 gpurun_out/r05_bench_joint.json
