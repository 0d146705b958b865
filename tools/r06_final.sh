#!/bin/bash
# Round 6: the profile set at .commit_for_profiles -- pBSRNN (counters, kernel stats, trace gaps, bench line with cpu_baseline),
# DPCCN, TF-GridNet, Conv-TasNet lines with their counters; the joint / SSA recipe variants
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_prof.sh bsrnn 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_joint.json 2> /dev/null
cut -c1-160 $O/r06_bench_joint.json
timeout 300 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r06_ssa_multi_bench.jsonl 2> $O/r06_ssa_multi.err
grep "^{" $O/r06_ssa_multi_bench.jsonl | cut -c1-200
bash tools/r06_prof.sh dpccn tfgridnet convtasnet 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
