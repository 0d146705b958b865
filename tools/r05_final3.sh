#!/bin/bash
# Round 5, the call after rfmt 2 became the default: profile set of the lines whose kernels changed (pBSRNN, TF-GridNet, the
# SSA / joint recipe variants) at .commit_for_profiles, and the recurrence step budget at that commit.  DPCCN / Conv-TasNet run
# no kernel that changed since their profile set (9dd7187).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r05_prof.sh bsrnn 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
timeout 200 python tools/r05_recur_probe.py > $O/r05_recurrence_step_budget.txt 2>&1
echo "== probe exit $?"; grep -E "^pair BPTT|^cluster" $O/r05_recurrence_step_budget.txt
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_joint.json 2> /dev/null
cut -c1-160 $O/r05_bench_joint.json
timeout 300 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r05_ssa_multi_bench.jsonl 2> $O/r05_ssa_multi.err
grep "^{" $O/r05_ssa_multi_bench.jsonl | cut -c1-200
bash tools/r05_prof.sh tfgridnet 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
