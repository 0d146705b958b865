#!/bin/bash
# Round 6: pBSRNN profile set again at .commit_for_profiles (after the carrier reordering / tail flush), the recipe variants, and
# the whole -m gpu suite + smoke() at the same code
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
bash tools/r06_prof.sh bsrnn 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_joint.json 2> /dev/null
cut -c1-160 $O/r06_bench_joint.json
timeout 300 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r06_ssa_multi_bench.jsonl 2> $O/r06_ssa_multi.err
grep "^{" $O/r06_ssa_multi_bench.jsonl | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/r06_full_gpu_suite.log 2>&1
echo "== full gpu suite exit $?"; tail -18 $O/r06_full_gpu_suite.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1
echo "== smoke exit $?"; tail -2 $O/r06_smoke.log | cut -c1-300
