#!/bin/bash
# Round 5, last call: the profile set of every bench line at the round's last code commit (.commit_for_profiles), the recipe
# variants, and the pBSRNN files of the -m gpu suite once more (the fall-back scratch is the only change since the full run)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
bash tools/r05_prof.sh bsrnn tfgridnet dpccn convtasnet 2>&1 | grep -E "exit|^\{|PMC traffic" | cut -c1-260
timeout 400 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r05_ssa_multi_bench.jsonl 2> $O/r05_ssa_multi.err
grep "^{" $O/r05_ssa_multi_bench.jsonl | cut -c1-200
timeout 200 python bench.py --joint --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_joint.json 2> /dev/null
cut -c1-160 $O/r05_bench_joint.json
timeout 900 python -m pytest tests/test_bsrnn_gpu.py tests/test_bsrnn_multi_gpu.py tests/test_cluster2_gpu.py tests/test_cluster_robustness_gpu.py tests/test_bptt_survival_gpu.py tests/test_tfgridnet_blocked_gpu.py -q > $O/r05_last_bsrnn_files.log 2>&1
echo "== pBSRNN / cluster files exit $?"; tail -3 $O/r05_last_bsrnn_files.log | cut -c1-200
