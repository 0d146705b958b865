#!/bin/bash
# Round 5, call 19: the recipe-variant lines again (the joint line of the profile call read 170 ms next to bench.py --joint's
# 127.5 ms for the same model): order swapped, and rfmt 1 for comparison
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 300 python tools/bench_ssa.py --what ssa,joint,multi --steps 6 --warmup 3 > $O/r05_c19_ssa_a.jsonl 2> $O/r05_c19_ssa_a.err
grep "^{" $O/r05_c19_ssa_a.jsonl | cut -c80-260; grep -v amdgpu $O/r05_c19_ssa_a.err | head -5
timeout 300 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r05_c19_ssa_b.jsonl 2> $O/r05_c19_ssa_b.err
grep "^{" $O/r05_c19_ssa_b.jsonl | cut -c80-260; grep -v amdgpu $O/r05_c19_ssa_b.err | head -5
WESEP_PAIR_RF=1 timeout 300 python tools/bench_ssa.py --what joint,ssa,multi --steps 6 --warmup 3 > $O/r05_c19_ssa_rf1.jsonl 2> $O/r05_c19_ssa_rf1.err
grep "^{" $O/r05_c19_ssa_rf1.jsonl | cut -c80-260
