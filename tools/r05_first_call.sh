#!/bin/bash
# What round 4 leaves for the first GPU call of the next round (nothing here is unvalidated product code: round 4's last
# commit ran the whole -m gpu suite, 348 tests, and smoke() green -- profiles/r04_full_gpu_suite.log).  Measurements that the
# round's GPU budget did not cover, cheapest first:
#   1. the SSA step's unexplained 81 ms on top of the plain joint step: kernel stats of
#      tools/bench_ssa.py --what ssa
#   2. the recipe variants at the last commit (joint through Executor.train, SSA, BSRNN_Multi: profiles/r04_ssa_multi_bench.jsonl
#      predates the fp16 A operand of gemm_tnb)
#   3. TF-GridNet with WESEP_TFG_TNB_A16=1 under the profiler (304 vs 310 ms for +8 GB: where do the other 30 ms of shorter
#      gemm_tnb launches go?)
#   4. gemm_tnb16<AF = 1> needs 80 KB of LDS, so TWO workgroups fit a CU now: WESEP_TNB_WGS=512 (64 splits x 8 column tiles) may
#      overlap one workgroup's loads with the other's MFMAs (with the 160 KB kernel it was slower: profiles/r04_ab_runs.md call 9)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ssa -- python $OLDPWD/tools/bench_ssa.py --what ssa --steps 3 --warmup 2 > $OLDPWD/$O/r05_ssa_under_rocprof.jsonl 2> /tmp/prof_ssa.err
cd $OLDPWD
cp "$(find /tmp/prof_ssa -name '*kernel_stats.csv' | head -1)" $O/r05_ssa_kernel_stats.csv 2>/dev/null
head -20 $O/r05_ssa_kernel_stats.csv | cut -c1-160
timeout 400 python tools/bench_ssa.py > $O/r05_ssa_multi.jsonl 2> $O/r05_ssa_multi.err
grep "^{" $O/r05_ssa_multi.jsonl | cut -c1-300
for v in 1 0; do
  WESEP_TFG_TNB_A16=$v timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 4 --warmup 3 > $O/r05_tfg_a16_$v.json 2> $O/r05_tfg_a16_$v.err
  python -c "import json;d=json.loads(open('$O/r05_tfg_a16_$v.json').read().strip().splitlines()[-1]);print('tfgridnet a16=$v', d['ms_per_step'], d['peak_mem_GB'], d['roofline']['kernel_ms_per_step'])"
done
for w in 256 512; do
  WESEP_TNB_WGS=$w timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_bench_wgs$w.json 2> $O/r05_bench_wgs$w.err
  python -c "import json;d=json.loads(open('$O/r05_bench_wgs$w.json').read().strip().splitlines()[-1]);print('bench WESEP_TNB_WGS=$w', d['ms_per_step'], d['value'])"
done
