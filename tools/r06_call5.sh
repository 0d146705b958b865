#!/bin/bash
# Round 6, call 5 (DPCCN at BASELINE config 3): split-lane slab reductions + the rounded-down split count of the halo weight
# gradient: bench, kernel stats, and the launches of ws_conv3x3_wgrad by grid (which layers cost what)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "transpose_and_reduce" > $O/r06_c5_reduce.log 2>&1
echo "== reduce test exit $?"; tail -2 $O/r06_c5_reduce.log
timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c5_dpccn_bench.json 2> $O/r06_c5_dpccn.err
echo "== dpccn bench exit $?"; cut -c1-300 $O/r06_c5_dpccn_bench.json
cd /tmp; rm -rf /tmp/prof_c5
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $ROOT/tools/bench_dpccn.py --rows 32 --joint --steps 3 --warmup 1 > $O/r06_c5_dpccn_under_rocprof.json 2> /tmp/prof_c5.err
echo "== rocprof exit $?"
cp "$(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1)" $O/r06_c5_dpccn_kernel_stats.csv
head -24 $O/r06_c5_dpccn_kernel_stats.csv | cut -c1-150
TR="$(find /tmp/prof_c5 -name '*kernel_trace.csv' | head -1)"
python - "$TR" > $O/r06_c5_wgrad_by_grid.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "conv3x3_wgrad" in n or "conv3x3_kernel" in n or "in_act_sums" in n or "reduce_slabs" in n:
        key = (n.split("(")[0][:40], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:60]:
    print(f"{k[0]:42s} grid {k[1]:>8s} {k[2]:>4s} {k[3]:>3s} wg {k[4]:>5s}  n {len(v):4d}  mean {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:8.2f} ms")
PY
head -45 $O/r06_c5_wgrad_by_grid.txt | cut -c1-170
cd $ROOT
timeout 900 python -m pytest tests/test_dpccn_gpu.py -q -x > $O/r06_c5_dpccn_tests.log 2>&1
echo "== dpccn tests exit $?"; tail -3 $O/r06_c5_dpccn_tests.log
