#!/bin/bash
# Round 3: the non-headline separators at HEAD -- bench line + rocprofv3 per-kernel stats of the same command
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
prof() {  # tag, command...
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- "$@" > $ROOT/gpurun_out/r03_${tag}_bench.json 2> $ROOT/gpurun_out/r03_${tag}.err
  echo "$tag exit $?"
  cp "$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03_${tag}_kernel_stats.csv
  tail -1 $ROOT/gpurun_out/r03_${tag}_bench.json | cut -c1-400
  head -16 $ROOT/gpurun_out/r03_${tag}_kernel_stats.csv | cut -c1-150
}
prof dpccn python $ROOT/tools/bench_dpccn.py --rows 32 --joint --steps 3
prof tfgridnet python $ROOT/tools/bench_tfgridnet.py --rows 8 --recipe --steps 2
prof convtasnet python $ROOT/tools/bench_convtasnet.py
