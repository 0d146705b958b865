#!/bin/bash
# Round 3: standalone cross-stream disturbance reproducer (tools/cbench/race_repro.hip): victim instruction classes,
# aggressor ingredient classes, CU-mask control.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r03_race_repro6.txt
: > $out
run() { echo "== $*" >> $out; timeout 150 "$@" >> $out 2>&1; echo "exit $?" >> $out; }
V=o_pk_add,o_pk_mul,o_pk_fma,o_pk_fma2,o_pk_mov,a_pk_add,mf_s_n0
run tools/cbench/race_repro --trials 8 --victims $V
cat $out
