#!/bin/bash
# Round 4: which ingredient of gemm_b2p disturbs packed FP32 with op_sel on src1 on the same CU?
# tools/cbench/b2p_clone.hip restates the kernel with one ingredient removable at a time; race_repro counts the victim
# launches (v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0], 2 048 per thread from inline asm) that differ from a launch
# made alone.   tools/r04_race_bisect.sh [variant ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r04_race_bisect.txt
[ -n "$APPEND" ] || : > $out
V=${VICTIMS:-o_pk_add}
for a in "${@:-b2p clone0 clone1 clone2 clone4 clone8 clone16 clone32 clone64 clone256 clone128 clone12 clone3 clone17 clone18 clone20 clone24 clone28 clone30 clone31 own15}"; do
  for v in $a; do
    echo "== --aggr $v ${DATA:+--data $DATA} ${OWNWPC:+--own-wpc $OWNWPC}" >> $out
    timeout 120 tools/cbench/race_repro --trials ${TRIALS:-6} --aggr $v --victims $V ${DATA:+--data $DATA} ${OWNWPC:+--own-wpc $OWNWPC} 2>&1 | grep -v "^# shader" >> $out
    echo "exit $?" >> $out
  done
done
cat $out
