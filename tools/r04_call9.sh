#!/bin/bash
# Round 4, call 9: gemm_tnb16 on the fp16 MFMA instruction (2 terms; WS_TNB_F16=0 = the 3-term bf16 form) and the
# granularity of the weight-gradient launches (WESEP_TNB_WGS / WESEP_TNB_MAXSPLIT), A/B on one box.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -k "tnb or formats or b2p" > $O/r04_c9_kernels.log 2>&1
echo "== format kernel tests exit $?"; tail -3 $O/r04_c9_kernels.log
timeout 200 python tools/r04_blk_probe.py --view time > $O/r04_c9_blk_probe.txt 2>&1
echo "== probe exit $?"; grep "^tnb" $O/r04_c9_blk_probe.txt
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c9_bench_$name.json 2> $O/r04_c9_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r04_c9_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"
}
run f16 WS_TNB_F16=1
run bf16x3 WS_TNB_F16=0
run f16_wgs512 WS_TNB_F16=1 WESEP_TNB_WGS=512
run f16_wgs1024 WS_TNB_F16=1 WESEP_TNB_WGS=1024 WESEP_TNB_MAXSPLIT=128
run f16_b WS_TNB_F16=1
timeout 600 python -m pytest tests/test_bsrnn_gpu.py -q -x -s > $O/r04_c9_bsrnn.log 2>&1
echo "== bsrnn exit $?"; grep -E "est rel|trajectory|passed|failed" $O/r04_c9_bsrnn.log | cut -c1-260
for i in 1 2; do
  timeout 200 python tools/bench_convtasnet.py --steps 20 --warmup 5 > $O/r04_c9_convtasnet_$i.json 2> $O/r04_c9_convtasnet_$i.err
  echo "== convtasnet run $i exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c9_convtasnet_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"
done
