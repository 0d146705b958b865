#!/bin/bash
# Round 4, call 15: fp16 copies of the weight-gradient GEMM's A operand (ABI v16: ws_gemm_p2b A_bl16, ws_gemm_b2p a16_out,
# ws_gemm_tnb a_fmt = 1) -- kernel tests, isolated timing, bench A/B on one box, the parity tests that bound the gradients.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 300 python -m pytest tests/test_gates_h2_gpu.py -q -x -k "fp16_operand or fp16_a_operand or tnb or b2p or formats" > $O/r04_c15_kernels.log 2>&1
echo "== kernel tests exit $?"; tail -3 $O/r04_c15_kernels.log | cut -c1-300
timeout 200 python tools/r04_blk_probe.py --view time > $O/r04_c15_blk_probe.txt 2>&1
echo "== probe exit $?"; grep "^tnb" $O/r04_c15_blk_probe.txt
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c15_bench_$name.json 2> $O/r04_c15_bench_$name.err
  echo "== bench $name exit $?: $(python -c "import json,sys;d=json.loads(open('$O/r04_c15_bench_$name.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])" 2>&1)"; tail -1 $O/r04_c15_bench_$name.err | cut -c1-200
}
run a16 WESEP_TNB_A16=1
run a32 WESEP_TNB_A16=0
run a16_b WESEP_TNB_A16=1
timeout 500 python -m pytest tests/test_bsrnn_gpu.py -q -x -s -k "resrnn_block or config2 or trajectory or training_step_matches" > $O/r04_c15_bsrnn.log 2>&1
echo "== bsrnn parity subset exit $?"; grep -E "est rel|trajectory|passed|failed|Error|assert " $O/r04_c15_bsrnn.log | cut -c1-260
for v in 1 0; do
  WESEP_TNB_A16=$v timeout 300 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r04_c15_tfgridnet_a16_$v.json 2> $O/r04_c15_tfgridnet_a16_$v.err
  echo "== tfgridnet a16=$v exit $?: $(python -c "import json;d=json.loads(open('$O/r04_c15_tfgridnet_a16_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['peak_mem_GB'])" 2>&1)"
done
