#!/bin/bash
# round 6, call 30: ws_gemm_b2p with LDS-direct weight stages, every format at two workgroups per CU: kernel tests, the lines
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_gates_h2_gpu.py -x -q -m gpu -k "b2p or resrnn or blocked or gemm" 2>&1 | tail -4 > $O/r06_c30_test.txt
cat $O/r06_c30_test.txt
timeout 600 python tools/r06_band_probe.py 2>&1 | grep -i "b2p" > $O/r06_c30_band_probe.txt; cat $O/r06_c30_band_probe.txt
for i in 1 2; do
  for f8 in 0 1; do
    WESEP_DXN_F8=$f8 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c30_bench_dxn${f8}_run${i}.json 2> $O/r06_c30_err.txt
    python -c "import json;d=json.load(open('$O/r06_c30_bench_dxn${f8}_run${i}.json'));print('DXN_F8=${f8} run ${i}:', d['ms_per_step'], d['value'], {k:round(v['ms_per_step'],2) for k,v in d['roofline_by_class'].items()})"
  done
done
timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c30_tfg.json 2> $O/r06_c30_tfg.err
python -c "import json;d=json.loads(open('$O/r06_c30_tfg.json').read().strip().splitlines()[-1]);print('tfgridnet:', d['ms_per_step'], d['value'], d.get('peak_mem_GB'), d['roofline']['kernel_ms_per_step'])"
