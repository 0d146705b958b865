#!/bin/bash
# round 6, call 29: occupancy fixes (launch bounds' second number = waves per SIMD): ws_gemm_b2p a_fmt 2 at 128 registers, the 128 x 128
# gemm_nt_bf16 tile at 162 -- all four models' lines
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out
mkdir -p $O
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c29_bench_run$i.json 2> $O/r06_c29_bench.err
  python -c "import json;d=json.load(open('$O/r06_c29_bench_run$i.json'));print('bsrnn run $i:', d['ms_per_step'], d['value'], {k:round(v['ms_per_step'],2) for k,v in d['roofline_by_class'].items()})"
done
timeout 500 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c29_dpccn.json 2> $O/r06_c29_dpccn.err
python -c "import json;d=json.loads(open('$O/r06_c29_dpccn.json').read().strip().splitlines()[-1]);print('dpccn:', d['ms_per_step'], d['value'])"
timeout 500 python tools/bench_convtasnet.py --steps 20 --warmup 5 > $O/r06_c29_convtasnet.json 2> $O/r06_c29_convtasnet.err
python -c "import json;d=json.loads(open('$O/r06_c29_convtasnet.json').read().strip().splitlines()[-1]);print('convtasnet:', d['ms_per_step'], d['value'])"
timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c29_tfg.json 2> $O/r06_c29_tfg.err
python -c "import json;d=json.loads(open('$O/r06_c29_tfg.json').read().strip().splitlines()[-1]);print('tfgridnet:', d['ms_per_step'], d['value'], d.get('peak_mem_GB'), d['roofline']['kernel_ms_per_step'])"
