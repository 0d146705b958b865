#!/bin/bash
# round 6, call 46: no stream-synchronising scalar uploads in the models' forward (torch.tensor(0.0, device=...) -> torch.zeros): host
# probe + all four lines
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out
mkdir -p $O
python tools/r06_host_probe.py --steps 10 2>&1 | grep -v amdgpu.ids | tee $O/r06_c46_host_probe.txt
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_c46_bench_run$i.json 2> $O/r06_c46_bench.err
  python -c "import json;d=json.load(open('$O/r06_c46_bench_run$i.json'));print('bsrnn run $i:', d['ms_per_step'], d['value'])"
done
timeout 500 python tools/bench_convtasnet.py --steps 20 --warmup 5 > $O/r06_c46_convtasnet.json 2> $O/r06_c46_convtasnet.err
python -c "import json;d=json.loads(open('$O/r06_c46_convtasnet.json').read().strip().splitlines()[-1]);print('convtasnet:', d['ms_per_step'], d['value'])"
timeout 500 python tools/bench_dpccn.py --rows 32 --joint --steps 5 --warmup 2 > $O/r06_c46_dpccn.json 2> $O/r06_c46_dpccn.err
python -c "import json;d=json.loads(open('$O/r06_c46_dpccn.json').read().strip().splitlines()[-1]);print('dpccn:', d['ms_per_step'], d['value'])"
timeout 500 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 3 --warmup 1 > $O/r06_c46_tfg.json 2> $O/r06_c46_tfg.err
python -c "import json;d=json.loads(open('$O/r06_c46_tfg.json').read().strip().splitlines()[-1]);print('tfgridnet:', d['ms_per_step'], d['value'])"
