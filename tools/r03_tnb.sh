#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q --tb=short -m gpu -k "tnb" 2>&1 | tail -5
WESEP_TNB_OLD=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q --tb=short -m gpu -k "tnb" 2>&1 | tail -2
timeout 300 python tools/gemm_probe.py 2>&1 | grep "tnb"
run() { "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'loss', d['config']['final_loss_dB'])"; }
echo "new tnb:"; run env
echo "old tnb:"; run env WESEP_TNB_OLD=1
echo "new tnb:"; run env
