mkdir -p gpurun_out
cd /root/repo
timeout 600 python tools/bench_dpccn.py --rows 32 --joint --steps 3 --cpu 2>&1 | grep '^{' | tee gpurun_out/r02_dpccn_bench.json
timeout 900 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --cpu 2>&1 | grep '^{' | tee gpurun_out/r02_tfgridnet_bench.json
