mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd /root/repo
timeout 400 python bench.py --steps 5 --warmup 2 2>&1 | grep '^{' > gpurun_out/r02_bench_r32.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_prof.log 2>&1
find gpurun_out/r02_prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r02_bench_r32_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  find gpurun_out/pmc_$c -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} gpurun_out/pmc_$c.csv
done
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mfma.log 2>&1
find gpurun_out/pmc_mfma -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} gpurun_out/pmc_mfma.csv
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv gpurun_out/r02_pmc_traffic.json "$(cat gpurun_out/../.commit 2>/dev/null)" "" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma.csv gpurun_out/r02_pmc_mfma.json "" "" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
find gpurun_out -name '*.csv' -size +6M -delete
timeout 300 python bench.py --joint --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r02_bench_joint.json
timeout 300 python tools/bench_convtasnet.py 2>&1 | grep '^{' > gpurun_out/r02_convtasnet_bench.json
ls -la gpurun_out/*.json
