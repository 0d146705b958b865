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
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv gpurun_out/r02_pmc_traffic.json 437959b424124464b4adb0bb01ef084ba17f5701 7dee508971cb5855 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma.csv gpurun_out/r02_pmc_mfma.json 437959b424124464b4adb0bb01ef084ba17f5701 7dee508971cb5855 "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline" > /dev/null
find gpurun_out -name '*.csv' -size +6M -delete
timeout 300 python bench.py --joint --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r02_bench_joint.json
timeout 400 python tools/bench_dpccn.py --rows 32 --joint --steps 3 --cpu 2>&1 | grep '^{' > gpurun_out/r02_dpccn_bench.json
python -c "
import json
for f in ('r02_bench_r32','r02_bench_joint','r02_dpccn_bench'):
    j=json.load(open('gpurun_out/'+f+'.json')); print(f, j['ms_per_step'], j['value'])
"
