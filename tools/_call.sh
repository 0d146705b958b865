mkdir -p gpurun_out
cd /root/repo
timeout 2400 python -X faulthandler -m pytest tests/ -x -q -m gpu > gpurun_out/full_gpu_suite_raw.log 2>&1
echo "rc=$?" >> gpurun_out/full_gpu_suite_raw.log
grep -v amdgpu.ids gpurun_out/full_gpu_suite_raw.log | grep -E "passed|failed|Fatal|rc=|FAILED|Error" | head -20 > gpurun_out/full_gpu_suite.log
tail -c 20000 gpurun_out/full_gpu_suite_raw.log > gpurun_out/full_gpu_suite_tail.log; rm gpurun_out/full_gpu_suite_raw.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 >> gpurun_out/full_gpu_suite.log
timeout 600 python tools/bench_tfgridnet.py --rows 8 --recipe --steps 2 --cpu 2>&1 | grep '^{' > gpurun_out/r02_tfgridnet_bench.json
timeout 300 python tools/bench_convtasnet.py 2>&1 | grep '^{' > gpurun_out/r02_convtasnet_bench.json
