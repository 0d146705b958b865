mkdir -p gpurun_out
cd /root/repo
timeout 2400 python -X faulthandler -m pytest tests/ -x -q -m gpu -v > gpurun_out/full_gpu_suite_raw.log 2>&1
echo "rc=$?" >> gpurun_out/full_gpu_suite_raw.log
grep -v amdgpu.ids gpurun_out/full_gpu_suite_raw.log | grep -n "Fatal\|Segmentation\|Current thread\|File \"/root/repo\|Aborted\|HSA\|hip\|rc=" | head -60 > gpurun_out/full_gpu_suite.log
grep -E "PASSED|FAILED" gpurun_out/full_gpu_suite_raw.log | tail -5 >> gpurun_out/full_gpu_suite.log
tail -c 200000 gpurun_out/full_gpu_suite_raw.log > gpurun_out/full_gpu_suite_tail.log; rm gpurun_out/full_gpu_suite_raw.log
