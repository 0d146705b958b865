mkdir -p gpurun_out
cd /root/repo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700 | tee gpurun_out/torchrun1.log
