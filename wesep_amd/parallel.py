"""Data parallelism for the pBSRNN path: one process per GPU, torch.distributed with the
"nccl" backend (= RCCL over xGMI on ROCm), gradients averaged by DistributedDataParallel's
bucketed all-reduce overlapped with backward on RCCL's own stream -- the only exchange step of
the path (wesep/bin/train.py:66-70,227-228).  Rows are independent, so
the batch is sharded by rank (seed + rank, train.py:88) and nothing else is communicated."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_distributed(backend=None):
    """env:// rendezvous (torchrun contract).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def wrap_ddp(model, local_rank=None, bucket_cap_mb=25):
    """DDP with the reference's settings (25 MB buckets; no buffers in BSRNN to broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    on_gpu = next(model.parameters()).device.type == "cuda"
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=[local_rank] if on_gpu else None, bucket_cap_mb=bucket_cap_mb,
        gradient_as_bucket_view=True)


def rank_seed(base_seed, rank):
    return base_seed + rank


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, device):
    """[value of rank 0, ..., value of rank N-1] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def comm_info():
    """What the collectives of this run actually ran on (for the scaling record)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1}
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    if info["backend"] == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            pass
    return info


def all_ranks_tensor_spread(t: torch.Tensor, device) -> float:
    """max over elements of (max over ranks - min over ranks) of a small tensor; 0.0 when every rank holds the same."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0.0
    hi, lo = t.clone().to(device), t.clone().to(device)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return float((hi - lo).abs().max().item())
