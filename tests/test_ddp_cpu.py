"""CPU, world_size 2, gloo: the N>1 logic of the path (wesep_amd/parallel.py) -- env rendezvous,
rank-sharded synthetic rows, DDP gradient averaging over the reference's parameter structure,
max-over-ranks timing.  The compute stand-in is the CPU oracle wrapped as an nn.Module (the product
model has no CPU path by design); on the GPU box the same wrap_ddp() carries the HIP model over RCCL."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleNet(torch.nn.Module):
    def __init__(self, cfg, params):
        super().__init__()
        from oracle import bsrnn_oracle as O
        self.O, self.cfg = O, cfg
        self.names = list(params)
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in params.values()])

    def forward(self, wav, emb):
        p = dict(zip(self.names, self.ps))
        return self.O.bsrnn_forward(p, self.cfg, wav, emb)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import bsrnn_oracle as O
    from wesep_amd.parallel import (all_ranks, all_ranks_tensor_spread, barrier, comm_info, init_distributed,
                                    max_over_ranks, rank_seed, wrap_ddp)
    from wesep_amd.utils.synthetic import synth_batch
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.BSRNNConfig(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
    params = O.synth_params(cfg, 3)
    net = OracleNet(cfg, params)
    ddp = wrap_ddp(net, lr)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    wav, tgt, emb = synth_batch(2, 1024, rank_seed(42, rank))
    loss = O.sisdr_loss(ddp(wav, emb), tgt)
    loss.backward()
    barrier()
    # local (un-averaged) gradient of this rank's shard, for the cross-check in the parent
    local = OracleNet(cfg, params)
    O.sisdr_loss(local(wav, emb), tgt).backward()
    t = max_over_ranks(1.0 + rank, torch.device("cpu"))
    # bench.py's scaling-record helpers: per-rank values in rank order, replica consistency, what the run ran on
    per_rank = all_ranks(10.0 + rank, torch.device("cpu"))
    same = all_ranks_tensor_spread(torch.stack([p.grad.double().sum() for p in net.ps]), torch.device("cpu"))
    differ = all_ranks_tensor_spread(torch.tensor([float(rank), 1.0], dtype=torch.float64), torch.device("cpu"))
    torch.save({"ddp": [p.grad.clone() for p in net.ps], "local": [p.grad.clone() for p in local.ps],
                "wav": wav, "tmax": t, "per_rank": per_rank, "same": same, "differ": differ, "comm": comm_info()},
               os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_world2_gloo_gradient_average(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{i}.pt") for i in range(2))
    assert not torch.equal(r0["wav"], r1["wav"])          # rank-sharded rows (seed + rank)
    assert r0["tmax"] == r1["tmax"] == 2.0                # max over ranks
    assert r0["per_rank"] == r1["per_rank"] == [10.0, 11.0]
    assert r0["same"] == 0.0 and r0["differ"] == 1.0      # averaged gradients identical on both replicas; a rank id is not
    assert r0["comm"] == {"backend": "gloo", "world_size": 2}
    for g0, g1, l0, l1 in zip(r0["ddp"], r1["ddp"], r0["local"], r1["local"]):
        assert torch.allclose(g0, g1, rtol=0, atol=0)     # replicas hold the same averaged gradient
        assert torch.allclose(g0, 0.5 * (l0 + l1), rtol=1e-5, atol=1e-7)


def test_single_process_helpers_are_noops():
    from wesep_amd.parallel import all_ranks, all_ranks_tensor_spread, barrier, comm_info, env_rank, max_over_ranks, wrap_ddp
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert env_rank() == (0, 0, 1)
    m = torch.nn.Linear(2, 2)
    assert wrap_ddp(m) is m
    barrier()
    assert max_over_ranks(3.5, torch.device("cpu")) == 3.5
    assert all_ranks(3.5, torch.device("cpu")) == [3.5] and all_ranks_tensor_spread(torch.ones(3), torch.device("cpu")) == 0.0
    assert comm_info()["world_size"] == 1
