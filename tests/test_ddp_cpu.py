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


# ----------------------------------------------------------------------------------------------------------------------
# The PRODUCT's host composition under DistributedDataParallel at world size 2: the module
# tree of wesep_amd.models.BSRNN with the blocked ResRNN (functional.ResRNNBlkFn), its weight-gradient carriers
# (functional.WGradCarrierFn: the LSTM / proj gradients reach autograd -- and DDP's bucket hooks -- through a node that
# runs at the very end of backward) and the deferred side-stream jobs, on the CPU emulation of the device entry points
# (tests/emu_dev.py + emu_blk.py + emu_bsrnn.py for the band split / mask decode; tests/emu_streams.py for streams).
# ----------------------------------------------------------------------------------------------------------------------
def _product_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WESEP_WGRAD_OVERLAP="1")
    torch.set_num_threads(2)
    import pytest as _pytest
    from oracle import bsrnn_oracle as O
    from tests import emu_blk, emu_bsrnn, emu_dev, emu_streams
    import wesep_amd.functional as f0
    from wesep_amd.models import get_model
    from wesep_amd.parallel import init_distributed, rank_seed, wrap_ddp
    from wesep_amd.utils.executor import Executor, ReplicaDivergence
    from wesep_amd.utils.synthetic import synth_batch
    mp_ = _pytest.MonkeyPatch()
    real_carrier, real_reset = f0.make_wgrad_carrier, f0.reset_deferred_wgrads
    emu_dev.install(mp_)
    emu_blk.install(mp_)
    emu_bsrnn.install(mp_, real_resrnn=True)
    mp_.setattr(f0, "make_wgrad_carrier", real_carrier)          # emu_bsrnn switches the carriers off: back on
    mp_.setattr(f0, "reset_deferred_wgrads", real_reset)
    emu_streams.install(mp_)
    mp_.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    r, lr, w = init_distributed(backend="gloo")
    cfg = O.BSRNNConfig(num_repeat=1, spk_fuse_type="multiply", multi_fuse=False)
    params = O.synth_params(cfg, 3)
    model = get_model("BSRNN")(spk_emb_dim=cfg.spk_emb_dim, sr=cfg.sr, win=cfg.win, stride=cfg.stride,
                               feature_dim=cfg.feature_dim, num_repeat=cfg.num_repeat,
                               use_spk_transform=cfg.use_spk_transform, spk_fuse_type=cfg.spk_fuse_type,
                               multi_fuse=cfg.multi_fuse, joint_training=False)
    model.load_state_dict(params, strict=True)
    model.train()
    ddp = wrap_ddp(model, lr)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    wav, tgt, emb = synth_batch(2, 2048, rank_seed(42, rank))
    carried = []
    orig_bwd = f0.WGradCarrierFn.backward

    def spy(ctx, g):
        res = orig_bwd(ctx, g)
        carried.append(sum(x is not None for x in res))
        return res
    mp_.setattr(f0.WGradCarrierFn, "backward", staticmethod(spy))
    est, _ = ddp(wav, emb)
    O.sisdr_loss(est, tgt).backward()
    named = dict(model.named_parameters())
    assert all(p.grad is not None for p in named.values())
    # this rank's own (un-averaged) gradient from the oracle, for the cross-check in the parent
    q = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    O.sisdr_loss(O.bsrnn_forward(q, cfg, wav, emb), tgt).backward()
    # the training self-check: identical replicas pass, a perturbed one is caught on every rank
    Executor._replica_check(ddp, torch.device("cpu"), "test")
    caught = False
    if rank == 1:
        with torch.no_grad():
            next(model.parameters()).view(-1)[0] += 1e-3
    try:
        Executor._replica_check(ddp, torch.device("cpu"), "test")
    except ReplicaDivergence:
        caught = True
    torch.save({"ddp": {k: p.grad.clone() for k, p in named.items()}, "local": {k: v.grad.clone() for k, v in q.items()},
                "carried": carried, "caught": caught}, os.path.join(out, f"p{rank}.pt"))
    torch.distributed.destroy_process_group()
    mp_.undo()


@pytest.mark.timeout(600)
def test_product_model_with_weight_gradient_carriers_under_ddp_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_product_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"p{i}.pt") for i in range(2))
    assert r0["carried"] == r1["carried"] == [10, 10]            # two ResRNNs, ten LSTM / proj tensors through each carrier
    assert r0["caught"] and r1["caught"]                         # a diverged replica is reported on every rank
    worst = 0.0
    for k in r0["ddp"]:
        g0, g1 = r0["ddp"][k], r1["ddp"][k]
        assert torch.equal(g0, g1), k                            # replicas hold the same averaged gradient, bit for bit
        want = 0.5 * (r0["local"][k] + r1["local"][k])           # ... = the mean of the two shards' gradients
        err = float((g0 - want).norm() / (want.norm() + 1e-30))
        worst = max(worst, err)
        assert err < 2e-3, (k, err)
    print("worst relative difference to the mean of the per-rank oracle gradients:", worst)


# ----------------------------------------------------------------------------------------------------------------------
# The widened separators under DDP: DPCCN (DenseBlockFn: one gradient per parameter from block convolutions; halo weight
# gradients with slab reductions) and TF-GridNet (QKVHeadsFn: gradients of the three projections' parameters through one
# concatenated weight) -- the same world-2 statement as above, gradients against the mean of the per-rank oracle
# gradients.
# ----------------------------------------------------------------------------------------------------------------------
def _separator_worker(rank, world, port, out, which):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import pytest as _pytest
    from oracle import bsrnn_oracle as O
    from tests import emu_dev
    from wesep_amd.models import get_model
    from wesep_amd.parallel import init_distributed, rank_seed, wrap_ddp
    from wesep_amd.utils.synthetic import synth_batch
    mp_ = _pytest.MonkeyPatch()
    emu_dev.install(mp_)
    mp_.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    r, lr, w = init_distributed(backend="gloo")
    if which == "dpccn":
        from oracle import dpccn_oracle as M
        cfg = M.DPCCNConfig(tcn_blocks=2, tcn_layers=1)
        model = get_model("DPCCN")(tcn_blocks=2, tcn_layers=1, joint_training=False)
        fwd = M.dpccn_forward
        T = 4480
    else:
        from oracle import tfgridnet_oracle as M
        kw = dict(n_layers=1, lstm_hidden_units=16, emb_dim=8, emb_ks=1, emb_hs=1, attn_n_head=2, attn_approx_qk_dim=260)
        cfg = M.TFGridNetConfig(**kw)
        model = get_model("TFGridNet")(**kw, joint_training=False)
        fwd = M.tfgridnet_forward
        T = 1280
    params = M.synth_params(cfg, 5)
    model.load_state_dict(params, strict=True)
    model.train()
    ddp = wrap_ddp(model, lr)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    wav, tgt, emb = synth_batch(2, T, rank_seed(42, rank))
    est, _ = ddp(wav, emb)
    O.sisdr_loss(est, tgt).backward()
    named = dict(model.named_parameters())
    assert all(p.grad is not None for p in named.values())
    q = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o = fwd(q, cfg, wav, emb)
    O.sisdr_loss(o[0] if isinstance(o, tuple) else o, tgt).backward()
    torch.save({"ddp": {k: p.grad.clone() for k, p in named.items()}, "local": {k: v.grad.clone() for k, v in q.items()}},
               os.path.join(out, f"s{rank}.pt"))
    torch.distributed.destroy_process_group()
    mp_.undo()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("which", ["dpccn", "tfgridnet"])
def test_widened_separators_under_ddp_world2(tmp_path, which):
    world, port = 2, _free_port()
    mp.spawn(_separator_worker, args=(world, port, str(tmp_path), which), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"s{i}.pt") for i in range(2))
    gmax = max(float(v.norm()) for v in r0["local"].values())
    for k in r0["ddp"]:
        g0, g1 = r0["ddp"][k], r1["ddp"][k]
        assert torch.equal(g0, g1), k                            # replicas hold the same averaged gradient, bit for bit
        want = 0.5 * (r0["local"][k] + r1["local"][k])
        err = float((g0 - want).norm() / (want.norm() + 1e-4 * gmax))
        assert err < 5e-3, (k, err)
