"""Training entry with the reference's configuration schema and flow (wesep/bin/train.py:51-395):

    torchrun --standalone --nproc-per-node 8 -m wesep_amd.bin.train --config confs/bsrnn.yaml --synthetic 2000
    python -m wesep_amd.bin.train --config confs/bsrnn.yaml --exp_dir exp/run1 --num_epochs 2 --synthetic 100

Same YAML keys (`model`, `model_args`, `optimizer(_args)`, `scheduler(_args)`, `loss`, `loss_args`, `dataloader_args`,
`dataset_args`, `clip_grad`, `num_epochs`, `num_avg`, `save_epoch_interval`, `log_batch_interval`, `exp_dir`,
`model_init`, `checkpoint`, `enable_amp`, `seed`); any key can be overridden with `--key value` (nested:
`--dataloader_args.batch_size 4`) like the reference's fire-style CLI.  Per epoch: `Executor.train`, `Executor.cv`,
`checkpoint_<epoch>.pt` in `<exp_dir>/models` under the reference's saving rule; one process per GPU over RCCL
(`parallel.init_distributed`), lr set from `scheduler_args.initial_lr` as the reference does.

What differs, visibly: `optimizer: Adam` becomes `FusedClipAdam` (same update, clip + step in two launches);
the CPU datapipe is NOT part of this package (north_star keeps it in wesep) -- data comes from
  * `--synthetic N`: N collated batches per epoch shaped like `tse_collate_fn`'s output for this config (2-speaker
    mixtures of `dataset_args.chunk_len` samples, one row per target speaker; enrollment as fixed embeddings, fbank or
    raw audio according to `joint_training` / `spk_feat`), or
  * the reference's own `wesep/bin/train.py` with the three import edits of INTEGRATION.md section 2, which keeps its
    shard readers / online mixing / collate functions and drives this package's models, optimizer and Executor."""
import argparse
import logging
import os
import re
import sys

import torch
import yaml

from .. import parallel
from ..models import get_model
from ..optim import FusedClipAdam
from ..utils import schedulers
from ..utils.checkpoint import load_checkpoint, load_pretrained_model, save_checkpoint
from ..utils.executor import Executor
from ..utils.losses import parse_loss


def parse_config(argv=None):
    ap = argparse.ArgumentParser(description="wesep_amd training entry (reference: wesep/bin/train.py)")
    ap.add_argument("--config", required=True)
    ap.add_argument("--synthetic", type=int, default=0, help="batches per epoch of synthetic collated data")
    args, extra = ap.parse_known_args(argv)
    with open(args.config) as f:
        configs = yaml.safe_load(f)
    if len(extra) % 2:
        raise SystemExit(f"override arguments come in '--key value' pairs: {extra}")
    for k, v in zip(extra[0::2], extra[1::2]):
        node, path = configs, k.lstrip("-").split(".")
        for p in path[:-1]:
            node = node.setdefault(p, {})
        node[path[-1]] = yaml.safe_load(v)
    configs["config"] = args.config
    return configs, args.synthetic


def setup_logger(rank, exp_dir):
    os.makedirs(os.path.join(exp_dir, "models"), exist_ok=True)
    logger = logging.getLogger(f"wesep_amd.train.{rank}")
    logger.setLevel(logging.INFO if rank == 0 else logging.WARNING)
    fmt = logging.Formatter("[ %(levelname)s : %(asctime)s ] - %(message)s")
    for h in list(logger.handlers):          # a second train() in the same process logs to its own exp_dir
        logger.removeHandler(h)
        h.close()
    for h in (logging.StreamHandler(sys.stdout), logging.FileHandler(os.path.join(exp_dir, "train.log"))):
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


class SyntheticTseLoader:
    """Collated batches with the keys and shapes `tse_collate_fn` produces (wesep/dataset/dataset.py:206-264)."""

    def __init__(self, configs, n_batches, seed, n_classes=251):
        margs = configs["model_args"]["tse_model"]
        dargs = configs.get("dataset_args", {})
        self.n = n_batches
        self.rows = 2 * int(configs["dataloader_args"]["batch_size"])        # one row per target speaker
        self.T = int(dargs.get("chunk_len", 48000))
        self.sr = int(dargs.get("resample_rate", 16000))
        self.joint = bool(margs.get("joint_training", False))
        self.spk_feat = bool(margs.get("spk_feat", dargs.get("speaker_feat", True)))
        self.emb_dim = int(margs.get("spk_emb_dim", 256))
        self.feat_dim = int((margs.get("spk_args") or {}).get("feat_dim", 80))
        self.n_classes = int(margs.get("spksInTrain", n_classes))
        self.seed, self.epoch = seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.n

    def __iter__(self):
        from ..utils.synthetic import synth_batch
        g = torch.Generator().manual_seed(self.seed * 100003 + self.epoch)
        for i in range(self.n):
            mix, tgt, emb = synth_batch(self.rows, self.T, int(torch.randint(1 << 30, (1,), generator=g)), self.emb_dim)
            if self.joint and self.spk_feat:          # mean-normalised fbank, 4 s of 10 ms frames
                fb = torch.randn(self.rows, 398, self.feat_dim, generator=g)
                emb = fb - fb.mean(1, keepdim=True)
            elif self.joint:                          # raw enrollment audio
                emb = 0.1 * torch.randn(self.rows, 4 * self.sr, generator=g)
            yield {"wav_mix": mix, "wav_targets": tgt, "spk_embeds": emb,
                   "spk_label": torch.randint(self.n_classes, (self.rows,), generator=g), "key": [str(i)] * self.rows}


def build_dataloaders(configs, n_synthetic, rank):
    if n_synthetic > 0:
        return (SyntheticTseLoader(configs, n_synthetic, configs.get("seed", 42) + rank),
                SyntheticTseLoader(configs, max(1, n_synthetic // 10), 7919 + rank), n_synthetic,
                max(1, n_synthetic // 10))
    raise SystemExit("wesep_amd.bin.train: no data source -- pass --synthetic N.  Real data comes from the reference's "
                     "CPU datapipe: run wesep/bin/train.py with the three edits of INTEGRATION.md section 2, which "
                     "hands its DataLoader to this package's Executor and models.")


def train(configs, n_synthetic=0):
    rank, local_rank, world = parallel.init_distributed()
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    exp_dir = configs["exp_dir"]
    model_dir = os.path.join(exp_dir, "models")
    logger = setup_logger(rank, exp_dir)
    torch.manual_seed(configs.get("seed", 42) + rank)          # train.py:88

    criterion = parse_loss(configs.get("loss") or "SISDR")
    loss_args = (configs.get("loss_args", {}).get("loss_posi", [[0]]),
                 configs.get("loss_args", {}).get("loss_weight", [[1.0]]))
    margs = configs["model_args"]["tse_model"]
    multi_task = bool(margs.get("multi_task", False))

    train_loader, val_loader, epoch_iter, val_iter = build_dataloaders(configs, n_synthetic, rank)
    model = get_model(configs["model"]["tse_model"])(**margs).to(device)
    if rank == 0:
        logger.info("tse_model size: {:.2f} M".format(sum(p.numel() for p in model.parameters()) / 1e6))
    ddp_model = parallel.wrap_ddp(model, local_rank)

    oargs = dict(configs["optimizer_args"]["tse_model"])
    sargs = dict(configs["scheduler_args"]["tse_model"])
    oargs["lr"] = sargs["initial_lr"]                           # train.py:236-237
    if configs["optimizer"]["tse_model"] == "Adam" and on_gpu:
        optimizer = FusedClipAdam(ddp_model.parameters(), **oargs)
    else:
        optimizer = getattr(torch.optim, configs["optimizer"]["tse_model"])(ddp_model.parameters(), **oargs)
    sargs.update(num_epochs=configs["num_epochs"], epoch_iter=epoch_iter)
    scheduler = getattr(schedulers, configs["scheduler"]["tse_model"])(optimizer, **sargs)

    checkpoint = configs.get("checkpoint")
    init = (configs.get("model_init") or {}).get("tse_model")
    start_epoch = 1
    if init:
        load_pretrained_model(ddp_model, init)
    if checkpoint:
        load_checkpoint([ddp_model], [optimizer], [scheduler], None, checkpoint)
        start_epoch = int(re.findall(r"(?<=checkpoint_)\d*(?=.pt)", checkpoint)[0]) + 1
    if rank == 0:
        with open(os.path.join(exp_dir, "config.yaml"), "w") as f:
            yaml.dump(configs, f)
        logger.info("start_epoch: {}".format(start_epoch))

    dargs = configs.get("dataset_args", {})
    executor = Executor()
    parallel.barrier()
    for epoch in range(start_epoch, configs["num_epochs"] + 1):
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(epoch)
        train_loss, _ = executor.train(
            train_loader, [ddp_model], epoch_iter, [optimizer], criterion, [scheduler], scaler=None, epoch=epoch,
            logger=logger, enable_amp=configs.get("enable_amp", False), clip_grad=configs.get("clip_grad", 5.0),
            log_batch_interval=configs.get("log_batch_interval", 100), device=device, se_loss_weight=loss_args,
            multi_task=multi_task, SSA_enroll_prob=dargs.get("SSA_enroll_prob", 0),
            fbank_args=dargs.get("fbank_args"), sample_rate=dargs.get("resample_rate", 16000),
            speaker_feat=dargs.get("speaker_feat", True),
            # every rank's loader is bounded by the same epoch_iter (build_dataloaders), so the in-loop replica check is a
            # well-formed collective; `replica_check_interval: 0` in the config turns it off
            replica_check_interval=int(configs.get("replica_check_interval", 10 * configs.get("log_batch_interval", 100))))
        val_loss, _ = executor.cv(val_loader, [ddp_model], val_iter, criterion, epoch=epoch, logger=logger,
                                  enable_amp=configs.get("enable_amp", False),
                                  log_batch_interval=configs.get("log_batch_interval", 100), device=device)
        if rank == 0:
            logger.info("Epoch {} Train info train_loss {}".format(epoch, train_loss))
            logger.info("Epoch {} Val info val_loss {}".format(epoch, val_loss))
            if (epoch % configs.get("save_epoch_interval", 1) == 0
                    or epoch >= configs["num_epochs"] - configs.get("num_avg", 2)):        # train.py:371-372
                save_checkpoint([ddp_model], [optimizer], [scheduler], None,
                                os.path.join(model_dir, "checkpoint_{}.pt".format(epoch)))
                _symlink("checkpoint_{}.pt".format(epoch), os.path.join(model_dir, "latest_checkpoint.pt"))
        parallel.barrier()
    if rank == 0:
        _symlink("checkpoint_{}.pt".format(configs["num_epochs"]), os.path.join(model_dir, "final_checkpoint.pt"))
    return executor


def _symlink(target, link):
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(target, link)


def main(argv=None):
    configs, n_synthetic = parse_config(argv)
    train(configs, n_synthetic)


if __name__ == "__main__":
    main()
