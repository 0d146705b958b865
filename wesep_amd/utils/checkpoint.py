"""Checkpoint files interchangeable with the reference's (wesep/utils/checkpoint.py:8-105):
`{"models": [state_dict], "optimizers": [...], "schedulers": [...], "scaler": ...}`; the
DDP/DataParallel `.module` wrapper is stripped on save and on load."""
import torch


def _unwrap(model):
    wrappers = (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)
    return model.module if isinstance(model, wrappers) else model


def _pick(states, mode):
    idx = {"all": None, "generator": 0, "discriminator": 1}[mode]
    keys = ("models", "optimizers", "schedulers")
    if idx is None:
        return tuple(states[k] for k in keys)
    return tuple([states[k][idx]] for k in keys)


def load_pretrained_model(model, path, type="generator"):
    assert type in ("generator", "discriminator")
    states = torch.load(path, map_location="cpu")
    if type == "discriminator":
        assert len(states["models"]) == 2
    _unwrap(model).load_state_dict(states["models"][0 if type == "generator" else 1])


def load_checkpoint(models, optimizers, schedulers, scaler, path, only_model=False, mode="all"):
    assert mode in ("all", "generator", "discriminator")
    states = torch.load(path, map_location="cpu")
    model_state, optim_state, sched_state = _pick(states, mode)
    for model, sd in zip(models, model_state):
        _unwrap(model).load_state_dict(sd, strict=False)
    if only_model:
        return
    for opt, sd in zip(optimizers, optim_state):
        opt.load_state_dict(sd)
    for sched, sd in zip(schedulers, sched_state):
        if sched is not None:
            sched.load_state_dict(sd)
    if scaler is not None and states.get("scaler") is not None:
        scaler.load_state_dict(states["scaler"])


def save_checkpoint(models, optimizers, schedulers, scaler, path):
    torch.save({
        "models": [_unwrap(m).state_dict() for m in models],
        "optimizers": [o.state_dict() for o in optimizers],
        "schedulers": [s.state_dict() if s is not None else None for s in schedulers],
        "scaler": scaler.state_dict() if scaler is not None else None,
    }, path)
