"""Checkpoint averaging with the reference tool's command line and file format (wesep/bin/average_model.py:25-99):

    python -m wesep_amd.bin.average_model --dst_model avg_model.pt --src_path exp/models --num 2
    python -m wesep_amd.bin.average_model --dst_model avg.pt --src_path exp/models --mode epochs --epochs 148,150

`--mode final` (default) averages the `--num` highest-numbered `checkpoint_<n>.pt` files of `--src_path` (files
whose names mark them as averaged / final / latest are left out); any other mode takes the comma-separated
`--epochs`.  Inputs may be full training checkpoints (`{"models": [state_dict], ...}`) or bare state dicts; the
output is `{"models": [averaged state_dict]}`, which `load_pretrained_model`, `infer.py` and `export_engine` read.
Every entry is summed and divided by the count with true division, integer buffers (`num_batches_tracked`) included
-- they become floating point, exactly as in the reference."""
import argparse
import os
import re

import torch


def select_checkpoints(src_path, mode="final", num=5, epochs="", min_epoch=0, max_epoch=65536):
    if mode != "final":
        return [os.path.join(src_path, f"checkpoint_{e.strip()}.pt") for e in epochs.split(",") if e.strip()]
    found = []
    for name in os.listdir(src_path):
        m = re.fullmatch(r"checkpoint_(\d+)\.pt", name)
        if m and min_epoch <= int(m.group(1)) <= max_epoch:
            found.append((int(m.group(1)), os.path.join(src_path, name)))
    return [p for _, p in sorted(found)][-num:]


def average_checkpoints(paths):
    if not paths:
        raise ValueError("average_model: no checkpoints to average")
    total = None
    for path in paths:
        states = torch.load(path, map_location="cpu")
        sd = states["models"][0] if "models" in states else states
        if total is None:
            total = {k: v.clone() if isinstance(v, torch.Tensor) else v for k, v in sd.items()}
        else:
            if sd.keys() != total.keys():
                raise ValueError(f"average_model: {path} holds a different set of tensors")
            for k in total:
                total[k] = total[k] + sd[k]
    return {k: torch.true_divide(v, len(paths)) for k, v in total.items()}


def main():
    ap = argparse.ArgumentParser(description="average model")
    ap.add_argument("--dst_model", required=True, help="averaged model")
    ap.add_argument("--src_path", required=True, help="src model path for average")
    ap.add_argument("--num", default=5, type=int, help="nums for averaged model")
    ap.add_argument("--min_epoch", default=0, type=int, help="min epoch used for averaging model")
    ap.add_argument("--max_epoch", default=65536, type=int, help="max epoch used for averaging model")
    ap.add_argument("--mode", default="final", type=str, help="final: the last --num epochs; else: --epochs")
    ap.add_argument("--epochs", default="1,2,3,4,5", type=str, help="epochs to average when --mode is not final")
    args = ap.parse_args()
    paths = select_checkpoints(args.src_path, args.mode, args.num, args.epochs, args.min_epoch, args.max_epoch)
    if args.mode == "final" and len(paths) != args.num:
        raise SystemExit(f"average_model: found {len(paths)} checkpoints, --num {args.num}")
    for p in paths:
        print("Processing", p)
    torch.save({"models": [average_checkpoints(paths)]}, args.dst_model)
    print("Saving to", args.dst_model)


if __name__ == "__main__":
    main()
