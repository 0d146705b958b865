"""Write a pBSRNN, Conv-TasNet / SpEx+, DPCCN or TF-GridNet checkpoint as the flat weight container of the native runtime (include/wesep_engine.h,
runtime/engine.cc) -- the counterpart of the reference's `wesep/bin/export_jit.py` (TorchScript archive for the
LibTorch runtime).

    python -m wesep_amd.bin.export_engine --config conf.yaml --checkpoint avg_model.pt --out model.wsw

Container layout (little endian):
    char magic[8] = "WSEPW001"
    u32 n_meta;    n_meta    x { char key[32]; i64 value }
    u32 n_tensors; n_tensors x { u32 name_len; char name[]; u32 ndim; i64 dims[ndim]; u64 offset_floats }
    u64 n_floats;  float32 data[n_floats]
Tensor names are the reference's `state_dict` keys, so any wesep BSRNN checkpoint exports unchanged."""
import argparse
import struct

import numpy as np
import torch

FUSE = {"concat": 0, "additive": 1, "multiply": 2, "FiLM": 3}
SKIP_PREFIXES = ("pred_linear.",)


def engine_meta(model):
    sep = model.separator
    fuse_layer = next(m for m in sep.separation if hasattr(m, "fuse_type"))
    meta = {
        "sample_rate": model.sr, "win": model.win, "stride": model.stride, "feature_dim": model.feature_dim,
        "num_repeat": sum(1 for m in sep.separation if not hasattr(m, "fuse_type")),
        "spk_emb_dim": model.spk_emb_dim, "spk_fuse_type": FUSE[fuse_layer.fuse_type],
        "multi_fuse": int(sep.multi_fuse), "use_spk_transform": int(not isinstance(model.spk_transform, torch.nn.Identity)),
        "joint_training": int(model.joint_training), "feat_dim": 80, "spk_feat": 1,
    }
    return speaker_meta(model, meta)


def speaker_meta(model, meta):
    """The speaker-encoder keys of the meta block (shared by the pBSRNN and DPCCN plans of the runtime)."""
    if model.joint_training:
        meta["spk_feat"] = int(bool(model.spk_feat))      # False: the in-model front-end's buffers are exported too
        spk = model.spk_model
        if type(spk).__name__ == "ECAPA_TDNN":            # wespeaker ECAPA-TDNN (the published bsrnn_ecapa_vox1 model)
            meta.update(spk_kind=1, spk_channels=spk.layer1.conv.out_channels, feat_dim=spk.layer1.conv.in_channels,
                        spk_glob=int(spk.pool.linear1.in_channels == 3 * spk.pool.linear2.out_channels),
                        spk_emb_bn=int(isinstance(getattr(spk, "bn2", None), torch.nn.BatchNorm1d)))
        elif hasattr(spk, "seg_1") and type(spk.layer1[0]).__name__ in ("BasicBlock", "Bottleneck") and \
                getattr(spk, "pooling_func", "TSTP") == "TSTP":
            ex = 4 if type(spk.layer1[0]).__name__ == "Bottleneck" else 1
            meta.update(spk_kind=0, spk_bottleneck=int(ex == 4), spk_two_emb=int(bool(getattr(spk, "two_emb_layer", False))))
            for i, layer in enumerate((spk.layer1, spk.layer2, spk.layer3, spk.layer4)):
                meta[f"spk_blocks{i}"] = len(layer)
            meta["feat_dim"] = int(spk.seg_1.weight.shape[1] // (2 * 32 * 8 * ex)) * 8
        elif type(spk).__name__ == "CAMPPlus":            # wespeaker CAM++ (round 5: spk_kind 2; the constructor's defaults)
            xv = spk.xvector
            layers = [len(getattr(xv, f"block{i}")) for i in (1, 2, 3)]
            first = xv.block1.tdnnd1
            if layers != [12, 24, 16] or xv.tdnn.linear.out_channels != 128 or first.linear1.out_channels != 128 or \
                    first.cam_layer.linear_local.out_channels != 32 or not hasattr(first.nonlinear1, "relu"):
                raise NotImplementedError("export_engine: CAM++ with a non-default backbone (growth_rate 32, bn_size 4, "
                                          "init_channels 128, 'batchnorm-relu') has no launch plan in the native runtime")
            meta.update(spk_kind=2, feat_dim=spk.feat_dim)
        else:
            raise NotImplementedError(f"export_engine: speaker encoder {type(spk).__name__} (a ResNet with a pooling layer "
                                      "other than TSTP) has no launch plan in the native runtime; the wespeaker ResNets "
                                      "with TSTP, ECAPA-TDNN and CAM++ do")
    return meta


def write_container(path, meta, state):
    names, blobs, off = [], [], 0
    for k, v in state.items():
        if k.startswith(SKIP_PREFIXES) or k.endswith("num_batches_tracked") or not torch.is_floating_point(v):
            continue
        a = np.ascontiguousarray(v.detach().cpu().float().numpy())
        names.append((k, a.shape, off))
        blobs.append(a.reshape(-1))
        off += -(-a.size // 4) * 4                      # every tensor starts on a 16-byte boundary
    data = np.zeros(off, dtype=np.float32)
    for (k, shape, o), b in zip(names, blobs):
        data[o:o + b.size] = b
    with open(path, "wb") as f:
        f.write(b"WSEPW001")
        f.write(struct.pack("<I", len(meta)))
        for k, v in meta.items():
            key = k.encode()
            assert len(key) < 32, k
            f.write(key.ljust(32, b"\0"))
            f.write(struct.pack("<q", int(v)))
        f.write(struct.pack("<I", len(names)))
        for k, shape, o in names:
            kb = k.encode()
            f.write(struct.pack("<I", len(kb)))
            f.write(kb)
            f.write(struct.pack("<I", len(shape)))
            for d in shape:
                f.write(struct.pack("<q", int(d)))
            f.write(struct.pack("<Q", o))
        f.write(struct.pack("<Q", data.size))
        f.write(data.tobytes())
    return len(names), data.size


def tasnet_meta(model):
    """Conv-TasNet / SpEx+ (arch 1): the runtime's launch plan covers the shipped configuration -- Multi encoder and
    decoder, gLN, non-causal, no skip connection, concatConv multi-fusion -- with fixed embeddings or the SpEx+ speaker
    encoder on the enrollment waveform.  Everything else is refused by name."""
    sep = model.separation
    first = sep.separation[0]
    problems = []
    if model.encoder_type != "Multi" or model.decoder_type != "Multi":
        problems.append(f"encoder / decoder type {model.encoder_type} / {model.decoder_type} (Multi only)")
    if sep.spk_fuse_type != "concatConv":
        problems.append(f"spk_fuse_type {sep.spk_fuse_type!r} (concatConv only)")
    if model.norm_type != "gLN":
        problems.append(f"norm {model.norm_type!r} (gLN only)")
    if getattr(first, "causal", False):
        problems.append("causal blocks")
    if any(getattr(b, "skip_con", False) for m in sep.separation if hasattr(m, "separation") for b in m.separation):
        problems.append("skip connections")
    if model.joint_training and model.spk_feat:
        problems.append("a wespeaker encoder on fbank enrollment (the SpEx+ encoder on the waveform is what the plan has)")
    if problems:
        raise NotImplementedError("export_engine: Conv-TasNet with " + "; ".join(problems) + " has no launch plan in the "
                                  "native runtime")
    enc = model.encoder
    blocks = sep.separation[1].separation
    return {
        "arch": 1, "sample_rate": 16000, "N": enc.encoder_1d_short.out_channels, "L": enc.L1,
        "B": enc.proj.out_channels, "H": first.conv1x1.out_channels, "P": first.dconv.kernel_size[0],
        "X": len(blocks) + 1, "R": len(sep.separation) // 2,
        "spk_emb_dim": first.conv1x1.in_channels - enc.proj.out_channels,
        "use_spk_transform": int(not isinstance(model.spk_transform, torch.nn.Identity)),
        "joint_training": int(model.joint_training), "spk_feat": 0,
    }


def dpccn_meta(model):
    """DPCCN (arch 2): the runtime's launch plan covers the reference constructor's defaults (win 512, stride 128, 257
    bins, 3 x 3 kernels, strides (1, 1) / (1, 2)) with any TCN depth, causal or not, multiply / additive / FiLM fusion,
    fixed embeddings or the speaker encoders the pBSRNN plan has; all four fusions (round 5: `concat`, a Linear over the
    frequency axis, as ws_freq_linear_fwd)."""
    fuse = model.spk_fuse.fuse_type
    blocks = model.tcn_layers[0]
    meta = {
        "arch": 2, "sample_rate": 16000, "win": model.win_len, "stride": model.hop_size, "feature_dim": model.win_len // 2 + 1,
        "tcn_layers": len(model.tcn_layers), "tcn_blocks": len(blocks), "causal": int(bool(blocks[0].causal)),
        "spk_emb_dim": model.spk_emb_dim, "spk_fuse_type": FUSE[fuse], "multi_fuse": 0,
        "use_spk_transform": int(not isinstance(model.spk_transform, torch.nn.Identity)),
        "joint_training": int(model.joint_training), "feat_dim": 80, "spk_feat": 1,
    }
    return speaker_meta(model, meta)


def gridnet_meta(model):
    """TF-GridNet (arch 3): the runtime's launch plan covers the shipped recipe's geometry -- one microphone, one source,
    emb_dim 128, emb_ks = emb_hs = 1, lstm_hidden_units <= 256, any of the four fusions -- with fixed embeddings or the
    speaker encoders the pBSRNN plan has.  Everything else is refused by name."""
    blk = model.blocks[0]
    problems = []
    if model.n_imics != 1 or model.n_srcs != 1:
        problems.append(f"n_imics {model.n_imics} / n_srcs {model.n_srcs} (1 / 1 only)")
    if blk.emb_dim != 128 or blk.emb_ks != 1 or blk.emb_hs != 1:
        problems.append(f"emb_dim {blk.emb_dim}, emb_ks {blk.emb_ks}, emb_hs {blk.emb_hs} (128 / 1 / 1: the blocked-layout recurrences)")
    if model.stride * 2 != model.n_fft:
        problems.append(f"stride {model.stride} != n_fft / 2")
    if problems:
        raise NotImplementedError("export_engine: TF-GridNet with " + "; ".join(problems) + " has no launch plan in the "
                                  "native runtime")
    meta = {
        "arch": 3, "sample_rate": 16000, "n_fft": model.n_fft, "stride": model.stride, "n_layers": model.n_layers,
        "emb_dim": blk.emb_dim, "emb_ks": blk.emb_ks, "emb_hs": blk.emb_hs, "lstm_hidden_units": blk.hidden,
        "attn_n_head": blk.n_head, "attn_E": blk.E, "n_srcs": model.n_srcs, "n_imics": model.n_imics,
        "spk_emb_dim": model.spk_emb_dim, "spk_fuse_type": FUSE[model.spk_fuse.fuse_type], "multi_fuse": 0,
        "use_spk_transform": int(not isinstance(model.spk_transform, torch.nn.Identity)),
        "joint_training": int(model.joint_training), "feat_dim": 80, "spk_feat": 1,
    }
    return speaker_meta(model, meta)


def export_engine(model, path):
    """model: a wesep_amd (or reference) `BSRNN` / `BSRNN_Multi` / `ConvTasNet` / `DPCCN` (or `TFGridNet`) instance -> container at `path`;
    returns (n_tensors, n_floats)."""
    name = type(model).__name__
    if name == "ConvTasNet":
        return write_container(path, tasnet_meta(model), model.state_dict())
    if name == "DPCCN":
        return write_container(path, dpccn_meta(model), model.state_dict())
    if name == "TFGridNet":
        return write_container(path, gridnet_meta(model), model.state_dict())
    if name not in ("BSRNN", "BSRNN_Multi"):
        raise NotImplementedError("export_engine: the native runtime runs pBSRNN (BSRNN / BSRNN_Multi), Conv-TasNet / "
                                  f"SpEx+, DPCCN and TF-GridNet checkpoints, not {name}")
    return write_container(path, engine_meta(model), model.state_dict())


def main():
    import yaml
    from ..models import get_model
    ap = argparse.ArgumentParser(description="export a pBSRNN, Conv-TasNet / SpEx+, DPCCN or TF-GridNet checkpoint for the native MI355X runtime")
    ap.add_argument("--config", required=True)
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    with open(args.config) as f:
        conf = yaml.safe_load(f)
    model = get_model(conf["model"]["tse_model"])(**conf["model_args"]["tse_model"])
    states = torch.load(args.checkpoint, map_location="cpu")
    model.load_state_dict(states["models"][0] if "models" in states else states)
    n, nf = export_engine(model, args.out)
    print(f"{args.out}: {n} tensors, {nf * 4 / 2 ** 20:.1f} MiB")


if __name__ == "__main__":
    main()
