"""ctypes binding of libwesep_hip.so (C ABI declared in include/wesep_hip.h).

The product path has NO fallback: if the shared library is missing or a call
fails, this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
(or `python -m wesep_amd.build`)."""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# WESEP_HIP_LIB: an alternate build of the same C ABI (kernel experiments); never a fallback
LIB_PATH = os.environ.get("WESEP_HIP_LIB") or os.path.join(_HERE, "libwesep_hip.so")

WS_OK = 0
PROF_LSTM_FWD, PROF_LSTM_BWD, PROF_GEMM_NT, PROF_GEMM_TN = 0, 1, 2, 3
LSTM_H = 256
ABI_VERSION = 20
GATES_F32, GATES_H2, GATES_H2S, GATES_H2F = 0, 1, 2, 3     # WS_GATES_* (wesep_hip.h): storage of the saved gates / d(gates)
DGATES_EXP = 8             # WS_DGATES_EXP: WS_GATES_H2F puts max |d(hcat)| into [2^8, 2^9)


def dgates_scale(amax_bits: int) -> float:
    """ws_dgates_scale (csrc/common.h) on the host: the power of two WS_GATES_H2F scales d(gates) by, from the float bits
    of max |d(hcat)|."""
    e = (int(amax_bits) >> 23) & 0xFF
    return 1.0 if e in (0, 255) else 2.0 ** (min(max(DGATES_EXP + 254 - e, 1), 253) - 127)


LSTM_F32_MT1, LSTM_F32_MT2, LSTM_BF16X3, LSTM_BF16X3_BLK, LSTM_BF16X3_BLK16 = 1, 2, 3, 4, 5
LSTM_PACK_FLOATS = 2 * 4 * LSTM_H * LSTM_H

_p = C.c_void_p
_ll = C.c_longlong
_i = C.c_int
_f = C.c_float


class ConvView(C.Structure):
    """ws_conv_view: the A operand of ws_gemm_nt / ws_gemm_tn as an implicit im2col matrix (include/wesep_hip.h)."""
    _fields_ = [(n, _i) for n in ("on", "mode", "H", "W", "C", "Ho", "Wo", "k", "sh", "sw", "p", "dil", "ldp")]


class Conv3x3PackSrc(C.Structure):
    _fields_ = [("w", _p), ("s_row", _ll), ("s_col", _ll), ("s_tap", _ll), ("col_off", _i), ("cols", _i)]


class Conv3x3PackArgs(C.Structure):                                                                  # ABI v19
    _fields_ = [("src", Conv3x3PackSrc * 5), ("out", _p), ("Cin", _i), ("Cout", _i), ("nsrc", _i), ("flip", _i)]


class Conv3x3Args(C.Structure):
    _fields_ = [(n, _p) for n in ("X", "W", "bias", "R", "Y")] + [(n, _ll) for n in ("ldx", "ldw", "ldy")] + \
               [(n, _i) for n in ("B", "H", "Wd", "Cin", "Cout", "pad_")]


class Conv3x3WgradArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("G", "X", "slab", "bslab")] + [(n, _ll) for n in ("ldg", "ldx", "slab_stride", "bslab_stride")] + \
               [(n, _i) for n in ("B", "H", "Wd", "Wx", "sw", "Cin", "Nn", "nsplit", "tiles_per_split", "pad_")]


class HeadsArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("x", "dy", "slope", "gamma", "beta", "y", "stats", "dx", "slab")] + \
               [(n, _ll) for n in ("ldx", "lddx")] + [(n, _i) for n in ("B", "T", "Tp", "Q", "nh", "ch", "nwg")] + \
               [("eps", C.c_float)]


class GemmNTArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("A", "W", "bias", "C", "R", "T", "stats", "gamma", "beta", "groups")] + \
               [(n, _ll) for n in ("a_s1", "a_s2", "c_s1", "c_s2", "st_m1", "st_m2", "st_base")] + \
               [(n, _i) for n in ("a_div", "c_div", "st_div1", "st_div2", "M", "N", "K", "ldw",
                                  "act", "ngroups", "max_n", "vec")] + [("conv", ConvView)]


class GemmTNArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("G", "A", "slab", "bslab", "stats", "gamma", "beta", "groups")] + \
               [(n, _ll) for n in ("g_s1", "g_s2", "a_s1", "a_s2", "st_m1", "st_m2", "st_base",
                                   "slab_stride", "bslab_stride", "out_off", "bout_off")] + \
               [(n, _i) for n in ("g_div", "a_div", "st_div1", "st_div2", "M", "Nn", "Kk",
                                  "rows_per_split", "nsplit", "shift_rows", "seq_div", "seq_len",
                                  "ngroups", "max_n", "max_k", "vec")] + [("conv", ConvView)]


class ConvWgradArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("G", "X", "slab", "bslab")] + \
               [(n, _ll) for n in ("ldg", "slab_stride", "bslab_stride")] + \
               [(n, _i) for n in ("M", "Nn", "nsplit", "tiles_per_split")] + [("conv", ConvView)]


class GroupsGeom(C.Structure):
    _fields_ = [("band_w", _p), ("band_off", _p), ("gs1", _ll), ("gs2", _ll), ("rs", _ll),
                ("ngroups", _i), ("gdiv", _i), ("L", _i), ("W", _i), ("nbands", _i), ("pad_", _i)]


class LstmArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("gates", "cbuf", "hcat", "dhcat", "wpack")] + \
               [(n, _ll) for n in ("sq_s1", "sq_s2", "step_rows")] + \
               [(n, _i) for n in ("nseq", "sq_div", "L", "mode")] + [("run_if", _p)] + \
               [("gates_in", _p), ("dgates", _p), ("gfmt", _i), ("rfmt", _i), ("amax", _p)] + \
               [("dxn", _p), ("dxn_dir_stride", _ll), ("wxpack", _p)]          # ABI v15; rfmt: v18; dxn ...: v19


class SeqMapC(C.Structure):
    _fields_ = [("sq_s1", _ll), ("sq_s2", _ll), ("step_rows", _ll), ("nseq", _i), ("sq_div", _i), ("L", _i), ("nvalid", _i)]


class GemmP2BArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("A", "Wpack", "bias", "C", "A_bl", "stats", "gamma", "beta")] + \
               [("sm", SeqMapC)] + [(n, _ll) for n in ("lda", "st_m1", "st_m2", "st_base")] + \
               [(n, _i) for n in ("st_div1", "st_div2", "N", "K")] + [("run_if", _p), ("amax", _p), ("A_bl16", _p)]


class GemmB2PArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("A", "Wpack", "bias", "R", "C")] + [("sm", SeqMapC), ("ldc", _ll), ("N", _i), ("K", _i),
                                                                         ("a_fmt", _i), ("pad_", _i), ("amax", _p), ("a16_out", _p)]


class GemmTNBArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("G", "A0", "A1", "slab", "bslab", "aslab")] + \
               [(n, _ll) for n in ("slab_stride", "bslab_stride", "aslab_stride")] + \
               [(n, _i) for n in ("g_width", "g_off", "g_cols", "a0_width", "a0_off", "a0_cols", "a0_shift",
                                  "a1_width", "a1_off", "a1_cols", "a1_shift", "nblk", "L", "nsplit",
                                  "blocks_per_split", "g_fmt")] + [("amax", _p), ("a_fmt", _i), ("pad_", _i)]


class LstmClusterArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("gates", "cbuf", "hcat", "dhcat", "whh_f", "whh_r", "xchg", "flags", "status")] + \
               [("nseq", _i), ("L", _i), ("dbg", _i), ("gfmt", _i), ("gates_in", _p)]


class LstmCluster2Args(C.Structure):
    _fields_ = [(n, _p) for n in ("gates", "cbuf", "hcat", "xn", "wcat", "bcat", "whh_f", "whh_r", "xchg", "tword", "status")] + \
               [("nseq", _i), ("L", _i), ("dbg", _i), ("rfmt", _i), ("dbg_buf", _p)]


class LstmPairArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("gates", "cbuf", "dhcat", "wpack", "xchg", "flags", "status", "dbg_buf")] + \
               [("nseq", _i), ("L", _i), ("dbg", _i), ("gfmt", _i), ("dgates", _p), ("amax", _p), ("rfmt", _i), ("pad_", _i)]


class Bands(C.Structure):
    _fields_ = [("band_of_bin", _p), ("band_f0", _p), ("band_bw", _p), ("nband", _i), ("nbins", _i)]


# device-resident descriptor arrays are built with numpy and uploaded as bytes
GROUP_NT_DTYPE = np.dtype([("W", "<u8"), ("bias", "<u8"), ("gamma", "<u8"), ("beta", "<u8"),
                           ("a_off", "<i8"), ("c_off", "<i8"), ("st_base", "<i8"),
                           ("K", "<i4"), ("N", "<i4"), ("ldw", "<i4"), ("pad_", "<i4")])
GROUP_TN_DTYPE = np.dtype([("gamma", "<u8"), ("beta", "<u8"), ("g_off", "<i8"), ("a_off", "<i8"),
                           ("st_base", "<i8"), ("out_off", "<i8"), ("bout_off", "<i8"),
                           ("Nn", "<i4"), ("Kk", "<i4"), ("pad0_", "<i4"), ("pad1_", "<i4")])
TENSOR_REF_DTYPE = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"),
                             ("exp_avg_sq", "<u8"), ("numel", "<i8")])
assert GROUP_NT_DTYPE.itemsize == 72 and GROUP_TN_DTYPE.itemsize == 72 and TENSOR_REF_DTYPE.itemsize == 40

class LstmFusedArgs(C.Structure):
    _fields_ = [("gates", _p), ("cbuf", _p), ("hcat", _p), ("xn", _p), ("wpack", _p), ("bias", _p),
                ("nseq", _i), ("L", _i), ("gfmt", _i), ("hfmt", _i)]            # hfmt: ABI v19 (the former pad_)


LSTM_FUSED_PACK_FLOATS = 2 * 8 * 24 * 4 * 2 * 64 * 4
LSTM_DX_PACK_FLOATS = 16 * (48 * 1024 + 64) // 4      # ws_lstm_pack_dx_f8 (ABI v19)

_SIGS = {
    "ws_abi_version": (_i, []),
    "ws_last_error": (C.c_char_p, []),
    "ws_prof_enable": (_i, [_i]),
    "ws_prof_collect": (_i, [_i, C.POINTER(C.c_double), C.POINTER(_ll)]),
    "ws_debug_dirty_lds": (_i, [_f, _i, _i, _p, _p]),
    "ws_debug_occupy": (_i, [_i, _i, _p, _p, _p]),
    "ws_pack_w_f16": (_i, [_p, _i, _i, _ll, _i, _i, _p, _p]),
    "ws_pack_w_f16f8": (_i, [_p, _i, _i, _ll, _i, _p, _p]),
    "ws_gemm_nt": (_i, [C.POINTER(GemmNTArgs), _p]),
    "ws_gemm_tn": (_i, [C.POINTER(GemmTNArgs), _p]),
    "ws_reduce_slabs": (_i, [_p, _i, _ll, _ll, _p, _i, _ll, _p]),
    "ws_transpose": (_i, [_p, _i, _i, _ll, _p, _p]),
    "ws_group_stats": (_i, [_p, C.POINTER(GroupsGeom), _f, _p, _p]),
    "ws_gn_bwd_reduce": (_i, [_p, _p, _p, _p, _p, C.POINTER(GroupsGeom), _p, _p]),
    "ws_gn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, C.POINTER(GroupsGeom), _p, _p]),
    "ws_gn_param_grad": (_i, [_p, _p, _p, C.POINTER(GroupsGeom), _i, _p, _p]),
    "ws_gn_bwd_fused": (_i, [_p, _p, _p, _p, _p, C.POINTER(GroupsGeom), _i, _p, _p, _p, _p, _p]),
    "ws_gn_bwd_fused2": (_i, [_p, _p, _p, _p, _p, _p, C.POINTER(GroupsGeom), _i, _p, _p, _p, _p, _p]),
    "ws_gn_bwd_apply_pg": (_i, [_p, _p, _p, _p, _p, _p, C.POINTER(GroupsGeom), _p, _p, _p, _p, _p]),
    "ws_lstm_pack": (_i, [_p, _p, _p, _p, _i, _p]),
    "ws_lstm_fwd": (_i, [C.POINTER(LstmArgs), _p]),
    "ws_lstm_bwd": (_i, [C.POINTER(LstmArgs), _p]),
    "ws_lstm_fwd_cluster": (_i, [C.POINTER(LstmClusterArgs), _p]),
    "ws_lstm_bwd_cluster": (_i, [C.POINTER(LstmClusterArgs), _p]),
    "ws_lstm_fwd_cluster2": (_i, [C.POINTER(LstmCluster2Args), _p]),
    "ws_lstm_pack_pair": (_i, [_p, _p, _p, _p]),
    "ws_lstm_pack_pair_f16": (_i, [_p, _p, _p, _p]),
    "ws_lstm_pack_pair_f8": (_i, [_p, _p, _p, _p]),
    "ws_lstm_pack_pair_f8mx": (_i, [_p, _p, _p, _p]),
    "ws_lstm_pack_bwd_f8": (_i, [_p, _p, _p, _p]),
    "ws_lstm_pack_dx_f8": (_i, [_p, _p, _p]),
    "ws_lstm_bwd_pair": (_i, [C.POINTER(LstmPairArgs), _p]),
    "ws_lstm_cat_ih": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "ws_pack_w": (_i, [_p, _i, _i, _ll, _i, _i, _p, _p]),
    "ws_gemm_p2b": (_i, [C.POINTER(GemmP2BArgs), _p]),
    "ws_gemm_b2p": (_i, [C.POINTER(GemmB2PArgs), _p]),
    "ws_gemm_tnb": (_i, [C.POINTER(GemmTNBArgs), _p]),
    "ws_stft_bandsplit": (_i, [_p, _i, _i, C.POINTER(Bands), _p, _p]),
    "ws_mask_istft_frames": (_i, [_p, _p, _i, _i, C.POINTER(Bands), _p, _p]),
    "ws_istft_ola": (_i, [_p, _i, _i, _i, _p, _p]),
    "ws_mask_istft_bwd": (_i, [_p, _p, _p, _i, _i, _i, C.POINTER(Bands), _p, _p]),
    "ws_affine_fwd": (_i, [_p, _p, _p, _f, _ll, _i, _i, _p, _p]),
    "ws_affine_bwd": (_i, [_p, _p, _p, _f, _ll, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "ws_sisdr_fwd": (_i, [_p, _p, _i, _i, _f, _p, _p, _p]),
    "ws_sisdr_bwd": (_i, [_p, _p, _p, _p, _i, _i, _p, _p]),
    # Conv-TasNet / SpEx+ (tasnet.hip)
    "ws_flat_stats": (_i, [_p, _i, _ll, C.c_float, _i, _p, _p, _p]),
    "ws_prelu_fwd": (_i, [_p, _p, _p, _ll, _i, _i, _p, _p]),
    "ws_prelu_bwd": (_i, [_p, _p, _p, _ll, _p, _p, _i, _p]),
    "ws_dwconv_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "ws_dwconv_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p]),
    "ws_dwconv_ex_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "ws_dwconv_ex_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p]),
    "ws_chan_sums": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_norm_ab": (_i, [_p, _p, _i, _i, _ll, _p, _p]),
    "ws_norm_bwd_apply_cl": (_i, [_p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _p]),
    "ws_maskmul_fwd": (_i, [_p, _ll, _p, _ll, _i, _p, _p]),
    "ws_maskmul_bwd": (_i, [_p, _p, _ll, _p, _ll, _i, _p, _ll, _p, _p]),
    "ws_relu_mask": (_i, [_p, _p, _ll, _p]),
    "ws_ola_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_ola_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_sum_partial": (_i, [_p, _ll, _p, _i, _p]),
    "ws_bn_stats": (_i, [_p, _ll, _i, C.c_float, C.c_float, _p, _p, _i, _p, _p, _p]),
    "ws_bn_prelu_fwd": (_i, [_p, _p, _p, _p, _p, _p, _ll, _i, _p, _p, _p]),
    "ws_bn_bwd": (_i, [_p, _p, _p, _p, _ll, _i, _i, _p, _p, _p, _p]),
    "ws_maxpool3_fwd": (_i, [_p, _i, _i, _i, _p, _p]),
    "ws_maxpool3_bwd": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "ws_bcast_rows": (_i, [_p, C.c_float, _i, _ll, _i, _p, _p]),
    "ws_cross_entropy": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "ws_im2col": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _ll, _p, _p]),
    "ws_col2im": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "ws_conv_wgrad": (_i, [C.POINTER(ConvWgradArgs), _p]),
    "ws_astp_fwd": (_i, [_p, _p, _i, _i, _i, C.c_float, _p, _p, _p]),
    "ws_astp_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, C.c_float, _p, _p, _p]),
    "ws_rowbias_act_fwd": (_i, [_p, _p, _ll, _i, _i, _i, _p, _p]),
    "ws_act_bwd": (_i, [_p, _p, _ll, _i, _p, _p]),
    "ws_seg_sums": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "ws_seg_scale": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "ws_tstp_fwd": (_i, [_p, _i, _i, _i, _i, C.c_float, _p, _p]),
    "ws_tstp_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "ws_im2col_hw": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _ll, _p, _p]),
    "ws_col2im_hw": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "ws_elu_fwd": (_i, [_p, _ll, _p, _p]),
    "ws_elu_bwd": (_i, [_p, _p, _ll, _p, _p]),
    "ws_inorm_finalize": (_i, [_p, _i, _i, _ll, C.c_float, _p, _p]),
    "ws_inorm_apply": (_i, [_p, _p, _ll, _i, _i, _p, _p]),
    "ws_inorm_bwd_apply": (_i, [_p, _p, _p, _p, _ll, _i, _i, _p, _p]),
    "ws_avgpool_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_avgpool_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_bilinear_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "ws_bilinear_bwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ws_scale_bf_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_freq_linear_fwd": (_i, [_p, _p, _ll, _p, _i, _i, _i, _i, _p, _p]),
    "ws_scale_bf_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ws_softmax_rows_fwd": (_i, [_p, _ll, _i, C.c_float, _p, _p]),
    "ws_softmax_rows_bwd": (_i, [_p, _p, _ll, _i, C.c_float, _p, _p]),
    "ws_heads_fwd": (_i, [C.POINTER(HeadsArgs), _p]),
    "ws_heads_bwd": (_i, [C.POINTER(HeadsArgs), _p]),
    "ws_conv3x3": (_i, [C.POINTER(Conv3x3Args), _p]),
    "ws_conv3x3_pack": (_i, [C.POINTER(Conv3x3PackArgs), _p]),
    "ws_conv3x3_wgrad": (_i, [C.POINTER(Conv3x3WgradArgs), _p]),
    "ws_in_act_sums": (_i, [_p, _p, _ll, _p, _i, _i, _i, _i, _i, _p, _p]),
    "ws_in_act_apply": (_i, [_p, _p, _ll, _i, _i, _i, _p, _ll, _p]),
    "ws_in_act_bwd_apply": (_i, [_p, _p, _ll, _p, _p, _ll, _i, _i, _i, _p, _ll, _p]),
    "ws_rowln_grid": (_i, [_ll, _i]),
    "ws_rowln_fwd": (_i, [_p, _p, _p, _ll, _i, C.c_float, _p, _p, _p]),
    "ws_rowln_bwd": (_i, [_p, _p, _p, _p, _p, _ll, _i, _p, _p, _p]),
    "ws_preemph_pad": (_i, [_p, _i, _i, _i, _i, C.c_float, _p, _p]),
    "ws_power_spec": (_i, [_p, _ll, _i, _i, _i, _p, _p]),
    "ws_log_eps": (_i, [_p, _ll, C.c_float, _p]),
    "ws_lstm_pack_fused": (_i, [_p, _p, _p, _p, _p, _p]),
    "ws_lstm_pack_fused_h16": (_i, [_p, _p, _p, _p, _p, _p]),
    "ws_lstm_pack_fused_h8": (_i, [_p, _p, _p, _p, _p, _p]),
    "ws_lstm_fwd_fused": (_i, [C.POINTER(LstmFusedArgs), _p]),
    "ws_grad_norms": (_i, [_p, _i, _p, _p, _p]),
    "ws_clip_adam_step": (_i, [_p, _i, _p, _f, _f, _f, _f, _f, _f, _i, _i, _p, _p, _p, _p]),
    "ws_guard_commit": (_i, [_p, _p, _p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib = None


class WesepHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WesepHipError(
                f"{LIB_PATH} is missing: the HIP extension is not built and wesep_amd has no "
                "CPU fallback. Run `python -m wesep_amd.build` (needs hipcc).")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.ws_abi_version() != ABI_VERSION:
            raise WesepHipError("libwesep_hip.so ABI version mismatch; rebuild")
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != WS_OK:
        msg = lib().ws_last_error().decode("utf-8", "replace")
        raise WesepHipError(f"{what} failed (rc={rc}): {msg}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a float32/int32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise WesepHipError("wesep_amd ops need CUDA (ROCm) tensors; there is no CPU path")
    return C.c_void_p(t.data_ptr())


def upload_struct_array(arr: np.ndarray, device, blocking=False) -> torch.Tensor:
    """numpy structured array -> uint8 device tensor holding the same bytes.  blocking: the pageable copy of rounds 1-5 (A/B)."""
    host = torch.from_numpy(np.frombuffer(arr.tobytes(), dtype=np.uint8).copy())
    if not blocking and torch.cuda.is_available() and torch.device(device).type == "cuda":
        # pinned + non_blocking: the host does not wait for the stream (a copy from pageable memory does); the pinned block's
        # reuse is the caching host allocator's business (it records the copy's stream)
        return host.pin_memory().to(device, non_blocking=True)
    return host.to(device)
