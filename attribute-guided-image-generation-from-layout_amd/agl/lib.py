"""ctypes binding of libagl.so (include/agl.h) plus tensor-level wrappers without autograd.

PyTorch is used here only as the owner of device memory and of the current HIP stream: every
wrapper allocates its outputs with torch.empty, passes raw device pointers + the current stream
handle through the C ABI and returns the tensors.  There is NO fallback: if libagl.so is missing,
or an operand is not a contiguous fp32 (int64 for indices) tensor on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagl.so")

_P, _I, _L, _F, _LL = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_longlong

# name -> (restype, argtypes); must mirror include/agl.h (tests/test_abi.py checks both directions)
SIGNATURES = {
    "agl_version": (_I, []),
    "agl_last_error": (C.c_char_p, []),
    "agl_conv2d_fwd_ws_bytes": (_L, [_I] * 9),
    "agl_conv2d_bwd_data_ws_bytes": (_L, [_I] * 10),
    "agl_conv2d_splitk_ws_bytes": (_L, [_I, _L, _I, _I, _L]),
    "agl_conv2d_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 13 + [_P]),
    "agl_conv2d_bwd_data": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 13 + [_P]),
    "agl_conv2d_fwd_packed_bytes": (_L, [_I] * 10),
    "agl_conv2d_bwd_data_packed_bytes": (_L, [_I] * 11),
    "agl_conv2d_pack_weights": (_I, [_P, _P, _L] + [_I] * 6 + [_P]),
    "agl_conv2d_pack_desc": (_I, [_P, _P, _L] + [_I] * 6 + [_P]),
    "agl_conv2d_pack_many": (_I, [_P, _I, _L, _P]),
    "agl_conv2d_last_pipe": (_I, []),
    "agl_conv2d_split_products": (_I, []),
    "agl_conv2d_bwd_weight_ws_bytes": (_L, [_I] * 6),
    "agl_conv2d_bwd_weight": (_I, [_P, _P, _P, _P, _I, _P, _P, _L] + [_I] * 14 + [_P]),
    "agl_conv2d_fwd_flops": (C.c_double, [_I] * 10),
    "agl_conv2d_bwd_data_flops": (C.c_double, [_I] * 11),
    "agl_conv2d_bwd_weight_flops": (C.c_double, [_I] * 13),
    "agl_bn_stats_ws_bytes": (_L, [_I] * 3),
    "agl_bn_stats": (_I, [_P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "agl_bn_running_update": (_I, [_P, _I, _F, _P, _P, _P, _P]),
    "agl_bn_running_update_multi": (_I, [_P, _I, _P]),
    "agl_bn_stats_eval": (_I, [_P, _P, _I, _F, _P, _P, _P]),
    "agl_bn_stats_from_partials": (_I, [_P, _I, _I, _L, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "agl_conv2d_fwd_stats_floats": (_L, [_I] * 4),
    "agl_conv2d_fwd_stats": (_I, [_P] * 7 + [_L] + [_I] * 11 + [_P, _L, _P, _P]),
    "agl_norm_apply_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P]),
    "agl_norm_apply_fwd_y16": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P]),
    "agl_norm_bwd_y16": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _L, _P]),
    "agl_norm_bwd_ws_bytes": (_L, [_I, _I]),
    "agl_norm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _L, _P]),
    "agl_crop_fwd": (_I, [_P, _P, _P, _P] + [_I] * 8 + [_P]),
    "agl_crop_bwd": (_I, [_P, _P, _P, _P] + [_I] * 8 + [_P]),
    "agl_crop_bwd_sorted": (_I, [_P, _P, _P, _P] + [_I] * 8 + [_P]),
    "agl_lstm_gates_fwd": (_I, [_P] * 7 + [_I] * 3 + [_P]),
    "agl_lstm_gates_bwd": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "agl_lstm_gates_fwd_sum": (_I, [_P, _P, _P, _I, _LL, _P, _P, _P, _P, _I, _I, _I, _P]),
    "agl_lstm_gates_bwd_sum": (_I, [_P, _P, _I, _LL, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "agl_conv2d_deferred": (_I, [_P, _P, _P]),
    "agl_spade_cells": (_I, [_P, _I, _I, _I, _P, _P]),
    "agl_conv2d_fwd_spade": (_I, [_P] * 5 + [_I] + [_P] * 6 + [_L] + [_I] * 10 + [_P, _L, _P, _P]),
    "agl_conv2d_fwd_spade_ok": (_I, [_I] * 9),
    "agl_conv2d_bwd_weight_spade": (_I, [_P] * 6 + [_I, _P, _P, _I, _P, _P, _L] + [_I] * 13 + [_P]),
    "agl_conv2d_bwd_weight_spade_ok": (_I, [_I] * 11),
    "agl_norm_bwd_spade": (_I, [_P] * 5 + [_I, _I, _P, _P, _I, _I, _I, _P, _P, _I, _I, _P, _L, _P]),
    "agl_relu_bwd": (_I, [_P, _P, _P, _L, _P]),
    "agl_axpby": (_I, [_P, _P, _F, _F, _P, _L, _P]),
    "agl_gather_rows": (_I, [_P, _P, _P, _L, _L, _I, _P]),
    "agl_scatter_rows": (_I, [_P, _P, _P, _L, _L, _P]),
    "agl_grid_gather_fwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "agl_grid_gather_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "agl_box2_fwd": (_I, [_P, _P, _L, _I, _I, _P]),
    "agl_box2_fwd_bf16": (_I, [_P, _P, _L, _I, _I, _P]),
    "agl_conv2d_bwd_weight_takes_bf16_x": (_I, [_I] * 11),
    "agl_conv2d_bwd_data_takes_bf16_mask": (_I, [_I] * 11),
    "agl_conv2d_fwd_writes_bf16_y": (_I, [_I] * 12),
    "agl_conv2d_fwd_takes_bf16_x": (_I, [_I] * 9),
    "agl_conv2d_fwd_takes_blocked": (_I, [_I] * 9),
    "agl_conv2d_bwd_data_takes_bf16_dy": (_I, [_I] * 11),
    "agl_conv2d_bwd_weight_takes_bf16_dy": (_I, [_I] * 11),
    "agl_norm_fold_table": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "agl_conv2d_fwd_fold_ok": (_I, [_I] * 9),
    "agl_conv2d_fwd_fold": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P, _L, _P, _P]),
    "agl_conv2d_bwd_weight_fold_ok": (_I, [_I] * 11),
    "agl_conv2d_bwd_weight_fold": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _L] + [_I] * 13 + [_P]),
    "agl_conv2d_fwd_addend": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 11 + [_P]),
    "agl_norm_bwd_fold": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "agl_box2_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "agl_avgpool2_fwd": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "agl_avgpool2_bwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "agl_avgpool2_fwd_x16": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "agl_avgpool2_bwd_x16": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "agl_to_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "agl_avgpool2_fwd_xblk": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "agl_avgpool2_bwd_xblk": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agl_conv2d_fwd_shortcut_ok": (_I, [_I] * 8),
    "agl_conv2d_fwd_shortcut": (_I, [_P] * 8 + [_I, _P, _P, _L] + [_I] * 10 + [_P]),
    "agl_upsample_nearest_fwd": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "agl_upsample_nearest_bwd": (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "agl_sum_hw_fwd": (_I, [_P, _P, _L, _I, _I, _F, _P]),
    "agl_sum_hw_bwd": (_I, [_P, _P, _P, _L, _I, _I, _F, _P]),
    "agl_channel_sum_ws_bytes": (_L, [_I]),
    "agl_channel_sum": (_I, [_P, _P, _I, _I, _I, _I, _P, _L, _P]),
    "agl_reparam_fwd": (_I, [_P, _P, _P, _P, _L, _P]),
    "agl_reparam_bwd": (_I, [_P, _P, _P, _P, _L, _P]),
    "agl_concat2_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "agl_concat2_bwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "agl_embedding_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "agl_mask_outer_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "agl_mask_outer_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "agl_pool_fuse_weight_fwd": (_I, [_P, _P, _L, _P]),
    "agl_pool_fuse_weight_bwd": (_I, [_P, _P, _L, _P]),
    "agl_layout1_levels": (_I, [_P] * 13 + [_I, _I, _I, _F, _F, _I, _P]),
    "agl_layout1_permute": (_I, [_P, _P, _I, _I, _I, _P]),
    "agl_layout1_pixels": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "agl_layout1_tapsum": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "agl_layout1_levels_bwd": (_I, [_P] * 12 + [_I, _I, _I, _I, _I, _P, _L, _P]),
    "agl_sn_layer_desc_bytes": (_L, []),
    "agl_sn_tmp_floats": (_L, [_I, _I]),
    "agl_sn_forward": (_I, [_P, _I, _I, _F, _P]),
    "agl_sn_backward": (_I, [_P, _I, _I, _P]),
    "agl_bce_logits_const": (_I, [_P, _L, _F, _F, _P, _P, _P]),
    "agl_bce_logits_posw": (_I, [_P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "agl_cross_entropy": (_I, [_P, _P, _L, _I, _F, _P, _P, _P]),
    "agl_loss_rows_ws_bytes": (_L, [_L]),
    "agl_cross_entropy_ws": (_I, [_P, _P, _L, _I, _F, _P, _P, _P, _L, _P]),
    "agl_bce_logits_posw_ws": (_I, [_P, _P, _P, _L, _I, _F, _P, _P, _P, _L, _P]),
    "agl_l1_rows_ws_bytes": (_L, []),
    "agl_l1_rows": (_I, [_P, _P, _P, _L, _L, _F, _F, _P, _P, _P, _L, _P]),
    "agl_hinge_loss": (_I, [_P, _L, _I, _F, _P, _P, _P]),
    "agl_kl_sum": (_I, [_P, _P, _L, _F, _P, _P, _P, _P]),
    "agl_rasterize_boxes": (_I, [_P, _P, _I, _I, _P]),
    "agl_attr_estimate": (_I, [_P, _P, _P, _I, _I, _P]),
    "agl_layout_from_boxes": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "agl_attr_edit": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "agl_topk_contains": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "agl_sigmoid_threshold": (_I, [_P, _P, _L, _F, _P]),
    "agl_deprocess_u8": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "agl_adam_step": (_I, [_P, _P, _P, _P, _L, C.c_double, C.c_double, C.c_double, C.c_double, _I, _F, _P]),
}


class SnLayer(C.Structure):
    """Mirror of struct AglSnLayer (include/agl.h)."""
    _fields_ = [("w", _P), ("u", _P), ("v", _P), ("w_sn", _P), ("sigma", _P), ("tmp", _P), ("u_used", _P),
                ("v_used", _P), ("g", _P), ("dw", _P), ("rows", _I), ("cols", _I)]


_lib = None
ABI_VERSION = 8     # = AGL_ABI_VERSION of include/agl.h; a library of another version is refused (shifted ctypes arguments fault on the GPU)


def load() -> C.CDLL:
    """Load libagl.so (built in-tree by `make -C csrc` / __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("AGL_LIBRARY", LIB_PATH)      # A/B builds of the same ABI (tools/); default = in-tree build
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build the HIP extension first "
                          f"(python -c 'import __graft_entry__ as g; g.build()' or make -C csrc). "
                          f"This package has no CPU or eager fallback.")
    lib = C.CDLL(path)
    lib.agl_version.restype, lib.agl_version.argtypes = _I, []
    if lib.agl_version() != ABI_VERSION:
        raise ImportError(f"{path} has ABI version {lib.agl_version()}, this binding needs {ABI_VERSION}: rebuild it (make -C csrc)")
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            raise ImportError(f"{path} does not export {name}: stale build, rebuild it (make -C csrc)")
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    assert lib.agl_sn_layer_desc_bytes() == C.sizeof(SnLayer), "AglSnLayer layout mismatch"
    _lib = lib
    return lib


# Optional per-launch timing (bench.py's roofline legs): when EVENT_LOG is a list, every call whose name is in
# EVENT_NAMES is bracketed by HIP events on the current stream (the stream it is launched on); each entry is
# (name, start, end, work, dims) with work = executed FLOPs for the convolution family (agl_conv2d_*_flops, i.e. the
# dense count minus the padded taps the position-major path skips) or algorithmic HBM bytes for the normalisation family.
EVENT_LOG = None
EVENT_NAMES = {"agl_conv2d_fwd", "agl_conv2d_fwd_stats", "agl_conv2d_bwd_data", "agl_conv2d_bwd_weight", "agl_bn_stats",
               "agl_bn_stats_from_partials", "agl_norm_apply_fwd", "agl_norm_bwd", "agl_conv2d_fwd_fold", "agl_conv2d_bwd_weight_fold",
               "agl_conv2d_fwd_addend", "agl_conv2d_fwd_shortcut", "agl_norm_bwd_fold", "agl_norm_apply_fwd_y16", "agl_norm_bwd_y16"}

# Convolution flags passed with every agl_conv2d_* call (include/agl.h AGL_CONV_*).  This is host-side state of the
# Python binding only — the C ABI has no process-wide switches.
CONV_BF16, CONV_NO_PATCH, CONV_NO_PATCH_S2, CONV_NO_POS, CONV_POS_ALL_KS, CONV_SPLIT3, CONV_ANY_GRID = 1, 2, 4, 8, 16, 32, 64
CONV_W8, CONV_PRIO = 128, 256
CONV_X_BF16 = 1 << 17      # per-call: x holds bf16 elements (set by conv2d_fwd / conv2d_bwd_weight from the tensor's dtype)
CONV_X_BLOCKED, CONV_Y_BLOCKED, CONV_MASK_BLOCKED = 1 << 21, 1 << 22, 1 << 23      # channel-blocked bf16 operands (include/agl.h)
CONV_BLOCKED = CONV_X_BLOCKED | CONV_Y_BLOCKED
CONV_DEFER_SUM = 1 << 24   # per-call: leave a plain reduction split unreduced for the caller's next kernel (include/agl.h)
CONV_Y_BF16, CONV_MASK_BF16, CONV_DY_BF16 = 1 << 18, 1 << 19, 1 << 20      # per-call: bf16 output of conv2d_fwd / bf16 pos_mask of conv2d_bwd_data (from dtypes)
CONV_FLAGS = 0


class conv_flags:
    """Context manager: `with conv_flags(CONV_BF16): ...` runs the enclosed convolutions with these flags."""

    def __init__(self, flags: int):
        self.flags = int(flags)

    def __enter__(self):
        global CONV_FLAGS
        self.prev, CONV_FLAGS = CONV_FLAGS, self.flags
        return self

    def __exit__(self, *exc):
        global CONV_FLAGS
        CONV_FLAGS = self.prev
        return False


def set_conv_precision(mode: str):
    """Default arithmetic of the convolutions this binding launches: 'f32' exact fp32 MFMA, 'bf16' bf16 MFMA operands
    with fp32 accumulation.  (Sets the AGL_CONV_BF16 bit of CONV_FLAGS, which travels with every call.)"""
    global CONV_FLAGS
    bit = {"f32": 0, "fp32": 0, "bf16": CONV_BF16}[mode]
    CONV_FLAGS = (CONV_FLAGS & ~CONV_BF16) | bit


def work_of(name, args) -> float:
    """Executed FLOPs (convolutions) or algorithmic HBM bytes (normalisation family) of one logged call."""
    lib = load()
    if name == "agl_conv2d_fwd":
        return lib.agl_conv2d_fwd_flops(*args[8:17], args[20])
    if name == "agl_conv2d_fwd_stats":
        return lib.agl_conv2d_fwd_flops(*args[8:17], args[18])
    if name == "agl_conv2d_bwd_data":
        return lib.agl_conv2d_bwd_data_flops(*args[9:19], args[21])
    if name == "agl_conv2d_bwd_weight":
        return lib.agl_conv2d_bwd_weight_flops(*args[8:20], args[21])
    if name == "agl_conv2d_fwd_fold":
        return lib.agl_conv2d_fwd_flops(*args[12:20], 0, args[21])
    if name == "agl_conv2d_fwd_shortcut":
        N, Cin, H, W, Cout, ks, pad = args[12:19]
        return lib.agl_conv2d_fwd_flops(N, Cin, H, W, Cout, ks, 1, pad, 0, args[21]) + 2.0 * N * Cout * H * W * args[8]
    if name == "agl_conv2d_fwd_addend":
        return lib.agl_conv2d_fwd_flops(*args[9:17], 0, args[19])
    if name == "agl_conv2d_bwd_weight_fold":
        return lib.agl_conv2d_bwd_weight_flops(*args[12:22], 0, args[22], args[24])
    if name == "agl_norm_bwd_fold":                  # dy, x read twice (sums, then apply); dx written — the mask costs no read
        return 4.0 * args[16] * args[17] * args[18] * 5
    if name == "agl_bn_stats":                       # one read of x (4*N*C*HW)
        return 4.0 * args[1] * args[2] * args[3]
    if name == "agl_bn_stats_from_partials":         # the statistics read of x the fused form avoids: algorithmic bytes 0
        return 0.0
    if name in ("agl_norm_apply_fwd_y16", "agl_norm_bwd_y16"):      # algorithmic bytes as the fp32 forms, whatever is stored
        return work_of(name[:-4], args)
    if name == "agl_norm_apply_fwd":                 # x read + y write (+ gamma|beta planes for SPADE, + residual)
        mode, res, N, Cc, HW = args[3], args[7], args[10], args[11], args[12]
        return 4.0 * N * Cc * HW * (2 + (2 if mode == 3 else 0) + (1 if res is not None else 0))
    if name == "agl_norm_bwd":                       # dy, x (+y if relu, + gamma|beta) read twice (sums, then apply); dx (+ d gamma|beta) written
        mode, relu, N, Cc, HW = args[5], args[9], args[14], args[15], args[16]
        reads = 2 + (1 if relu else 0) + (2 if mode == 3 else 0)
        return 4.0 * N * Cc * HW * (2 * reads + 1 + (2 if mode == 3 else 0))
    return 0.0


def bytes_of(name, args) -> float:
    """Algorithmic HBM bytes of one logged convolution call: every operand once at its stored width (x / dy / mask / addend read, y /
    dx / dw written; weights fp32 as the parameter is stored).  bench.py's roofline leg prices a launch against
    min(pipe peak, bytes-bound) with it.  0 for calls that are not convolutions."""
    def osz(h, ks, stride, pad, up=0):
        return ((h << up) + 2 * pad - ks) // stride + 1
    fl = lambda a: a if isinstance(a, int) else 0
    if name in ("agl_conv2d_fwd", "agl_conv2d_fwd_stats"):
        N, Cin, H, W, Cout, ks, stride, pad, up = args[8:17]
        flags = fl(args[20]) if name == "agl_conv2d_fwd" else fl(args[18])
        ex, ey = (2 if flags & CONV_X_BF16 else 4), (2 if flags & CONV_Y_BF16 else 4)
        acc = 1 if (name == "agl_conv2d_fwd" and args[19]) else 0
        return N * Cin * H * W * ex + Cout * Cin * ks * ks * 4 + N * Cout * osz(H, ks, stride, pad, up) * osz(W, ks, stride, pad, up) * ey * (1 + acc)
    if name == "agl_conv2d_fwd_fold":
        N, Cin, H, W, Cout, ks, stride, pad = args[12:20]
        return N * Cin * H * W * 4 + Cout * Cin * ks * ks * 4 + N * Cout * osz(H, ks, stride, pad) * osz(W, ks, stride, pad) * 4
    if name == "agl_conv2d_fwd_addend":
        N, Cin, H, W, Cout, ks, stride, pad = args[9:17]
        flags = fl(args[19])
        ex, ey = (2 if flags & CONV_X_BF16 else 4), (2 if flags & CONV_Y_BF16 else 4)
        no = N * Cout * osz(H, ks, stride, pad) * osz(W, ks, stride, pad)
        return N * Cin * H * W * ex + Cout * Cin * ks * ks * 4 + no * (ey + 4)
    if name == "agl_conv2d_fwd_shortcut":
        N, Cin, H, W, Cout, ks, pad = args[12:19]
        flags = fl(args[21])
        ex, ey = (2 if flags & CONV_X_BF16 else 4), (2 if flags & CONV_Y_BF16 else 4)
        return N * Cin * H * W * ex + Cout * Cin * ks * ks * 4 + N * Cout * H * W * ey + N * args[8] * H * W * 4
    if name == "agl_conv2d_bwd_data":
        N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad = args[9:19]
        flags = fl(args[21])
        edy, em = (2 if flags & CONV_DY_BF16 else 4), (2 if flags & CONV_MASK_BF16 else 4)
        return N * Cout * OH * OW * edy + Cout * Cin * ks * ks * 4 + N * Cin * IH * IW * (4 + (em if args[5] else 0) + (4 if args[20] else 0))
    if name == "agl_conv2d_bwd_weight":
        N, Cin, H, W, Cout, OH, OW, ks = args[8:16]
        flags = fl(args[21])
        ex, edy = (2 if flags & CONV_X_BF16 else 4), (2 if flags & CONV_DY_BF16 else 4)
        return N * Cin * H * W * ex + N * Cout * OH * OW * edy + Cout * Cin * ks * ks * 4
    if name == "agl_conv2d_bwd_weight_fold":
        N, Cin, H, W, Cout, OH, OW, ks = args[12:20]
        return N * Cin * H * W * 4 + N * Cout * OH * OW * 4 + Cout * Cin * ks * ks * 4
    return 0.0


CALL_COUNT = 0      # C-ABI calls issued through call() (bench.py reports calls per training iteration)


def call(name: str, *args):
    global CALL_COUNT
    CALL_COUNT += 1
    lib = load()
    log = EVENT_LOG
    if log is not None and name in EVENT_NAMES:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        pipe = lib.agl_conv2d_last_pipe() if name.startswith("agl_conv2d") else -1
        log.append((name, e0, e1, work_of(name, args), tuple(a for a in args if isinstance(a, int) and abs(a) < 100000)[-15:], pipe,
                    bytes_of(name, args)))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.agl_last_error().decode()}")


# --------------------------------------------------------------------------- helpers
def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("agl: operand is not on a HIP device (no CPU fallback exists)")
    if t.dtype != dtype:
        raise RuntimeError(f"agl: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("agl: operand must be contiguous")
    return t.data_ptr()


_ws_cache = {}


# Weight-gradient side streams (set by agl.trainer.Trainer around its backward passes): dw / db of a convolution depend on dy and x
# only and nothing in the backward chain reads them, so they run beside the input-gradient chain and fill its tails (partial last
# rounds of workgroups, small grids of the deep layers).  Used only for gradients accumulated straight into arena slots.
# One side stream per stream the backward chains run on: {chain stream handle: side stream}.
WGRAD_STREAMS = None


def on_wgrad_stream(fn, *tensors):
    """Run fn() on the weight-gradient stream of the current stream (after everything enqueued so far on it) and keep
    `tensors` alive for it."""
    side = WGRAD_STREAMS.get(torch.cuda.current_stream().cuda_stream) if WGRAD_STREAMS else None
    if side is None:
        return fn()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    return out


def used_on(stream, *tensors):
    """Tensors allocated on one stream and read by work queued on `stream` (chains on several streams): tell the caching
    allocator, so that freeing them does not recycle their memory before that work has run.  Ordering is the caller's job."""
    if stream is None:
        return
    for t in tensors:
        if t is not None and torch.is_tensor(t) and t.is_cuda:
            t.record_stream(stream)


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): the launches that share one are ordered on that stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def conv_out_size(h, ks, stride, pad, up=0):
    return ((h << up) + 2 * pad - ks) // stride + 1


# --------------------------------------------------------------------------- packed-weight cache
# The matrix-core patch kernels read their weights in a packed bf16 layout (include/agl.h agl_conv2d_pack_weights).  Packing per
# call cost ~640 launches / 5.6 ms per training iteration although the generator's weights are constant for a whole iteration and
# a discriminator's weight_orig between two optimiser updates; only sigma of the spectral norm changes per forward call, and the
# kernels divide by it in their epilogue.  A WeightSrc names where the values of a convolution weight come from; the packed
# buffers live on the owning parameter and are re-made when its version changes.
PACK_STATS = {"packs": 0, "hits": 0, "fused": 0, "dropped": 0}
PACK_DESC_WORDS = 14       # = AGL_PACK_DESC_WORDS of include/agl.h

# Cross-stream contract of a pack-cache entry (owner.__dict__["_agl_packs"][key] = (version, buffer, event, stream handle)):
#   * WRITERS.  An entry's buffer is written by exactly two launch sequences: the pack that creates it (WeightSrc.packed on a miss:
#     a NEW buffer, written on the current stream) and PackPlan.repack (IN PLACE, on the optimiser's stream, right behind the Adam
#     launch that changed the weights).  Both record the entry's event AFTER their last launch, so everything that lives in the buffer
#     (packed planes, the chunk shifts behind them) is covered by the event.  Nothing else may write into a cached buffer — in
#     particular no consumer call may build side tables inside it at call time: such a write is ordered by no event, and three chains
#     that share one layer (the generator's rec / rand / shift branches all read layout_encoder.c3's packs) would race on it.
#   * READERS on the writer's stream are ordered by the stream; readers on any other stream wait for the entry's event first
#     (WeightSrc.packed on a hit) and mark the buffer with record_stream.
#   * The in-place re-pack is a second writer of a buffer that earlier readers on OTHER streams may still be using: it is legal only
#     because whoever calls FlatParams.adam_step has ordered every reader of the arena's weights before the current stream (the weights
#     themselves are rewritten in place by the Adam launch, which needs exactly the same ordering).  Trainer joins every chain and
#     weight-gradient stream into the optimiser's stream before adam_step; FlatParams.adam_step documents the requirement.
# AGL_DEBUG_PACK=1 checks the contract at run time: an entry is only ever re-written on a stream that has been told (note_joined)
# that all readers were joined, and a hit never finds an entry whose event is missing.
DEBUG_PACK = os.environ.get("AGL_DEBUG_PACK", "0") == "1"
_joined_streams = set()      # (debug) stream handles on which the caller declared "all readers of the arena are ordered before this point"


def note_joined(stream=None):
    """Debug bookkeeping for the contract above: Trainer calls this after joining every chain into the optimiser's stream."""
    if DEBUG_PACK:
        _joined_streams.add((stream or torch.cuda.current_stream()).cuda_stream)


class PackPlan:
    """The packs of one parameter arena, re-made in ONE launch right after the optimiser step (agl_conv2d_pack_many) instead of one
    launch per weight and form at their first use in the next iteration (~130-150 launches per training iteration).  Entries are noted
    by WeightSrc.packed when it packs a parameter of the arena (not a derived tensor); a re-pack writes the entries' buffers IN
    PLACE on the optimiser's stream: whatever ordered the readers of the old weights before the optimiser's in-place update orders
    them before this, and later readers on other streams wait for the cache entry's event as before."""

    def __init__(self):
        self.entries = {}          # (id(owner), key) -> (owner, key, wsrc, pass_, Cin, Cout, ks, stride, flags, buf, src)
        self.noted_ptr = {}        # (id(owner), key) -> (device address of src, of buf) as they go into the descriptor table
        self.table = None          # device int64 (n, PACK_DESC_WORDS)
        self.blocks = 0
        self.order = []

    def note(self, owner, key, wsrc, pass_, Cin, Cout, ks, stride, flags, buf, src):
        k = (id(owner), key)
        old = self.entries.get(k)
        if old is None or old[9] is not buf or old[8] != flags:
            self.entries[k] = (owner, key, wsrc, pass_, Cin, Cout, ks, stride, flags, buf, src)
            self.noted_ptr[k] = (src.data_ptr(), buf.data_ptr())
            self.table = None

    def _build(self, device):
        import ctypes as C_
        rows = []
        self.order = list(self.entries.values())
        first = 0
        for (owner, key, wsrc, pass_, Cin, Cout, ks, stride, flags, buf, src) in self.order:
            row = (C_.c_longlong * PACK_DESC_WORDS)()
            call("agl_conv2d_pack_desc", ptr(src.detach()), buf.data_ptr(), buf.numel(), pass_, Cin, Cout, ks, stride, flags, C_.addressof(row))
            row[12] = first
            first += row[13]
            rows.append(list(row))
        self.blocks = first
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)

    def repack(self):
        """Re-pack every noted entry whose cache entry still holds the noted buffer AND whose parameter still lives where the descriptor
        table says (the table holds raw device pointers); returns the number of packs done.  An entry whose parameter storage moved
        (module.to(), a re-flattened arena, a .data reassignment) is dropped together with its cache entry, so the next use packs the
        new storage on the ordinary miss path instead of convolving with stale weights."""
        live = {}
        for k, e in self.entries.items():
            owner, key, wsrc, buf, src = e[0], e[1], e[2], e[9], e[10]
            cache = owner.__dict__.get("_agl_packs", {})
            hit = cache.get(key)
            if hit is None or hit[1] is not buf:
                continue
            cur_src = wsrc.base if wsrc.base is not None else wsrc.owner
            if (cur_src.data_ptr(), buf.data_ptr()) != self.noted_ptr.get(k) or cur_src.device != buf.device:
                cache.pop(key, None)
                PACK_STATS["dropped"] += 1
                continue
            live[k] = e
        if len(live) != len(self.entries):
            self.entries, self.table = live, None
            self.noted_ptr = {k: v for k, v in self.noted_ptr.items() if k in live}
        if not self.entries:
            return 0
        dev = next(iter(self.entries.values()))[9].device
        if self.table is None:
            self._build(dev)
        cur = torch.cuda.current_stream()
        if DEBUG_PACK:
            assert cur.cuda_stream in _joined_streams, "PackPlan.repack on a stream nobody declared joined (agl.lib.note_joined): readers of the old packs may still run"
            _joined_streams.discard(cur.cuda_stream)
        call("agl_conv2d_pack_many", self.table.data_ptr(), len(self.order), self.blocks, stream())
        ev = torch.cuda.Event()
        ev.record(cur)
        for (owner, key, wsrc, *_rest) in self.order:
            buf = _rest[6]
            owner.__dict__["_agl_packs"][key] = (wsrc.version(), buf, ev, cur.cuda_stream)
        PACK_STATS["fused"] += 1
        return len(self.order)



class WeightSrc:
    """Handle of the packed forms of one convolution weight.

    owner   : the nn.Parameter (of a flat arena, agl.flat.FlatParams) the values derive from; holds the cache
    version : callable -> hashable that changes whenever the values change (arena epoch, tensor version counters)
    base    : tensor to pack on a miss (None: the weight tensor handed to the call, e.g. a fresh concatenation)
    div     : optional 1-element device tensor with w = base / div (spectral norm's sigma of THIS forward call)
    tag     : distinguishes derived forms that share an owner"""

    __slots__ = ("owner", "version", "base", "div", "tag")

    def __init__(self, owner, version, base=None, div=None, tag=""):
        self.owner, self.version, self.base, self.div, self.tag = owner, version, base, div, tag

    def derived(self, tag):
        return WeightSrc(self.owner, self.version, self.base, self.div, self.tag + tag)

    def packed(self, pass_, nbytes, w, Cin, Cout, ks, stride, make=None):
        """Packed buffer for (pass, arithmetic) — from the cache, or packed now from `make()` / base / w."""
        cache = self.owner.__dict__.setdefault("_agl_packs", {})
        arith = CONV_FLAGS & (CONV_BF16 | CONV_SPLIT3)
        key = (self.tag, pass_, arith, ks, 2 if (pass_ == 1 and stride == 2) else 1)
        ver = self.version()
        hit = cache.get(key)
        cur = torch.cuda.current_stream()
        if hit is not None and hit[0] == ver and hit[1].numel() >= nbytes:
            PACK_STATS["hits"] += 1
            if DEBUG_PACK:
                assert hit[2] is not None and hit[1].device == cur.device, "pack-cache entry without an event / on another device"
            if hit[3] != cur.cuda_stream:      # packed on another stream (concurrent chains share layers): order this stream behind it,
                cur.wait_event(hit[2])         # and tell the allocator that this stream reads the buffer too (its memory
                hit[1].record_stream(cur)      # must not be handed out again before the reads queued here have run)
            return hit[1]
        src = make() if make is not None else (self.base if self.base is not None else w)
        # A miss gets a NEW buffer: the old one may still be read by launches queued on other chains' streams, which nothing orders
        # before a re-pack issued from HERE (an arbitrary consumer stream); dropping the cache entry hands it back to the stream-aware
        # allocator instead (its record_stream marks keep the memory out of circulation until those reads are done).  The one in-place
        # writer is PackPlan.repack, on the optimiser's stream, under the contract stated above PACK_STATS.
        buf = torch.empty(nbytes, dtype=torch.uint8, device=src.device)
        call("agl_conv2d_pack_weights", ptr(src.detach()), buf.data_ptr(), buf.numel(), pass_, Cin, Cout, ks, stride, CONV_FLAGS, stream())
        ev = torch.cuda.Event()
        ev.record(cur)
        PACK_STATS["packs"] += 1
        cache[key] = (ver, buf, ev, cur.cuda_stream)
        flat = getattr(self.owner, "_agl_flat", None)
        if make is None and flat is not None and getattr(flat, "pack_plan", None) is not None and src.data_ptr() == (self.base if self.base is not None else self.owner).data_ptr():
            # a parameter of a flat arena packed as it is (not a derived tensor): the arena re-packs it after its optimiser step
            flat.pack_plan.note(self.owner, key, self, pass_, Cin, Cout, ks, stride, CONV_FLAGS & (CONV_BF16 | CONV_SPLIT3), buf, src)
        return buf


_packed_bytes_memo = {}


def _packed_bytes(fn, *dims):
    key = (fn,) + dims
    v = _packed_bytes_memo.get(key)
    if v is None:
        v = _packed_bytes_memo[key] = getattr(load(), fn)(*dims)
    return v


# --------------------------------------------------------------------------- raw ops (no autograd)
def _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, stride, pad, up, make=None):
    if wsrc is None or not (CONV_FLAGS & (CONV_BF16 | CONV_SPLIT3)):
        return None, None
    nb = _packed_bytes("agl_conv2d_fwd_packed_bytes", N, Cin, H, W, Cout, ks, stride, pad, up, CONV_FLAGS)
    if not nb:
        return None, None
    return wsrc.packed(0, nb, w, Cin, Cout, ks, stride, make=make), wsrc.div


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, up=0, in_relu=False, relu=False, out=None, accumulate=False, wsrc=None, out_bf16=False,
               w_shape=None, make_base=None, out_blk=False, defer=False):
    """out_bf16: y as a torch.bfloat16 tensor (agl_conv2d_fwd_writes_bf16_y says when the kernel that runs can write it).
    defer: returns a DeferredSum instead of the tensor (AGL_CONV_DEFER_SUM: a reduction split's partial outputs may be left for the
    caller's NEXT kernel on this stream to add — lstm_gates_fwd).
    w may be None when w_shape + wsrc + make_base are given: a derived weight that only exists in packed form (cached under wsrc;
    make_base() builds the tensor to pack on a miss) — the call must then run on the matrix-core patch kernel."""
    xblk = is_blk(x)
    N, Cin, H, W = nchw_shape(x)
    Cout, Cin_w, ks, ks2 = w.shape if w is not None else w_shape
    assert Cin_w == Cin and ks == ks2, (x.shape, w_shape if w is None else w.shape)
    OH, OW = conv_out_size(H, ks, stride, pad, up), conv_out_size(W, ks, stride, pad, up)
    if out is None:
        assert not accumulate
        if out_blk:      # channel-blocked bf16 output (agl_conv2d_fwd_takes_blocked says when the kernel that runs can write it)
            out = torch.empty((N, Cout // 8, OH, OW, 8), dtype=torch.bfloat16, device=x.device)
        else:
            out = torch.empty((N, Cout, OH, OW), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    else:
        assert nchw_shape(out) == (N, Cout, OH, OW)
        out_blk = is_blk(out)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, stride, pad, up)
    ws = workspace(need, x.device) if need else None
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, stride, pad, up, make=make_base) if not (relu and accumulate) else (None, None)
    assert w is not None or pk is not None, "a derived weight needs the packed path"
    xb16 = x.dtype == torch.bfloat16      # a tensor its producer wrote in bf16 (box2_fwd(bf16=True)): matrix-core kernel only
    yb16 = out.dtype == torch.bfloat16
    call("agl_conv2d_fwd", ptr(x, x.dtype if xb16 else torch.float32), ptr(w), pk.data_ptr() if pk is not None else None, ptr(pdiv), ptr(bias),
         ptr(out, out.dtype if yb16 else torch.float32), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
         N, Cin, H, W, Cout, ks, stride, pad, up, int(in_relu), int(relu), int(accumulate),
         CONV_FLAGS | (CONV_X_BF16 if xb16 else 0) | (CONV_Y_BF16 if yb16 else 0) | (CONV_X_BLOCKED if xblk else 0) | (CONV_Y_BLOCKED if out_blk else 0)
         | (CONV_DEFER_SUM if defer else 0), stream())
    return DeferredSum(out, ws) if defer else out


class DeferredSum:
    """Result of a convolution called with defer=True: either the tensor (splits == 0) or `splits` partial outputs `stride` floats apart
    in the stream's workspace, to be added by the next kernel on that stream (agl_conv2d_deferred; include/agl.h AGL_CONV_DEFER_SUM).
    Valid until the workspace is used again — pass it to the consuming call at once."""
    __slots__ = ("out", "ws", "addr", "splits", "stride")

    def __init__(self, out, ws):
        a, n, st = C.c_void_p(0), C.c_int(0), C.c_longlong(0)
        call("agl_conv2d_deferred", C.addressof(a), C.addressof(n), C.addressof(st))
        self.out, self.ws = out, ws
        self.splits, self.stride = int(n.value), int(st.value)
        self.addr = a.value if self.splits >= 2 else out.data_ptr()
        assert self.splits == 0 or (ws is not None and ws.data_ptr() <= self.addr < ws.data_ptr() + ws.numel())


def is_blk(t) -> bool:
    """A channel-blocked bf16 tensor (N, C/8, H, W, 8) — include/agl.h AGL_CONV_X_BLOCKED."""
    return t is not None and torch.is_tensor(t) and t.dim() == 5 and t.dtype == torch.bfloat16 and t.shape[-1] == 8


def nchw_shape(t):
    """(N, C, H, W) of an NCHW or a channel-blocked tensor."""
    if is_blk(t):
        N, G, H, W, _ = t.shape
        return N, 8 * G, H, W
    return tuple(t.shape)


def to_blocked(x):
    """(N, C, H, W) fp32 / bf16 -> channel-blocked bf16 (N, C/8, H, W, 8) by torch ops (tests)."""
    N, C, H, W = x.shape
    assert C % 8 == 0
    return x.to(torch.bfloat16).view(N, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()


def to_blocked_dev(x):
    """The same by one launch (agl_to_blocked): fp32 values are rounded to bf16 (RNE), bf16 values are moved."""
    N, C, H, W = x.shape
    assert C % 8 == 0 and x.is_contiguous()
    y = torch.empty((N, C // 8, H, W, 8), dtype=torch.bfloat16, device=x.device)
    b16 = x.dtype == torch.bfloat16
    call("agl_to_blocked", ptr(x, x.dtype if b16 else torch.float32), ptr(y, torch.bfloat16), N, C, H, W, int(b16), stream())
    return y


def from_blocked(xb):
    """Channel-blocked (N, C/8, H, W, 8) -> (N, C, H, W), same dtype."""
    N, G, H, W, _ = xb.shape
    return xb.permute(0, 1, 4, 2, 3).reshape(N, G * 8, H, W).contiguous()


def conv2d_fwd_blocked(xb, w, bias=None, in_relu=False, relu=False, wsrc=None):
    """PROTOTYPE: 3x3 stride-1 "same" convolution on channel-blocked bf16 tensors (include/agl.h AGL_CONV_BLOCKED): xb (N, Cin/8, H, W, 8)
    bf16 -> (N, Cout/8, H, W, 8) bf16.  bf16 arithmetic only; raises when the kernel does not take the extents."""
    N, G, H, W, e = xb.shape
    Cin = G * 8
    Cout, Cin_w, ks, _ = w.shape
    assert e == 8 and Cin_w == Cin and xb.dtype == torch.bfloat16 and xb.is_contiguous()
    pad = (ks - 1) // 2
    flags = CONV_FLAGS | CONV_X_BF16 | CONV_Y_BF16 | CONV_BLOCKED
    if not load().agl_conv2d_fwd_takes_blocked(N, Cin, H, W, Cout, ks, 1, pad, flags):
        raise ValueError(f"conv2d_fwd_blocked: extents not taken ({N}, {Cin}, {H}, {W}) -> {Cout}, ks {ks}")
    out = torch.empty((N, Cout // 8, H, W, 8), dtype=torch.bfloat16, device=xb.device)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, 1, pad, 0)
    ws = workspace(need, xb.device) if need else None
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, 1, pad, 0)
    call("agl_conv2d_fwd", ptr(xb, torch.bfloat16), ptr(w), pk.data_ptr() if pk is not None else None, ptr(pdiv), ptr(bias),
         ptr(out, torch.bfloat16), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
         N, Cin, H, W, Cout, ks, 1, pad, 0, int(in_relu), int(relu), 0, flags, stream())
    return out


def conv2d_fwd_addend(x, w, bias, addend, stride=1, pad=0, in_relu=False, relu=False, wsrc=None, out_bf16=False, out_blk=False):
    """y = conv(x) + addend (+ bias, ReLU) out of place (agl_conv2d_fwd_addend); y bf16 on request (the fp32 sum rounded once),
    channel-blocked with out_blk."""
    N, Cin, H, W = x.shape
    Cout, _, ks, _ = w.shape
    OH, OW = conv_out_size(H, ks, stride, pad), conv_out_size(W, ks, stride, pad)
    assert tuple(addend.shape) == (N, Cout, OH, OW) and addend.dtype == torch.float32
    if out_blk:
        out_bf16 = True
        out = torch.empty((N, Cout // 8, OH, OW, 8), dtype=torch.bfloat16, device=x.device)
    else:
        out = torch.empty((N, Cout, OH, OW), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, stride, pad, 0)
    ws = workspace(need, x.device) if need else None
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, stride, pad, 0)
    xb16 = x.dtype == torch.bfloat16
    call("agl_conv2d_fwd_addend", ptr(x, x.dtype if xb16 else torch.float32), ptr(w), pk.data_ptr() if pk is not None else None, ptr(pdiv), ptr(bias),
         ptr(addend), ptr(out, out.dtype), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
         N, Cin, H, W, Cout, ks, stride, pad, int(in_relu), int(relu),
         CONV_FLAGS | (CONV_X_BF16 if xb16 else 0) | (CONV_Y_BF16 if out_bf16 else 0) | (CONV_Y_BLOCKED if out_blk else 0), stream())
    return out


def conv2d_fwd_shortcut(x, w, bias, sc_x, sc_w, sc_bias, pad=1, in_relu=False, relu=False, wsrc=None, out_bf16=False, out_blk=False):
    """y = conv3x3(x) + bias + conv1x1(sc_x; sc_w, sc_bias) with the few-channel shortcut evaluated in the epilogue (agl_conv2d_fwd_shortcut).
    x may be channel-blocked (then y can be, with out_blk)."""
    xblk = is_blk(x)
    N, Cin, H, W = nchw_shape(x)
    Cout, _, ks, _ = w.shape
    sc_cin = sc_x.shape[1]
    assert tuple(sc_x.shape) == (N, sc_cin, H, W) and sc_w.numel() == Cout * sc_cin
    if out_blk:
        out_bf16 = True
        out = torch.empty((N, Cout // 8, H, W, 8), dtype=torch.bfloat16, device=x.device)
    else:
        out = torch.empty((N, Cout, H, W), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, 1, pad, 0)
    ws = workspace(need, x.device) if need else None
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, 1, pad, 0)
    xb16 = x.dtype == torch.bfloat16
    call("agl_conv2d_fwd_shortcut", ptr(x, x.dtype if xb16 else torch.float32), ptr(w), pk.data_ptr() if pk is not None else None, ptr(pdiv),
         ptr(bias), ptr(sc_x), ptr(sc_w), ptr(sc_bias), sc_cin, ptr(out, out.dtype), ws.data_ptr() if ws is not None else None,
         ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, ks, pad, int(in_relu), int(relu),
         CONV_FLAGS | (CONV_X_BF16 if xb16 else 0) | (CONV_Y_BF16 if out_bf16 else 0) | (CONV_X_BLOCKED if xblk else 0) | (CONV_Y_BLOCKED if out_blk else 0),
         stream())
    return out


def conv2d_fwd_stats(x, w, bias=None, stride=1, pad=0, up=0, in_relu=False, wsrc=None):
    """conv2d_fwd that also returns the BatchNorm partial rows of its output: (y, partials or None, rows)."""
    N, Cin, H, W = x.shape
    Cout, Cin_w, ks, ks2 = w.shape
    assert Cin_w == Cin and ks == ks2, (x.shape, w.shape)
    OH, OW = conv_out_size(H, ks, stride, pad, up), conv_out_size(W, ks, stride, pad, up)
    out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, stride, pad, up)
    ws = workspace(need, x.device) if need else None
    nst = load().agl_conv2d_fwd_stats_floats(N, Cout, OH, OW)
    stats = torch.empty(nst, dtype=torch.float32, device=x.device)
    rows = C.c_int(0)
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, stride, pad, up)
    xb16 = x.dtype == torch.bfloat16
    call("agl_conv2d_fwd_stats", ptr(x, x.dtype if xb16 else torch.float32), ptr(w), pk.data_ptr() if pk is not None else None, ptr(pdiv),
         ptr(bias), ptr(out), ws.data_ptr() if ws is not None else None,
         ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, ks, stride, pad, up, int(in_relu), CONV_FLAGS | (CONV_X_BF16 if xb16 else 0),
         stats.data_ptr(), nst, C.addressof(rows), stream())
    return out, (stats if rows.value > 0 else None), rows.value


# ---- normalise-modulate folded into the consuming convolution (include/agl.h: agl_norm_fold_table, agl_conv2d_fwd_fold, ...)
class Fold:
    """Tables of a folded normalise-modulate: v = (x - mean[c]) * scale[r][c] + shift[r][c], r = object (per_n) or 0."""
    __slots__ = ("mean", "scale", "shift", "per_n")

    def __init__(self, mean, scale, shift, per_n):
        self.mean, self.scale, self.shift, self.per_n = mean, scale, shift, int(per_n)


def norm_fold_table(mean, rstd, mode, p0, p1, labels, N):
    Cc = mean.numel()
    rows = N if mode == 2 else 1
    scale = torch.empty((rows, Cc), dtype=torch.float32, device=mean.device)
    shift = torch.empty_like(scale)
    call("agl_norm_fold_table", ptr(mean), ptr(rstd), mode, ptr(p0), ptr(p1), ptr(labels, torch.int64), N, Cc, ptr(scale), ptr(shift), stream())
    return Fold(mean, scale, shift, mode == 2)


def conv_fold_ok(N, Cin, H, W, Cout, ks, stride, pad, need_bww=True):
    """True when conv(relu?(norm(x))) can run with the norm folded into the convolution's staging (forward and weight gradient)."""
    if not (CONV_FLAGS & (CONV_BF16 | CONV_SPLIT3)) or (CONV_FLAGS & CONV_NO_PATCH):
        return False
    lib = load()
    if not lib.agl_conv2d_fwd_fold_ok(N, Cin, H, W, Cout, ks, stride, pad, CONV_FLAGS):
        return False
    OH, OW = conv_out_size(H, ks, stride, pad), conv_out_size(W, ks, stride, pad)
    return (not need_bww) or bool(lib.agl_conv2d_bwd_weight_fold_ok(N, Cin, H, W, Cout, OH, OW, ks, stride, pad, CONV_FLAGS))


def conv2d_fwd_fold(x, fold, w, bias=None, stride=1, pad=0, in_relu=False, wsrc=None, want_stats=False):
    """conv(relu?(fold(x))) -> (y, BatchNorm partial rows of y or None, row count)."""
    N, Cin, H, W = x.shape
    Cout, Cin_w, ks, _ = w.shape
    assert Cin_w == Cin
    OH, OW = conv_out_size(H, ks, stride, pad), conv_out_size(W, ks, stride, pad)
    out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, stride, pad, 0)
    ws = workspace(need, x.device) if need else None
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, stride, pad, 0)
    stats = rows = None
    nst = 0
    if want_stats:
        nst = load().agl_conv2d_fwd_stats_floats(N, Cout, OH, OW)
        stats = torch.empty(nst, dtype=torch.float32, device=x.device)
        rows = C.c_int(0)
    call("agl_conv2d_fwd_fold", ptr(x), ptr(fold.mean), ptr(fold.scale), ptr(fold.shift), fold.per_n, ptr(w),
         pk.data_ptr() if pk is not None else None, ptr(pdiv), ptr(bias), ptr(out), ws.data_ptr() if ws is not None else None,
         ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, ks, stride, pad, int(in_relu), CONV_FLAGS,
         stats.data_ptr() if stats is not None else None, nst, C.addressof(rows) if rows is not None else None, stream())
    if rows is not None and rows.value > 0:
        return out, stats, rows.value
    return out, None, 0


def conv2d_bwd_weight_fold(dy, x, fold, ks, stride=1, pad=0, in_relu=False, out=None, accumulate=False, dbias=None, dbias_accumulate=None):
    if dbias_accumulate is None:
        dbias_accumulate = accumulate
    N, Cout, OH, OW = dy.shape
    _, Cin, H, W = x.shape
    if out is None:
        assert not accumulate
        out = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=dy.device)
    need = load().agl_conv2d_bwd_weight_ws_bytes(N, Cin, Cout, ks, OH, OW)
    if accumulate:
        need = max(need, Cout * Cin * ks * ks * 4)
    ws = workspace(need, dy.device) if need else None
    done = C.c_int(0)
    call("agl_conv2d_bwd_weight_fold", ptr(dy), ptr(x), ptr(fold.mean), ptr(fold.scale), ptr(fold.shift), fold.per_n, ptr(out), ptr(dbias),
         int(dbias_accumulate), C.addressof(done) if dbias is not None else None, ws.data_ptr() if ws is not None else None,
         ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, OH, OW, ks, stride, pad, int(in_relu), int(accumulate), CONV_FLAGS, stream())
    if dbias is not None and not done.value:
        channel_sum(dy, out=dbias, accumulate=dbias_accumulate)
    return out


# ---- SPADE applied by its consumer's staging pass (include/agl.h: agl_spade_cells, agl_conv2d_fwd_spade, ...)
class SpadeFold:
    """Operands of a SPADE modulation applied inside the convolution that reads it: per-channel batch mean / rstd, the blocked cell
    table of (1 + gamma | beta) (agl_spade_cells) and the pixel -> class-grid map."""
    __slots__ = ("mean", "rstd", "cells", "map", "G")

    def __init__(self, mean, rstd, gb, gmap):
        N, C2, G, G2 = gb.shape
        assert G == G2 and C2 % 16 == 0 and gmap.dtype == torch.int32
        cells = torch.empty((N, C2 // 16, G * G, 16), dtype=torch.float32, device=gb.device)
        call("agl_spade_cells", ptr(gb), N, C2 // 2, G, ptr(cells), stream())
        self.mean, self.rstd, self.cells, self.map, self.G = mean, rstd, cells, gmap, G


def conv_spade_ok(N, Cin, H, W, Cout, ks, stride, pad, need_bww=True):
    """True when conv(relu?(SPADE(x))) runs with the modulation applied while the convolution stages x (forward and weight gradient):
    bf16 arithmetic, the 5x5 and the few-output-channel 7x7 forms."""
    if not (CONV_FLAGS & CONV_BF16) or (CONV_FLAGS & CONV_NO_PATCH):
        return False
    lib = load()
    if not lib.agl_conv2d_fwd_spade_ok(N, Cin, H, W, Cout, ks, stride, pad, CONV_FLAGS):
        return False
    return (not need_bww) or bool(lib.agl_conv2d_bwd_weight_spade_ok(N, Cin, H, W, Cout, H, W, ks, stride, pad, CONV_FLAGS))


def conv2d_fwd_spade(x, sp, w, bias=None, stride=1, pad=0, in_relu=False, wsrc=None, want_stats=False):
    """conv(relu?(SPADE(x))) -> (y, BatchNorm partial rows of y or None, row count)."""
    N, Cin, H, W = x.shape
    Cout, Cin_w, ks, _ = w.shape
    assert Cin_w == Cin and sp.map.numel() == H == W
    OH, OW = conv_out_size(H, ks, stride, pad), conv_out_size(W, ks, stride, pad)
    out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device)
    need = load().agl_conv2d_fwd_ws_bytes(N, Cin, H, W, Cout, ks, stride, pad, 0)
    ws = workspace(need, x.device) if need else None
    pk, pdiv = _fwd_pack(wsrc, w, N, Cin, H, W, Cout, ks, stride, pad, 0) if Cout > 4 else (None, None)
    stats = rows = None
    nst = 0
    if want_stats and Cout > 4:
        nst = load().agl_conv2d_fwd_stats_floats(N, Cout, OH, OW)
        stats = torch.empty(nst, dtype=torch.float32, device=x.device)
        rows = C.c_int(0)
    call("agl_conv2d_fwd_spade", ptr(x), ptr(sp.mean), ptr(sp.rstd), ptr(sp.cells), ptr(sp.map, torch.int32), sp.G, ptr(w),
         pk.data_ptr() if pk is not None else None, ptr(pdiv), ptr(bias), ptr(out), ws.data_ptr() if ws is not None else None,
         ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, ks, stride, pad, int(in_relu), CONV_FLAGS,
         stats.data_ptr() if stats is not None else None, nst, C.addressof(rows) if rows is not None else None, stream())
    if rows is not None and rows.value > 0:
        return out, stats, rows.value
    return out, None, 0


def conv2d_bwd_weight_spade(dy, x, sp, ks, stride=1, pad=0, in_relu=False, out=None, accumulate=False, dbias=None, dbias_accumulate=None):
    if dbias_accumulate is None:
        dbias_accumulate = accumulate
    N, Cout, OH, OW = dy.shape
    _, Cin, H, W = x.shape
    if out is None:
        assert not accumulate
        out = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=dy.device)
    need = load().agl_conv2d_bwd_weight_ws_bytes(N, Cin, Cout, ks, OH, OW)
    if accumulate:
        need = max(need, Cout * Cin * ks * ks * 4)
    ws = workspace(need, dy.device) if need else None
    done = C.c_int(0)
    call("agl_conv2d_bwd_weight_spade", ptr(dy), ptr(x), ptr(sp.mean), ptr(sp.rstd), ptr(sp.cells), ptr(sp.map, torch.int32), sp.G, ptr(out),
         ptr(dbias), int(dbias_accumulate), C.addressof(done) if dbias is not None else None, ws.data_ptr() if ws is not None else None,
         ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, OH, OW, ks, stride, pad, int(in_relu), int(accumulate), CONV_FLAGS, stream())
    if dbias is not None and not done.value:
        channel_sum(dy, out=dbias, accumulate=dbias_accumulate)
    return out


def norm_bwd_spade(dy, x, mean, rstd, gb, relu, batch_stats, dgb, gb_map=None, gb_lo=None):
    """Backward of SPADE's modulate(+ReLU) whose output was never stored (the consumer applied it): the mask is recomputed."""
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    dx = torch.empty_like(x)
    nb = load().agl_norm_bwd_ws_bytes(N, Cc)
    ws = workspace(nb, x.device)
    call("agl_norm_bwd_spade", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gb), int(relu), int(batch_stats), ptr(dx), ptr(dgb), N, Cc, HW,
         ptr(gb_map, torch.int32), ptr(gb_lo, torch.int32), x.shape[-1] if gb_map is not None else 0,
         gb.shape[-1] if gb_map is not None else 0, ws.data_ptr(), ws.numel(), stream())
    return dx


def norm_bwd_fold(dy, x, mean, rstd, fold, mode, p0, p1, labels, relu, batch_stats, dp0=None, dp1=None, param_accumulate=False):
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    dx = torch.empty_like(x)
    nb = load().agl_norm_bwd_ws_bytes(N, Cc)
    ws = workspace(nb, x.device)
    call("agl_norm_bwd_fold", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(fold.scale), ptr(fold.shift), fold.per_n, mode, ptr(p0), ptr(p1),
         ptr(labels, torch.int64), int(relu), int(batch_stats), ptr(dx), ptr(dp0), ptr(dp1), N, Cc, HW, p0.shape[0] if mode == 2 else 0,
         int(param_accumulate), ws.data_ptr(), ws.numel(), stream())
    return dx


def bn_stats_from_partials(partials, rows, Cc, count, eps, momentum, running_mean=None, running_var=None, nbt=None, moments=None):
    mean = torch.empty(Cc, dtype=torch.float32, device=partials.device)
    rstd = torch.empty_like(mean)
    call("agl_bn_stats_from_partials", partials.data_ptr(), rows, Cc, count, eps, momentum, ptr(running_mean), ptr(running_var),
         ptr(nbt, torch.int64), ptr(mean), ptr(rstd), ptr(moments, torch.float64), stream())
    return mean, rstd


def bn_running_update(moments, momentum, running_mean, running_var, nbt=None):
    """Re-apply the running-statistics update of the statistics call that wrote `moments` (bit-identical to recomputing it)."""
    call("agl_bn_running_update", ptr(moments, torch.float64), running_mean.numel(), momentum, ptr(running_mean), ptr(running_var),
         ptr(nbt, torch.int64), stream())


class BnUpdate(C.Structure):
    """AglBnUpdate of include/agl.h."""
    _fields_ = [("moments", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
                ("C", C.c_int), ("momentum", C.c_float)]


BN_UPDATE_MAX = 24


def bn_running_update_many(entries, momentum):
    """bn_running_update for a list of (moments, running_mean, running_var, nbt) in as few launches as possible (array order kept)."""
    for i in range(0, len(entries), BN_UPDATE_MAX):
        chunk = entries[i:i + BN_UPDATE_MAX]
        arr = (BnUpdate * len(chunk))()
        for k, (mom, rm, rv, nbt) in enumerate(chunk):
            arr[k].moments, arr[k].running_mean, arr[k].running_var = ptr(mom, torch.float64), ptr(rm), ptr(rv)
            arr[k].num_batches_tracked = ptr(nbt, torch.int64)
            arr[k].C, arr[k].momentum = rm.numel(), momentum
        call("agl_bn_running_update_multi", C.addressof(arr), len(chunk), stream())


def bwd_data_packed_bytes(N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad):
    """Bytes of the packed weights when this input-gradient call runs on the matrix-core patch kernel, else 0."""
    if not (CONV_FLAGS & (CONV_BF16 | CONV_SPLIT3)):
        return 0
    return _packed_bytes("agl_conv2d_bwd_data_packed_bytes", N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad, CONV_FLAGS)


def conv2d_bwd_data(dy, w, in_hw, stride=1, pad=0, pos_mask=None, out=None, accumulate=False, wsrc=None, w_shape=None, make_w=None,
                    make_base=None, defer=False):
    """dx (N,Cin,IH,IW) from dy (N,Cout,OH,OW), w (Cout,Cin,ks,ks).  Also ConvTranspose2d forward.
    w may be None when w_shape + make_w are given: the weights are then a derived tensor that is only materialised when needed —
    make_w() builds w itself (for launches that read it), make_base() the tensor whose packed form is cached under wsrc
    (w = make_base() / wsrc.div)."""
    N, Cout, OH, OW = dy.shape
    Cout_w, Cin, ks, _ = w.shape if w is not None else w_shape
    assert Cout_w == Cout, (dy.shape, w_shape if w is None else w.shape)
    IH, IW = in_hw
    if out is None:
        assert not accumulate
        out = torch.empty((N, Cin, IH, IW), dtype=torch.float32, device=dy.device)
    need = load().agl_conv2d_bwd_data_ws_bytes(N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad)
    ws = workspace(need, dy.device) if need else None
    pk = pdiv = None
    if wsrc is not None:
        nb = bwd_data_packed_bytes(N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad)
        if nb:
            pk, pdiv = wsrc.packed(1, nb, w, Cin, Cout, ks, stride, make=make_base), wsrc.div
    if w is None and pk is None:
        w = make_w()
    mb16 = pos_mask is not None and pos_mask.dtype == torch.bfloat16
    mblk = is_blk(pos_mask)                # (a channel-blocked bf16 mask: the blocked h / block input of a discriminator block)
    db16 = dy.dtype == torch.bfloat16      # (the bf16-stored input of a transposed convolution: this call is its forward)
    call("agl_conv2d_bwd_data", ptr(dy, dy.dtype if db16 else torch.float32), ptr(w), pk.data_ptr() if pk is not None else None, ptr(pdiv), None,
         ptr(pos_mask, torch.bfloat16 if mb16 else torch.float32), ptr(out),
         ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad,
         0, int(accumulate), CONV_FLAGS | (CONV_MASK_BF16 if mb16 else 0) | (CONV_X_BF16 if db16 else 0) | (CONV_MASK_BLOCKED if mblk else 0)
         | (CONV_DEFER_SUM if defer else 0), stream())
    return DeferredSum(out, ws) if defer else out


def conv2d_bwd_weight(dy, x, ks, stride=1, pad=0, up=0, in_relu=False, out=None, accumulate=False, dbias=None, dbias_accumulate=None):
    """dw (+)= weight gradient.  dbias (optional, (Cout,) tensor): also the bias gradient sum_{n,oh,ow} dy — by the weight-gradient
    kernel itself where it stages dy anyway, else by agl_channel_sum; added in place when dbias_accumulate (default: like dw)."""
    if dbias_accumulate is None:
        dbias_accumulate = accumulate
    N, Cout, OH, OW = dy.shape
    xblk = is_blk(x)
    _, Cin, H, W = nchw_shape(x)
    if out is None:
        assert not accumulate
        out = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=dy.device)
    need = load().agl_conv2d_bwd_weight_ws_bytes(N, Cin, Cout, ks, OH, OW)
    if accumulate:
        need = max(need, Cout * Cin * ks * ks * 4)
    ws = workspace(need, dy.device) if need else None
    done = C.c_int(0)
    xb16, db16 = x.dtype == torch.bfloat16, dy.dtype == torch.bfloat16
    call("agl_conv2d_bwd_weight", ptr(dy, dy.dtype if db16 else torch.float32), ptr(x, x.dtype if xb16 else torch.float32), ptr(out), ptr(dbias),
         int(dbias_accumulate), C.addressof(done) if dbias is not None else None,
         ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, N, Cin, H, W, Cout, OH, OW, ks, stride, pad, up,
         int(in_relu), int(accumulate), CONV_FLAGS | (CONV_X_BF16 if xb16 else 0) | (CONV_DY_BF16 if db16 else 0) | (CONV_X_BLOCKED if xblk else 0),
         stream())
    if dbias is not None and not done.value:
        assert not db16
        channel_sum(dy, out=dbias, accumulate=dbias_accumulate)
    return out


def norm_output_as_bf16(N, Cc, H, W, consumer, Cout, ks=0, stride=1, pad=0, need_bww=True):
    """True when the normalised (+ReLU'd) tensor (N, C, H, W) that feeds ONLY `consumer` — "conv": a convolution C -> Cout (forward
    and weight gradient read it as x); "convT": ConvTranspose2d(4, 2, 1) C -> Cout (its forward is an input-gradient call with the
    tensor in the dy role, its weight gradient has it in the dy role too) — can be stored as bf16 in bf16 arithmetic: every kernel
    that would read it runs on the matrix cores and takes that form."""
    if not (CONV_FLAGS & CONV_BF16) or (CONV_FLAGS & CONV_NO_PATCH) or W % 8 != 0:
        return False
    lib = load()
    if consumer == "conv":
        OH, OW = conv_out_size(H, ks, stride, pad), conv_out_size(W, ks, stride, pad)
        if not lib.agl_conv2d_fwd_takes_bf16_x(N, Cc, H, W, Cout, ks, stride, pad, CONV_FLAGS):
            return False
        return (not need_bww) or bool(lib.agl_conv2d_bwd_weight_takes_bf16_x(N, Cc, H, W, Cout, OH, OW, ks, stride, pad, CONV_FLAGS))
    if consumer == "convT":      # C-ABI roles: dx (N, Cout, 2H, 2W) from dy = the tensor (N, C, H, W), weights [C][Cout][4][4]
        if not lib.agl_conv2d_bwd_data_takes_bf16_dy(N, Cout, 2 * H, 2 * W, Cc, H, W, 4, 2, 1, CONV_FLAGS):
            return False
        return (not need_bww) or bool(lib.agl_conv2d_bwd_weight_takes_bf16_dy(N, Cout, 2 * H, 2 * W, Cc, H, W, 4, 2, 1, CONV_FLAGS))
    return False


def channel_sum(x, out=None, accumulate=False):
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    if out is None:
        out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = workspace(load().agl_channel_sum_ws_bytes(Cc), x.device)
    call("agl_channel_sum", ptr(x), ptr(out), N, Cc, HW, int(accumulate), ws.data_ptr(), ws.numel(), stream())
    return out


def bn_stats(x, eps, momentum, running_mean=None, running_var=None, nbt=None, moments=None):
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    nb = load().agl_bn_stats_ws_bytes(N, Cc, HW)
    ws = workspace(nb, x.device)
    call("agl_bn_stats", ptr(x), N, Cc, HW, eps, momentum, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
         ptr(nbt, torch.int64), ptr(moments, torch.float64), ws.data_ptr(), ws.numel(), stream())
    return mean, rstd


def bn_stats_eval(running_mean, running_var, eps):
    mean = torch.empty_like(running_mean)
    rstd = torch.empty_like(running_mean)
    call("agl_bn_stats_eval", ptr(running_mean), ptr(running_var), running_mean.numel(), eps, ptr(mean), ptr(rstd), stream())
    return mean, rstd


def norm_apply_fwd(x, mean, rstd, mode, p0, p1, labels, residual, relu, gb_map=None, out_bf16=False):
    """gb_map (mode 3 only): int32 device tensor of W entries — p0 is then (N, 2C, s, s) on a coarser grid, read through the map.
    out_bf16: y as a torch.bfloat16 tensor (for consumers that are bf16-mode convolutions; the caller asked their predicates)."""
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    y = torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    call("agl_norm_apply_fwd_y16" if out_bf16 else "agl_norm_apply_fwd", ptr(x), ptr(mean), ptr(rstd), mode, ptr(p0), ptr(p1),
         ptr(labels, torch.int64), ptr(residual), int(relu), ptr(y, y.dtype), N, Cc, HW, ptr(gb_map, torch.int32),
         x.shape[-1] if gb_map is not None else 0, p0.shape[-1] if gb_map is not None else 0, stream())
    return y


def norm_bwd(dy, x, y, mean, rstd, mode, p0, p1, labels, relu, batch_stats, dp0=None, dp1=None, param_accumulate=False, gb_map=None,
             gb_lo=None):
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    dx = torch.empty_like(x)
    nb = load().agl_norm_bwd_ws_bytes(N, Cc)
    ws = workspace(nb, x.device)
    y16 = y is not None and y.dtype == torch.bfloat16
    call("agl_norm_bwd_y16" if y16 else "agl_norm_bwd", ptr(dy), ptr(x), ptr(y, torch.bfloat16 if y16 else torch.float32), ptr(mean), ptr(rstd),
         mode, ptr(p0), ptr(p1),
         ptr(labels, torch.int64), int(relu), int(batch_stats), ptr(dx), ptr(dp0), ptr(dp1), N, Cc, HW,
         p0.shape[0] if mode == 2 else 0, int(param_accumulate), ptr(gb_map, torch.int32), ptr(gb_lo, torch.int32),
         x.shape[-1] if gb_map is not None else 0,
         p0.shape[-1] if gb_map is not None else 0, ws.data_ptr(), ws.numel(), stream())
    return dx


def crop_fwd(feats, boxes, o2i, HH, WW, align=False):
    N, Cc, H, W = feats.shape
    B = boxes.shape[0]
    out = torch.empty((B, Cc, HH, WW), dtype=torch.float32, device=feats.device)
    call("agl_crop_fwd", ptr(feats), ptr(boxes), ptr(o2i, torch.int64), ptr(out), N, B, Cc, H, W, HH, WW, int(align), stream())
    return out


def box_map_to_device(box_to_img, device):
    """The box -> image map on the device.  When it arrives on the CPU (as in the reference's loop, train64.py:149-151) its order is
    checked here for free: a non-decreasing map lets the crop backward run as a fixed-order gather (agl_crop_bwd_sorted)."""
    t = box_to_img.to(device).long()
    if not box_to_img.is_cuda:
        t._agl_sorted = bool(box_to_img.numel() < 2 or bool((box_to_img[1:] >= box_to_img[:-1]).all()))
    return t


def crop_bwd(dout, boxes, o2i, feat_shape, align=False):
    """Sorted box -> image map (marked by box_map_to_device): a gather in fixed order; else a scatter with float atomics."""
    N, Cc, H, W = feat_shape
    B, _, HH, WW = dout.shape
    dfeats = torch.zeros(feat_shape, dtype=torch.float32, device=dout.device)
    name = "agl_crop_bwd_sorted" if getattr(o2i, "_agl_sorted", False) else "agl_crop_bwd"
    call(name, ptr(dout), ptr(boxes), ptr(o2i, torch.int64), ptr(dfeats), N, B, Cc, H, W, HH, WW, int(align), stream())
    return dfeats


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    call("agl_relu_bwd", ptr(dy), ptr(y), ptr(dx), dy.numel(), stream())
    return dx


def axpby(a, b, alpha=1.0, beta=1.0, out=None):
    if out is None:
        out = torch.empty_like(a)
    call("agl_axpby", ptr(a), ptr(b), alpha, beta, ptr(out), a.numel(), stream())
    return out


def gather_rows(src, rows, out=None, accumulate=False):
    R = rows.numel()
    ln = src.numel() // src.shape[0]
    if out is None:
        out = torch.empty((R,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    call("agl_gather_rows", ptr(src), ptr(rows, torch.int64), ptr(out), R, ln, int(accumulate), stream())
    return out


def scatter_rows(src, rows, out):
    ln = src.numel() // src.shape[0]
    call("agl_scatter_rows", ptr(src), ptr(rows, torch.int64), ptr(out), rows.numel(), ln, stream())
    return out


def box2_fwd(x, bf16=False):
    """bf16=True: the filtered map as a torch.bfloat16 tensor, for convolutions called with it as x (CONV_X_BF16)."""
    N, Cc, H, W = x.shape
    if bf16:
        xb = torch.empty((N, Cc, H + 1, W + 1), dtype=torch.bfloat16, device=x.device)
        call("agl_box2_fwd_bf16", ptr(x), ptr(xb, torch.bfloat16), N * Cc, H, W, stream())
        return xb
    xb = torch.empty((N, Cc, H + 1, W + 1), dtype=torch.float32, device=x.device)
    call("agl_box2_fwd", ptr(x), ptr(xb), N * Cc, H, W, stream())
    return xb


def first_conv_output_as_bf16(N, Cin0, H, W, C, Cout, ks0, need_bww, need_bwd_data):
    """True when h = relu(conv_ks0(x (N, Cin0 <= 4, H, W) -> C channels)), consumed only by a 3x3 / stride-1 / pad-1 convolution
    C -> Cout (forward, weight gradient, and — as the ReLU mask — its input gradient), can be stored as bf16 in bf16 arithmetic with
    identical results: the few-input-channel stream kernel writes it, the matrix-core kernels read it."""
    if not (CONV_FLAGS & CONV_BF16) or (CONV_FLAGS & CONV_NO_PATCH) or Cin0 > 4 or W % 4 != 0 or ks0 not in (1, 3):
        return False
    lib = load()
    if not lib.agl_conv2d_fwd_packed_bytes(N, C, H, W, Cout, 3, 1, 1, 0, CONV_FLAGS):
        return False
    if need_bww and not lib.agl_conv2d_bwd_weight_takes_bf16_x(N, C, H, W, Cout, H, W, 3, 1, 1, CONV_FLAGS):
        return False
    if need_bwd_data and not lib.agl_conv2d_bwd_data_takes_bf16_mask(N, C, H, W, Cout, H, W, 3, 1, 1, CONV_FLAGS):
        return False
    return True


def box_input_as_bf16(N, Cin, H, W, Cout, need_bww):
    """True when the 3x3 / stride-2 / unpadded convolution of an (N, Cin, H, W) box-filtered map (and, with need_bww, its weight
    gradient) runs on the matrix-core kernels in bf16 arithmetic — the producer may then write the map in bf16 (identical results)."""
    if not (CONV_FLAGS & CONV_BF16) or (CONV_FLAGS & CONV_NO_PATCH) or W % 4 != 1:
        return False
    OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    if not load().agl_conv2d_fwd_packed_bytes(N, Cin, H, W, Cout, 3, 2, 0, 0, CONV_FLAGS):
        return False
    return (not need_bww) or bool(load().agl_conv2d_bwd_weight_takes_bf16_x(N, Cin, H, W, Cout, OH, OW, 3, 2, 0, CONV_FLAGS))


def box2_bwd(dxb, mask=None):
    N, Cc, HB, WB = dxb.shape
    dx = torch.empty((N, Cc, HB - 1, WB - 1), dtype=torch.float32, device=dxb.device)
    call("agl_box2_bwd", ptr(dxb), ptr(mask), ptr(dx), N * Cc, HB - 1, WB - 1, stream())
    return dx


def avgpool2_fwd(x, in_relu=False):
    """x fp32 or (a discriminator block output stored as bf16: NCHW or channel-blocked) bfloat16; the result is fp32 NCHW."""
    N, Cc, H, W = nchw_shape(x)
    y = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
    if is_blk(x):
        call("agl_avgpool2_fwd_xblk", ptr(x, torch.bfloat16), ptr(y), N, Cc, H, W, int(in_relu), stream())
    elif x.dtype == torch.bfloat16:
        call("agl_avgpool2_fwd_x16", ptr(x, torch.bfloat16), ptr(y), N * Cc, H, W, int(in_relu), stream())
    else:
        call("agl_avgpool2_fwd", ptr(x), ptr(y), N * Cc, H, W, int(in_relu), stream())
    return y


def avgpool2_bwd(dy, x_or_shape, in_relu=False, out=None, accumulate=False):
    x = x_or_shape if isinstance(x_or_shape, torch.Tensor) else None
    shape = nchw_shape(x) if x is not None else x_or_shape
    N, Cc, H, W = shape
    if out is None:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=dy.device)
    if is_blk(x):
        assert in_relu
        call("agl_avgpool2_bwd_xblk", ptr(dy), ptr(x, torch.bfloat16), ptr(out), N, Cc, H, W, int(accumulate), stream())
    elif x is not None and x.dtype == torch.bfloat16:
        assert in_relu
        call("agl_avgpool2_bwd_x16", ptr(dy), ptr(x, torch.bfloat16), ptr(out), N * Cc, H, W, int(accumulate), stream())
    else:
        call("agl_avgpool2_bwd", ptr(dy), ptr(x), ptr(out), N * Cc, H, W, int(in_relu), int(accumulate), stream())
    return out


def upsample_fwd(x, k):
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, H << k, W << k), dtype=torch.float32, device=x.device)
    call("agl_upsample_nearest_fwd", ptr(x), ptr(y), N * Cc, H, W, k, stream())
    return y


def upsample_bwd(dy, k, out=None, accumulate=False):
    N, Cc, OH, OW = dy.shape
    H, W = OH >> k, OW >> k
    if out is None:
        out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dy.device)
    call("agl_upsample_nearest_bwd", ptr(dy), ptr(out), N * Cc, H, W, k, int(accumulate), stream())
    return out


def sum_hw_fwd(x, in_relu=False, scale=1.0):
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    y = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
    call("agl_sum_hw_fwd", ptr(x), ptr(y), N * Cc, HW, int(in_relu), scale, stream())
    return y


def sum_hw_bwd(dy, x, in_relu=False, scale=1.0):
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    dx = torch.empty_like(x)
    call("agl_sum_hw_bwd", ptr(dy), ptr(x), ptr(dx), N * Cc, HW, int(in_relu), scale, stream())
    return dx


def _sum_args(t):
    """(address, splits, stride) of a tensor or a DeferredSum operand."""
    if isinstance(t, DeferredSum):
        return t.addr, t.splits, t.stride
    return ptr(t), 1, 0


def lstm_gates_fwd(ccx, rows, cch, c_prev, h, c, gates, B, hid, S):
    """cch: tensor, None, or the DeferredSum of the recurrence convolution (its slabs are added here)."""
    a, n, st = _sum_args(cch)
    call("agl_lstm_gates_fwd_sum", ptr(ccx), ptr(rows, torch.int64), a, n, st, ptr(c_prev), ptr(h), ptr(c), ptr(gates), B, hid, S, stream())


def lstm_gates_bwd(dh_a, dh_b, Bb, dc_next, Bc, gates, c_prev, c, dcc, dc_prev, B, hid, S):
    """dh_b: tensor, None, or the DeferredSum of the recurrence's input-gradient convolution."""
    a, n, st = _sum_args(dh_b)
    call("agl_lstm_gates_bwd_sum", ptr(dh_a), a, n, st, Bb, ptr(dc_next), Bc, ptr(gates), ptr(c_prev), ptr(c), ptr(dcc),
         ptr(dc_prev), B, hid, S, stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    call("agl_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, stream())


def attr_estimate(logits, attribute):
    """train64.py:156-166 on device: un-annotated rows get one-hot(argmax(logits)); annotated rows keep their attributes."""
    est = torch.empty_like(attribute)
    O, A = attribute.shape
    call("agl_attr_estimate", ptr(logits.contiguous()), ptr(attribute), ptr(est), O, A, stream())
    return est


def rasterize_boxes(boxes, R):
    """(O,4) [x0,y0,x1,y1] in [0,1] -> (O,1,R,R) {0,1} masks (data/vg_custom_mask.py:136 semantics)."""
    O = boxes.shape[0]
    masks = torch.empty((O, 1, R, R), dtype=torch.float32, device=boxes.device)
    call("agl_rasterize_boxes", ptr(boxes.contiguous()), ptr(masks), O, R, stream())
    return masks


def layout_from_boxes(boxes, R):
    """(O,4) boxes -> (boxes_shift (O,4), masks (O,1,R,R), masks_shift (O,1,R,R)) on device (data/vg_custom_mask.py:136-158)."""
    O = boxes.shape[0]
    boxes = boxes.contiguous()
    bs = torch.empty_like(boxes)
    masks = torch.empty((O, 1, R, R), dtype=torch.float32, device=boxes.device)
    ms = torch.empty_like(masks)
    call("agl_layout_from_boxes", ptr(boxes), ptr(bs), ptr(masks), ptr(ms), O, R, stream())
    return bs, masks, ms


def attr_edit_(attribute, cols_dev, tgt):
    """In place: attribute[:, cols] = 0; attribute[:, tgt] = 1 (test64.py:160-167).  cols_dev: int32 device tensor."""
    O, A = attribute.shape
    call("agl_attr_edit", ptr(attribute), ptr(cols_dev, torch.int32), cols_dev.numel(), int(tgt), O, A, stream())
    return attribute


def topk_contains(logits, k, tgt):
    O, A = logits.shape
    out = torch.empty(O, dtype=torch.uint8, device=logits.device)
    call("agl_topk_contains", ptr(logits.contiguous()), ptr(out, torch.uint8), O, A, int(k), int(tgt), stream())
    return out


def sigmoid_threshold(logits, thr):
    pred = torch.empty(logits.shape, dtype=torch.uint8, device=logits.device)
    call("agl_sigmoid_threshold", ptr(logits.contiguous()), ptr(pred, torch.uint8), logits.numel(), float(thr), stream())
    return pred


def deprocess_u8(imgs, inv_std, mean, rescale=True):
    """(N,3,H,W) fp32 -> (N,3,H,W) uint8 on device (data/utils.py:47-66)."""
    N, Cc, H, W = imgs.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.uint8, device=imgs.device)
    a = (C.c_float * 3)(*inv_std)
    b = (C.c_float * 3)(*mean)
    call("agl_deprocess_u8", ptr(imgs.contiguous()), ptr(out, torch.uint8), N, Cc, H * W, int(bool(rescale)), a, b, stream())
    return out
