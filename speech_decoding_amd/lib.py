"""ctypes binding of libsdamd.so (C ABI declared in include/sd_amd.h).

There is NO fallback: if the shared library is missing or an entry point fails, this raises.  The
library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdamd.so")

ABI_VERSION = 4          # SDA_ABI_VERSION of include/sd_amd.h this binding was written against
F32, BF16, F16 = 0, 1, 2
ROW_PAD = 16
CH_ALIGN = 64
EPI_GELU, EPI_GLU, EPI_GLU_BWD = 1, 2, 4
EPI_GELU_BWD, EPI_ROW_SUMSQ, EPI_BN_STORE_DG = 65536, 131072, 262144          # epilogues of conv1_flat only (need CONV_FLAT_TILES)
CONV_SINGLE_TILE, CONV_PAIR_TILES, CONV_FLAT_TILES, CONV_ONE_PER_CU = 4096, 8192, 16384, 32768
CONV_WAVE_PRIO = 1048576      # the launch's waves at s_setprio 3
CONV_WIDE_TILES = 524288      # kernel size 1 on 256-row x 256 / 320-channel tiles, one 8-wave workgroup per CU (conv1_wide.hip)
WGRAD_FLAT_ROWS = 1

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double


class ConvArgs(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("bias", vp), ("res", vp), ("y", vp), ("y_pre", vp), ("widx", vp),
                ("stats", vp), ("partial", vp), ("bn_x", vp), ("bn_coef", vp), ("glu_out", vp), ("glu_gate", vp),
                ("B", i32), ("T", i32), ("Cin_p", i32), ("Cout_p", i32), ("KS", i32), ("dil", i32),
                ("x_pitch", i64), ("w_pitch", i64), ("x_row0", i64), ("x_sample_rows", i64),
                ("x_rows_limit", i64), ("w_rows_limit", i32), ("ksplit", i32), ("flags", i32), ("dtype", i32)]


class WgradArgs(C.Structure):
    _fields_ = [("dy", vp), ("x", vp), ("g", vp), ("out_e", vp), ("sub", vp), ("rscale", vp), ("out_scale", vp), ("perm", vp),
                ("seg_start", vp),
                ("nseg", i32), ("B", i32), ("T", i32), ("Cout_p", i32), ("Cin_p", i32), ("KS", i32), ("dil", i32),
                ("dy_pitch", i64), ("x_pitch", i64), ("out_pitch", i64), ("row0", i64), ("sample_rows", i64),
                ("rows_limit", i64), ("dy_zero_row", i64), ("co_valid", i32), ("dtype", i32), ("acc_scale", vp), ("flags", i32)]


class PackDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("nW", i32), ("Cout", i32), ("Cin", i32), ("KS", i32), ("Cout_p", i32),
                ("Cin_p", i32), ("mode", i32), ("glu_half", i32), ("glu_half_p", i32), ("is_vector", i32), ("glu_tile", i32), ("total", i64)]


class PgemmArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
                ("a_i", i64), ("a_k", i64), ("a_b", i64), ("b_k", i64), ("b_j", i64), ("b_b", i64),
                ("c_i", i64), ("c_j", i64), ("c_b", i64), ("c_dtype", i32)]


class AdamDesc(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("n", i64), ("aligned", i32)]


# name -> (restype, argtypes); every symbol declared in include/sd_amd.h
SIGNATURES = {
    "sda_abi_version": (i32, []),
    "sda_last_error": (C.c_char_p, []),
    "sda_rows_alloc": (i64, [i32, i32]),
    "sda_pad_channels": (i32, [i32]),
    "sda_device_count": (i32, []),
    "sda_stream_create_cumask": (i32, [vp, i32, vp]),
    "sda_sim_gemm_ksplit": (i32, [i32, i32, i64, i32]),
    "sda_sim_gemm": (i32, [vp, vp, vp, i32, i32, i32, i64, i64, i32, i32, vp]),
    "sda_stream_create_priority": (i32, [i32, vp]),
    "sda_stream_destroy": (i32, [vp]),
    "sda_set_cu_limit": (i32, [i32]),
    "sda_upload_words": (i32, [vp, vp, i64, vp]),
    "sda_pack_rows": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "sda_pack_rows_ones": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sda_unpack_rows": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "sda_rows_sumsq": (i32, [vp, vp, vp, i32, i64, i64, i32, vp]),
    "sda_rows_sumsq_from_stats": (i32, [vp, i32, i32, vp, i32, vp]),
    "sda_rows_sumsq_from_row_parts": (i32, [vp, i32, vp, i32, i32, vp]),
    "sda_pack_conv_weight": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "sda_adam_multi": (i32, [vp, i32, i64, f32, f32, f32, f32, i64, vp]),
    "sda_pack_multi": (i32, [vp, i32, i64, i32, vp]),
    "sda_reduce_unpack_wgrad": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "sda_pack_vector": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "sda_unpack_conv_wgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "sda_unpack_vector": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "sda_conv_gemm": (i32, [C.POINTER(ConvArgs), vp]),
    "sda_conv_n_t_tiles": (i32, [i32]),
    "sda_conv_stats_rows": (i32, [i32, i32, i32, i32, i32]),
    "sda_bn_finalize": (i32, [vp, i32, f64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "sda_reduce_stats": (i32, [vp, i32, vp, vp, i32, vp]),
    "sda_bn_gelu_backward_from_stats": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, f64, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_bn_gelu_backward_from_stats_dg": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, f64, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_bn_gelu_forward": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_bn_gelu_backward_reduce": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_bn_gelu_backward_apply": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, f64, vp, vp, i32, i32, i32, i32, vp]),
    "sda_bn_gelu_backward_apply_dg": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, f64, vp, vp, i32, i32, i32, i32, vp]),
    "sda_reduce_scratch_floats": (i32, [i32]),
    "sda_glu_forward": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "sda_glu_backward": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_gelu_backward": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_reduce_scratch_rows": (i32, [i32, i32]),
    "sda_glu_backward_colsum": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_w2v_conv0": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]),
    "sda_layernorm_rows": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp]),
    "sda_w2v_group_split": (i32, [vp, vp, i32, i32, i32, i32, i32, i64, i32, i32, vp]),
    "sda_w2v_group_merge_add": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i64, vp, i32, i32, vp]),
    "sda_w2v_attention": (i32, [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, f32, i32, vp]),
    "sda_splitk_epilogue": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_w2v_mean4": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_glu_backward_colsum_og": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_gelu_backward_colsum": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_colsum": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_wgrad_gemm": (i32, [C.POINTER(WgradArgs), vp]),
    "sda_reduce_slabs": (i32, [vp, vp, i32, i64, vp]),
    "sda_sa_weights_forward": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sda_sa_scratch_floats": (i32, [i32, i32, i32]),
    "sda_sa_softmax_pack": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "sda_sa_softmax_backward": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, vp]),
    "sda_sa_weights_backward": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sda_clip_logits_stats": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "sda_clip_grad": (i32, [vp, vp, vp, vp, vp, vp, f32, i32, vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]),
    "sda_collate_rows": (i32, [vp, vp, i64, i32, i32, f32, i32, vp]),
    "sda_collate_windows": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "sda_clip_ranks": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "sda_clip_dz_supported": (i32, [i32, i32, i64, i32]),
    "sda_clip_dz": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    "sda_param_gemm": (i32, [C.POINTER(PgemmArgs), vp]),
    "sda_zero_pad_rows": (i32, [vp, i32, i32, i32, i32, vp]),
    "sda_scalar_mul": (i32, [vp, vp, vp, i32, vp]),
    "sda_fill_zero": (i32, [vp, i64, vp]),
    "sda_gather_samples": (i32, [vp, vp, vp, i32, i64, vp]),
    "sda_clip_merge_rows": (i32, [vp, i32, i32, vp, vp, vp]),
    "sda_copy3d": (i32, [vp, i64, i64, i64, vp, i64, i64, i64, i32, i32, i32, vp]),
}

_lib = None


class SdaError(RuntimeError):
    pass


def load():
    """Load libsdamd.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdaError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C speech_decoding_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sda_abi_version() != ABI_VERSION:
        raise SdaError("libsdamd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sda_last_error()
        raise SdaError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def pad_channels(c: int) -> int:
    return (c + CH_ALIGN - 1) // CH_ALIGN * CH_ALIGN


def rows_tp(T: int) -> int:
    return T + ROW_PAD


def rows_alloc(B: int, T: int) -> int:
    return B * rows_tp(T) + ROW_PAD + 128 + 2 * ROW_PAD
