"""ctypes binding of libbdetr.so (the C ABI declared in include/bdetr.h).

The product path has NO fallback: if the HIP library is missing or a symbol cannot be
resolved, importing this module's ``lib()`` raises.  Nothing here imports ``oracle``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(os.environ.get("BDETR_LIB") or Path(__file__).resolve().parent / "csrc" / "libbdetr.so")   # BDETR_LIB: kernel experiments
_lib = None

c_f32p = C.c_void_p     # device pointers are passed as integers (tensor.data_ptr())
c_i32p = C.c_void_p
c_i64p = C.c_void_p
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "OH", "OW")]


class GemmDesc(C.Structure):
    _fields_ = [
        ("I", C.c_int), ("J", C.c_int), ("R", C.c_int), ("nb0", C.c_int), ("nb1", C.c_int),
        ("a", C.c_void_p), ("lda", C.c_int64), ("sa0", C.c_int64), ("sa1", C.c_int64), ("a_rcontig", C.c_int),
        ("b", C.c_void_p), ("ldb", C.c_int64), ("sb0", C.c_int64), ("sb1", C.c_int64), ("b_rcontig", C.c_int),
        ("c", C.c_void_p), ("ldc", C.c_int64), ("sc0", C.c_int64), ("sc1", C.c_int64),
        ("bias", C.c_void_p), ("alpha", C.c_float), ("act", C.c_int), ("accumulate", C.c_int), ("splitk", C.c_int),
        ("grad", C.c_int),
    ]


class BnAffine(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p)]


class BnBwdFuse(C.Structure):
    _fields_ = [("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("relu", C.c_int),
                ("part_g", C.c_void_p), ("part_gx", C.c_void_p), ("relu_mask", C.c_void_p),
                ("y2", C.c_void_p), ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("part_gx2", C.c_void_p)]


class RowchainFwdDesc(C.Structure):
    _fields_ = ([("M", C.c_int64), ("nstages", C.c_int), ("eps", C.c_float), ("rate", C.c_float), ("ctx", C.c_void_p), ("resid", C.c_void_p),
                 ("w", C.c_void_p * 3), ("bias", C.c_void_p * 3)]
                + [(n, C.c_void_p) for n in ("g1", "b1", "g2", "b2", "pre1", "x1", "mean1", "rstd1", "h", "pre2", "x2", "mean2", "rstd2")]
                + [("seed1", C.c_uint64), ("seed2", C.c_uint64), ("seed_base", C.c_void_p)])


class RowchainBwdDesc(C.Structure):
    _fields_ = ([("M", C.c_int64), ("nstages", C.c_int), ("rate", C.c_float)]
                + [(n, C.c_void_p) for n in ("dout", "pre2", "mean2", "rstd2", "g2", "h", "pre1", "mean1", "rstd1", "g1")]
                + [("wt", C.c_void_p * 3)]
                + [(n, C.c_void_p) for n in ("G2", "G1", "G0", "dresid", "dctx", "partials")]
                + [("seed1", C.c_uint64), ("seed2", C.c_uint64), ("seed_base", C.c_void_p)])


class LossDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("M", C.c_int), ("N", C.c_int), ("C", C.c_int), ("A", C.c_int),
                ("category_weight", C.c_float), ("attribute_weight", C.c_float),
                ("box_weight", C.c_float), ("exist_weight", C.c_float)]


P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
U64 = C.c_uint64

# name -> (restype, argtypes); every symbol include/bdetr.h declares
SIGNATURES = {
    "bdetr_abi_version": (I, []),
    "bdetr_last_error": (C.c_char_p, []),
    "bdetr_device_cus": (I, []),
    "bdetr_low_priority_stream_create": (I, [P]),
    "bdetr_side_stream_candidates": (I, [P, I, I, I, P, P, P]),
    "bdetr_stream_destroy": (I, [P]),
    "bdetr_stream_priority_range": (I, [P, P]),
    "bdetr_set_gemm_precision": (I, [I]),
    "bdetr_get_gemm_precision": (I, []),
    "bdetr_prof_enable": (I, [I]),
    "bdetr_prof_read": (I, [P, P, P]),
    "bdetr_prof_read_arith": (I, [I, P, P, P]),
    "bdetr_prof_dump": (I, [C.c_char_p]),
    "bdetr_image_prep": (I, [P, I, I, I, P, I, I, P]),
    "bdetr_tokens_prepare": (I, [P, P, L, I, I, I, P, P, P]),
    "bdetr_augment_ws_floats": (I, [I]),
    "bdetr_augment": (I, [P, P, P, P, I, I, I, P, P]),
    "bdetr_jpeg_quality_ws_bytes": (C.c_int64, [I, I, I]),
    "bdetr_jpeg_quality": (I, [P, P, P, I, I, I, P, P]),
    "bdetr_augment_jpeg": (I, [P, P, P, P, P, I, I, I, P, P, P]),
    "bdetr_conv2d_fwd": (I, [P, P, P, P, C.POINTER(ConvDesc), I, P, P, P]),
    "bdetr_conv2d_fwd_stat_chunks": (I, [C.POINTER(ConvDesc)]),
    "bdetr_conv2d_bwd_data": (I, [P, P, P, C.POINTER(ConvDesc), I, P]),
    "bdetr_conv2d_bwd_weight": (I, [P, P, P, C.POINTER(ConvDesc), I, P]),
    "bdetr_conv2d_bwd_weight_splitk": (I, [C.POINTER(ConvDesc)]),
    "bdetr_bn_apply_p16": (I, [P, P, P, P, P, P, I, C.POINTER(BnAffine), I, P, P, P, P, P, L, I, P]),
    "bdetr_bn_bwd_p16": (I, [P, P, I, P, P, P, P, P, I, I, P, P, P, P, P, P, P, P, I, L, I, P]),
    "bdetr_bn_bwd_p16_even_pixels": (I, [P, P, I, P, P, P, P, P, I, I, P, P, P, P, P, P, I, I, I, I, I, P]),
    "bdetr_p16_supported": (I, [C.POINTER(ConvDesc)]),
    "bdetr_p16_pack": (I, [P, L, P, P, P, P]),
    "bdetr_p16_unpack": (I, [P, I, L, P, P]),
    "bdetr_p16_pack_conv_weights": (I, [P, I, I, I, I, P, P, P, P]),
    "bdetr_p16_pack_conv_weights_multi": (I, [P, I, P, P]),
    "bdetr_p16_conv2d_fwd_stat_chunks": (I, [C.POINTER(ConvDesc)]),
    "bdetr_p16_conv2d_fwd": (I, [P, P, P, P, C.POINTER(ConvDesc), I, P, P, P]),
    "bdetr_p16_conv2d_bwd_data": (I, [P, P, P, C.POINTER(ConvDesc), I, P]),
    "bdetr_p16_conv2d_bwd_data_stat_chunks": (I, [C.POINTER(ConvDesc)]),
    "bdetr_p16_conv2d_bwd_data_masked_accum": (I, [P, P, P, P, C.POINTER(ConvDesc), C.POINTER(BnBwdFuse), P]),
    "bdetr_p16_conv2d_bwd_data_masked_accum_compact": (I, [P, P, P, P, P, C.POINTER(ConvDesc), C.POINTER(BnBwdFuse), P]),
    "bdetr_relu_mask_apply": (I, [P, P, C.c_int64, P]),
    "bdetr_p16_conv2d_bwd_data_bnstats": (I, [P, P, P, C.POINTER(ConvDesc), C.POINTER(BnBwdFuse), P]),
    "bdetr_p16_conv2d_bwd_weight_splitk": (I, [C.POINTER(ConvDesc)]),
    "bdetr_p16_conv2d_bwd_weight": (I, [P, P, P, C.POINTER(ConvDesc), I, P]),
    "bdetr_p16_conv2d_bwd_weight_xf16": (I, [P, P, P, C.POINTER(ConvDesc), I, P]),
    "bdetr_gemm": (I, [C.POINTER(GemmDesc), P]),
    "bdetr_gemm_ws": (I, [C.POINTER(GemmDesc), P, L, P]),
    "bdetr_splitk_workspace_elems": (L, [L, L, I]),
    "bdetr_conv2d_bwd_weight_ws": (I, [P, P, P, C.POINTER(ConvDesc), I, P, L, P]),
    "bdetr_p16_conv2d_bwd_weight_ws": (I, [P, I, P, P, C.POINTER(ConvDesc), I, P, L, P]),
    "bdetr_gemm_grouped": (I, [C.POINTER(GemmDesc), I, P]),
    "bdetr_colsum_chunks": (I, [L]),
    "bdetr_colsum": (I, [P, L, I, P, P, P]),
    "bdetr_colsum_accumulate": (I, [P, L, I, P, P]),
    "bdetr_colsum_accumulate_group": (I, [P, P, I, P, I, P]),
    "bdetr_colstats": (I, [P, L, I, P, P, P]),
    "bdetr_bn_stats": (I, [P, L, I, P, P, I, F, F, I, P, P, P, P, P, P, P]),
    "bdetr_flag_nonfinite": (I, [P, L, P, P]),
    "bdetr_flag_snapshot": (I, [P, P, P, I, P]),
    "bdetr_debug_checksum": (I, [P, L, P, P, P, I, U64, P]),
    "bdetr_graph_node_census": (I, [P, P, I]),
    "bdetr_bn_stats_fold_rows": (I, []),
    "bdetr_bn_stats_frozen": (I, [P, P, I, F, P, P, P]),
    "bdetr_bn_apply": (I, [P, P, P, P, P, P, I, P, L, I, P]),
    "bdetr_bn_bwd_chunks": (I, [L]),
    "bdetr_bn_bwd": (I, [P, P, P, P, P, P, P, I, I, P, P, P, P, P, L, I, P]),
    "bdetr_maxpool3x3s2_fwd": (I, [P, P, I, I, I, I, I, I, P]),
    "bdetr_maxpool3x3s2_bwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "bdetr_stem_pool_bwd_chunks": (I, [L]),
    "bdetr_stem_pool_fwd": (I, [P, P, P, P, P, I, I, I, I, P, P, P, P, P]),
    "bdetr_stem_pool_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P, I, P, P, P, P]),
    "bdetr_p16_s2d_pack_bf16": (I, [P, I, I, I, P, P]),
    "bdetr_p16_stem_bwd_weight": (I, [P, P, P, I, I, I, I, P]),
    "bdetr_p16_s2d_unpack_dw": (I, [P, P, I, P]),
    "bdetr_attention_head_dim": (I, []),
    "bdetr_attention_fwd": (I, [P, P, P, P, P, I, I, I, I, F, P]),
    "bdetr_attention_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, P]),
    "bdetr_softmax_rows_fwd": (I, [P, P, L, I, F, P]),
    "bdetr_softmax_rows_bwd": (I, [P, P, P, L, I, F, P]),
    "bdetr_add_dropout_layernorm_fwd": (I, [P, P, P, P, P, P, P, L, I, F, F, U64, P, P]),
    "bdetr_ln_bwd_chunks": (I, [L]),
    "bdetr_add_dropout_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, L, I, F, U64, P, I, P]),
    "bdetr_rowchain_fwd": (I, [C.POINTER(RowchainFwdDesc), P]),
    "bdetr_rowchain_bwd": (I, [C.POINTER(RowchainBwdDesc), P]),
    "bdetr_rowchain_reduce": (I, [P, I, C.POINTER(C.c_void_p * 7), C.POINTER(C.c_int * 7), P]),
    "bdetr_rowchain_width": (I, []),
    "bdetr_rowchain_pack_elems": (L, []),
    "bdetr_rowchain_partial_rows": (I, [L]),
    "bdetr_rowchain_pack_weights": (I, [P, I, P, P]),
    "bdetr_resize_bilinear_nhwc": (I, [P, I, I, I, I, P, I, I, P]),
    "bdetr_layernorm_act_fwd": (I, [P, L, I, I, P, P, F, F, P, I, P]),
    "bdetr_copy_cols": (I, [P, L, I, I, P, I, I, P]),
    "bdetr_nhwc_to_nchw": (I, [P, I, I, I, I, P, P]),
    "bdetr_softmax_lastdim_fwd": (I, [P, P, L, I, P]),
    "bdetr_softmax_lastdim_bwd": (I, [P, P, P, L, I, P]),
    "bdetr_sigmoid_fwd": (I, [P, P, L, P]),
    "bdetr_sigmoid_bwd": (I, [P, P, P, L, P]),
    "bdetr_boxsigmoid_fwd": (I, [P, P, L, P]),
    "bdetr_boxsigmoid_bwd": (I, [P, P, P, L, P]),
    "bdetr_zero": (I, [P, L, P]),
    "bdetr_add": (I, [P, P, P, L, P]),
    "bdetr_add_bcast_rows": (I, [P, P, P, L, L, P]),
    "bdetr_sum_over_batch": (I, [P, P, L, L, I, P]),
    "bdetr_tanh_bwd": (I, [P, P, P, L, P]),
    "bdetr_relu_bwd": (I, [P, P, P, L, P]),
    "bdetr_axpy": (I, [F, P, P, L, P]),
    "bdetr_cost_matrix": (I, [C.POINTER(LossDesc), P, P, P, P, P, P, P, P, P, P, P, P]),
    "bdetr_lsa": (I, [P, P, I, I, I, P, P]),
    "bdetr_set_loss": (I, [C.POINTER(LossDesc), P, P, P, P, P, P, P, P, P, P, P, P, F, P]),
    "bdetr_match_to_mask": (I, [P, P, I, I, I, P]),
    "bdetr_sgd_slab_elems": (I, []),
    "bdetr_sgd_nesterov_clipnorm": (I, [P, P, I, P, P, I, P, P, P, F, F, F, P, P]),
}


class BdetrError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def lib():
    """Load libbdetr.so (once) and bind every symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise BdetrError(
            f"{_LIB_PATH} not found: the HIP extension is not built. Run `python -m boosted_detr_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback in the product path.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and must load it first, so that this
    # library's dependency resolves to the SAME copy (the host hands over torch's device pointers and
    # stream handles).  Loaded the other way round, the second runtime finds "no ROCm-capable device".
    import torch  # noqa: F401
    h = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if h.bdetr_abi_version() != 8:
        raise BdetrError("libbdetr.so ABI version mismatch; rebuild")
    _lib = h
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().bdetr_last_error().decode(errors="replace")
        raise BdetrError(f"{what or 'libbdetr'} failed (status {status}): {msg}")
