"""Thin typed wrappers over the C ABI (include/bdetr.h).

PyTorch-ROCm tensors are used ONLY as device-memory containers (``data_ptr()``) and for the
current HIP stream; every arithmetic operation below is a hand-written HIP kernel in
``csrc/``.  There is no eager/PyTorch fallback: a missing library raises in ``_lib.lib()``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BnAffine, BnBwdFuse, ConvDesc, GemmDesc, LossDesc, RowchainBwdDesc, RowchainFwdDesc, check

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


# Raw hipStream_t the kernels are launched on.  Asking torch for the current stream costs ~8 us per
# launch (a third of the host's enqueue time at ~1500 launches per step), so the training step pins
# the handle for its duration (engine.on_side_stream swaps it while side-stream work is queued).
_LAUNCH_STREAM = [None]


def set_launch_stream(handle: Optional[int]) -> Optional[int]:
    prev = _LAUNCH_STREAM[0]
    _LAUNCH_STREAM[0] = handle
    return prev


GEMM_FP32, GEMM_BF16X3, GEMM_MIXED, GEMM_SPLIT, GEMM_BF16X6 = 0, 1, 2, 3, 4
_GEMM_MODES = {"fp32": GEMM_FP32, "bf16x3": GEMM_BF16X3, "mixed": GEMM_MIXED, "split": GEMM_SPLIT, "bf16x6": GEMM_BF16X6}


def set_gemm_precision(mode) -> int:
    """Arithmetic of the conv/GEMM family: 'fp32' (exact fp32 MFMA everywhere), 'bf16x3' (split-bf16
    everywhere) or 'mixed' (default: exact forward, split-bf16 gradient products).  Returns the previous mode."""
    code = _GEMM_MODES.get(mode, mode)
    prev = _lib.lib().bdetr_get_gemm_precision()
    check(_lib.lib().bdetr_set_gemm_precision(int(code)), "set_gemm_precision")
    return prev


def get_gemm_precision() -> str:
    code = _lib.lib().bdetr_get_gemm_precision()
    return next(k for k, v in _GEMM_MODES.items() if v == code)


def demote_split_forward() -> None:
    """Frozen BatchNormalization (moving statistics) does not guarantee normalised activations, so the rest of this
    forward must not take the range-limited split-fp16 path: 'split' -> 'mixed' (exact-fp32 forward, same gradient
    arithmetic).  The enclosing ``gemm_precision`` scope restores the policy afterwards."""
    L = _lib.lib()
    if L.bdetr_get_gemm_precision() == GEMM_SPLIT:
        check(L.bdetr_set_gemm_precision(GEMM_MIXED), "set_gemm_precision")


# Deterministic mode (BDETR_DETERMINISTIC=1 or set_deterministic(True)): the only run-to-run variation of a training step is
# the ORDER in which float atomics land - the split-K slices of the weight gradients and the one-launch bias-gradient column
# sums.  In this mode the split-K launches store their slices into slabs of a workspace and a second launch folds them in a fixed
# order (the *_ws entry points of include/bdetr.h), and column sums take the two-level fixed-order reduction: two runs of the same
# step - eager or replayed from hipGraphs - give bit-identical weights.  Cost: one fold launch per split-K launch and the slabs'
# round trip through HBM (reported by bench.py as `deterministic_cost`).
import os as _os

_DETERMINISTIC = [_os.environ.get("BDETR_DETERMINISTIC", "0") == "1"]


def set_deterministic(on: bool) -> bool:
    prev = _DETERMINISTIC[0]
    _DETERMINISTIC[0] = bool(on)
    return prev


def deterministic() -> bool:
    return _DETERMINISTIC[0]


def _splitk_ws(rows: int, cols: int, sk: int, like: torch.Tensor):
    """(workspace tensor, its element count) of a deterministic split-K launch, or (None, 0).  Allocated from the caching allocator on
    the stream the launch goes to (side tasks run with the side stream current), so a later launch on that stream may reuse it."""
    if not _DETERMINISTIC[0] or sk <= 1:
        return None, 0
    n = int(_lib.lib().bdetr_splitk_workspace_elems(rows, cols, sk))
    return torch.empty(n, dtype=torch.float32, device=like.device), n


class gemm_precision:
    """``with gemm_precision('split'): ...`` - scoped arithmetic policy (None leaves it alone)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_gemm_precision(self.mode) if self.mode is not None else None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            set_gemm_precision(self.prev)
        return False


def _stream() -> int:
    h = _LAUNCH_STREAM[0]
    return h if h is not None else torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _chk(*tensors, dtype=torch.float32):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.BdetrError("kernel operands must live in HBM (cuda tensors); no CPU path exists")
        if t.dtype != dtype:
            raise _lib.BdetrError(f"expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise _lib.BdetrError("kernel operands must be contiguous")


def empty(*shape, like: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


# --------------------------------------------------------------------------------------
# K1
# --------------------------------------------------------------------------------------
def image_prep(image: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """backbone.py:49-56.  image [B,h,w,3] in [0,1] -> [B,H,W,4] (BGR - mean, pad channel 0)."""
    _chk(image)
    B, h, w, c = image.shape
    assert c == 3
    out = empty(B, H, W, 4, like=image)
    check(_lib.lib().bdetr_image_prep(_p(image), B, h, w, _p(out), H, W, _stream()), "image_prep")
    return out


def tokens_prepare(cat_ids: torch.Tensor, att_ids: torch.Tensor, C_: int, A: int):
    """tokenizers.py:40-82 for targets that are already int32 ids in HBM: (range-checked category ids [B,M],
    multi-hot attributes f32 [B,M,A])."""
    _chk(cat_ids, att_ids, dtype=torch.int32)
    if cat_ids.dim() == 3:
        cat_ids = cat_ids[..., 0].contiguous()
    B, M = cat_ids.shape
    slots = att_ids.shape[2] if att_ids.dim() == 3 else 1
    cat = torch.empty_like(cat_ids)
    hot = torch.empty((B, M, A), dtype=torch.float32, device=cat_ids.device)
    check(_lib.lib().bdetr_tokens_prepare(_p(cat_ids), _p(att_ids), B * M, slots, C_, A, _p(cat), _p(hot), _stream()), "tokens_prepare")
    return cat, hot


def augment(image: torch.Tensor, iparams: torch.Tensor, fparams: torch.Tensor, quality: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pipeline.py:274-341 on the GPU.  image [B,H,W,3]; iparams int32 [B,4]; fparams f32 [B,3]; quality int32 [B] or None
    (random_jpeg_quality between brightness and saturation, pipeline.py:319-325)."""
    _chk(image, fparams)
    _chk(iparams, dtype=torch.int32)
    B, H, W, c = image.shape
    assert c == 3
    L = _lib.lib()
    out = torch.empty_like(image)
    ws = empty(L.bdetr_augment_ws_floats(B), like=image)
    if quality is None:
        check(L.bdetr_augment(_p(image), _p(out), _p(iparams), _p(fparams), B, H, W, _p(ws), _stream()), "augment")
        return out
    _chk(quality, dtype=torch.int32)
    jws = torch.empty(int(L.bdetr_jpeg_quality_ws_bytes(B, H, W)), dtype=torch.uint8, device=image.device)
    check(L.bdetr_augment_jpeg(_p(image), _p(out), _p(iparams), _p(fparams), _p(quality), B, H, W, _p(ws), _p(jws), _stream()), "augment_jpeg")
    return out


def jpeg_quality(image: torch.Tensor, quality: torch.Tensor) -> torch.Tensor:
    """tf.image.adjust_jpeg_quality per image: float [B,H,W,3] in [0,1], quality int32 [B] (device)."""
    _chk(image)
    _chk(quality, dtype=torch.int32)
    B, H, W, c = image.shape
    assert c == 3 and quality.numel() == B
    L = _lib.lib()
    out = torch.empty_like(image)
    jws = torch.empty(int(L.bdetr_jpeg_quality_ws_bytes(B, H, W)), dtype=torch.uint8, device=image.device)
    check(L.bdetr_jpeg_quality(_p(image), _p(out), _p(quality), B, H, W, _p(jws), _stream()), "jpeg_quality")
    return out


# --------------------------------------------------------------------------------------
# convolution family
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ConvGeom:
    N: int
    H: int
    W: int
    C: int
    K: int
    R: int
    S: int
    stride: int
    pad: int

    @property
    def OH(self) -> int:
        return (self.H + 2 * self.pad - self.R) // self.stride + 1

    @property
    def OW(self) -> int:
        return (self.W + 2 * self.pad - self.S) // self.stride + 1

    @property
    def M(self) -> int:
        return self.N * self.OH * self.OW

    def desc(self) -> ConvDesc:
        return ConvDesc(self.N, self.H, self.W, self.C, self.K, self.R, self.S, self.stride, self.pad, self.OH, self.OW)


def conv2d_fwd(x, w, bias, g: ConvGeom, act: int = ACT_NONE, want_stats: bool = False):
    """x [N,H,W,C], w [K,R,S,C] (OHWI), bias [K] -> y [N,OH,OW,K] (+ partial column stats)."""
    _chk(x, w, bias)
    L = _lib.lib()
    d = g.desc()
    y = empty(g.N, g.OH, g.OW, g.K, like=x)
    psum = psq = None
    nparts = 0
    if want_stats:
        nparts = L.bdetr_conv2d_fwd_stat_chunks(C.byref(d))
        if nparts <= 0:
            check(-1, "conv2d_fwd_stat_chunks")
        psum = empty(nparts, g.K, like=x)
        psq = empty(nparts, g.K, like=x)
    check(L.bdetr_conv2d_fwd(_p(x), _p(w), _p(bias), _p(y), C.byref(d), act, _p(psum), _p(psq), _stream()), "conv2d_fwd")
    return y, (psum, psq, nparts)


def conv2d_bwd_data(dy, w, g: ConvGeom, dx: Optional[torch.Tensor] = None, accumulate: bool = False):
    _chk(dy, w, dx)
    d = g.desc()
    if dx is None:
        assert not accumulate
        dx = empty(g.N, g.H, g.W, g.C, like=dy)
    check(_lib.lib().bdetr_conv2d_bwd_data(_p(dy), _p(w), _p(dx), C.byref(d), int(accumulate), _stream()), "conv2d_bwd_data")
    return dx


def conv2d_bwd_weight(x, dy, g: ConvGeom, dw: Optional[torch.Tensor] = None, prezeroed: bool = False):
    """prezeroed: dw already holds zeros (the optimizer's flat gradient buffer is cleared once per step),
    so the split-K atomics need no per-tensor memset."""
    _chk(x, dy, dw)
    L = _lib.lib()
    d = g.desc()
    if dw is None:
        dw = empty(g.K, g.R, g.S, g.C, like=x)
        prezeroed = False
    sk = L.bdetr_conv2d_bwd_weight_splitk(C.byref(d))
    if sk > 1 and not prezeroed:
        check(L.bdetr_zero(_p(dw), dw.numel(), _stream()), "zero")
    ws, n = _splitk_ws(g.K, g.R * g.S * g.C, sk, x)
    check(L.bdetr_conv2d_bwd_weight_ws(_p(x), _p(dy), _p(dw), C.byref(d), sk, _p(ws), n, _stream()), "conv2d_bwd_weight")
    return dw


# --------------------------------------------------------------------------------------
# pre-split (P16) operand path of the convolutions - csrc/sgemm.hip, csrc/p16.hip
# --------------------------------------------------------------------------------------
# A P16 tensor is carried as a float32 torch tensor of the logical shape (4 bytes per element, never read as
# floats): f16 pairs for forward operands, bf16 pairs for gradient operands (include/bdetr.h).
_OVERFLOW = [None]


def overflow_flag() -> torch.Tensor:
    """Device int32 the P16-f16 producers set when a value leaves the f16 pair's range (|x| >= 65504 / not finite)."""
    if _OVERFLOW[0] is None:
        _OVERFLOW[0] = torch.zeros(1, dtype=torch.int32, device="cuda")
    return _OVERFLOW[0]


_GUARD_ACTIVE = [True]


def set_guard_active(on: bool) -> None:
    """Whether the running step reads and clears the overflow flag (Model: policy 'split').  Only then do BatchNorm
    statistics watch it: under another policy nobody clears it, and a flag left up by an earlier guarded step (or by a stray
    p16_pack of a tool) would stop the moving-statistics updates for the rest of the process."""
    _GUARD_ACTIVE[0] = bool(on)


def flag_nonfinite(x: torch.Tensor) -> None:
    """Set the step's range guard when x holds a non-finite value (the loss vectors: a NaN born in a kernel without
    its own range check, e.g. the in-kernel split-fp16 products of the transformer layers)."""
    _chk(x)
    check(_lib.lib().bdetr_flag_nonfinite(_p(x), x.numel(), _p(overflow_flag()), _stream()), "flag_nonfinite")


def flag_snapshot(ordinal: torch.Tensor, host_ring: torch.Tensor) -> None:
    """Log the overflow flag against the step ordinal in a pinned int32[1 + n] host tensor: ordinal += 1 (device int32),
    host_ring[1 + ordinal % n] = flag, host_ring[0] = ordinal.  One lane on the launch stream, capturable, no D2H memcpy, no sync."""
    assert host_ring.dtype == torch.int32 and host_ring.numel() >= 2 and host_ring.is_pinned()
    assert ordinal.dtype == torch.int32 and ordinal.is_cuda
    check(_lib.lib().bdetr_flag_snapshot(_p(overflow_flag()), _p(ordinal), host_ring.data_ptr(), host_ring.numel() - 1, _stream()), "flag_snapshot")


def read_and_clear_overflow() -> bool:
    """Host read of the range guard (synchronises the device)."""
    f = overflow_flag()
    tripped = bool(int(f.item()))
    if tripped:
        f.zero_()
    return tripped


def p16_supported(g: ConvGeom) -> bool:
    d = g.desc()
    return bool(_lib.lib().bdetr_p16_supported(C.byref(d)))


def p16_pack(x, want_f16=True, want_bf16=True):
    """fp32 tensor (last dim % 8 == 0) -> (P16-f16 or None, P16-bf16 or None)."""
    _chk(x)
    f = torch.empty_like(x) if want_f16 else None
    b = torch.empty_like(x) if want_bf16 else None
    check(_lib.lib().bdetr_p16_pack(_p(x), x.numel(), _p(f), _p(b), _p(overflow_flag()) if want_f16 else None, _stream()), "p16_pack")
    return f, b


def p16_unpack(p, is_f16: bool):
    _chk(p)
    out = torch.empty_like(p)
    check(_lib.lib().bdetr_p16_unpack(_p(p), int(is_f16), p.numel(), _p(out), _stream()), "p16_unpack")
    return out


def p16_pack_conv_weights(w, want_fwd=True, want_bwd=True):
    """w [K,R,S,C] fp32 -> (P16-f16 [K,R,S,C] forward copy, P16-bf16 [C,R,S,K] transposed tap-flipped copy)."""
    _chk(w)
    K_, R, S, Cc = w.shape
    wf = torch.empty_like(w) if want_fwd else None
    wt = torch.empty((Cc, R, S, K_), dtype=torch.float32, device=w.device) if want_bwd else None
    check(_lib.lib().bdetr_p16_pack_conv_weights(_p(w), K_, R, S, Cc, _p(wf), _p(wt), _p(overflow_flag()) if want_fwd else None, _stream()),
          "p16_pack_conv_weights")
    return wf, wt


def bn_apply_p16(x2d, mean, rstd, gamma, beta, residual=None, relu=False, want_fp32=True, want_f16=True, want_bf16=True,
                 residual_p16=False, want_mask=False, residual_bn=None):
    """bn_apply with P16 outputs: returns (out32 | None, out_f16 | None, out_bf16 | None[, relu bit mask]).  residual_p16:
    `residual` is the f16 pair copy of the shortcut tensor.  residual_bn = (mean, rstd, gamma, beta): `residual` is the RAW
    output of the projection shortcut's convolution, normalised here.  want_mask: also return the 1-bit-per-element ReLU mask
    (int64 words) the backward pass of a residual unit reads instead of the forward output."""
    _chk(x2d, mean, rstd, gamma, beta, residual)
    rbn = None
    if residual_bn is not None:
        _chk(*residual_bn)
        assert residual is not None and not residual_p16
        rbn = C.byref(BnAffine(*(_p(t) for t in residual_bn)))
        residual_p16 = 2
    rows, Cc = x2d.shape
    o32 = torch.empty_like(x2d) if want_fp32 else None
    of = torch.empty_like(x2d) if want_f16 else None
    ob = torch.empty_like(x2d) if want_bf16 else None
    mask = torch.empty(((rows * Cc // 4 + 63) // 64) * 4, dtype=torch.int64, device=x2d.device) if want_mask else None
    check(_lib.lib().bdetr_bn_apply_p16(_p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(residual), int(residual_p16), rbn, int(relu), _p(o32),
                                        _p(of), _p(ob), _p(mask), _p(overflow_flag()) if want_f16 else None, rows, Cc, _stream()), "bn_apply_p16")
    return (o32, of, ob, mask) if want_mask else (o32, of, ob)


def bn_bwd_p16(dout, out, x2d, mean, rstd, gamma, relu, frozen, want_residual_grad=False, dgamma=None, dbeta=None, beta=None, want_fp32=False,
               out_p16=False, pre=None, even_pixels=None, dout_compact=False):
    """bn_bwd writing the input gradient as a bf16 pair: returns (dx_bf16, dx32 | None, dgamma, dbeta, dres | None).
    out_p16: what the ReLU mask source `out` is - 0 / False the fp32 forward output, 1 / True its bf16 pair copy, 2 the bit
    mask of bn_apply_p16(want_mask=True)."""
    _chk(dout, x2d, mean, rstd, gamma, dgamma, dbeta)
    if out is not None and int(out_p16) != 2:
        _chk(out)
    L = _lib.lib()
    rows, Cc = x2d.shape
    dxb = torch.empty_like(x2d)
    dx32 = torch.empty_like(x2d) if want_fp32 else None
    dgamma = empty(Cc, like=x2d) if dgamma is None else dgamma
    dbeta = empty(Cc, like=x2d) if dbeta is None else dbeta
    dres = torch.empty_like(x2d) if want_residual_grad else None
    # pre = (part_g, part_gx, nparts): the reduction already done by the backward-data epilogue that produced dout
    ws = empty(2 * Cc * L.bdetr_bn_bwd_chunks(rows), like=x2d) if pre is None else None
    if even_pixels is not None and pre is None:
        # dout [N,H,W,C] is zero outside the pixels (2i, 2j) (ops.conv_bn tags such gradients): the reduction visits those only
        N, H, W = even_pixels
        assert N * H * W == rows
        # dout_compact: dout is the [N, H/2, W/2, C] tensor of the even pixels alone (the stride-2 backward-data products wrote it densely)
        assert not dout_compact or (dres is None and dout.numel() == N * (H // 2) * (W // 2) * Cc and H % 2 == 0 and W % 2 == 0)
        check(L.bdetr_bn_bwd_p16_even_pixels(_p(dout), _p(out), int(out_p16), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), int(frozen),
                                             _p(dx32), _p(dxb), _p(dgamma), _p(dbeta), _p(dres), _p(ws), N, H, W, Cc, int(dout_compact), _stream()), "bn_bwd_p16_even_pixels")
        return dxb, dx32, dgamma, dbeta, dres
    assert not dout_compact, "a compact dout goes through the even-pixel reduction (even_pixels=(N, H, W), no pre)"
    pg, pgx, pn = pre if pre is not None else (None, None, 0)
    check(L.bdetr_bn_bwd_p16(_p(dout), _p(out), int(out_p16), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), int(frozen), _p(dx32),
                             _p(dxb), _p(dgamma), _p(dbeta), _p(dres), _p(ws), _p(pg), _p(pgx), int(pn), rows, Cc, _stream()), "bn_bwd_p16")
    return dxb, dx32, dgamma, dbeta, dres


def p16_pack_conv_weights_multi(table: torch.Tensor) -> None:
    """table: int64 [n,7] device tensor of {w, w_f16, wt_bf16, K, R, S, C} rows."""
    _chk(table, dtype=torch.int64)
    check(_lib.lib().bdetr_p16_pack_conv_weights_multi(_p(table), table.shape[0], _p(overflow_flag()), _stream()), "p16_pack_conv_weights_multi")


def p16_conv2d_fwd(x_f16, w_f16, bias, g: ConvGeom, act: int = ACT_NONE, want_stats: bool = False):
    _chk(x_f16, w_f16, bias)
    L = _lib.lib()
    d = g.desc()
    y = empty(g.N, g.OH, g.OW, g.K, like=x_f16)
    psum = psq = None
    nparts = 0
    if want_stats:
        nparts = L.bdetr_p16_conv2d_fwd_stat_chunks(C.byref(d))
        if nparts <= 0:
            check(-1, "p16_conv2d_fwd_stat_chunks")
        psum = empty(nparts, g.K, like=x_f16)
        psq = empty(nparts, g.K, like=x_f16)
    check(L.bdetr_p16_conv2d_fwd(_p(x_f16), _p(w_f16), _p(bias), _p(y), C.byref(d), act, _p(psum), _p(psq), _stream()), "p16_conv2d_fwd")
    return y, (psum, psq, nparts)


def p16_conv2d_bwd_data(dy_bf16, wt_bf16, g: ConvGeom, dx: Optional[torch.Tensor] = None, accumulate: bool = False):
    _chk(dy_bf16, wt_bf16, dx)
    d = g.desc()
    if dx is None:
        assert not accumulate
        dx = empty(g.N, g.H, g.W, g.C, like=dy_bf16)
    check(_lib.lib().bdetr_p16_conv2d_bwd_data(_p(dy_bf16), _p(wt_bf16), _p(dx), C.byref(d), int(accumulate), _stream()), "p16_conv2d_bwd_data")
    return dx


def relu_mask_apply_(x: torch.Tensor, relu_mask: torch.Tensor) -> torch.Tensor:
    """x <- x * relu_mask (bn_apply_p16's bit mask) in place."""
    _chk(x)
    _chk(relu_mask, dtype=torch.int64)
    check(_lib.lib().bdetr_relu_mask_apply(_p(x), _p(relu_mask), x.numel(), _stream()), "relu_mask_apply")
    return x


def p16_conv2d_bwd_data_masked_accum(dy_bf16, wt_bf16, g: ConvGeom, dx: torch.Tensor, relu_mask: torch.Tensor, bn_ctx=None, bn_ctx2=None, old_even=None):
    """dx (the gradient of a residual unit's output on entry) <- conv_transpose(dy) + dx * relu_mask, in place: the unit's skip
    branch merged inside the 1x1 backward-data epilogue of its first convolution.  bn_ctx = (y_prev, mean, rstd, gamma, beta,
    relu_mask_prev): the result is the complete output gradient of the PREVIOUS residual unit - also emit that unit's
    BatchNorm-backward partial sums; returns (dx, (part_g, part_gx, nparts)) then."""
    _chk(dy_bf16, wt_bf16, dx)
    _chk(relu_mask, dtype=torch.int64)
    L = _lib.lib()
    d = g.desc()
    if old_even is not None:
        # the unit's output gradient exists at the even pixels only: old_even [N, H/2, W/2, C] (compact), dx is a FRESH dense tensor
        _chk(old_even)
        assert bn_ctx2 is None and old_even.numel() * 4 == dx.numel() and g.H % 2 == 0 and g.W % 2 == 0

        def launch(fuse):
            check(L.bdetr_p16_conv2d_bwd_data_masked_accum_compact(_p(dy_bf16), _p(wt_bf16), _p(dx), _p(old_even), _p(relu_mask), C.byref(d), fuse, _stream()),
                  "p16_conv2d_bwd_data_masked_accum_compact")
    else:
        def launch(fuse):
            check(L.bdetr_p16_conv2d_bwd_data_masked_accum(_p(dy_bf16), _p(wt_bf16), _p(dx), _p(relu_mask), C.byref(d), fuse, _stream()),
                  "p16_conv2d_bwd_data_masked_accum")
    if bn_ctx is None:
        launch(None)
        return dx
    y_prev, mean, rstd, gamma, beta, bits_prev = bn_ctx
    _chk(y_prev, mean, rstd, gamma, beta)
    _chk(bits_prev, dtype=torch.int64)
    n = L.bdetr_p16_conv2d_bwd_data_stat_chunks(C.byref(d))
    if n <= 0:
        check(-1, "p16_conv2d_bwd_data_stat_chunks")
    pg, pgx = empty(n, g.C, like=dy_bf16), empty(n, g.C, like=dy_bf16)
    f = BnBwdFuse(_p(y_prev), _p(mean), _p(rstd), _p(gamma), _p(beta), 1, _p(pg), _p(pgx), _p(bits_prev))
    pgx2 = None
    if bn_ctx2 is not None:
        # the previous unit is a stage's FIRST one: its projection shortcut's BatchNorm sees the same gradient - (y0, mean0, rstd0):
        # also emit that BatchNorm's sum(g * xhat0); returns ((part_g, part_gx, n), (part_g, part_gx0, n))
        y0, mean0, rstd0 = bn_ctx2
        _chk(y0, mean0, rstd0)
        pgx2 = empty(n, g.C, like=dy_bf16)
        f.y2, f.mean2, f.rstd2, f.part_gx2 = _p(y0), _p(mean0), _p(rstd0), _p(pgx2)
    launch(C.byref(f))
    if pgx2 is not None:
        return dx, (pg, pgx, n), (pg, pgx2, n)
    return dx, (pg, pgx, n)


def p16_conv2d_bwd_data_bnstats(dy_bf16, wt_bf16, g: ConvGeom, y_prev, mean, rstd, gamma, beta, relu: bool):
    """Backward-data whose output is the gradient of the BatchNorm(+ReLU) output that fed this conv (pre-normalisation tensor
    y_prev, no residual): returns (dx, (part_g, part_gx, nparts)) - that layer's backward reduction, done in this epilogue."""
    _chk(dy_bf16, wt_bf16, y_prev, mean, rstd, gamma, beta)
    L = _lib.lib()
    d = g.desc()
    dx = empty(g.N, g.H, g.W, g.C, like=dy_bf16)
    n = L.bdetr_p16_conv2d_bwd_data_stat_chunks(C.byref(d))
    if n <= 0:
        check(-1, "p16_conv2d_bwd_data_stat_chunks")
    pg, pgx = empty(n, g.C, like=dy_bf16), empty(n, g.C, like=dy_bf16)
    f = BnBwdFuse(_p(y_prev), _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), _p(pg), _p(pgx), None)
    check(L.bdetr_p16_conv2d_bwd_data_bnstats(_p(dy_bf16), _p(wt_bf16), _p(dx), C.byref(d), C.byref(f), _stream()), "p16_conv2d_bwd_data_bnstats")
    return dx, (pg, pgx, n)


def p16_conv2d_bwd_weight(x_bf16, dy_bf16, g: ConvGeom, dw: Optional[torch.Tensor] = None, prezeroed: bool = False, x_f16: bool = False):
    """x_f16: `x_bf16` is the P16-f16 tensor the forward read (converted to bf16 pairs inside the kernel)."""
    _chk(x_bf16, dy_bf16, dw)
    L = _lib.lib()
    d = g.desc()
    if dw is None:
        dw = empty(g.K, g.R, g.S, g.C, like=x_bf16)
        prezeroed = False
    sk = L.bdetr_p16_conv2d_bwd_weight_splitk(C.byref(d))
    if sk > 1 and not prezeroed:
        check(L.bdetr_zero(_p(dw), dw.numel(), _stream()), "zero")
    ws, n = _splitk_ws(g.K, g.R * g.S * g.C, sk, x_bf16)
    check(L.bdetr_p16_conv2d_bwd_weight_ws(_p(x_bf16), int(x_f16), _p(dy_bf16), _p(dw), C.byref(d), sk, _p(ws), n, _stream()), "p16_conv2d_bwd_weight")
    return dw


# --------------------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------------------
def gemm_raw(I, J, R, a, lda, a_rc, b, ldb, b_rc, c, ldc, *, nb0=1, nb1=1, sa=(0, 0), sb=(0, 0), sc=(0, 0),
             bias=None, alpha=1.0, act=ACT_NONE, accumulate=False, splitk=1, grad=False):
    _chk(a, b, c, bias)
    g = GemmDesc(I, J, R, nb0, nb1, _p(a), lda, sa[0], sa[1], int(a_rc), _p(b), ldb, sb[0], sb[1], int(b_rc),
                 _p(c), ldc, sc[0], sc[1], _p(bias), float(alpha), act, int(accumulate), splitk, int(grad))
    ws, n = _splitk_ws(I, J, splitk, c) if ldc == J else (None, 0)
    check(_lib.lib().bdetr_gemm_ws(C.byref(g), _p(ws), n, _stream()), "gemm")
    return c


def gemm_grouped(descs):
    """descs: list of (I, J, R, a, lda, a_rc, b, ldb, b_rc, c, ldc, bias, act, accumulate); a product with an
    x-contiguous operand is a gradient product (dy @ w)."""
    arr = (GemmDesc * len(descs))()
    for k, (I, J, R, a, lda, a_rc, b, ldb, b_rc, c, ldc, bias, act, acc) in enumerate(descs):
        _chk(a, b, c, bias)
        arr[k] = GemmDesc(I, J, R, 1, 1, _p(a), lda, 0, 0, int(a_rc), _p(b), ldb, 0, 0, int(b_rc), _p(c), ldc, 0, 0, _p(bias), 1.0, act,
                          int(acc), 1, int(not (a_rc and b_rc)))
    check(_lib.lib().bdetr_gemm_grouped(arr, len(descs), _stream()), "gemm_grouped")


def linear_fwd_group(xs, ws, biases, act=ACT_NONE):
    """[x_k @ w_k^T + b_k for k]: same in/out widths, row counts may differ; one launch."""
    K_, O = xs[0].shape[1], ws[0].shape[0]
    ys = [empty(x.shape[0], O, like=x) for x in xs]
    gemm_grouped([(x.shape[0], O, K_, x, K_, True, w, K_, True, y, O, b, act, False) for x, w, b, y in zip(xs, ws, biases, ys)])
    return ys


def linear_bwd_data_group(dys, ws):
    """[dy_k @ w_k for k] in one launch."""
    O, K_ = ws[0].shape
    dxs = [empty(dy.shape[0], K_, like=dy) for dy in dys]
    gemm_grouped([(dy.shape[0], K_, O, dy, O, True, w, K_, False, dx, K_, None, ACT_NONE, False) for dy, w, dx in zip(dys, ws, dxs)])
    return dxs


def linear_fwd(x2d, w, bias, act=ACT_NONE):
    """y[m][o] = act(sum_i x[m][i] * w[o][i] + bias[o]);  w is [out][in]."""
    M, K = x2d.shape
    O = w.shape[0]
    y = empty(M, O, like=x2d)
    return gemm_raw(M, O, K, x2d, K, True, w, K, True, y, O, bias=bias, act=act)


def linear_bwd_data(dy2d, w, dx=None, accumulate=False):
    """dx[m][i] = sum_o dy[m][o] * w[o][i]."""
    M, O = dy2d.shape
    K = w.shape[1]
    if dx is None:
        dx = empty(M, K, like=dy2d)
    return gemm_raw(M, K, O, dy2d, O, True, w, K, False, dx, K, accumulate=accumulate, grad=True)


def _auto_splitk(I, J, R) -> int:
    cus = _lib.lib().bdetr_device_cus()
    tiles = max(1, ((I + 63) // 64) * ((J + 63) // 64))
    sk = max(1, min((2 * cus) // tiles, R // 128))      # floor: tiles x slices must not spill into a second round of workgroups
    return sk


def linear_bwd_weight(dy2d, x2d, dw=None, prezeroed: bool = False):
    """dw[o][i] = sum_m dy[m][o] * x[m][i]."""
    M, O = dy2d.shape
    K = x2d.shape[1]
    if dw is None:
        dw = empty(O, K, like=dy2d)
        prezeroed = False
    sk = _auto_splitk(O, K, M)
    if sk > 1 and not prezeroed:
        check(_lib.lib().bdetr_zero(_p(dw), dw.numel(), _stream()), "zero")
    return gemm_raw(O, K, M, dy2d, O, False, x2d, K, False, dw, K, splitk=sk, grad=True)


def linear_bwd_weight_group(dys, xs, dws, prezeroed):
    """dws[k][o][i] += sum_m dys[k][m][o] * xs[k][m][i] for up to 4 Dense layers of one shape in ONE split-K launch (float atomics);
    prezeroed[k]: dws[k] already holds zeros or a running sum.  Deterministic mode and ragged groups fall back to one launch each."""
    same = all(dy.shape == dys[0].shape and x.shape == xs[0].shape for dy, x in zip(dys, xs))
    if _DETERMINISTIC[0] or not same or len(dys) == 1 or len(dys) > 4:
        for dy, x, dw, pz in zip(dys, xs, dws, prezeroed):
            linear_bwd_weight(dy, x, dw=dw, prezeroed=pz)
        return dws
    M, O = dys[0].shape
    K_ = xs[0].shape[1]
    L = _lib.lib()
    for dw, pz in zip(dws, prezeroed):
        if not pz:
            check(L.bdetr_zero(_p(dw), dw.numel(), _stream()), "zero")
    tiles = len(dys) * max(1, ((O + 63) // 64) * ((K_ + 63) // 64))
    sk = max(1, min((2 * L.bdetr_device_cus()) // tiles, M // 128))
    arr = (GemmDesc * len(dys))()
    for k, (dy, x, dw) in enumerate(zip(dys, xs, dws)):
        _chk(dy, x, dw)
        arr[k] = GemmDesc(O, K_, M, 1, 1, _p(dy), O, 0, 0, 0, _p(x), K_, 0, 0, 0, _p(dw), K_, 0, 0, None, 1.0, ACT_NONE, 0, sk, 1)
    check(L.bdetr_gemm_grouped(arr, len(dys), _stream()), "gemm_grouped(weight gradients)")
    return dws


def colsum(x2d, out=None, prezeroed=False):
    """Column sums.  prezeroed: `out` already holds zeros (a slice of the step's zero-filled flat gradient buffer) - one
    launch that adds with float atomics instead of the two-kernel deterministic reduction."""
    _chk(x2d, out)
    L = _lib.lib()
    rows, cols = x2d.shape
    if prezeroed and out is not None and not _DETERMINISTIC[0]:
        check(L.bdetr_colsum_accumulate(_p(x2d), rows, cols, _p(out), _stream()), "colsum_accumulate")
        return out
    if out is None:
        out = empty(cols, like=x2d)
    ws = empty(L.bdetr_colsum_chunks(rows) * cols, like=x2d)
    check(L.bdetr_colsum(_p(x2d), rows, cols, _p(out), _p(ws), _stream()), "colsum")
    return out


def colsum_group(xs, outs):
    """out[m] += column sums of xs[m] for up to four [rows_m, cols] tensors of one width in ONE launch (float atomics: every out[m] must
    hold zeros or the running sum; the callers fall back to `colsum` per tensor in deterministic mode)."""
    _chk(*xs, *outs)
    n, cols = len(xs), xs[0].shape[1]
    assert 1 <= n <= 4 and all(x.dim() == 2 and x.shape[1] == cols for x in xs) and all(o.numel() == cols for o in outs)
    PA, LA = C.c_void_p * n, C.c_int64 * n
    check(_lib.lib().bdetr_colsum_accumulate_group(PA(*[x.data_ptr() for x in xs]), LA(*[x.shape[0] for x in xs]), cols, PA(*[o.data_ptr() for o in outs]), n,
                                                   _stream()), "colsum_accumulate_group")


# --------------------------------------------------------------------------------------
# BatchNorm
# --------------------------------------------------------------------------------------
def colstats(x2d):
    _chk(x2d)
    L = _lib.lib()
    rows, Cc = x2d.shape
    n = L.bdetr_bn_bwd_chunks(rows)
    psum, psq = empty(n, Cc, like=x2d), empty(n, Cc, like=x2d)
    check(L.bdetr_colstats(_p(x2d), rows, Cc, _p(psum), _p(psq), _stream()), "colstats")
    return psum, psq, n


def bn_stats(rows, Cc, parts, eps, momentum, bessel, moving_mean, moving_var, like):
    psum, psq, n = parts
    L = _lib.lib()
    mean, rstd = empty(Cc, like=like), empty(Cc, like=like)
    fold = L.bdetr_bn_stats_fold_rows()
    ws = empty(2 * fold * Cc, like=like) if n > 4 * fold else None
    check(L.bdetr_bn_stats(None, rows, Cc, _p(psum), _p(psq), n, eps, momentum, int(bessel), _p(mean), _p(rstd),
                           _p(moving_mean), _p(moving_var), _p(ws), _p(overflow_flag()) if _GUARD_ACTIVE[0] else None, _stream()), "bn_stats")
    return mean, rstd


def bn_stats_frozen(moving_mean, moving_var, eps):
    _chk(moving_mean, moving_var)
    Cc = moving_mean.numel()
    mean, rstd = empty(Cc, like=moving_mean), empty(Cc, like=moving_mean)
    check(_lib.lib().bdetr_bn_stats_frozen(_p(moving_mean), _p(moving_var), Cc, eps, _p(mean), _p(rstd), _stream()), "bn_stats_frozen")
    return mean, rstd


def bn_apply(x2d, mean, rstd, gamma, beta, residual=None, relu=False, out=None):
    _chk(x2d, mean, rstd, gamma, beta, residual, out)
    rows, Cc = x2d.shape
    if out is None:
        out = torch.empty_like(x2d)
    check(_lib.lib().bdetr_bn_apply(_p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(residual), int(relu), _p(out), rows, Cc, _stream()),
          "bn_apply")
    return out


def bn_bwd(dout, out, x2d, mean, rstd, gamma, relu, frozen, want_residual_grad=False, dgamma=None, dbeta=None, beta=None):
    _chk(dout, out, x2d, mean, rstd, gamma, dgamma, dbeta)
    L = _lib.lib()
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    dgamma = empty(Cc, like=x2d) if dgamma is None else dgamma
    dbeta = empty(Cc, like=x2d) if dbeta is None else dbeta
    dres = torch.empty_like(x2d) if want_residual_grad else None
    ws = empty(2 * Cc * L.bdetr_bn_bwd_chunks(rows), like=x2d)
    check(L.bdetr_bn_bwd(_p(dout), _p(out), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu), int(frozen), _p(dx), _p(dgamma), _p(dbeta),
                         _p(dres), _p(ws), rows, Cc, _stream()), "bn_bwd")
    return dx, dgamma, dbeta, dres


# --------------------------------------------------------------------------------------
# pooling / softmax / layernorm / activations
# --------------------------------------------------------------------------------------
def maxpool_fwd(x):
    _chk(x)
    N, H, W, Cc = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = empty(N, OH, OW, Cc, like=x)
    check(_lib.lib().bdetr_maxpool3x3s2_fwd(_p(x), _p(y), N, H, W, Cc, OH, OW, _stream()), "maxpool_fwd")
    return y


def maxpool_bwd(x, y, dy):
    _chk(x, y, dy)
    N, H, W, Cc = x.shape
    _, OH, OW, _ = y.shape
    dx = torch.empty_like(x)
    check(_lib.lib().bdetr_maxpool3x3s2_bwd(_p(x), _p(y), _p(dy), _p(dx), N, H, W, Cc, OH, OW, _stream()), "maxpool_bwd")
    return dx


def stem_pool_fwd(y, mean, rstd, gamma, beta, want_fp32=False):
    """BatchNorm -> ReLU -> pad 1 -> MaxPool 3x3/2 of the raw stem convolution output y [N,H,W,C]: (pooled fp32 | None, pooled f16
    pair, tap bytes) - csrc/norm.hip stem_pool_fwd_kernel."""
    _chk(y, mean, rstd, gamma, beta)
    N, H, W, Cc = y.shape
    PH, PW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    o32 = empty(N, PH, PW, Cc, like=y) if want_fp32 else None
    of = empty(N, PH, PW, Cc, like=y)
    tap = torch.empty((N, PH, PW, Cc), dtype=torch.uint8, device=y.device)
    check(_lib.lib().bdetr_stem_pool_fwd(_p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), N, H, W, Cc, _p(o32), _p(of), _p(tap),
                                         _p(overflow_flag()), _stream()), "stem_pool_fwd")
    return o32, of, tap


def stem_pool_bwd(dpool, tap, y, mean, rstd, gamma, beta, dgamma=None, dbeta=None, dy_p16=False):
    """The backward of stem_pool_fwd + BatchNorm (batch statistics): (dy [N,H,W,C], dgamma, dbeta).  dy_p16: dy as the bf16 pair
    of the P16 layout (what stem_bwd_weight_s2d reads) instead of fp32 - same shape, never read as floats."""
    _chk(dpool, y, mean, rstd, gamma, beta, dgamma, dbeta)
    assert tap.is_cuda and tap.is_contiguous() and tap.dtype == torch.uint8 and tuple(tap.shape) == tuple(dpool.shape)
    L = _lib.lib()
    N, H, W, Cc = y.shape
    dy = torch.empty_like(y)
    dgamma = empty(Cc, like=y) if dgamma is None else dgamma
    dbeta = empty(Cc, like=y) if dbeta is None else dbeta
    ws = empty(2 * Cc * L.bdetr_stem_pool_bwd_chunks(N * H * W), like=y)
    check(L.bdetr_stem_pool_bwd(_p(dpool), _p(tap), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), N, H, W, Cc, _p(dy), int(dy_p16), _p(dgamma), _p(dbeta),
                                _p(ws), _stream()), "stem_pool_bwd")
    return dy, dgamma, dbeta


def stem_bwd_weight_s2d(x4, dy_bf16, dw):
    """dw [K,7,7,4] <- the weight gradient of the ResNet stem (7x7 / stride 2 / pad 3 over the 4-channel image x4 [N,H,W,4], H and W even)
    from dy's bf16 pair [N,H/2,W/2,K]: space-to-depth image pack, the pre-split XX kernel on a 4x4 / 16-channel view, unpack (include/bdetr.h)."""
    _chk(x4, dy_bf16, dw)
    L = _lib.lib()
    N, H, W, Cc = x4.shape
    Kout = dw.shape[0]
    assert Cc == 4 and H % 2 == 0 and W % 2 == 0 and tuple(dw.shape) == (Kout, 7, 7, 4) and tuple(dy_bf16.shape) == (N, H // 2, W // 2, Kout)
    x2 = empty(N, H // 2, W // 2, 16, like=x4)
    check(L.bdetr_p16_s2d_pack_bf16(_p(x4), N, H, W, _p(x2), _stream()), "p16_s2d_pack_bf16")
    dw2 = empty(Kout, 4, 4, 16, like=x4)
    check(L.bdetr_zero(_p(dw2), dw2.numel(), _stream()), "zero")
    check(L.bdetr_p16_stem_bwd_weight(_p(x2), _p(dy_bf16), _p(dw2), N, H // 2, W // 2, Kout, _stream()), "p16_stem_bwd_weight")
    check(L.bdetr_p16_s2d_unpack_dw(_p(dw2), _p(dw), Kout, _stream()), "p16_s2d_unpack_dw")
    return dw


def attention_fwd(q, k, v, heads, scale):
    """q [B,nq,h*32], k/v [B,nk,h*32] -> o [B,h,nq,32], lse [B,h,nq] (fused; scores never hit HBM)."""
    _chk(q, k, v)
    B, nq, D = q.shape
    nk = k.shape[1]
    o = empty(B, heads, nq, D // heads, like=q)
    lse = empty(B, heads, nq, like=q)
    check(_lib.lib().bdetr_attention_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), B, heads, nq, nk, scale, _stream()), "attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, d_o, lse, heads, scale):
    _chk(q, k, v, o, d_o, lse)
    B, nq, D = q.shape
    nk = k.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = empty(B * heads * nq, like=q)
    check(_lib.lib().bdetr_attention_bwd(_p(q), _p(k), _p(v), _p(o), _p(d_o), _p(lse), _p(dq), _p(dk), _p(dv), _p(ws), B, heads, nq, nk,
                                         scale, _stream()), "attention_bwd")
    return dq, dk, dv


def softmax_rows_fwd(s2d, scale=1.0, out=None):
    _chk(s2d, out)
    rows, cols = s2d.shape
    if out is None:
        out = torch.empty_like(s2d)
    check(_lib.lib().bdetr_softmax_rows_fwd(_p(s2d), _p(out), rows, cols, scale, _stream()), "softmax_fwd")
    return out


def softmax_rows_bwd(p2d, dp2d, scale=1.0, out=None):
    _chk(p2d, dp2d, out)
    rows, cols = p2d.shape
    if out is None:
        out = torch.empty_like(p2d)
    check(_lib.lib().bdetr_softmax_rows_bwd(_p(p2d), _p(dp2d), _p(out), rows, cols, scale, _stream()), "softmax_bwd")
    return out


def add_dropout_layernorm_fwd(x2d, y2d, gamma, beta, eps, rate=0.0, seed=0, seed_base=None):
    """seed_base: device int64 [1] holding the per-step seed (folded into `seed` on the device)."""
    _chk(x2d, y2d, gamma, beta)
    rows, D = x2d.shape
    out = torch.empty_like(x2d)
    mean, rstd = empty(rows, like=x2d), empty(rows, like=x2d)
    check(_lib.lib().bdetr_add_dropout_layernorm_fwd(_p(x2d), _p(y2d), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), rows, D, eps, rate,
                                                     seed, _p(seed_base), _stream()), "add_dropout_layernorm_fwd")
    return out, mean, rstd


def add_dropout_layernorm_bwd(dout, x2d, y2d, gamma, mean, rstd, rate=0.0, seed=0, dgamma=None, dbeta=None, seed_base=None):
    _chk(dout, x2d, y2d, gamma, mean, rstd, dgamma, dbeta)
    L = _lib.lib()
    rows, D = x2d.shape
    dx, dy = torch.empty_like(x2d), torch.empty_like(x2d)
    dgamma = empty(D, like=x2d) if dgamma is None else dgamma
    dbeta = empty(D, like=x2d) if dbeta is None else dbeta
    ws = empty(2 * D * L.bdetr_ln_bwd_chunks(rows), like=x2d)
    check(L.bdetr_add_dropout_layernorm_bwd(_p(dout), _p(x2d), _p(y2d), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dy), _p(dgamma), _p(dbeta),
                                            _p(ws), rows, D, rate, seed, _p(seed_base), 0, _stream()), "add_dropout_layernorm_bwd")
    return dx, dy, dgamma, dbeta


# --------------------------------------------------------------------------------------
# row chain (csrc/rowchain.hip): out-projection + LayerNorm (+ FFN + LayerNorm) of a transformer layer in one launch
# --------------------------------------------------------------------------------------
ROWCHAIN_WIDTH = 256


def rowchain_pack_weights(table: torch.Tensor) -> None:
    """table: int64 [n,3] device tensor of {W [256,256] fp32, forward copy | 0, backward copy | 0} rows."""
    _chk(table, dtype=torch.int64)
    check(_lib.lib().bdetr_rowchain_pack_weights(_p(table), table.shape[0], _p(overflow_flag()), _stream()), "rowchain_pack_weights")


def rowchain_fwd(ctx2d, resid2d, packs, biases, ln1, ln2, eps, rate, seed1, seed2, seed_base):
    """packs / biases: 1 or 3 packed forward copies / bias vectors (Wo[, W1, W2]); ln1 / ln2: (gamma, beta) value tensors (ln2 None for
    one stage).  Returns a dict of the outputs and saved tensors."""
    n = len(packs)
    _chk(ctx2d, resid2d, *biases, *ln1, *(ln2 or ()))
    M, D = ctx2d.shape
    assert D == ROWCHAIN_WIDTH and n in (1, 3) and resid2d.shape == ctx2d.shape
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=ctx2d.device)
    o = {"pre1": e(M, D), "x1": e(M, D), "mean1": e(M), "rstd1": e(M)}
    if n == 3:
        o.update({"h": e(M, D), "pre2": e(M, D), "x2": e(M, D), "mean2": e(M), "rstd2": e(M)})
    d = RowchainFwdDesc()
    d.M, d.nstages, d.eps, d.rate = M, n, float(eps), float(rate)
    d.ctx, d.resid = _p(ctx2d), _p(resid2d)
    for k in range(n):
        d.w[k], d.bias[k] = _p(packs[k]), _p(biases[k])
    d.g1, d.b1 = _p(ln1[0]), _p(ln1[1])
    if n == 3:
        d.g2, d.b2 = _p(ln2[0]), _p(ln2[1])
    for k, v in o.items():
        setattr(d, k, _p(v))
    d.seed1, d.seed2, d.seed_base = int(seed1), int(seed2), _p(seed_base)
    check(_lib.lib().bdetr_rowchain_fwd(C.byref(d), _stream()), "rowchain_fwd")
    return o


def rowchain_bwd(dout2d, saved, packs_t, gammas, rate, seed1, seed2, seed_base):
    """saved: rowchain_fwd's dict; packs_t: the packed backward copies; gammas: (gamma1[, gamma2]).  Returns (dctx, dresid, [G0, G1, G2],
    partials, nparts)."""
    n = len(packs_t)
    _chk(dout2d)
    M, D = dout2d.shape
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dout2d.device)
    L = _lib.lib()
    nparts = L.bdetr_rowchain_partial_rows(M)
    G = [e(M, D) for _ in range(n)]
    dctx, dresid, partials = e(M, D), e(M, D), e(nparts, 7, D)
    d = RowchainBwdDesc()
    d.M, d.nstages, d.rate, d.dout = M, n, float(rate), _p(dout2d)
    d.pre1, d.mean1, d.rstd1, d.g1 = _p(saved["pre1"]), _p(saved["mean1"]), _p(saved["rstd1"]), _p(gammas[0])
    if n == 3:
        d.pre2, d.mean2, d.rstd2, d.g2, d.h = _p(saved["pre2"]), _p(saved["mean2"]), _p(saved["rstd2"]), _p(gammas[1]), _p(saved["h"])
        d.G1, d.G2 = _p(G[1]), _p(G[2])
    for k in range(n):
        d.wt[k] = _p(packs_t[k])
    d.G0, d.dresid, d.dctx, d.partials = _p(G[0]), _p(dresid), _p(dctx), _p(partials)
    d.seed1, d.seed2, d.seed_base = int(seed1), int(seed2), _p(seed_base)
    check(L.bdetr_rowchain_bwd(C.byref(d), _stream()), "rowchain_bwd")
    return dctx, dresid, G, partials, nparts


def rowchain_reduce(partials, nparts, dsts, accumulate):
    """dsts: 7 tensors or None ({dgamma2, dbeta2, dbias2, dbias1, dgamma1, dbeta1, dbias0}); accumulate: 7 flags."""
    ptrs = (C.c_void_p * 7)(*[_p(t) for t in dsts])
    acc = (C.c_int * 7)(*[int(bool(a)) for a in accumulate])
    check(_lib.lib().bdetr_rowchain_reduce(_p(partials), nparts, C.byref(ptrs), C.byref(acc), _stream()), "rowchain_reduce")


def _unary(name, x, out=None):
    _chk(x, out)
    if out is None:
        out = torch.empty_like(x)
    check(getattr(_lib.lib(), name)(_p(x), _p(out), x.numel(), _stream()), name)
    return out


def _binary(name, a, b, out=None):
    _chk(a, b, out)
    if out is None:
        out = torch.empty_like(a)
    check(getattr(_lib.lib(), name)(_p(a), _p(b), _p(out), a.numel(), _stream()), name)
    return out


def sigmoid_fwd(x): return _unary("bdetr_sigmoid_fwd", x)
def sigmoid_bwd(y, dy): return _binary("bdetr_sigmoid_bwd", y, dy)
def boxsigmoid_fwd(x): return _unary("bdetr_boxsigmoid_fwd", x)
def boxsigmoid_bwd(y, dy): return _binary("bdetr_boxsigmoid_bwd", y, dy)
def tanh_bwd(y, dy): return _binary("bdetr_tanh_bwd", y, dy)
def relu_bwd(y, dy): return _binary("bdetr_relu_bwd", y, dy)
def add(a, b, out=None): return _binary("bdetr_add", a, b, out)


def zero_(t):
    _chk(t)
    check(_lib.lib().bdetr_zero(_p(t), t.numel(), _stream()), "zero")
    return t


def axpy_(alpha, x, y):
    _chk(x, y)
    check(_lib.lib().bdetr_axpy(float(alpha), _p(x), _p(y), x.numel(), _stream()), "axpy")
    return y


def add_bcast_rows(a, row):
    """a [B, *rest] + row [*rest]"""
    _chk(a, row)
    out = torch.empty_like(a)
    rowlen = row.numel()
    check(_lib.lib().bdetr_add_bcast_rows(_p(a), _p(row), _p(out), a.numel() // rowlen, rowlen, _stream()), "add_bcast_rows")
    return out


def sum_over_batch(x, n, out=None):
    _chk(x, out)
    if out is None:
        out = empty(n, like=x)
    check(_lib.lib().bdetr_sum_over_batch(_p(x), _p(out), x.numel() // n, n, 0, _stream()), "sum_over_batch")
    return out


# --------------------------------------------------------------------------------------
# set criterion
# --------------------------------------------------------------------------------------
def loss_desc(B, M, N, Cc, A, category_weight, attribute_weight, box_weight, exist_weight) -> LossDesc:
    return LossDesc(B, M, N, Cc, A, category_weight, attribute_weight, box_weight, exist_weight)


def cost_matrix(d: LossDesc, cat_pred, att_pred, box_pred, cat_ids, att_hot, bbox, num_objects, components=False):
    _chk(cat_pred, att_pred, box_pred, att_hot, bbox)
    _chk(cat_ids, num_objects, dtype=torch.int32)
    cost = empty(d.B, d.M, d.N, like=cat_pred)
    comps = [empty(d.B, d.M, d.N, like=cat_pred) for _ in range(3)] if components else [None] * 3
    check(_lib.lib().bdetr_cost_matrix(C.byref(d), _p(cat_pred), _p(att_pred), _p(box_pred), _p(cat_ids), _p(att_hot), _p(bbox),
                                       _p(num_objects), _p(cost), _p(comps[0]), _p(comps[1]), _p(comps[2]), _stream()), "cost_matrix")
    return (cost, comps) if components else cost


def lsa(cost, num_objects):
    _chk(cost)
    _chk(num_objects, dtype=torch.int32)
    B, M, N = cost.shape
    match = empty(B, M, like=cost, dtype=torch.int32)
    check(_lib.lib().bdetr_lsa(_p(cost), _p(num_objects), B, M, N, _p(match), _stream()), "lsa")
    return match


def match_to_mask(match, N):
    _chk(match, dtype=torch.int32)
    B, M = match.shape
    mask = empty(B, M, N, like=match, dtype=torch.float32)
    check(_lib.lib().bdetr_match_to_mask(_p(match), _p(mask), B, M, N, _stream()), "match_to_mask")
    return mask


def set_loss(d: LossDesc, cat_pred, att_pred, box_pred, cat_ids, att_hot, bbox, num_objects, match, loss_scale=1.0, want_grads=True):
    _chk(cat_pred, att_pred, box_pred, att_hot, bbox)
    _chk(cat_ids, num_objects, match, dtype=torch.int32)
    losses = empty(6, d.B, like=cat_pred)
    d_cat = torch.empty_like(cat_pred) if want_grads else None
    d_att = torch.empty_like(att_pred) if (want_grads and att_pred is not None) else None
    d_box = torch.empty_like(box_pred) if want_grads else None
    check(_lib.lib().bdetr_set_loss(C.byref(d), _p(cat_pred), _p(att_pred), _p(box_pred), _p(cat_ids), _p(att_hot), _p(bbox), _p(num_objects),
                                    _p(match), _p(losses), _p(d_cat), _p(d_att), _p(d_box), float(loss_scale), _stream()), "set_loss")
    return losses, d_cat, d_att, d_box


# --------------------------------------------------------------------------------------
# panoptic head pieces (forward only) - csrc/panoptic.hip
# --------------------------------------------------------------------------------------
def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def resize_bilinear(x, H: int, W: int):
    _chk(x)
    B, h, w, Cc = x.shape
    out = empty(B, H, W, Cc, like=x)
    check(_lib.lib().bdetr_resize_bilinear_nhwc(_p(x), B, h, w, Cc, _p(out), H, W, _stream()), "resize_bilinear")
    return out


def layernorm_act(x, C_true: int, gamma, beta, eps: float, slope: float = 1.0, ld_out: Optional[int] = None):
    """LayerNormalization over the first C_true channels of x [..., ld] (+ leaky ReLU with `slope`); output [..., ld_out], pad = 0."""
    _chk(x, gamma, beta)
    ld = x.shape[-1]
    ldo = pad4(C_true) if ld_out is None else ld_out
    out = empty(*x.shape[:-1], ldo, like=x)
    rows = x.numel() // ld
    check(_lib.lib().bdetr_layernorm_act_fwd(_p(x), rows, C_true, ld, _p(gamma), _p(beta), eps, slope, _p(out), ldo, _stream()), "layernorm_act")
    return out


def copy_cols(src, C_true: int, dst, col0: int):
    _chk(src, dst)
    rows = src.numel() // src.shape[-1]
    check(_lib.lib().bdetr_copy_cols(_p(src), rows, C_true, src.shape[-1], _p(dst), dst.shape[-1], col0, _stream()), "copy_cols")
    return dst


def nhwc_to_nchw(x, C_true: int):
    _chk(x)
    B, H, W, ld = x.shape
    out = empty(B, C_true, H * W, like=x)
    check(_lib.lib().bdetr_nhwc_to_nchw(_p(x), B, H * W, C_true, ld, _p(out), _stream()), "nhwc_to_nchw")
    return out
