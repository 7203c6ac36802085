"""Tape-recorded composite ops of the hot path.  Each function runs HIP kernels for the
forward and, when a Tape is recording, registers the closure that runs the backward kernels.

Activation gradients travel through the Tape; PARAMETER gradients do not: each backward
closure writes them straight into the variable's slice of the optimizer's flat gradient buffer
(``GradSink``), so there is no per-tensor staging copy before the all-reduce / optimizer.

Tensors are NHWC / [B,T,D]; ops view them as 2-D row matrices internally.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import kernels as K
from .engine import WEIGHTS_VERSION, Variable, current_tape, materialise, side_task

# Dropout masks are keyed by (per-step seed, dropout site, element index).  The per-step seed lives in HBM
# (one int64 the host rewrites before every step), the site salt is a launch argument: a captured step graph then
# draws fresh masks on every replay.
_dropout_site = [0]
_dropout_seed_dev = [None]


def dropout_seed_tensor() -> torch.Tensor:
    if _dropout_seed_dev[0] is None:
        _dropout_seed_dev[0] = torch.full((1,), 0x5EED, dtype=torch.int64, device="cuda")
    return _dropout_seed_dev[0]


def set_dropout_seed(seed: int, write: bool = True) -> None:
    """Start a step: reset the site counter and (unless a graph replay already staged it) store the step's seed."""
    if write:
        dropout_seed_tensor().fill_(int(seed) & 0x7FFFFFFFFFFF)
    _dropout_site[0] = 0


def _next_dropout_seed() -> int:
    _dropout_site[0] += 1
    return (_dropout_site[0] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


def _rec(outputs, inputs, fn):
    t = current_tape()
    if t is not None:
        t.record(outputs, inputs, fn)


def _2d(t: torch.Tensor) -> torch.Tensor:
    return t.view(-1, t.shape[-1])


def _own(t: torch.Tensor) -> torch.Tensor:
    """Mark a freshly produced gradient tensor as solely owned by the Tape (safe to accumulate into)."""
    t._bdetr_owned = True
    return t


_live_flat_grad = [None]


def set_live_flat_grad(flat: Optional[torch.Tensor]) -> None:
    """The flat gradient buffer that was zero-filled for THIS step (Model.forward_backward), or None.  A
    Variable.grad_buf is only written in place when it is a slice of exactly this buffer: after compile(new
    optimizer) or a freeze/unfreeze cycle a variable may still carry a slice of a retired buffer that holds an
    earlier step's gradients, and the split-K atomics would add onto them."""
    _live_flat_grad[0] = flat


_grad_ready_hook = [None]


def set_grad_ready_hook(fn) -> None:
    """fn(var) is called after every gradient contribution to a parameter has been enqueued (data-parallel training
    launches a bucket's all-reduce once all of its gradients are complete and live in the flat buffer)."""
    _grad_ready_hook[0] = fn


class GradSink:
    """Where a backward kernel writes one parameter's gradient.

    direct: first contribution of the step and the variable owns a slice of the flat buffer zeroed this step -> in place.
    temp  : later contribution (shared layer), no optimizer yet, or a stale slice -> temporary, added on commit.
    drop  : the variable is frozen but the kernel always produces the value -> scratch."""

    __slots__ = ("var", "buf", "mode")

    def __init__(self, var: Variable):
        self.var = var
        live = _live_flat_grad[0]
        if not var.needs_grad:
            self.mode, self.buf = "drop", torch.empty_like(var.value)
        elif var.grad is None and var.grad_buf is not None and live is not None and getattr(var, "_grad_flat", None) is live:
            self.mode, self.buf = "direct", var.grad_buf      # zero-filled at the start of the step (Model.forward_backward)
        else:
            self.mode, self.buf = "temp", torch.empty_like(var.value)

    def commit(self) -> None:
        v = self.var
        if self.mode == "direct":
            v.grad, v._grad_fresh = v.grad_buf, True
        elif self.mode == "temp":
            if v.grad is None:
                v.grad = self.buf
            else:
                K.axpy_(1.0, self.buf.view(v.grad.shape), v.grad)
        if self.mode != "drop" and _grad_ready_hook[0] is not None:
            _grad_ready_hook[0](v)


# ----------------------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------------------
def image_prep(image: torch.Tensor, H: int, W: int) -> torch.Tensor:
    return K.image_prep(image, H, W)     # input images need no gradient


COMPACT_S2 = os.environ.get("BDETR_COMPACT_S2", "1") != "0"       # stride-2 1x1 input gradients as compact even-pixel tensors (conv_bn backward)


class BNState:
    """gamma/beta/moving stats of one BatchNormalization layer + its hyper-parameters."""

    def __init__(self, gamma: Variable, beta: Variable, moving_mean: Variable, moving_var: Variable, eps: float,
                 momentum: float = 0.99):
        self.gamma, self.beta, self.moving_mean, self.moving_var = gamma, beta, moving_mean, moving_var
        self.eps, self.momentum = eps, momentum


def _bn_forward(y2d, rows, Cc, parts, bn: BNState, use_batch_stats: bool, bessel: bool, residual2d, relu: bool):
    if use_batch_stats:
        if parts is None or parts[0] is None:
            parts = K.colstats(y2d)
        mean, rstd = K.bn_stats(rows, Cc, parts, bn.eps, bn.momentum, bessel, bn.moving_mean.value, bn.moving_var.value, like=y2d)
    else:
        K.demote_split_forward()
        mean, rstd = K.bn_stats_frozen(bn.moving_mean.value, bn.moving_var.value, bn.eps)
    out = K.bn_apply(y2d, mean, rstd, bn.gamma.value, bn.beta.value, residual2d, relu)
    return out, mean, rstd


def _bn_backward(g2d, out2d, x2d, mean, rstd, bn: BNState, relu: bool, frozen: bool, want_residual_grad: bool):
    sg, sb = GradSink(bn.gamma), GradSink(bn.beta)
    # without a residual the ReLU mask is a function of x alone: the kernel recomputes it instead of reading `out`
    mask_src = out2d if (relu and want_residual_grad) else None
    dx, _, _, dres = K.bn_bwd(g2d, mask_src, x2d, mean, rstd, bn.gamma.value, relu, frozen, want_residual_grad=want_residual_grad,
                              dgamma=sg.buf, dbeta=sb.buf, beta=bn.beta.value)
    sg.commit()
    sb.commit()
    return dx, dres


# ---- pre-split (P16) operand path of the backbone convolutions (csrc/sgemm.hip) ----
# Under the 'split' policy a training-mode Conv+BN unit whose channel counts allow it runs on operands that their
# producers already wrote as 16-bit pairs: BatchNorm apply writes the f16 pair (forward operand of the next conv)
# and the bf16 pair (its weight-gradient operand), BatchNorm backward writes the bf16 pair of dy, the weights are
# packed once per optimizer step.  A tensor handle carries its packed companions as attributes:
#   t._p16f / t._p16b   P16-f16 / P16-bf16 copies (fp32-shaped torch tensors, never read as floats)
#   t._p16_only         the handle IS the f16 copy: no fp32 tensor was materialised (links inside a bottleneck)
P16_ENABLED = [os.environ.get("BDETR_P16", "1") != "0"]
# Round 3: the weight gradient reads the f16 pair the FORWARD of its convolution read (converted to bf16 pairs inside the kernel,
# bdetr_p16_conv2d_bwd_weight_xf16), so activations have no bf16 pair copy at all: bn_apply_p16 writes 4 bytes per element less.
WGRAD_XF16 = os.environ.get("BDETR_WGRAD_XF16", "1") != "0"
BF16_FOR_3X3 = os.environ.get("BDETR_BF16_3X3", "1") != "0"       # ... except in front of a 3x3 convolution (conv_bn want_bf16)
EVEN_PIXELS = os.environ.get("BDETR_EVEN_PIXELS", "1") != "0"  # stage-last BatchNorm backward reduces over the even pixels only (conv_bn.backward)
LAZY_SKIP = os.environ.get("BDETR_LAZY_SKIP", "1") != "0"      # residual units hand their skip gradient on unmasked (conv_bn.backward)


def _p16_active() -> bool:
    return P16_ENABLED[0] and K.get_gemm_precision() == "split"


def as_fp32(t: torch.Tensor) -> torch.Tensor:
    """The fp32 tensor behind a handle (unpacks a P16-only handle: 2^-23 relative round trip; applies a deferred
    BatchNorm: conv_bn(defer_apply=True))."""
    if getattr(t, "_p16_only", False):
        return K.p16_unpack(t._p16f, True).view(t.shape)
    dbn = getattr(t, "_deferred_bn", None)
    if dbn is not None:
        return K.bn_apply(_2d(t), *dbn, None, False).view(t.shape)
    return t


def _packed_input(x: torch.Tensor, need_bf16: bool):
    """(f16 pair, bf16 pair | None) of an activation, packing an fp32 handle once and caching the result on it."""
    xf, xb = getattr(x, "_p16f", None), getattr(x, "_p16b", None)
    need_bf16 = need_bf16 and not WGRAD_XF16
    if xf is None or (need_bf16 and xb is None):
        f, b = K.p16_pack(x, want_f16=xf is None, want_bf16=need_bf16 and xb is None)
        xf, xb = (f if xf is None else xf), (b if b is not None else xb)
        x._p16f, x._p16b = xf, xb
    return xf, xb


class _PackedWeights:
    """Both P16 copies of every conv kernel that takes the pre-split path, in persistent buffers, refreshed by ONE
    multi-tensor launch the first time a copy is asked for after the weights changed (optimizer step / assign).
    The table keeps the buffers of its rows alive; rows of variables that no longer exist are dropped whenever the
    table is rebuilt (a new variable registers)."""

    def __init__(self):
        self.rows, self.table, self.version = [], None, -1       # rows: (weakref(var), value, wf, wt)

    def get(self, w: Variable):
        import weakref
        c = getattr(w, "_p16", None)
        if c is None or c[0].data_ptr() != w.value.data_ptr():
            Kout, R, S, Cin = w.value.shape
            c = (w.value, torch.empty_like(w.value), torch.empty((Cin, R, S, Kout), dtype=torch.float32, device=w.value.device))
            w._p16 = c
            self.rows = [r for r in self.rows if r[0]() is not None and r[0]() is not w]
            self.rows.append((weakref.ref(w),) + c)
            self.table, self.version = None, -1
        if self.version != WEIGHTS_VERSION[0]:
            if self.table is None:
                import numpy as np
                rows = np.array([[v.data_ptr(), wf.data_ptr(), wt.data_ptr(), *v.shape] for _, v, wf, wt in self.rows], np.int64)
                self.table = torch.from_numpy(rows).to(w.value.device)
            K.p16_pack_conv_weights_multi(self.table)
            self.version = WEIGHTS_VERSION[0]
        return c[1], c[2]


_PACKED = _PackedWeights()


def packed_weights(w: Variable, need_bwd: bool = True):
    """(P16-f16 forward copy [K,R,S,C], P16-bf16 transposed tap-flipped copy [C,R,S,K]) of a conv kernel."""
    return _PACKED.get(w)


def conv_bn(x: torch.Tensor, w: Variable, b: Variable, bn: BNState, stride: int, pad: int, relu: bool,
            residual: Optional[torch.Tensor] = None, training: bool = False, bn_batch_stats: Optional[bool] = None,
            x_needs_grad: bool = True, want_fp32: bool = True, want_p16: bool = False, defer_apply: bool = False,
            sole_consumer_is_identity_unit: bool = False, want_bf16: bool = False) -> torch.Tensor:
    """Conv2D(+bias) -> BatchNormalization -> [+ residual] -> [ReLU]  (keras ResNet-50 block unit).

    sole_consumer_is_identity_unit (residual units): the only consumers of this output are the next unit's first 1x1
    convolution and its identity skip - the masked accumulate that completes this output's gradient there can then also do
    THIS unit's BatchNorm-backward reduction (ctx attached to the handle as `_bn_ctx_bits`).
    defer_apply (projection shortcut, no ReLU, no residual; P16 path only): the statistics are reduced but the normalised
    tensor is not written - the returned handle is the RAW convolution output tagged `_deferred_bn`, and the unit that takes
    it as `residual` applies both BatchNorms in its one pass (bn_apply_p16 residual_bn; as_fp32 is the fallback).

    want_p16: the consumer is another conv_bn - also emit the packed copy of the output (P16 path only): the f16 pair, which feeds
    the consumer's forward AND (converted in registers) its weight gradient; want_bf16: additionally a bf16 pair copy for that weight
    gradient - worth its 4 bytes per element only for the quarter-width tensor in front of a 3x3 convolution;
    want_fp32=False: no fp32 output at all, the returned handle is the f16 copy (valid on the P16 path only,
    otherwise ignored)."""
    N, H, W, Cin = x.shape
    Kout, R, S, _ = w.value.shape
    g = K.ConvGeom(N, H, W, Cin, Kout, R, S, stride, pad)
    use_batch = training if bn_batch_stats is None else bn_batch_stats
    p16 = training and use_batch and _p16_active() and K.p16_supported(g)
    x_handle, res_handle = x, residual                 # the Tape keys gradients by the handles their producers returned
    res_p16 = p16 and residual is not None and getattr(residual, "_p16_only", False)
    res_bn = getattr(residual, "_deferred_bn", None) if (p16 and residual is not None) else None
    if residual is not None and not res_p16 and res_bn is None:
        residual = as_fp32(residual)
    res2d = _2d(residual) if residual is not None else None     # a P16-only handle IS its f16 pair copy
    xb = ob = relu_bits = None
    if p16:
        xf, xb = _packed_input(x, need_bf16=w.needs_grad)
        wf, _ = packed_weights(w, need_bwd=False)
        y, parts = K.p16_conv2d_fwd(xf, wf, b.value, g, K.ACT_NONE, want_stats=True)
        y2d = _2d(y)
        mean, rstd = K.bn_stats(g.M, Kout, parts, bn.eps, bn.momentum, True, bn.moving_mean.value, bn.moving_var.value, like=y2d)
        fp32_out = want_fp32 or not want_p16
        # a residual unit without an fp32 output: its backward ReLU mask is a 1-bit-per-element by-product of this pass
        want_mask = relu and residual is not None and not fp32_out
        if defer_apply and not relu and residual is None:
            out2d = None
            out = y.view(N, g.OH, g.OW, Kout)            # (a fresh handle: y itself stays this unit's own tensor)
            out._deferred_bn = (mean, rstd, bn.gamma.value, bn.beta.value)
        else:
            o32, of, ob, *rest = K.bn_apply_p16(y2d, mean, rstd, bn.gamma.value, bn.beta.value, res2d, relu, want_fp32=fp32_out,
                                                want_f16=want_p16, want_bf16=want_p16 and ((want_bf16 and BF16_FOR_3X3) or not WGRAD_XF16), residual_p16=res_p16, want_mask=want_mask,
                                                residual_bn=res_bn)
            relu_bits = rest[0] if want_mask else None
            out2d = o32
            out = (o32 if fp32_out else of).view(N, g.OH, g.OW, Kout)
        if want_p16:
            out._p16f, out._p16b, out._p16_only = of.view(out.shape), (ob.view(out.shape) if ob is not None else None), not fp32_out
        if relu_bits is not None and sole_consumer_is_identity_unit and os.environ.get("BDETR_BN_FUSE", "1") != "0":
            out._bn_ctx_bits = (y2d, mean, rstd, bn.gamma.value, bn.beta.value, relu_bits)
            if res_bn is not None and os.environ.get("BDETR_BN_FUSE2", "1") != "0":
                # a stage's first unit: the projection shortcut's BatchNorm (deferred: res2d is its RAW conv output) gets the same gradient -
                # its sum(g * xhat0) rides the same epilogue (round 4)
                out._bn_ctx_bits2 = (res2d, res_bn[0], res_bn[1])
        if not fp32_out and residual is None and os.environ.get("BDETR_BN_FUSE", "1") != "0":
            # a link with exactly one consumer (the next conv of the bottleneck): that conv's backward-data epilogue can do
            # THIS BatchNorm's backward reduction while it stores the gradient (ops: see `backward` below)
            out._bn_ctx = (y2d, mean, rstd, bn.gamma.value, bn.beta.value, relu)
    else:
        x = as_fp32(x)
        y, parts = K.conv2d_fwd(x, w.value, b.value, g, K.ACT_NONE, want_stats=use_batch)
        y2d = _2d(y)
        out2d, mean, rstd = _bn_forward(y2d, g.M, Kout, parts, bn, use_batch, True, res2d, relu)
        out = out2d.view(N, g.OH, g.OW, Kout)

    def backward(g_out, acc=None):
        want_res = residual is not None
        # A gradient that exists at the pixels (2i, 2j) only arrives as the compact [N, OH/2, OW/2, K] tensor its producers - the next
        # stage's stride-2 1x1 backward-data products, below - wrote densely (round 5: no zero-filled dense tensor, no scatter).  A
        # residual unit on the pre-split path reads it through the pixel map in both of its consumers (this BatchNorm backward and
        # the skip merge of the unit's first convolution); every other case gets the dense form (engine.materialise).
        compact = getattr(g_out, "_compact_even", None)
        if compact is not None and not (p16 and _p16_active() and relu and want_res and relu_bits is not None and LAZY_SKIP and compact == (N, g.OH, g.OW)
                                         and getattr(g_out, "_bdetr_owned", False) and getattr(g_out, "_lazy_mask", None) is None
                                         and getattr(g_out, "_bnb_parts", None) is None and g_out.is_contiguous()):
            g_out, compact = materialise(g_out), None
        lazy_bits = getattr(g_out, "_lazy_mask", None)      # g_out still needs its producer's ReLU mask (see below)
        if lazy_bits is not None and not (p16 and _p16_active() and not relu and residual is None):
            g_out, lazy_bits = materialise(g_out), None
        g2d = _2d(g_out.contiguous())
        if p16 and _p16_active():
            sg, sb = GradSink(bn.gamma), GradSink(bn.beta)
            # ReLU mask: recomputed from y when there is no residual, else read from the forward output (fp32, or the
            # hi halves of its bf16 pair copy when the output was never materialised in fp32)
            # ReLU mask: recomputed from y when there is no residual, else the forward output (fp32), or - when that was never
            # materialised in fp32 - the bit mask bn_apply wrote (else the hi halves of the bf16 pair copy)
            mask_src, mode = None, 0
            if relu and want_res:
                mask_src, mode = (out2d, 0) if out2d is not None else ((relu_bits, 2) if relu_bits is not None else (ob, 1))
                assert mask_src is not None, "a residual unit without an fp32 output keeps its ReLU bit mask"
            pre = getattr(g_out, "_bnb_parts", None)        # the reduction came with the gradient (fused into the consumer's epilogue)
            # The skip gradient of a residual unit is g_out * mask.  With the bit mask at hand it is not written out: the
            # incoming gradient tensor itself is handed to the shortcut's producer tagged with the mask, and the consumers
            # that know the tag (the 1x1 backward-data below, the projection shortcut's BatchNorm backward) fold the mask into
            # their own kernels - one 4-byte-per-element write per unit less (engine.materialise is the fallback).
            lazy_skip = (want_res and mode == 2 and LAZY_SKIP and getattr(g_out, "_bdetr_owned", False) and lazy_bits is None
                         and g_out.is_contiguous())
            bn_relu = relu
            if lazy_bits is not None:                       # this BatchNorm has no ReLU of its own: apply the incoming mask instead
                mask_src, mode, bn_relu = lazy_bits, 2, True
            even = getattr(g_out, "_even_pixels", None) if EVEN_PIXELS else None     # zero off the even pixels: a quarter-size reduction pass
            if even is not None and even != (N, g.OH, g.OW):
                even = None
            if compact is not None:
                assert lazy_skip and pre is None, "a compact even-pixel gradient is handed on to the skip merge as it is"
                even = compact
            dyb, _, _, _, dres = K.bn_bwd_p16(g2d, mask_src, y2d, mean, rstd, bn.gamma.value, bn_relu, False,
                                              want_residual_grad=want_res and not lazy_skip,
                                              dgamma=sg.buf, dbeta=sb.buf, beta=bn.beta.value, out_p16=mode, pre=pre, even_pixels=even,
                                              dout_compact=compact is not None)
            if lazy_skip:
                dres = _own(g_out.view(residual.shape)) if compact is None else _own(g_out)      # (compact: keeps its `_compact_even` tag)
                dres._lazy_mask = relu_bits
                sc_parts = getattr(g_out, "_bnb_parts_shortcut", None)
                if sc_parts is not None and res_bn is not None:
                    dres._bnb_parts = sc_parts          # the shortcut's BatchNorm backward finds its reduction done (it reads them as `pre`)
            sg.commit()
            sb.commit()
            dyb4 = dyb.view(N, g.OH, g.OW, Kout)
            if w.needs_grad or b.needs_grad:
                xw, xw_f16 = (xb, False) if xb is not None else (xf, True)     # the weight gradient's x operand: a bf16 pair copy where the producer wrote one, else the forward's f16 pair

                def param_grads(xw=xw, xw_f16=xw_f16, dyb4=dyb4):
                    if w.needs_grad:
                        s = GradSink(w)
                        K.p16_conv2d_bwd_weight(xw, dyb4, g, dw=s.buf, prezeroed=s.mode == "direct", x_f16=xw_f16)
                        s.commit()
                    if b.needs_grad:
                        s = GradSink(b)             # a bias in front of a batch-statistics BN has an exactly zero gradient (see below)
                        if s.mode != "direct":
                            K.zero_(s.buf)
                        s.commit()
                side_task(param_grads, xw, dyb)
            dx = None
            s2 = R == 1 and S == 1 and stride == 2 and pad == 0
            # a stride-2 1x1 convolution's input gradient lives at the pixels (2i, 2j): on an even map it is produced as the compact
            # [N, H/2, W/2, C] tensor - a plain dense product over the OUTPUT grid - and tagged; see the top of this function
            s2c = s2 and COMPACT_S2 and H % 2 == 0 and W % 2 == 0
            gc = K.ConvGeom(N, g.OH, g.OW, Cin, Kout, 1, 1, 1, 0) if s2c else None
            if x_needs_grad:
                _, wt = packed_weights(w, need_bwd=True)
                ctx = getattr(x_handle, "_bn_ctx", None)
                have_c = getattr(acc[0], "_compact_even", None) if acc is not None and acc[0] is not None else None
                if have_c is not None and s2c and have_c == (N, H, W) and getattr(acc[0], "_lazy_mask", None) is None:
                    # the other stride-2 consumer of x was first: add into its compact tensor
                    K.p16_conv2d_bwd_data(dyb4, wt, gc, dx=acc[0].view(N, g.OH, g.OW, Cin), accumulate=True)
                    dx = acc[0]
                elif have_c is not None and getattr(acc[0], "_lazy_mask", None) is not None and R == 1 and S == 1 and stride == 1 and pad == 0 \
                        and have_c == (N, H, W) and getattr(x_handle, "_bn_ctx_bits2", None) is None:
                    # the skip merge of a stage's last unit: conv_transpose(dy) + expand(compact gradient) * mask into a FRESH dense tensor
                    ctx_bits = getattr(x_handle, "_bn_ctx_bits", None)
                    fresh = K.empty(N, H, W, Cin, like=dyb)
                    r = K.p16_conv2d_bwd_data_masked_accum(dyb4, wt, g, fresh, acc[0]._lazy_mask, bn_ctx=ctx_bits, old_even=acc[0])
                    dx = _own(fresh)
                    if ctx_bits is not None:
                        dx._bnb_parts = r[1]
                    dx._replaces_acc = True                 # (engine.Tape: this tensor IS the accumulated gradient now)
                elif have_c is not None:
                    base = materialise(acc[0])              # any other pairing: the dense form (a new tensor), then the plain accumulate
                    K.p16_conv2d_bwd_data(dyb4, wt, g, dx=base.view(N, H, W, Cin), accumulate=True)
                    if hasattr(base, "_even_pixels") and not (s2 and base._even_pixels == (N, H, W)):
                        del base._even_pixels
                    dx = base
                    dx._replaces_acc = True
                elif acc is not None and acc[0] is not None:
                    skip_bits = getattr(acc[0], "_lazy_mask", None)
                    if skip_bits is not None and R == 1 and S == 1 and stride == 1 and pad == 0:
                        ctx_bits = getattr(x_handle, "_bn_ctx_bits", None)
                        if ctx_bits is not None:        # this merge completes the previous unit's output gradient: do its BN-backward sums too
                            ctx2 = getattr(x_handle, "_bn_ctx_bits2", None)
                            r = K.p16_conv2d_bwd_data_masked_accum(dyb4, wt, g, acc[0].view(N, H, W, Cin), skip_bits, bn_ctx=ctx_bits, bn_ctx2=ctx2)
                            acc[0]._bnb_parts = r[1]
                            if ctx2 is not None:
                                acc[0]._bnb_parts_shortcut = r[2]      # ... and those of its projection shortcut's BatchNorm
                        else:
                            K.p16_conv2d_bwd_data_masked_accum(dyb4, wt, g, acc[0].view(N, H, W, Cin), skip_bits)
                        del acc[0]._lazy_mask
                    else:
                        K.p16_conv2d_bwd_data(dyb4, wt, g, dx=materialise(acc[0]).view(N, H, W, Cin), accumulate=True)
                    if hasattr(acc[0], "_even_pixels") and not (s2 and acc[0]._even_pixels == (N, H, W)):
                        del acc[0]._even_pixels         # this contribution is dense
                    dx = acc[0]
                elif ctx is not None and stride == 1:
                    dx, parts = K.p16_conv2d_bwd_data_bnstats(dyb4, wt, g, *ctx)
                    dx = _own(dx)
                    dx._bnb_parts = parts
                elif s2c:
                    dx = _own(K.p16_conv2d_bwd_data(dyb4, wt, gc))
                    dx._compact_even = (N, H, W)
                else:
                    dx = _own(K.p16_conv2d_bwd_data(dyb4, wt, g))
                    if s2:
                        # a stride-2 1x1 convolution's input gradient: zero-filled, then written at the pixels (2i, 2j) only.  The tag lets
                        # the producer's BatchNorm backward reduce over those pixels alone (it survives further stride-2 1x1 contributions)
                        dx._even_pixels = (N, H, W)
            return dx, ((dres if lazy_skip else _own(dres.view(residual.shape))) if want_res else None)
        x32 = as_fp32(x)
        out32 = out2d if (out2d is not None or not (relu and want_res)) else _2d(as_fp32(out))
        dy, dres = _bn_backward(g2d, out32, y2d, mean, rstd, bn, relu, not use_batch, want_res)
        dy4 = dy.view(N, g.OH, g.OW, Kout)
        if w.needs_grad or b.needs_grad:
            def param_grads(x32=x32, dy=dy, dy4=dy4):
                if w.needs_grad:
                    s = GradSink(w)
                    K.conv2d_bwd_weight(x32, dy4, g, dw=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
                if b.needs_grad:
                    s = GradSink(b)
                    if use_batch:
                        # A bias in front of a batch-statistics BN has an exactly zero gradient: sum_rows(dy) =
                        # -rstd*gamma*mean(g*xhat)*sum(xhat) and sum(xhat) == 0.  (The fp64 oracle gives ~1e-15.)
                        if s.mode != "direct":
                            K.zero_(s.buf)
                    else:
                        K.colsum(dy, out=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
            side_task(param_grads, x32, dy)
        dx = None
        if x_needs_grad:
            if acc is not None and acc[0] is not None:
                # residual merge fused into the GEMM epilogue: dx += conv_transpose(dy) (no separate add pass)
                base = materialise(acc[0])                  # (a compact even-pixel gradient comes back as a NEW dense tensor)
                K.conv2d_bwd_data(dy4, w.value, g, dx=base.view(N, H, W, Cin), accumulate=True)
                if hasattr(base, "_even_pixels"):
                    del base._even_pixels
                dx = base
                if base is not acc[0]:
                    dx._replaces_acc = True                 # (engine.Tape re-binds the accumulated gradient to it)
            else:
                dx = _own(K.conv2d_bwd_data(dy4, w.value, g))
        dr = _own(dres.view(residual.shape)) if want_res else None
        return dx, dr

    backward.wants_acc = True
    backward.accepts_lazy = True
    _rec([out], [x_handle, res_handle], backward)
    return out


def conv_act(x: torch.Tensor, w: Variable, b: Variable, stride: int, pad: int, act: int) -> torch.Tensor:
    """Conv2D + bias + activation (BackboneNeck.conv2d_downscaler: 1x1, tanh)."""
    N, H, W, Cin = x.shape
    Kout, R, S, _ = w.value.shape
    g = K.ConvGeom(N, H, W, Cin, Kout, R, S, stride, pad)
    y, _ = K.conv2d_fwd(x, w.value, b.value, g, act)

    def backward(g_out):
        g_out = g_out.contiguous()
        if act == K.ACT_TANH:
            dpre = K.tanh_bwd(y, g_out)
        elif act == K.ACT_RELU:
            dpre = K.relu_bwd(y, g_out)
        else:
            dpre = g_out
        if w.needs_grad or b.needs_grad:
            def param_grads(dpre=dpre):
                if w.needs_grad:
                    s = GradSink(w)
                    K.conv2d_bwd_weight(x, dpre, g, dw=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
                if b.needs_grad:
                    s = GradSink(b)
                    K.colsum(_2d(dpre), out=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
            side_task(param_grads, x, dpre)
        return (_own(K.conv2d_bwd_data(dpre, w.value, g)),)

    _rec([y], [x], backward)
    return y


def batchnorm(x: torch.Tensor, bn: BNState, training: bool, bessel: bool) -> torch.Tensor:
    """Stand-alone BatchNormalization over the last axis (neck: 4-D fused path -> bessel moving
    variance; heads: 3-D path -> biased)."""
    x2d = _2d(x)
    rows, Cc = x2d.shape
    out2d, mean, rstd = _bn_forward(x2d, rows, Cc, None, bn, training, bessel, None, False)
    out = out2d.view(x.shape)

    def backward(g_out):
        dx, _ = _bn_backward(_2d(g_out.contiguous()), None, x2d, mean, rstd, bn, False, not training, False)
        return (_own(dx.view(x.shape)),)

    _rec([out], [x], backward)
    return out


def maxpool(x: torch.Tensor) -> torch.Tensor:
    y = K.maxpool_fwd(x)
    _rec([y], [x], lambda g: (_own(K.maxpool_bwd(x, y, g.contiguous())),))
    return y


STEM_FUSE = os.environ.get("BDETR_STEM_FUSE", "1") != "0"
STEM_S2D = os.environ.get("BDETR_STEM_S2D", "1") != "0"            # the stem's weight gradient on the pre-split path (conv_bn_relu_maxpool backward)


def conv_bn_relu_maxpool(x: torch.Tensor, w: Variable, b: Variable, bn: BNState, stride: int, pad: int, training: bool,
                         bn_batch_stats: bool) -> torch.Tensor:
    """The ResNet stem: Conv2D(+bias) -> BatchNormalization -> ReLU -> ZeroPadding2D(1) -> MaxPool 3x3/2; the input image needs no
    gradient.  Training with batch statistics under the 'split' policy, the tail runs fused (csrc/norm.hip stem_*): the pooled tensor is
    written once, as the f16 pair its consumers read (the returned handle IS that copy, like a link inside a bottleneck), and the
    normalised full-resolution tensor and its gradient are never materialised.  Otherwise: conv_bn + maxpool."""
    N, H, W, Cin = x.shape
    Kout, R, S, _ = w.value.shape
    if not (STEM_FUSE and training and bn_batch_stats and _p16_active() and Kout % 8 == 0):
        return maxpool(conv_bn(x, w, b, bn, stride, pad, True, training=training, bn_batch_stats=bn_batch_stats, x_needs_grad=False))
    g = K.ConvGeom(N, H, W, Cin, Kout, R, S, stride, pad)
    x32 = as_fp32(x)
    y, parts = K.conv2d_fwd(x32, w.value, b.value, g, K.ACT_NONE, want_stats=True)
    if parts is None or parts[0] is None:
        parts = K.colstats(_2d(y))
    mean, rstd = K.bn_stats(g.M, Kout, parts, bn.eps, bn.momentum, True, bn.moving_mean.value, bn.moving_var.value, like=y)
    _, out, tap = K.stem_pool_fwd(y, mean, rstd, bn.gamma.value, bn.beta.value)
    # `out` IS the f16 pair copy.  The alias must not lead back to `out`: `out` itself is a cycle through its own __dict__ (freed only by
    # the cyclic GC: 105 MB per step at batch 16), and a VIEW is worse - its C-level `_base` edge is invisible to the GC, so the cycle
    # out -> view -> out is never collected (round 5's first fix leaked 100 MB per eager step: profiles/r05_soak_2000steps_eager_leak.txt).
    # detach() shares the storage and holds no reference to the tensor object.
    out._p16f, out._p16b, out._p16_only = out.detach(), None, True

    def backward(g_out):
        sg, sb = GradSink(bn.gamma), GradSink(bn.beta)
        # The weight gradient of the 7x7 / stride-2 / pad-3 stem over a 4-channel image with even sides runs on the pre-split XX kernel
        # through a space-to-depth view (kernels.stem_bwd_weight_s2d, round 5): it is the LAST kernel of the backward pass, alone on the chip
        # with the optimizer waiting for it.  dy is then written as its bf16 pair.  Gradient products under 'split' are bf16 pairs on
        # both paths; any other backward policy, deterministic mode and other geometries keep igemm.hip's kernel.
        s2d = (STEM_S2D and w.needs_grad and (R, S, stride, pad, Cin) == (7, 7, 2, 3, 4) and H % 2 == 0 and W % 2 == 0 and Kout % 64 == 0
               and K.get_gemm_precision() in ("split", "mixed") and not K.deterministic())
        dy, _, _ = K.stem_pool_bwd(materialise(g_out).contiguous(), tap, y, mean, rstd, bn.gamma.value, bn.beta.value, dgamma=sg.buf, dbeta=sb.buf,
                                   dy_p16=s2d)
        sg.commit()
        sb.commit()
        if w.needs_grad or b.needs_grad:
            def param_grads(dy=dy):
                if w.needs_grad:
                    s = GradSink(w)
                    if s2d:
                        K.stem_bwd_weight_s2d(x32, dy, s.buf)
                    else:
                        K.conv2d_bwd_weight(x32, dy, g, dw=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
                if b.needs_grad:
                    s = GradSink(b)             # exactly zero in front of a batch-statistics BatchNorm (see conv_bn)
                    if s.mode != "direct":
                        K.zero_(s.buf)
                    s.commit()
            side_task(param_grads, x32, dy)
        return (None,)

    _rec([out], [x], backward)
    return out


# ----------------------------------------------------------------------------------------
# dense / elementwise
# ----------------------------------------------------------------------------------------
def dense(x: torch.Tensor, w: Variable, b: Variable, act: int = K.ACT_NONE) -> torch.Tensor:
    """tf.keras.layers.Dense on the last axis; w is stored [out][in]."""
    x2d = _2d(x)
    y2d = K.linear_fwd(x2d, w.value, b.value, act)
    y = y2d.view(*x.shape[:-1], w.value.shape[0])

    def backward(g_out, acc=None):
        g2d = _2d(g_out.contiguous())
        if act == K.ACT_RELU:
            g2d = K.relu_bwd(y2d, g2d)
        elif act == K.ACT_TANH:
            g2d = K.tanh_bwd(y2d, g2d)
        if w.needs_grad or b.needs_grad:
            def param_grads(g2d=g2d):
                if w.needs_grad:
                    s = GradSink(w)
                    K.linear_bwd_weight(g2d, x2d, dw=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
                if b.needs_grad:
                    s = GradSink(b)
                    K.colsum(g2d, out=s.buf, prezeroed=s.mode == "direct")
                    s.commit()
            side_task(param_grads, x2d, g2d)
        have = acc[0] if acc is not None else None
        if have is not None and have.is_contiguous() and have.shape == x.shape:
            # another consumer's gradient of x is already there: add into it in the GEMM epilogue (no separate axpy pass)
            K.linear_bwd_data(g2d, w.value, dx=_2d(materialise(have)), accumulate=True)
            if hasattr(have, "_even_pixels"):
                del have._even_pixels
            return (have,)
        return (_own(K.linear_bwd_data(g2d, w.value).view(x.shape)),)

    backward.wants_acc = True
    _rec([y], [x], backward)
    return y


def dense_group(xs, ws, bs):
    """Independent Dense layers of identical width (the Q/K/V projections of an attention block,
    transformers.py:68-70) as ONE grouped GEMM launch forward and one for the input gradients."""
    x2 = [_2d(x) for x in xs]
    aligned = all(t.data_ptr() % 16 == 0 for t in x2) and x2[0].shape[1] % 4 == 0 and ws[0].value.shape[0] % 4 == 0
    same = all(w.value.shape == ws[0].value.shape for w in ws)
    if not (aligned and same) or len(xs) > 4:
        return [dense(x, w, b) for x, w, b in zip(xs, ws, bs)]
    y2 = K.linear_fwd_group(x2, [w.value for w in ws], [b.value for b in bs])
    ys = [y.view(*x.shape[:-1], y.shape[1]) for y, x in zip(y2, xs)]

    def backward(*gs):
        g2 = [_2d(g.contiguous()) for g in gs]

        def param_grads():
            # weight gradients of the projections that share a shape go out as one grouped split-K launch (self-attention: all three;
            # cross-attention: key + value, the query projection has its own row count)
            groups = {}
            for g, x, w in zip(g2, x2, ws):
                if w.needs_grad:
                    groups.setdefault((tuple(g.shape), tuple(x.shape)), []).append((g, x, GradSink(w)))
            for members in groups.values():
                K.linear_bwd_weight_group([m[0] for m in members], [m[1] for m in members], [m[2].buf for m in members], [m[2].mode == "direct" for m in members])
                for m in members:
                    m[2].commit()
            # bias gradients: the members whose sink is a zero-filled slice of the flat gradient buffer share ONE launch
            sinks = [(g, GradSink(b)) for g, b in zip(g2, bs) if b.needs_grad]
            direct = [(g, s) for g, s in sinks if s.mode == "direct"]
            if len(direct) > 1 and not K.deterministic() and all(g.shape[1] == direct[0][0].shape[1] for g, _ in direct):
                K.colsum_group([g for g, _ in direct], [s.buf for _, s in direct])
                for _, s in direct:
                    s.commit()
                sinks = [(g, s) for g, s in sinks if s.mode != "direct"]
            for g, s in sinks:
                K.colsum(g, out=s.buf, prezeroed=s.mode == "direct")
                s.commit()
        if any(w.needs_grad or b.needs_grad for w, b in zip(ws, bs)):
            side_task(param_grads, *x2, *g2)
        dxs = K.linear_bwd_data_group(g2, [w.value for w in ws])
        return tuple(_own(dx.view(x.shape)) for dx, x in zip(dxs, xs))

    _rec(ys, list(xs), backward)
    return ys


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = K.add(a, b)
    _rec([out], [a, b], lambda g: (g, g))        # shared tensor: NOT owned
    return out


def _param_or_tensor_grad(p, fn):
    """Gradient of a broadcast operand that may be a Variable's value (positional table, queries)."""
    if isinstance(p, Variable):
        if p.needs_grad:
            s = GradSink(p)
            fn(s.buf)
            s.commit()
        return None
    return fn(None)


def add_bcast(x: torch.Tensor, row) -> torch.Tensor:
    """x [B, ...] + row [...]  (positional encoding tiled over the batch, transformers.py:299-300).
    ``row`` is a tensor or a Variable."""
    rv = row.value if isinstance(row, Variable) else row
    out = K.add_bcast_rows(x, rv)
    n = rv.numel()

    def backward(g):
        g = g.contiguous()
        grow = _param_or_tensor_grad(row, lambda out_: K.sum_over_batch(g, n, out=out_))
        return g, (grow.view(rv.shape) if grow is not None else None)

    _rec([out], [x, None if isinstance(row, Variable) else row], backward)
    return out


def tile_batch(row: Variable, B: int) -> torch.Tensor:
    """row [...] -> [B, ...]  (DecoderPrep's tiled queries, transformers.py:445-447)."""
    rv = row.value
    zeros = torch.empty((B,) + tuple(rv.shape), dtype=rv.dtype, device=rv.device)
    K.zero_(zeros)
    out = K.add_bcast_rows(zeros, rv)
    n = rv.numel()
    from . import engine as _engine
    if _engine._DEBUG_LOG[0] is not None:
        _engine._DEBUG_LOG[0].emit("L", out)        # diagnostic: the zero-filled + broadcast tensor (see bdetr_zero_bytes)

    def backward(g):
        g = g.contiguous()
        _param_or_tensor_grad(row, lambda out_: K.sum_over_batch(g, n, out=out_))
        return ()

    _rec([out], [], backward)
    return out


def reshape(x: torch.Tensor, shape) -> torch.Tensor:
    y = x.view(shape)

    def backward(g):
        r = g.contiguous().view(x.shape)
        if getattr(g, "_bdetr_owned", False):
            r._bdetr_owned = True
        return (r,)

    _rec([y], [x], backward)
    return y


# ----------------------------------------------------------------------------------------
# transformer pieces
# ----------------------------------------------------------------------------------------
def attention_core(Q: torch.Tensor, Kt: torch.Tensor, V: torch.Tensor, heads: int) -> torch.Tensor:
    """softmax(Q K^T / sqrt(d)) V per head (transformers.py:86-97).  Q [B,q,h*d], K/V [B,k,h*d].
    Returns [B,h,q,d] contiguous - the layout the reference reshapes WITHOUT permuting (line 100)."""
    B, q, D = Q.shape
    kk = Kt.shape[1]
    d = D // heads
    scale = 1.0 / math.sqrt(float(d))
    if d == 32 and FUSED_ATTENTION[0]:
        O, lse = K.attention_fwd(Q, Kt, V, heads, scale)

        def backward_fused(gO):
            dQ, dK, dV = K.attention_bwd(Q, Kt, V, O, gO.contiguous(), lse, heads, scale)
            return _own(dQ), _own(dK), _own(dV)

        _rec([O], [Q, Kt, V], backward_fused)
        return O
    S = K.empty(B, heads, q, kk, like=Q)
    K.gemm_raw(q, kk, d, Q, D, True, Kt, D, True, S, kk, nb0=B, nb1=heads, sa=(q * D, d), sb=(kk * D, d), sc=(heads * q * kk, q * kk))
    P = K.softmax_rows_fwd(S.view(-1, kk), scale, out=S.view(-1, kk)).view(B, heads, q, kk)   # in place; Rescaling is fused
    O = K.empty(B, heads, q, d, like=Q)
    K.gemm_raw(q, d, kk, P, kk, True, V, D, False, O, d, nb0=B, nb1=heads, sa=(heads * q * kk, q * kk), sb=(kk * D, d), sc=(heads * q * d, q * d))

    def backward(gO):
        gO = gO.contiguous()
        sP, sO = (heads * q * kk, q * kk), (heads * q * d, q * d)
        # dP[q][k] = sum_d dO[q][d] V[k][d]
        dP = K.empty(B, heads, q, kk, like=Q)
        K.gemm_raw(q, kk, d, gO, d, True, V, D, True, dP, kk, nb0=B, nb1=heads, sa=sO, sb=(kk * D, d), sc=sP, grad=True)
        # dV[k][d] = sum_q P[q][k] dO[q][d]
        dV = K.empty(B, kk, D, like=Q)
        K.gemm_raw(kk, d, q, P, kk, False, gO, d, False, dV, D, nb0=B, nb1=heads, sa=sP, sb=sO, sc=(kk * D, d), grad=True)
        dS = K.softmax_rows_bwd(P.view(-1, kk), dP.view(-1, kk), scale, out=dP.view(-1, kk)).view(B, heads, q, kk)
        # dQ[q][d] = sum_k dS[q][k] K[k][d] ;  dK[k][d] = sum_q dS[q][k] Q[q][d]
        dQ = K.empty(B, q, D, like=Q)
        K.gemm_raw(q, d, kk, dS, kk, True, Kt, D, False, dQ, D, nb0=B, nb1=heads, sa=sP, sb=(kk * D, d), sc=(q * D, d), grad=True)
        dK = K.empty(B, kk, D, like=Q)
        K.gemm_raw(kk, d, q, dS, kk, False, Q, D, False, dK, D, nb0=B, nb1=heads, sa=sP, sb=(q * D, d), sc=(kk * D, d), grad=True)
        return _own(dQ), _own(dK), _own(dV)

    _rec([O], [Q, Kt, V], backward)
    return O


FUSED_ATTENTION = [True]     # head dim 32 -> csrc/attention.hip; other widths use batched GEMMs + softmax


def add_dropout_layernorm(x: torch.Tensor, y: torch.Tensor, gamma: Variable, beta: Variable, eps: float, rate: float,
                          training: bool) -> torch.Tensor:
    """LayerNormalization(x + Dropout(y))  (transformers.py:135-137, 178-180)."""
    r = rate if training else 0.0
    seed = _next_dropout_seed() if r > 0.0 else 0
    x2d, y2d = _2d(x), _2d(y)
    base = dropout_seed_tensor() if r > 0.0 else None
    out2d, mean, rstd = K.add_dropout_layernorm_fwd(x2d, y2d, gamma.value, beta.value, eps, r, seed, seed_base=base)
    out = out2d.view(x.shape)

    def backward(g_out):
        sg, sb = GradSink(gamma), GradSink(beta)
        dx, dy, _, _ = K.add_dropout_layernorm_bwd(_2d(g_out.contiguous()), x2d, y2d, gamma.value, mean, rstd, r, seed,
                                                   dgamma=sg.buf, dbeta=sb.buf, seed_base=base)
        sg.commit()
        sb.commit()
        return _own(dx.view(x.shape)), _own(dy.view(y.shape))

    _rec([out], [x, y], backward)
    return out


# ----------------------------------------------------------------------------------------
# row chain: OutputProjection + Add/Dropout/LayerNorm (+ FeedForwardBlock) of a transformer layer as one launch per direction
# ----------------------------------------------------------------------------------------
ROWCHAIN = [os.environ.get("BDETR_ROWCHAIN", "1") != "0"]


def rowchain_active(width: int) -> bool:
    """The fused path exists for model width 256 under the 'split' policy (csrc/rowchain.hip: f16-pair forward, bf16-pair gradient
    products); every other case runs the separate GEMM / LayerNorm launches."""
    return ROWCHAIN[0] and width == K.ROWCHAIN_WIDTH and K.get_gemm_precision() == "split"


class _RowchainPacks:
    """Forward (f16 pairs of 2^8 W, MFMA fragment order) and backward (bf16 pairs of W^T) copies of every Dense kernel on a row
    chain, in persistent buffers, refreshed by ONE multi-matrix launch the first time a copy is asked for after the weights changed
    (the same scheme as _PackedWeights for the conv kernels)."""

    def __init__(self):
        self.rows, self.table, self.version = [], None, -1

    def get(self, w: Variable):
        import weakref
        c = getattr(w, "_rc", None)
        if c is None or c[0].data_ptr() != w.value.data_ptr():
            from . import _lib
            n = int(_lib.lib().bdetr_rowchain_pack_elems())
            dev = w.value.device
            c = (w.value, torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev))
            w._rc = c
            self.rows = [r for r in self.rows if r[0]() is not None and r[0]() is not w]
            self.rows.append((weakref.ref(w),) + c)
            self.table, self.version = None, -1
        if self.version != WEIGHTS_VERSION[0]:
            if self.table is None:
                import numpy as np
                rows = np.array([[v.data_ptr(), f.data_ptr(), b.data_ptr()] for _, v, f, b in self.rows], np.int64)
                self.table = torch.from_numpy(rows).to(w.value.device)
            K.rowchain_pack_weights(self.table)
            self.version = WEIGHTS_VERSION[0]
        return c[1], c[2]


_RC_PACKED = _RowchainPacks()


def attention_out_chain(ctx: torch.Tensor, resid: torch.Tensor, out_proj, ln1, ffn, eps: float, rate: float, training: bool) -> torch.Tensor:
    """LN1(resid + Dropout(ctx Wo^T + bo)) and, with ``ffn`` = (W1, b1, W2, b2, gamma2, beta2), the FeedForwardBlock behind it
    (transformers.py:101,135-137,174-180) as ONE forward launch; the backward closure runs one launch for every data gradient, LayerNorm
    gradient and bias gradient of the chain, one for their partial sums, and the three weight gradients as side tasks.
    out_proj = (Wo, bo), ln1 = (gamma1, beta1): Variables.  ctx, resid: [B, T, 256]."""
    r = rate if training else 0.0
    seed1 = _next_dropout_seed() if r > 0.0 else 0                  # the same site order as the unfused path: identical masks
    seed2 = _next_dropout_seed() if (r > 0.0 and ffn is not None) else 0
    base = dropout_seed_tensor() if r > 0.0 else None
    ctx2d, res2d = _2d(ctx.contiguous()), _2d(resid.contiguous())
    Wo, bo = out_proj
    g1, b1 = ln1
    ws = [Wo] + ([ffn[0], ffn[2]] if ffn is not None else [])
    bs = [bo] + ([ffn[1], ffn[3]] if ffn is not None else [])
    packs = [_RC_PACKED.get(w) for w in ws]
    saved = K.rowchain_fwd(ctx2d, res2d, [p[0] for p in packs], [b.value for b in bs], (g1.value, b1.value),
                           (ffn[4].value, ffn[5].value) if ffn is not None else None, eps, r, seed1, seed2, base)
    out = (saved["x2"] if ffn is not None else saved["x1"]).view(resid.shape)

    def backward_unfused(g_out):
        """The same backward on the separate kernels, from the tensors the fused forward saved - taken when the backward runs under
        another arithmetic policy than the forward did (Model.replay_backward('fp32'): exact-fp32 gradient products)."""
        g2 = _2d(g_out.contiguous())
        zeros = torch.zeros_like(ctx2d)                   # add_dropout_layernorm_bwd recomputes x + keep * y: x = the saved sum, y = 0
        todo = []
        if ffn is not None:
            W1, bb1, W2, bb2, g2v, b2v = ffn
            sg, sb = GradSink(g2v), GradSink(b2v)
            dh2, df, _, _ = K.add_dropout_layernorm_bwd(g2, saved["pre2"], zeros, g2v.value, saved["mean2"], saved["rstd2"], r, seed2,
                                                        dgamma=sg.buf, dbeta=sb.buf, seed_base=base)
            sg.commit(); sb.commit()
            dpre1 = K.relu_bwd(saved["h"], K.linear_bwd_data(df, W2.value))
            dx1 = K.linear_bwd_data(dpre1, W1.value, dx=dh2, accumulate=True)
            todo += [(W2, bb2, df, saved["h"]), (W1, bb1, dpre1, saved["x1"])]
        else:
            dx1 = g2
        sg, sb = GradSink(g1), GradSink(b1)
        dres, da, _, _ = K.add_dropout_layernorm_bwd(dx1, saved["pre1"], zeros, g1.value, saved["mean1"], saved["rstd1"], r, seed1,
                                                    dgamma=sg.buf, dbeta=sb.buf, seed_base=base)
        sg.commit(); sb.commit()
        dctx = K.linear_bwd_data(da, Wo.value)
        todo.append((Wo, bo, da, ctx2d))
        for w, b, g, x in todo:
            if w.needs_grad:
                s_ = GradSink(w)
                K.linear_bwd_weight(g, x, dw=s_.buf, prezeroed=s_.mode == "direct")
                s_.commit()
            if b.needs_grad:
                s_ = GradSink(b)
                K.colsum(g, out=s_.buf, prezeroed=s_.mode == "direct")
                s_.commit()
        return _own(dctx.view(ctx.shape)), _own(dres.view(resid.shape))

    def backward(g_out):
        if K.get_gemm_precision() != "split":
            return backward_unfused(g_out)
        gammas = (g1.value,) + ((ffn[4].value,) if ffn is not None else ())
        packs_t = [_RC_PACKED.get(w)[1] for w in ws]
        dctx, dres, G, partials, nparts = K.rowchain_bwd(_2d(g_out.contiguous()), saved, packs_t, gammas, r, seed1, seed2, base)
        xs = [ctx2d] + ([saved["x1"], saved["h"]] if ffn is not None else [])
        vec_vars = ([ffn[4], ffn[5], ffn[3], ffn[1]] if ffn is not None else [None] * 4) + [g1, b1, bo]

        def param_grads():
            sinks = [GradSink(v) if v is not None else None for v in vec_vars]
            K.rowchain_reduce(partials, nparts, [s_.buf if s_ is not None else None for s_ in sinks], [0] * 7)
            for s_ in sinks:
                if s_ is not None:
                    s_.commit()
            todo = [(w, g, x, GradSink(w)) for w, g, x in zip(ws, G, xs) if w.needs_grad]
            if todo:                                           # the chain's weight gradients: one grouped split-K launch
                K.linear_bwd_weight_group([t[1] for t in todo], [t[2] for t in todo], [t[3].buf for t in todo], [t[3].mode == "direct" for t in todo])
                for t in todo:
                    t[3].commit()
        side_task(param_grads, partials, *G, *xs)
        return _own(dctx.view(ctx.shape)), _own(dres.view(resid.shape))

    _rec([out], [ctx, resid], backward)
    return out


# ----------------------------------------------------------------------------------------
# head activations
# ----------------------------------------------------------------------------------------
def softmax_lastdim(x: torch.Tensor) -> torch.Tensor:
    p = K.softmax_rows_fwd(_2d(x), 1.0).view(x.shape)
    _rec([p], [x], lambda g: (_own(K.softmax_rows_bwd(_2d(p), _2d(g.contiguous()), 1.0).view(x.shape)),))
    return p


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    y = K.sigmoid_fwd(x)
    _rec([y], [x], lambda g: (_own(K.sigmoid_bwd(y, g.contiguous())),))
    return y


def box_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """3*sigmoid(x/100) - 1  (prediction_heads.py:44)."""
    y = K.boxsigmoid_fwd(x)
    _rec([y], [x], lambda g: (_own(K.boxsigmoid_bwd(y, g.contiguous())),))
    return y
