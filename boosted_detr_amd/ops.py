"""Tape-recorded composite ops of the hot path.  Each function runs HIP kernels for the
forward and, when a Tape is recording, registers the closure that runs the backward kernels.

Tensors are NHWC / [B,T,D]; ops view them as 2-D row matrices internally.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import kernels as K
from .engine import Variable, current_tape

_dropout_site = [0]
_dropout_base_seed = [0x5EED]


def set_dropout_seed(seed: int) -> None:
    _dropout_base_seed[0] = int(seed) & 0xFFFFFFFFFFFF
    _dropout_site[0] = 0


def _next_dropout_seed() -> int:
    _dropout_site[0] += 1
    return (_dropout_base_seed[0] * 0x100000001B3 + _dropout_site[0] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


def _rec(outputs, inputs, fn):
    t = current_tape()
    if t is not None:
        t.record(outputs, inputs, fn)


def _2d(t: torch.Tensor) -> torch.Tensor:
    return t.view(-1, t.shape[-1])


# ----------------------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------------------
def image_prep(image: torch.Tensor, H: int, W: int) -> torch.Tensor:
    return K.image_prep(image, H, W)     # input images need no gradient


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
        mean, rstd = K.bn_stats_frozen(bn.moving_mean.value, bn.moving_var.value, bn.eps)
    out = K.bn_apply(y2d, mean, rstd, bn.gamma.value, bn.beta.value, residual2d, relu)
    return out, mean, rstd


def conv_bn(x: torch.Tensor, w: Variable, b: Variable, bn: BNState, stride: int, pad: int, relu: bool,
            residual: Optional[torch.Tensor] = None, training: bool = False, bn_batch_stats: Optional[bool] = None,
            x_needs_grad: bool = True) -> torch.Tensor:
    """Conv2D(+bias) -> BatchNormalization -> [+ residual] -> [ReLU]  (keras ResNet-50 block unit)."""
    N, H, W, Cin = x.shape
    Kout, R, S, _ = w.value.shape
    g = K.ConvGeom(N, H, W, Cin, Kout, R, S, stride, pad)
    use_batch = training if bn_batch_stats is None else bn_batch_stats
    y, parts = K.conv2d_fwd(x, w.value, b.value, g, K.ACT_NONE, want_stats=use_batch)
    y2d = _2d(y)
    res2d = _2d(residual) if residual is not None else None
    out2d, mean, rstd = _bn_forward(y2d, g.M, Kout, parts, bn, use_batch, True, res2d, relu)
    out = out2d.view(N, g.OH, g.OW, Kout)

    def backward(g_out):
        dy, dgamma, dbeta, dres = K.bn_bwd(_2d(g_out.contiguous()), out2d, y2d, mean, rstd, bn.gamma.value, relu, not use_batch,
                                           want_residual_grad=residual is not None)
        dy4 = dy.view(N, g.OH, g.OW, Kout)
        dw = K.conv2d_bwd_weight(x, dy4, g)
        db = K.colsum(dy)
        dx = K.conv2d_bwd_data(dy4, w.value, g) if x_needs_grad else None
        dr = dres.view(residual.shape) if residual is not None else None
        return dx, dr, dw, db, dgamma, dbeta

    _rec([out], [x, residual, w.value, b.value, bn.gamma.value, bn.beta.value], backward)
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
        dw = K.conv2d_bwd_weight(x, dpre, g)
        db = K.colsum(_2d(dpre))
        dx = K.conv2d_bwd_data(dpre, w.value, g)
        return dx, dw, db

    _rec([y], [x, w.value, b.value], backward)
    return y


def batchnorm(x: torch.Tensor, bn: BNState, training: bool, bessel: bool) -> torch.Tensor:
    """Stand-alone BatchNormalization over the last axis (neck: 4-D fused path -> bessel moving
    variance; heads: 3-D path -> biased)."""
    x2d = _2d(x)
    rows, Cc = x2d.shape
    out2d, mean, rstd = _bn_forward(x2d, rows, Cc, None, bn, training, bessel, None, False)
    out = out2d.view(x.shape)

    def backward(g_out):
        dx, dgamma, dbeta, _ = K.bn_bwd(_2d(g_out.contiguous()), None, x2d, mean, rstd, bn.gamma.value, False, not training)
        return dx.view(x.shape), dgamma, dbeta

    _rec([out], [x, bn.gamma.value, bn.beta.value], backward)
    return out


def maxpool(x: torch.Tensor) -> torch.Tensor:
    y = K.maxpool_fwd(x)
    _rec([y], [x], lambda g: (K.maxpool_bwd(x, y, g.contiguous()),))
    return y


# ----------------------------------------------------------------------------------------
# dense / elementwise
# ----------------------------------------------------------------------------------------
def dense(x: torch.Tensor, w: Variable, b: Variable, act: int = K.ACT_NONE) -> torch.Tensor:
    """tf.keras.layers.Dense on the last axis; w is stored [out][in]."""
    x2d = _2d(x)
    y2d = K.linear_fwd(x2d, w.value, b.value, act)
    y = y2d.view(*x.shape[:-1], w.value.shape[0])

    def backward(g_out):
        g2d = _2d(g_out.contiguous())
        if act == K.ACT_RELU:
            g2d = K.relu_bwd(y2d, g2d)
        elif act == K.ACT_TANH:
            g2d = K.tanh_bwd(y2d, g2d)
        dx = K.linear_bwd_data(g2d, w.value).view(x.shape)
        dw = K.linear_bwd_weight(g2d, x2d)
        db = K.colsum(g2d)
        return dx, dw, db

    _rec([y], [x, w.value, b.value], backward)
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = K.add(a, b)
    _rec([out], [a, b], lambda g: (g, g))
    return out


def add_bcast(x: torch.Tensor, row: torch.Tensor) -> torch.Tensor:
    """x [B, ...] + row [...]  (positional encoding tiled over the batch, transformers.py:299-300)."""
    out = K.add_bcast_rows(x, row)
    n = row.numel()
    _rec([out], [x, row], lambda g: (g, K.sum_over_batch(g.contiguous(), n).view(row.shape)))
    return out


def tile_batch(row: torch.Tensor, B: int) -> torch.Tensor:
    """row [...] -> [B, ...]  (DecoderPrep's tiled queries, transformers.py:445-447)."""
    zeros = torch.empty((B,) + tuple(row.shape), dtype=row.dtype, device=row.device)
    K.zero_(zeros)
    out = K.add_bcast_rows(zeros, row)
    n = row.numel()
    _rec([out], [row], lambda g: (K.sum_over_batch(g.contiguous(), n).view(row.shape),))
    return out


def reshape(x: torch.Tensor, shape) -> torch.Tensor:
    y = x.view(shape)
    _rec([y], [x], lambda g: (g.contiguous().view(x.shape),))
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
        K.gemm_raw(q, kk, d, gO, d, True, V, D, True, dP, kk, nb0=B, nb1=heads, sa=sO, sb=(kk * D, d), sc=sP)
        # dV[k][d] = sum_q P[q][k] dO[q][d]
        dV = K.empty(B, kk, D, like=Q)
        K.gemm_raw(kk, d, q, P, kk, False, gO, d, False, dV, D, nb0=B, nb1=heads, sa=sP, sb=sO, sc=(kk * D, d))
        dS = K.softmax_rows_bwd(P.view(-1, kk), dP.view(-1, kk), scale, out=dP.view(-1, kk)).view(B, heads, q, kk)
        # dQ[q][d] = sum_k dS[q][k] K[k][d] ;  dK[k][d] = sum_q dS[q][k] Q[q][d]
        dQ = K.empty(B, q, D, like=Q)
        K.gemm_raw(q, d, kk, dS, kk, True, Kt, D, False, dQ, D, nb0=B, nb1=heads, sa=sP, sb=(kk * D, d), sc=(q * D, d))
        dK = K.empty(B, kk, D, like=Q)
        K.gemm_raw(kk, d, q, dS, kk, False, Q, D, False, dK, D, nb0=B, nb1=heads, sa=sP, sb=(q * D, d), sc=(kk * D, d))
        return dQ, dK, dV

    _rec([O], [Q, Kt, V], backward)
    return O


def add_dropout_layernorm(x: torch.Tensor, y: torch.Tensor, gamma: Variable, beta: Variable, eps: float, rate: float,
                          training: bool) -> torch.Tensor:
    """LayerNormalization(x + Dropout(y))  (transformers.py:135-137, 178-180)."""
    r = rate if training else 0.0
    seed = _next_dropout_seed() if r > 0.0 else 0
    x2d, y2d = _2d(x), _2d(y)
    out2d, mean, rstd = K.add_dropout_layernorm_fwd(x2d, y2d, gamma.value, beta.value, eps, r, seed)
    out = out2d.view(x.shape)

    def backward(g_out):
        dx, dy, dg, db = K.add_dropout_layernorm_bwd(_2d(g_out.contiguous()), x2d, y2d, gamma.value, mean, rstd, r, seed)
        return dx.view(x.shape), dy.view(y.shape), dg, db

    _rec([out], [x, y, gamma.value, beta.value], backward)
    return out


# ----------------------------------------------------------------------------------------
# head activations
# ----------------------------------------------------------------------------------------
def softmax_lastdim(x: torch.Tensor) -> torch.Tensor:
    p = K.softmax_rows_fwd(_2d(x), 1.0).view(x.shape)
    _rec([p], [x], lambda g: (K.softmax_rows_bwd(_2d(p), _2d(g.contiguous()), 1.0).view(x.shape),))
    return p


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    y = K.sigmoid_fwd(x)
    _rec([y], [x], lambda g: (K.sigmoid_bwd(y, g.contiguous()),))
    return y


def box_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """3*sigmoid(x/100) - 1  (prediction_heads.py:44)."""
    y = K.boxsigmoid_fwd(x)
    _rec([y], [x], lambda g: (K.boxsigmoid_bwd(y, g.contiguous()),))
    return y
