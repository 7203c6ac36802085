"""GPU parity of the pre-split ("P16") operand path (csrc/sgemm.hip, csrc/p16.hip) through the C ABI.

* the packers against a bit-exact torch-CPU restatement of the layout (include/bdetr.h: groups of 8 elements =
  [8 x hi][8 x lo]; f16 pair hi = f16(x), lo = f16(x - hi) unscaled - the weights' forward copy holds 2^8 w -, bf16 pair);
* the three convolution products (forward on f16 pairs, backward-data / backward-weight on bf16 pairs) against fp64
  F.conv2d and its autograd: tolerance 2e-5 x max|ref| forward (fp32-grade), 6e-5 gradients (2^-18 per product) -
  the same bars as the in-kernel split arithmetic (tests/test_precision_gpu.py);
* edge cases: ragged tiles in every dimension, padding borders, strided 1x1, C = 64 3x3 (two taps per 128 gathered
  columns), images smaller than one K-step, split-K tails, the accumulate epilogue."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, dev, rnd

pytestmark = pytest.mark.gpu


def p16_ref(x: torch.Tensor, f16: bool) -> torch.Tensor:
    """Bit-exact CPU restatement of the P16 layout; returned as int32 words (4 bytes per element)."""
    x = x.float().contiguous()
    if f16:
        hi = x.half()
        lo = (x - hi.float()).half()                    # unscaled: a subnormal f16 for |x| < 2^-3 (csrc/p16.h)
    else:
        hi = x.bfloat16()
        lo = (x - hi.float()).bfloat16()
    g = x.shape[:-1] + (x.shape[-1] // 8, 8)
    words = torch.stack([hi.view(torch.int16).reshape(g), lo.view(torch.int16).reshape(g)], dim=-2).contiguous()   # [..., C/8, 2, 8]
    return words.view(torch.int32).reshape(x.shape)


def bits(t: torch.Tensor) -> torch.Tensor:
    return t.cpu().view(torch.int32)


def test_pack_unpack_bit_exact(cuda):
    from boosted_detr_amd import kernels as k
    x = rnd(37, 11, 64, seed=1) * 3
    x[0, 0, :8] = torch.tensor([0.0, -0.0, 1e-8, -3e-5, 65000.0, 1e-3, 255.5, -1.0])
    f, b = k.p16_pack(dev(x))
    assert torch.equal(bits(f), p16_ref(x, True))
    assert torch.equal(bits(b), p16_ref(x, False))
    # the f16 pair carries 22 significant bits down to |x| = 2^-3; below, its (unscaled) lo half is an f16 subnormal and the
    # absolute error is bounded by half a subnormal step, 2^-25 (csrc/p16.h); the bf16 pair carries 16 bits at any magnitude
    xf, xb = k.p16_unpack(f, True).cpu(), k.p16_unpack(b, False).cpu()
    def worst(got, rel, floor):
        viol = (got - x).abs() - (x.abs() * rel + floor)
        i = int(viol.argmax())
        return float(viol.max()), float(x.reshape(-1)[i]), float(got.reshape(-1)[i])
    assert worst(xf, 2.0 ** -21, 2.0 ** -25)[0] <= 0, worst(xf, 2.0 ** -21, 2.0 ** -25)
    assert worst(xb, 2.0 ** -15, 1e-30)[0] <= 0, worst(xb, 2.0 ** -15, 1e-30)
    assert int(k.overflow_flag().item()) == 0
    big = x.clone(); big[3, 3, 3] = 7e4
    k.p16_pack(dev(big))
    assert int(k.overflow_flag().item()) == 1          # beyond the f16 pair's range: flagged, never silent
    k.overflow_flag().zero_()


def test_weight_packs_bit_exact(cuda):
    from boosted_detr_amd import kernels as k
    for K_, R, C in ((64, 3, 64), (256, 1, 128), (32, 3, 32)):
        w = rnd(K_, R, R, C, seed=K_) * 0.1
        wf, wt = k.p16_pack_conv_weights(dev(w))
        assert torch.equal(bits(wf), p16_ref(w * 256.0, True))          # the forward copy holds 2^8 w (csrc/p16.h)
        want_t = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()         # [C][R][S][K], taps flipped
        assert torch.equal(bits(wt), p16_ref(want_t, False))


# N, H, W, C, K, R, stride, pad
P16_CONVS = [
    (2, 20, 20, 64, 64, 3, 1, 1),        # C = 64: a 128-column weight-gradient tile spans two taps
    (2, 20, 20, 64, 256, 1, 1, 0),
    (2, 20, 20, 256, 128, 1, 2, 0),      # strided 1x1: patch loader forward, row-scatter backward-data
    (1, 9, 11, 128, 32, 3, 1, 1),        # odd spatial sizes, K = 32 (ragged 64-wide tile)
    (3, 7, 7, 512, 2048, 1, 1, 0),
    (2, 14, 14, 256, 256, 3, 1, 1),
    (5, 3, 3, 64, 96, 3, 1, 1),          # 9-pixel images: several images inside one 32-pixel K-step of the weight gradient
    (1, 5, 6, 40, 72, 1, 1, 0),          # channel counts that are multiples of 8 only (1x1)
    (4, 40, 40, 256, 256, 3, 1, 1),
]
# the benchmark's batch-16 layer shapes: every tile / loader combination the training step launches
P16_BIG_CONVS = [
    (8, 80, 80, 128, 128, 3, 1, 1), (16, 40, 40, 256, 256, 3, 1, 1), (16, 20, 20, 512, 512, 3, 1, 1),
    (4, 160, 160, 64, 64, 3, 1, 1), (4, 80, 80, 256, 512, 1, 1, 0), (16, 80, 80, 512, 256, 1, 2, 0),
    (16, 20, 20, 2048, 512, 1, 1, 0), (16, 20, 20, 512, 2048, 1, 1, 0), (4, 160, 160, 256, 64, 1, 1, 0),
]


def p16_conv_case(N, H, W, C, K, R, stride, pad, accumulate=True):
    from boosted_detr_amd import kernels as k
    x, w, b = rnd(N, H, W, C, seed=1), rnd(K, R, R, C, seed=2, scale=(R * R * C) ** -0.5), rnd(K, seed=3)
    g = k.ConvGeom(N, H, W, C, K, R, R, stride, pad)
    assert k.p16_supported(g)
    xf, xb = k.p16_pack(dev(x))
    wf, wt = k.p16_pack_conv_weights(dev(w))
    y, (ps, pq, n) = k.p16_conv2d_fwd(xf, wf, dev(b), g, 0, want_stats=True)
    xt = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wtt = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xt, wtt, b.double(), stride=stride, padding=pad)
    r2 = ref.permute(0, 2, 3, 1).reshape(-1, K)
    close(y, ref.permute(0, 2, 3, 1), rtol=2e-5)
    close(ps.sum(0), r2.sum(0), rtol=1e-4)
    close(pq.sum(0), (r2 * r2).sum(0), rtol=1e-4)
    dy = rnd(*ref.shape, seed=4).double()
    ref.backward(dy)
    dyn = dy.permute(0, 2, 3, 1).float().contiguous()
    _, dyb = k.p16_pack(dev(dyn), want_f16=False)
    close(k.p16_conv2d_bwd_data(dyb, wt, g), xt.grad.permute(0, 2, 3, 1), rtol=6e-5)
    if accumulate:
        base = rnd(N, H, W, C, seed=9)
        dx2 = dev(base)
        k.p16_conv2d_bwd_data(dyb, wt, g, dx=dx2, accumulate=True)
        close(dx2, base.double() + xt.grad.permute(0, 2, 3, 1), rtol=6e-5)
    close(k.p16_conv2d_bwd_weight(xb, dyb, g), wtt.grad.permute(0, 2, 3, 1), rtol=6e-5)
    # the same from the FORWARD's f16 pair of x (converted to bf16 pairs inside the kernel): what the training step uses
    close(k.p16_conv2d_bwd_weight(xf, dyb, g, x_f16=True), wtt.grad.permute(0, 2, 3, 1), rtol=6e-5)
    # deterministic split-K (bdetr_p16_conv2d_bwd_weight_ws): slabs + fixed-order fold - the same values, and bit-identical twice
    prev = k.set_deterministic(True)
    try:
        d1 = k.p16_conv2d_bwd_weight(xf, dyb, g, x_f16=True)
        d2 = k.p16_conv2d_bwd_weight(xf, dyb, g, x_f16=True)
        base = dev(rnd(K, R, R, C, seed=12))
        d3 = k.p16_conv2d_bwd_weight(xb, dyb, g, dw=base.clone(), prezeroed=True)       # "+=" onto a running sum, like the atomics
    finally:
        k.set_deterministic(prev)
    close(d1, wtt.grad.permute(0, 2, 3, 1), rtol=6e-5)
    assert torch.equal(d1, d2)
    import ctypes
    from boosted_detr_amd import _lib
    split = _lib.lib().bdetr_p16_conv2d_bwd_weight_splitk(ctypes.byref(g.desc())) > 1       # an unsplit launch stores, a split one adds (include/bdetr.h)
    close(d3, (base.double().cpu() if split else 0) + wtt.grad.permute(0, 2, 3, 1), rtol=6e-5)


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", P16_CONVS + P16_BIG_CONVS)
def test_p16_conv_fwd_bwd(cuda, N, H, W, C, K, R, stride, pad):
    p16_conv_case(N, H, W, C, K, R, stride, pad)


@pytest.mark.parametrize("tile", ["128x128", "128x64", "64x64"])
def test_p16_conv_every_tile(cuda, tile, monkeypatch):
    """Each tile configuration on shapes with ragged edges in both tile dimensions (the chooser would not pick the
    big tiles for problems this small): BDETR_STILE is read once per process, so this runs in a child process."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_p16_gpu import p16_conv_case\n"
            "for shp in [(2, 13, 9, 64, 224, 3, 1, 1), (3, 10, 10, 96, 136, 1, 1, 0), (2, 12, 12, 128, 72, 1, 2, 0)]:\n"
            "    p16_conv_case(*shp)\n"
            "print('TILE_OK')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BDETR_STILE=tile), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TILE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_p16_unsupported_shapes_are_reported(cuda):
    from boosted_detr_amd import kernels as k
    assert not k.p16_supported(k.ConvGeom(2, 32, 32, 4, 64, 7, 7, 2, 3))          # the stem: 4 input channels
    assert not k.p16_supported(k.ConvGeom(2, 8, 8, 40, 64, 3, 3, 1, 1))           # 3x3 needs C % 32 == 0
    assert k.p16_supported(k.ConvGeom(2, 8, 8, 40, 64, 1, 1, 1, 0))


@pytest.mark.parametrize("rows,C,relu,res", [(800, 64, True, False), (3000, 256, True, True), (100, 1024, False, False), (98, 2048, True, True), (51, 8, True, True)])
def test_bn_p16_producers_match_the_fp32_kernels(cuda, rows, C, relu, res):
    """bdetr_bn_apply_p16 / bdetr_bn_bwd_p16 against bdetr_bn_apply / bdetr_bn_bwd: the fp32 outputs are bit-identical,
    the pair outputs are the bit-exact P16 image of them, and the P16 forms of the residual (f16 pair) and of the ReLU
    mask source (bf16 pair) give the result of their fp32 counterparts."""
    from boosted_detr_amd import kernels as k
    x = rnd(rows, C, seed=1) * 2 + 0.5
    gamma, beta = dev(1 + 0.1 * rnd(C, seed=2)), dev(0.1 * rnd(C, seed=3))
    resid = rnd(rows, C, seed=6) if res else None
    xd = dev(x)
    parts = k.colstats(xd)
    mean, rstd = k.bn_stats(rows, C, parts, 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=xd)
    want = k.bn_apply(xd, mean, rstd, gamma, beta, dev(resid) if res else None, relu)
    o32, of, ob = k.bn_apply_p16(xd, mean, rstd, gamma, beta, dev(resid) if res else None, relu)
    assert torch.equal(o32, want)
    assert torch.equal(bits(of), p16_ref(want.cpu(), True)) and torch.equal(bits(ob), p16_ref(want.cpu(), False))
    if res:
        # shortcut handed over as its f16 pair: equals applying the fp32 kernel to the round-tripped shortcut
        rf, _ = k.p16_pack(dev(resid), want_bf16=False)
        rt = k.p16_unpack(rf, True)
        o2, _, _ = k.bn_apply_p16(xd, mean, rstd, gamma, beta, rf, relu, want_f16=False, want_bf16=False, residual_p16=True)
        assert torch.equal(o2, k.bn_apply(xd, mean, rstd, gamma, beta, rt, relu))
    dout = dev(rnd(rows, C, seed=7))
    dx, dg, db, dres = k.bn_bwd(dout, want if (relu and res) else None, xd, mean, rstd, gamma, relu, False, want_residual_grad=res, beta=beta)
    dxb, dx32, dg2, db2, dres2 = k.bn_bwd_p16(dout, want if (relu and res) else None, xd, mean, rstd, gamma, relu, False, want_residual_grad=res,
                                              beta=beta, want_fp32=True)
    assert torch.equal(dx32, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
    assert torch.equal(bits(dxb), p16_ref(dx.cpu(), False))
    if res:
        assert torch.equal(dres2, dres)
        if relu:     # mask from the hi halves of the bf16 pair copy of the forward output
            dxb3, dx33, dg3, db3, dres3 = k.bn_bwd_p16(dout, ob, xd, mean, rstd, gamma, relu, False, want_residual_grad=True, beta=beta,
                                                       want_fp32=True, out_p16=True)
            assert torch.equal(dx33, dx) and torch.equal(dres3, dres) and torch.equal(dg3, dg) and torch.equal(db3, db)
            # ... and from the 1-bit-per-element mask that bn_apply_p16 writes on request
            o4, _, _, bits_ = k.bn_apply_p16(xd, mean, rstd, gamma, beta, dev(resid), relu, want_f16=False, want_bf16=False, want_mask=True)
            assert torch.equal(o4, want)
            dxb4, dx34, dg4, db4, dres4 = k.bn_bwd_p16(dout, bits_, xd, mean, rstd, gamma, relu, False, want_residual_grad=True, beta=beta,
                                                       want_fp32=True, out_p16=2)
            assert torch.equal(dx34, dx) and torch.equal(dres4, dres) and torch.equal(dg4, dg) and torch.equal(db4, db)


@pytest.mark.parametrize("N,H,W,C,K,R,relu", [(2, 20, 20, 64, 256, 1, True), (4, 40, 40, 256, 256, 3, True), (3, 9, 11, 128, 64, 3, False), (16, 40, 40, 256, 1024, 1, True)])
def test_bn_backward_reduction_fused_into_the_bwd_data_epilogue(cuda, N, H, W, C, K, R, relu):
    """bdetr_p16_conv2d_bwd_data_bnstats: the backward-data product of conv l+1 also emits the BatchNorm-backward partial sums of
    layer l (whose output fed the conv).  dx is bit-identical to the plain launch, and bn_bwd_p16 fed with the partials gives the
    dgamma / dbeta / dx of the two-pass form (same arithmetic, different summation order: 1e-5)."""
    from boosted_detr_amd import kernels as k
    g = k.ConvGeom(N, H, W, C, K, R, R, 1, R // 2)
    rows = N * H * W
    y_prev = dev(rnd(rows, C, seed=1) * 2 + 0.3)                      # layer l's conv output
    gamma, beta = dev(1 + 0.1 * rnd(C, seed=2)), dev(0.1 * rnd(C, seed=3))
    mean, rstd = k.bn_stats(rows, C, k.colstats(y_prev), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=y_prev)
    w = rnd(K, R, R, C, seed=4, scale=(R * R * C) ** -0.5)
    _, wt = k.p16_pack_conv_weights(dev(w), want_fwd=False)
    _, dyb = k.p16_pack(dev(rnd(N, g.OH, g.OW, K, seed=5)), want_f16=False)
    dx_plain = k.p16_conv2d_bwd_data(dyb, wt, g)
    dx, parts = k.p16_conv2d_bwd_data_bnstats(dyb, wt, g, y_prev, mean, rstd, gamma, beta, relu)
    assert torch.equal(dx, dx_plain)
    ref = k.bn_bwd_p16(dx.view(rows, C), None, y_prev, mean, rstd, gamma, relu, False, beta=beta, want_fp32=True)
    got = k.bn_bwd_p16(dx.view(rows, C), None, y_prev, mean, rstd, gamma, relu, False, beta=beta, want_fp32=True, pre=parts)
    close(got[2], ref[2], rtol=1e-5); close(got[3], ref[3], rtol=1e-5)          # dgamma, dbeta
    close(got[1], ref[1], rtol=1e-5)                                             # dx of the BatchNorm


@pytest.mark.parametrize("N,H,W,C,K", [(2, 20, 20, 64, 256), (3, 9, 11, 256, 64), (1, 5, 6, 40, 72), (16, 40, 40, 1024, 256), (4, 160, 160, 256, 64)])
def test_masked_skip_accumulate_in_the_bwd_data_epilogue(cuda, N, H, W, C, K):
    """bdetr_p16_conv2d_bwd_data_masked_accum: dx <- conv_transpose(dy) + dx * relu_mask with bn_apply_p16's bit mask (the skip
    branch of a residual unit merged in the first convolution's backward-data epilogue), and bdetr_relu_mask_apply, its
    stand-alone fallback (bit-exact)."""
    from boosted_detr_amd import kernels as k
    g = k.ConvGeom(N, H, W, C, K, 1, 1, 1, 0)
    rows = N * H * W
    # the unit's output: relu(bn(y) + shortcut); only its sign pattern matters here
    y = dev(rnd(rows, C, seed=1) * 2 + 0.3)
    gamma, beta = dev(1 + 0.1 * rnd(C, seed=2)), dev(0.1 * rnd(C, seed=3))
    mean, rstd = k.bn_stats(rows, C, k.colstats(y), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=y)
    shortcut = dev(rnd(rows, C, seed=4))
    out32, _, _, bits = k.bn_apply_p16(y, mean, rstd, gamma, beta, shortcut, True, want_fp32=True, want_f16=False, want_bf16=False, want_mask=True)
    on = (out32 > 0).double().cpu()
    d_out = rnd(N, H, W, C, seed=5)
    w = rnd(K, 1, 1, C, seed=6, scale=C ** -0.5)
    dy = rnd(N, H, W, K, seed=7)
    _, wt = k.p16_pack_conv_weights(dev(w), want_fwd=False)
    _, dyb = k.p16_pack(dev(dy), want_f16=False)
    dx = dev(d_out)
    k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dx, bits)
    want = dy.double().reshape(rows, K) @ w.double().reshape(K, C) + d_out.double().reshape(rows, C) * on
    close(dx.view(rows, C), want, rtol=6e-5)
    masked = k.relu_mask_apply_(dev(d_out).view(rows, C), bits)
    assert torch.equal(masked.cpu().double(), d_out.double().reshape(rows, C) * on)


@pytest.mark.parametrize("rows,C", [(800, 64), (3000, 256), (98, 2048)])
def test_bn_apply_p16_with_the_shortcut_batchnorm_folded_in(cuda, rows, C):
    """bdetr_bn_apply_p16(residual_p16 = 2): relu(bn(y) + bn_shortcut(y_shortcut)) in one pass is bit-identical to
    normalising the projection shortcut first and adding it as an fp32 residual."""
    from boosted_detr_amd import kernels as k
    y, ys = dev(rnd(rows, C, seed=1) * 2 + 0.5), dev(rnd(rows, C, seed=2) * 3 - 0.2)
    params = []
    for t, seed in ((y, 3), (ys, 5)):
        gamma, beta = dev(1 + 0.1 * rnd(C, seed=seed)), dev(0.1 * rnd(C, seed=seed + 1))
        mean, rstd = k.bn_stats(rows, C, k.colstats(t), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=t)
        params.append((mean, rstd, gamma, beta))
    shortcut = k.bn_apply(ys, *params[1], None, False)
    want32, wantf, wantb, wantm = k.bn_apply_p16(y, *params[0], shortcut, True, want_mask=True)
    got32, gotf, gotb, gotm = k.bn_apply_p16(y, *params[0], ys, True, want_mask=True, residual_bn=params[1])
    assert torch.equal(got32, want32) and torch.equal(gotf, wantf) and torch.equal(gotb, wantb) and torch.equal(gotm, wantm)


@pytest.mark.parametrize("N,H,W,C,K", [(2, 20, 20, 64, 256), (3, 9, 11, 256, 64), (16, 40, 40, 1024, 256)])
def test_masked_accumulate_also_reduces_the_previous_units_batchnorm_backward(cuda, N, H, W, C, K):
    """bdetr_p16_conv2d_bwd_data_masked_accum with a BatchNorm context: the launch that completes a residual unit's output
    gradient (product + skip gradient * this unit's mask) also emits the partial sums of the PREVIOUS unit's closing
    BatchNorm (ReLU decision from that unit's bit mask).  dx is bit-identical to the plain masked accumulate, and bn_bwd_p16 fed
    with the partials agrees with its own two-pass reduction (summation order: 1e-5)."""
    from boosted_detr_amd import kernels as k
    g = k.ConvGeom(N, H, W, C, K, 1, 1, 1, 0)
    rows = N * H * W

    def unit(seed):          # a residual unit's closing BatchNorm: conv output y, its statistics, the ReLU bit mask of relu(bn(y) + shortcut)
        y = dev(rnd(rows, C, seed=seed) * 2 + 0.3)
        gamma, beta = dev(1 + 0.1 * rnd(C, seed=seed + 1)), dev(0.1 * rnd(C, seed=seed + 2))
        mean, rstd = k.bn_stats(rows, C, k.colstats(y), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=y)
        _, _, _, bits = k.bn_apply_p16(y, mean, rstd, gamma, beta, dev(rnd(rows, C, seed=seed + 3)), True, want_fp32=False, want_f16=True,
                                       want_bf16=False, want_mask=True)
        return y, mean, rstd, gamma, beta, bits

    prev, this = unit(10), unit(20)
    d_out = rnd(N, H, W, C, seed=5)
    _, wt = k.p16_pack_conv_weights(dev(rnd(K, 1, 1, C, seed=6, scale=C ** -0.5)), want_fwd=False)
    _, dyb = k.p16_pack(dev(rnd(N, H, W, K, seed=7)), want_f16=False)
    plain = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dev(d_out), this[5])
    dx, parts = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dev(d_out), this[5], bn_ctx=prev)
    assert torch.equal(dx, plain)
    y, mean, rstd, gamma, beta, bits = prev
    ref = k.bn_bwd_p16(dx.view(rows, C), bits, y, mean, rstd, gamma, True, False, beta=beta, want_fp32=True, out_p16=2)
    got = k.bn_bwd_p16(dx.view(rows, C), bits, y, mean, rstd, gamma, True, False, beta=beta, want_fp32=True, out_p16=2, pre=parts)
    close(got[2], ref[2], rtol=1e-5); close(got[3], ref[3], rtol=1e-5)          # dgamma, dbeta
    close(got[1], ref[1], rtol=1e-5)                                             # dx of the BatchNorm
    # ... and, when the previous unit is a stage's first one, the sums of its PROJECTION SHORTCUT's BatchNorm over the same masked
    # gradient (round 4): same dx, same first sums, and bn_bwd_p16 of the shortcut's BatchNorm (ReLU decision = the unit's bits,
    # out_p16 = 2) fed with the second pair agrees with its own reduction pass
    y0 = dev(rnd(rows, C, seed=31) * 1.5 - 0.2)
    g0 = dev(1 + 0.1 * rnd(C, seed=32))
    b0 = dev(0.1 * rnd(C, seed=33))
    mean0, rstd0 = k.bn_stats(rows, C, k.colstats(y0), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=y0)
    dx2, parts_a, parts_b = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dev(d_out), this[5], bn_ctx=prev, bn_ctx2=(y0, mean0, rstd0))
    assert torch.equal(dx2, plain) and torch.equal(parts_a[0], parts[0]) and torch.equal(parts_a[1], parts[1]) and parts_b[0] is parts_a[0]
    ref0 = k.bn_bwd_p16(dx.view(rows, C), bits, y0, mean0, rstd0, g0, True, False, beta=b0, want_fp32=True, out_p16=2)
    got0 = k.bn_bwd_p16(dx.view(rows, C), bits, y0, mean0, rstd0, g0, True, False, beta=b0, want_fp32=True, out_p16=2, pre=parts_b)
    close(got0[2], ref0[2], rtol=1e-5); close(got0[3], ref0[3], rtol=1e-5); close(got0[1], ref0[1], rtol=1e-5)


@pytest.mark.parametrize("rms", [1e-3, 1e-5])
def test_p16_forward_on_small_activations(cuda, rms):
    """The f16 pair keeps its lo half UNSCALED (p16.h): for |x| < 2^-3 the pair's error is absolute, 2^-25, not relative.  An activation
    tensor whose RMS is far below 1 therefore loses relative accuracy: bound 2^-25 / RMS per element, and - errors of the K products
    adding incoherently - about that over the output's RMS too.  RMS 1e-3 (3e-5 per element): the forward product still meets 1e-4 of
    the output's scale, 10x inside the 1e-3 parity bar; RMS 1e-5: 3e-3 per element, the bound below is what DESIGN.md section 4 states
    (BatchNorm / LayerNorm outputs, the only forward operands of the training step, have RMS ~1)."""
    from boosted_detr_amd import kernels as k
    N, H, W, C, K, R = 2, 20, 20, 256, 128, 3
    x, w = rnd(N, H, W, C, seed=1, scale=rms), rnd(K, R, R, C, seed=2, scale=(R * R * C) ** -0.5)
    g = k.ConvGeom(N, H, W, C, K, R, R, 1, 1)
    xf, _ = k.p16_pack(dev(x), want_bf16=False)
    wf, _ = k.p16_pack_conv_weights(dev(w), want_bwd=False)
    y = k.p16_conv2d_fwd(xf, wf, None, g, 0)
    y = y[0] if isinstance(y, tuple) else y
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    bound = max(2e-5, 4 * 2.0 ** -25 / rms)
    assert err <= bound, (err, bound)


def test_halo_resident_weight_gradient_kernel_matches_too(cuda):
    """csrc/hwgrad.hip (3x3 weight gradients with both operands as sliding LDS windows) is off by default - it measured slower than
    the im2col kernel (profiles/r04_hwgrad_ab.json) - but stays correct: the same conv cases, fp64 references and deterministic-mode
    checks with BDETR_HWGRAD=1 (read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BDETR_HWGRAD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_p16_gpu.py"), "-q", "-x", "-k",
                        "test_p16_conv_fwd_bwd and (40-40-256 or 20-20-512 or 80-80-128 or 14-14-256)"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    assert int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1]) >= 4, r.stdout[-500:]


@pytest.mark.parametrize("N,H,W,C", [(2, 20, 20, 256), (3, 9, 7, 64), (16, 80, 80, 512), (1, 1, 1, 8)])
def test_bn_backward_reduces_over_the_even_pixels_of_a_strided_gradient(cuda, N, H, W, C):
    """bdetr_bn_bwd_p16_even_pixels: the gradient of a stage's last unit arrives through stride-2 1x1 convolutions and is zero off the
    pixels (2i, 2j); the reduction pass over those pixels alone gives the dgamma / dbeta / dx of the dense pass (the skipped rows add
    exact zeros: 1e-6 for the summation order), and the real producer of such a gradient - the stride-2 backward-data - is what
    ops.conv_bn tags."""
    from boosted_detr_amd import kernels as k
    rows = N * H * W
    y = dev(rnd(rows, C, seed=1) * 2 + 0.3)
    gamma, beta = dev(1 + 0.1 * rnd(C, seed=2)), dev(0.1 * rnd(C, seed=3))
    mean, rstd = k.bn_stats(rows, C, k.colstats(y), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=y)
    resid = dev(rnd(rows, C, seed=4))
    _, _, _, bits_ = k.bn_apply_p16(y, mean, rstd, gamma, beta, resid, True, want_f16=False, want_bf16=False, want_mask=True)
    g = rnd(N, H, W, C, seed=5)
    keep = torch.zeros(N, H, W, 1)
    keep[:, ::2, ::2] = 1
    dout = dev(g * keep).view(rows, C)
    ref = k.bn_bwd_p16(dout, bits_, y, mean, rstd, gamma, True, False, want_residual_grad=True, beta=beta, want_fp32=True, out_p16=2)
    got = k.bn_bwd_p16(dout, bits_, y, mean, rstd, gamma, True, False, want_residual_grad=True, beta=beta, want_fp32=True, out_p16=2, even_pixels=(N, H, W))
    close(got[2], ref[2], rtol=1e-6)
    close(got[3], ref[3], rtol=1e-6)
    close(got[1], ref[1], rtol=1e-6)
    assert torch.equal(got[4], ref[4])


def test_stride2_backward_data_gradients_carry_the_even_pixel_tag(cuda):
    """Two stride-2 1x1 consumers of one tensor (the next stage's c1 and projection shortcut).  Round 5: on an even map their accumulated
    input gradient is the COMPACT [N, H/2, W/2, C] tensor of the pixels (2i, 2j), tagged `_compact_even`; engine.materialise turns it into
    the dense zero-filled tensor (tag `_even_pixels`), which equals what the scatter form (BDETR_COMPACT_S2 off) writes, bit for bit; a
    dense contribution from a third consumer yields an untagged dense gradient with the same values either way."""
    from boosted_detr_amd import kernels as k, ops
    from boosted_detr_amd.backbone import _ConvBN
    from boosted_detr_amd.engine import Tape, join_side_stream, materialise, recording
    N, H, C = 2, 12, 64
    a, b, d = (_ConvBN("t/", f"c{i}", f"b{i}", C, 128, 1, s, 0, seed=i) for i, s in ((1, 2), (2, 2), (3, 1)))
    x32 = dev(rnd(N, H, H, C, seed=1)).relu_()
    seen = {}
    keep = ops.COMPACT_S2
    try:
        with k.gemm_precision("split"):
            for compact in (True, False):
                ops.COMPACT_S2 = compact
                for layers in ((a, b), (d, a, b), (a, b, d)):
                    x = x32.clone()
                    xf, _ = k.p16_pack(x, want_f16=True, want_bf16=False)
                    x._p16f, x._p16b = xf, None
                    tape = Tape()
                    with recording(tape):
                        outs = [l([x], training=True, relu=True, want_fp32=True) for l in layers]
                    seeds = {l: dev(rnd(*o.shape, seed=7 + (a, b, d).index(l))) for l, o in zip(layers, outs)}
                    grads = tape.backward({id(o): seeds[l] for l, o in zip(layers, outs)})
                    join_side_stream()
                    seen[(compact, layers)] = grads[id(x)]
    finally:
        ops.COMPACT_S2 = keep
    c2, s2 = seen[(True, (a, b))], seen[(False, (a, b))]
    assert getattr(c2, "_compact_even", None) == (N, H, H) and tuple(c2.shape) == (N, H // 2, H // 2, C)
    assert getattr(s2, "_even_pixels", None) == (N, H, H) and tuple(s2.shape) == (N, H, H, C)
    sparse = materialise(c2)
    assert getattr(sparse, "_even_pixels", None) == (N, H, H) and not hasattr(sparse, "_compact_even") and torch.equal(sparse, s2)
    for order in ((d, a, b), (a, b, d)):            # the dense consumer processed last / first in the backward pass
        dc, ds = seen[(True, order)], seen[(False, order)]
        assert not hasattr(dc, "_even_pixels") and not hasattr(dc, "_compact_even") and not hasattr(ds, "_even_pixels")
        close(dc, ds, rtol=1e-6)
    off = sparse.clone()
    off[:, ::2, ::2] = 0
    assert float(off.abs().max()) == 0.0 and float(sparse.abs().max()) > 0
    # a FROZEN consumer (moving statistics: the fp32 path, not the pre-split one) next to a trainable stride-2 one, in either order: the
    # compact tensor meets a convolution that cannot take it compact, and the merged gradient equals the scatter form's
    fz = _ConvBN("t/", "c9", "b9", C, 128, 1, 2, 0, seed=9)
    fz.trainable = False
    try:
        with k.gemm_precision("split"):
            for compact in (True, False):
                ops.COMPACT_S2 = compact
                for layers in ((a, fz), (fz, a)):
                    x = x32.clone()
                    xf, _ = k.p16_pack(x, want_f16=True, want_bf16=False)
                    x._p16f, x._p16b = xf, None
                    tape = Tape()
                    with recording(tape):
                        outs = [l([x], training=True, relu=True, want_fp32=True) for l in layers]
                    grads = tape.backward({id(o): dev(rnd(*o.shape, seed=17 + i)) for i, o in enumerate(outs)})
                    join_side_stream()
                    seen[("fz", compact, layers)] = materialise(grads[id(x)])
    finally:
        ops.COMPACT_S2 = keep
    for layers in ((a, fz), (fz, a)):
        close(seen[("fz", True, layers)], seen[("fz", False, layers)], rtol=1e-6)


@pytest.mark.parametrize("N,H,W,C,K", [(2, 20, 20, 64, 256), (3, 8, 12, 256, 64), (16, 40, 40, 1024, 256)])
def test_compact_even_pixel_gradient_through_masked_accumulate_and_batchnorm_backward(cuda, N, H, W, C, K):
    """Round 5: the gradient of a stage's last unit reaches it only through the next stage's stride-2 1x1 convolutions, i.e. at the pixels
    (2i, 2j).  It now exists as the COMPACT [N, H/2, W/2, C] tensor those products write densely - no zero-filled [N, H, W, C] tensor - and
    its two consumers read it through the pixel map: bdetr_p16_conv2d_bwd_data_masked_accum_compact (the unit's skip merge, into a fresh
    dense tensor, with and without the previous unit's fused BatchNorm sums) and bdetr_bn_bwd_p16_even_pixels(dout_compact=1).  Both must
    equal the dense-gradient forms bit for bit (same arithmetic, same summation order)."""
    from boosted_detr_amd import kernels as k
    g = k.ConvGeom(N, H, W, C, K, 1, 1, 1, 0)
    rows = N * H * W

    def unit(seed):
        y = dev(rnd(rows, C, seed=seed) * 2 + 0.3)
        gamma, beta = dev(1 + 0.1 * rnd(C, seed=seed + 1)), dev(0.1 * rnd(C, seed=seed + 2))
        mean, rstd = k.bn_stats(rows, C, k.colstats(y), 1.001e-5, 0.99, True, dev(torch.zeros(C)), dev(torch.ones(C)), like=y)
        _, _, _, bits = k.bn_apply_p16(y, mean, rstd, gamma, beta, dev(rnd(rows, C, seed=seed + 3)), True, want_fp32=False, want_f16=True,
                                       want_bf16=False, want_mask=True)
        return y, mean, rstd, gamma, beta, bits

    prev, this = unit(10), unit(20)
    compact = rnd(N, H // 2, W // 2, C, seed=5)
    dense = torch.zeros(N, H, W, C)
    dense[:, ::2, ::2] = compact
    _, wt = k.p16_pack_conv_weights(dev(rnd(K, 1, 1, C, seed=6, scale=C ** -0.5)), want_fwd=False)
    _, dyb = k.p16_pack(dev(rnd(N, H, W, K, seed=7)), want_f16=False)
    # the skip merge
    want = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dev(dense), this[5])
    got = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, torch.empty(N, H, W, C, device="cuda"), this[5], old_even=dev(compact))
    assert torch.equal(got, want)
    want2, parts_w = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dev(dense), this[5], bn_ctx=prev)
    got2, parts_g = k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, torch.empty(N, H, W, C, device="cuda"), this[5], bn_ctx=prev, old_even=dev(compact))
    assert torch.equal(got2, want) and torch.equal(want2, want)
    assert torch.equal(parts_g[0], parts_w[0]) and torch.equal(parts_g[1], parts_w[1]) and parts_g[2] == parts_w[2]
    # the unit's own BatchNorm backward (ReLU decision = its bit mask): reduction over the even pixels, apply over every pixel
    y, mean, rstd, gamma, beta, bits = this
    ref = k.bn_bwd_p16(dev(dense).view(rows, C), bits, y, mean, rstd, gamma, True, False, beta=beta, want_fp32=True, out_p16=2, even_pixels=(N, H, W))
    cmp_ = k.bn_bwd_p16(dev(compact).view(-1, C), bits, y, mean, rstd, gamma, True, False, beta=beta, want_fp32=True, out_p16=2, even_pixels=(N, H, W),
                        dout_compact=True)
    assert torch.equal(cmp_[0], ref[0]) and torch.equal(cmp_[1], ref[1]) and torch.equal(cmp_[2], ref[2]) and torch.equal(cmp_[3], ref[3])
