"""The fused stem tail (csrc/norm.hip stem_*: BatchNorm -> ReLU -> ZeroPadding2D(1) -> MaxPool 3x3/2 of the raw conv1 output, and its
backward) against (a) the unfused HIP kernels it replaces - bit for bit, the arithmetic is the same - and (b) an fp64 PyTorch-CPU
restatement of the same Keras layers (ResNet50 conv1_bn .. pool1_pool, reference backbone.py:79-80)."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, dev, rnd

pytestmark = pytest.mark.gpu
C = 64


def _inputs(N, H, W, seed):
    y = rnd(N, H, W, C, seed=seed, scale=1.5) + 0.3
    gamma, beta = rnd(C, seed=seed + 1, scale=0.3) + 1.0, rnd(C, seed=seed + 2, scale=0.3)
    mean = y.double().mean((0, 1, 2))
    var = y.double().var((0, 1, 2), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1.001e-5)
    return y, mean.float(), rstd.float(), gamma, beta


@pytest.mark.parametrize("N,H,W", [(2, 16, 20, ), (3, 13, 17), (1, 1, 1), (1, 2, 3), (2, 64, 64)])
def test_stem_pool_matches_the_unfused_kernels_and_fp64(cuda, N, H, W):
    from boosted_detr_amd import kernels as k
    y, mean, rstd, gamma, beta = _inputs(N, H, W, 3)
    yd, md, rd, gd, bd = (dev(t) for t in (y, mean, rstd, gamma, beta))
    o32, of, tap = k.stem_pool_fwd(yd, md, rd, gd, bd, want_fp32=True)
    # (a) the unfused launches
    a = k.bn_apply(yd.view(-1, C), md, rd, gd, bd, None, True).view(N, H, W, C)
    p = k.maxpool_fwd(a)
    assert torch.equal(o32, p)
    pf, _ = k.p16_pack(p, want_f16=True, want_bf16=False)
    assert torch.equal(of.view(torch.int32), pf.view(torch.int32))
    assert int(tap.max()) <= 8
    PH, PW = p.shape[1:3]
    dpool = dev(rnd(N, PH, PW, C, seed=9))
    dy, dg, db = k.stem_pool_bwd(dpool, tap, yd, md, rd, gd, bd)
    da = k.maxpool_bwd(a, p, dpool)
    dy0, dg0, db0, _ = k.bn_bwd(da.view(-1, C), None, yd.view(-1, C), md, rd, gd, True, False, beta=bd)
    assert torch.equal(dg, dg0) and torch.equal(db, db0)
    assert torch.equal(dy.view(-1, C), dy0)
    # (b) fp64
    y64 = y.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mu = y64.mean((0, 1, 2))
    xh = (y64 - mu) / torch.sqrt(y64.var((0, 1, 2), unbiased=False) + 1.001e-5)
    act = F.relu(xh * g64 + b64).permute(0, 3, 1, 2)
    ref = F.max_pool2d(F.pad(act, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
    close(o32, ref, rtol=1e-5)
    close(k.p16_unpack(of, True), ref, rtol=1e-5)
    ref.backward(dpool.cpu().double())
    if N * H * W > 8:                # (a handful of rows: the statistics' own gradient is ill-conditioned in fp32)
        close(dy, y64.grad, rtol=2e-4)
        close(dg, g64.grad, rtol=1e-4)
        close(db, b64.grad, rtol=1e-4)


def test_ties_go_to_the_first_tap(cuda):
    """A constant positive plane: every window's maximum is tied nine ways; the gradient lands on the first real tap of each window
    (TensorFlow's MaxPoolGrad routes to the arg max, one element per window) and nothing is counted twice."""
    from boosted_detr_amd import kernels as k
    N, H, W = 1, 6, 6
    y = torch.ones(N, H, W, C)
    mean, rstd, gamma, beta = torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C)
    yd, md, rd, gd, bd = (dev(t) for t in (y, mean, rstd, gamma, beta))
    _, of, tap = k.stem_pool_fwd(yd, md, rd, gd, bd)
    t = tap.cpu()[0, :, :, 0]
    assert t.tolist() == [[4, 3, 3], [1, 0, 0], [1, 0, 0]]
    dpool = torch.ones(N, 3, 3, C).cuda()
    # d(sum of pooled)/d(normalised activation): one unit per window, at that tap (BatchNorm's own backward is exercised above)
    dy, dg, db = k.stem_pool_bwd(dpool, tap, yd, md, rd, gd, bd)
    assert float(db[0]) == 9.0 and float(dg[0]) == 9.0      # sum g = 9 windows; xhat = 1 everywhere


@pytest.mark.parametrize("B,H", [(2, 64), (3, 40)])
def test_stem_op_fused_equals_unfused(cuda, B, H):
    """ops.conv_bn_relu_maxpool with the fused tail against conv_bn + maxpool: output (as the consumers read it) and the gradients
    of the stem's kernel, gamma and beta - same launches up to the fused ones, which agree bit for bit."""
    from boosted_detr_amd import kernels as k, ops
    from boosted_detr_amd.backbone import ResNet
    from boosted_detr_amd.engine import Tape, join_side_stream, recording
    net = ResNet(stages=[(64, 1, 1)], name="r", seed=3)
    st = net.stem
    x = dev(rnd(B, H, H, 4, seed=1))
    x[..., 3] = 0
    gout = dev(rnd(B, H // 4, H // 4, C, seed=2))
    res = {}
    keep = ops.STEM_FUSE
    try:
        with k.gemm_precision("split"):
            for fused in (False, True):
                ops.STEM_FUSE = fused
                for v in st.variables:
                    v.reset_grad()
                mm = [st.bn.moving_mean.value.clone(), st.bn.moving_var.value.clone()]
                tape = Tape()
                with recording(tape):
                    out = ops.conv_bn_relu_maxpool(x, st.kernel, st.bias, st.bn, st.stride, st.pad, True, True)
                assert bool(getattr(out, "_p16_only", False)) == fused
                tape.backward({id(out): gout.clone()})
                join_side_stream()
                torch.cuda.synchronize()
                res[fused] = (ops.as_fp32(out).clone(), {v.name: v.grad.clone() for v in st.variables if v.grad is not None},
                              st.bn.moving_mean.value.clone(), st.bn.moving_var.value.clone())
                st.bn.moving_mean.value.copy_(mm[0])
                st.bn.moving_var.value.copy_(mm[1])
    finally:
        ops.STEM_FUSE = keep
    a, b = res[False], res[True]
    close(b[0], a[0], rtol=3e-7)                 # the f16 pair round trip: 2^-22
    assert set(a[1]) == set(b[1]) and len(a[1]) >= 3
    for n in a[1]:
        close(b[1][n], a[1][n], rtol=1e-6, atol=1e-6 * max(float(a[1][n].abs().max()), 1e-3))
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


@pytest.mark.parametrize("N,H,W", [(2, 32, 48), (1, 64, 64), (3, 18, 22), (4, 224, 224)])
def test_stem_weight_gradient_on_the_space_to_depth_path(cuda, N, H, W):
    """Round 5: the 7x7 / stride-2 / pad-3 stem's weight gradient through a space-to-depth view of the 4-channel image (x2 [N,H/2,W/2,16], a
    4x4 size-preserving convolution with low-side padding 2) on the pre-split XX kernel: against the fp64 gradient of the same convolution
    (bf16-pair products: 2e-5 of the tensor's scale, the igemm kernel's own bound) and against igemm.hip's kernel under the same policy."""
    from boosted_detr_amd import kernels as k
    K_ = 64
    x = rnd(N, H, W, 4, seed=1, scale=40.0)
    x[..., 3] = 0                                                       # the padding channel image_prep adds
    dy = rnd(N, H // 2, W // 2, K_, seed=2)
    xt = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    wt = torch.zeros(K_, 4, 7, 7, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(xt, wt, stride=2, padding=3)
    assert tuple(out.shape) == (N, K_, H // 2, W // 2)
    out.backward(dy.double().permute(0, 3, 1, 2))
    want = wt.grad.permute(0, 2, 3, 1).contiguous()                     # [K,7,7,4]
    with k.gemm_precision("split"):
        _, dyb = k.p16_pack(dev(dy), want_f16=False)
        got = k.stem_bwd_weight_s2d(dev(x), dyb, torch.full((K_, 7, 7, 4), 7.0, device="cuda"))       # (every element is overwritten)
        g = k.ConvGeom(N, H, W, 4, K_, 7, 7, 2, 3)
        old = k.conv2d_bwd_weight(dev(x), dev(dy), g)
    scale = float(want.abs().max())
    assert float((got.double().cpu() - want).abs().max()) <= 3e-5 * scale, float((got.double().cpu() - want).abs().max()) / scale
    assert float((old.double().cpu() - want).abs().max()) <= 3e-5 * scale
    assert float(got[..., 3].abs().max()) == 0.0
