"""The arithmetic policies of the conv/GEMM family (include/bdetr.h: BDETR_GEMM_FP32 / BF16X3 / SPLIT; MIXED is
covered by test_default_policy_is_mixed) against fp64 products.  Tolerances: exact-fp32 MFMA and split-fp16 forward
products 2e-5 x max|ref| (fp32-grade round-off over the sum), split-bf16 6e-5 x max|ref| (2^-18 per product).

Which kernel variant a shape reaches depends on the tile chooser (csrc/igemm.hip choose_tile): the 128x128 split
tiles are only picked when the grid still fills the chip, i.e. at the benchmark's batch-16 shapes.  BIG_CONVS /
BIG_LINEARS below are sized to cross that threshold, and test_bench_step_variants_are_all_oracle_tested asserts
that EVERY (arithmetic, loaders, layouts, tile) tuple a config-2 batch-16 training step launches is also launched -
and compared with an fp64 reference - by the kernel-level cases of this file."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, dev, rnd

pytestmark = pytest.mark.gpu
TOL = {"fp32": 2e-5, "bf16x3": 6e-5, "split": 6e-5, "bf16x6": 2e-5}          # gradient products (and everything under bf16x3 / bf16x6)
TOL_FWD = {"fp32": 2e-5, "bf16x3": 6e-5, "split": 2e-5, "bf16x6": 2e-5}      # forward products: split = fp16x3, fp32-grade; bf16x6 = three bf16 terms, six products, fp32-grade

LINEARS = [(300, 256, 256), (6400, 256, 1024), (1600, 1024, 256), (130, 52, 48), (4000, 64, 64), (25600, 128, 512),
           (1600, 1024, 82), (1600, 256, 3), (1600, 256, 4)]          # the heads: unaligned 82- / 3-wide outputs, the narrow 128x32 tile
BIG_LINEARS = [(102400, 256, 128), (25600, 1024, 256), (6400, 2048, 512)]
CONVS = [(2, 20, 20, 64, 64, 3, 1, 1), (2, 20, 20, 256, 128, 1, 2, 0), (1, 9, 11, 128, 32, 3, 1, 1),
         (3, 7, 7, 512, 2048, 1, 1, 0), (4, 40, 40, 256, 256, 3, 1, 1), (2, 32, 32, 4, 64, 7, 2, 3)]
# shapes that reach the 128x128 split tiles (>= 0.75 workgroups per CU): the batch-16 3x3 convolutions of ResNet
# stages 3-5, a batch-16 stage-2 1x1, a strided 1x1 and the batch-16 stem
BIG_CONVS = [(8, 80, 80, 128, 128, 3, 1, 1), (16, 40, 40, 256, 256, 3, 1, 1), (16, 20, 20, 512, 512, 3, 1, 1),
             (4, 80, 80, 256, 512, 1, 1, 0), (16, 80, 80, 512, 256, 1, 2, 0), (16, 160, 160, 64, 64, 3, 1, 1),
             (4, 320, 320, 4, 64, 7, 2, 3)]


@pytest.fixture(params=["fp32", "bf16x3", "split", "bf16x6"])
def mode(request, cuda):
    from boosted_detr_amd import kernels as k
    prev = k.set_gemm_precision(request.param)
    assert k.get_gemm_precision() == request.param
    yield request.param
    k.set_gemm_precision(prev)


def test_default_policy_is_mixed(cuda):
    from boosted_detr_amd import kernels as k
    assert k.get_gemm_precision() == "mixed"
    # forward product exact, gradient product split: on a long reduction the split error is ~8x the fp32 one
    x, w = rnd(512, 4096, seed=1), rnd(256, 4096, seed=2)
    ref = x.double() @ w.double().T
    e_fwd = (k.linear_fwd(dev(x), dev(w), None, 0).cpu().double() - ref).norm() / ref.norm()
    dy = rnd(512, 256, seed=3)
    ref_b = dy.double() @ w.double()
    e_bwd = (k.linear_bwd_data(dev(dy), dev(w)).cpu().double() - ref_b).norm() / ref_b.norm()
    assert e_fwd < 1.5e-6 and 1.5e-6 < e_bwd < 1e-5, (float(e_fwd), float(e_bwd))


def test_three_term_bf16_products_are_fp32_grade(cuda):
    """BDETR_GEMM_BF16X6 (x = hi + mid + lo in bf16, six products) on a long reduction, relative L2 error against fp64: within 1.5x
    the exact-fp32 MFMA's own and well under the two-term split's 2^-18 - for the forward, backward-data and
    weight-gradient flavours and for operands spread over 12 decades of magnitude (bf16 keeps fp32's exponent range)."""
    from boosted_detr_amd import kernels as k
    x, w, dy = rnd(512, 4096, seed=1), rnd(256, 4096, seed=2), rnd(512, 256, seed=3)
    scale = 10.0 ** torch.linspace(-6, 6, 4096)
    xs = x * scale                                       # columns from 1e-6 to 1e6
    refs = {"fwd": x.double() @ w.double().T, "bwd": dy.double() @ w.double(), "wgrad": dy.double().T @ x.double(),
            "wide": xs.double() @ (w / scale).double().T}
    err = {}
    for mode in ("fp32", "bf16x3", "bf16x6"):
        prev = k.set_gemm_precision(mode)
        try:
            got = {"fwd": k.linear_fwd(dev(x), dev(w), None, 0), "bwd": k.linear_bwd_data(dev(dy), dev(w)), "wgrad": k.linear_bwd_weight(dev(dy), dev(x)),
                   "wide": k.linear_fwd(dev(xs), dev(w / scale), None, 0)}
        finally:
            k.set_gemm_precision(prev)
        err[mode] = {n: float((got[n].cpu().double() - refs[n]).norm() / refs[n].norm()) for n in refs}
    print(err)
    for n in refs:
        assert err["bf16x6"][n] <= 1.5 * err["fp32"][n] + 1e-7, (n, err)
        assert err["bf16x6"][n] < 0.5 * err["bf16x3"][n], (n, err)      # (measured 0.07-0.22: on the 4096-long sums the fp32 ACCUMULATION is what is left - 1.0e-6 against the fp32 MFMA's 1.1e-6)


def linear_case(mode, M, K, O):
    from boosted_detr_amd import kernels as k
    x, w, b, dy = rnd(M, K, seed=1), rnd(O, K, seed=2, scale=K ** -0.5), rnd(O, seed=3), rnd(M, O, seed=4)
    close(k.linear_fwd(dev(x), dev(w), dev(b), 1), (x.double() @ w.double().T + b.double()).relu(), rtol=TOL_FWD[mode])
    close(k.linear_bwd_data(dev(dy), dev(w)), dy.double() @ w.double(), rtol=TOL[mode])
    close(k.linear_bwd_weight(dev(dy), dev(x)), dy.double().T @ x.double(), rtol=TOL[mode])


def conv_case(mode, N, H, W, C, K, R, stride, pad):
    from boosted_detr_amd import kernels as k
    x, w, b = rnd(N, H, W, C, seed=1), rnd(K, R, R, C, seed=2, scale=(R * R * C) ** -0.5), rnd(K, seed=3)
    g = k.ConvGeom(N, H, W, C, K, R, R, stride, pad)
    y, (ps, pq, n) = k.conv2d_fwd(dev(x), dev(w), dev(b), g, 0, want_stats=True)
    xt = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wt = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xt, wt, b.double(), stride=stride, padding=pad)
    close(y, ref.permute(0, 2, 3, 1), rtol=TOL_FWD[mode])
    close(ps.sum(0), ref.permute(0, 2, 3, 1).reshape(-1, K).sum(0), rtol=1e-4)
    dy = rnd(*ref.shape, seed=4).double()
    ref.backward(dy)
    dyn = dy.permute(0, 2, 3, 1).float()
    if R != 7:
        close(k.conv2d_bwd_data(dev(dyn), dev(w), g), xt.grad.permute(0, 2, 3, 1), rtol=TOL[mode])
        base = rnd(N, H, W, C, seed=9)
        dx2 = dev(base)
        k.conv2d_bwd_data(dev(dyn), dev(w), g, dx=dx2, accumulate=True)       # the residual-merge epilogue
        close(dx2, base.double() + xt.grad.permute(0, 2, 3, 1), rtol=TOL[mode])
    close(k.conv2d_bwd_weight(dev(x), dev(dyn), g), wt.grad.permute(0, 2, 3, 1), rtol=max(TOL[mode], 5e-5))


@pytest.mark.parametrize("M,K,O", LINEARS + BIG_LINEARS)
def test_linear_all_flavours(mode, M, K, O):
    linear_case(mode, M, K, O)


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", CONVS + BIG_CONVS)
def test_conv_all_flavours(mode, N, H, W, C, K, R, stride, pad):
    conv_case(mode, N, H, W, C, K, R, stride, pad)


def _prof_tuples(path):
    """{(kind, bm, bn)} of the igemm launches recorded since bdetr_prof_enable(1); kind = arithmetic * 10000 +
    loader/layout code of both operands (csrc/igemm.hip launch_cfg)."""
    import csv
    import torch as _t
    from boosted_detr_amd import _lib
    _t.cuda.synchronize()
    _lib.check(_lib.lib().bdetr_prof_dump(path.encode()), "prof_dump")
    with open(path) as f:
        return {(int(r["kind"]), int(r["bm"]), int(r["bn"])) for r in csv.DictReader(f)}


def test_bench_step_variants_are_all_oracle_tested(cuda, tmp_path):
    """Coverage of the kernels the benchmark actually runs: every igemm variant launched by one BASELINE configs[1]
    training step at the bench's batch 16 must also be launched - and compared with an fp64 reference, in this very
    test - by the kernel-level cases above under the same 'split' policy."""
    import os
    import sys
    from boosted_detr_amd import _lib, engine
    from boosted_detr_amd import kernels as k
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    L = _lib.lib()
    side_was = engine._SIDE["enabled"]
    engine.set_side_stream_enabled(False)            # hipEvent brackets on one stream
    try:
        with k.gemm_precision("split"):
            L.bdetr_prof_enable(1)
            for shp in LINEARS + BIG_LINEARS:
                linear_case("split", *shp)
            for shp in CONVS + BIG_CONVS:
                conv_case("split", *shp)
            _grouped_case("split")
            from test_p16_gpu import P16_BIG_CONVS, P16_CONVS, p16_conv_case
            for shp in P16_CONVS + P16_BIG_CONVS:              # the pre-split operand kernels (csrc/sgemm.hip)
                p16_conv_case(*shp, accumulate=True)
            from test_rowchain_gpu import test_rowchain_kernels_match_fp64 as rowchain_case      # the fused row chains (csrc/rowchain.hip)
            for M_, stages in ((1600, 3), (98, 1)):
                rowchain_case(None, M_, stages)
            tested = _prof_tuples(str(tmp_path / "tested.csv"))
        args = type("A", (), dict(model="detr", fashionpedia=False, backbone="ResNet", image=640, image_w=0, layers=6, queries=100,
                                  learners=3, batch=16))()
        model = bench.build_model(args)
        assert model.train_gemm_precision == "split"
        host = bench.make_batch(16, 640, 640, 100, 82, seed=1234)
        model.train_step(host)                      # build-by-first-call (not recorded: first-step staging differs)
        L.bdetr_prof_enable(1)
        model.train_step(host)
        launched = _prof_tuples(str(tmp_path / "bench.csv"))
    finally:
        L.bdetr_prof_enable(0)
        engine.set_side_stream_enabled(side_was)
    assert len(launched) >= 8
    missing = sorted(launched - tested)
    assert not missing, f"bench-step igemm variants never compared with an oracle (kind, bm, bn): {missing}; tested: {sorted(tested)}"


def _grouped_case(mode):
    from boosted_detr_amd import kernels as k
    xs = [rnd(m, 256, seed=m) for m in (400, 400, 100)]
    ws = [rnd(256, 256, seed=7 + i, scale=1 / 16) for i in range(3)]
    bs = [rnd(256, seed=11 + i) for i in range(3)]
    ys = k.linear_fwd_group([dev(x) for x in xs], [dev(w) for w in ws], [dev(b) for b in bs])
    for y, x, w, b in zip(ys, xs, ws, bs):
        close(y, x.double() @ w.double().T + b.double(), rtol=TOL_FWD[mode])
    dxs = k.linear_bwd_data_group([dev(x) for x in xs], [dev(w) for w in ws])
    for dx, x, w in zip(dxs, xs, ws):
        close(dx, x.double() @ w.double(), rtol=TOL[mode])
    # the encoder's shapes (6400-row q/k/v projections of a batch-16 step)
    xs = [rnd(6400, 256, seed=20 + i) for i in range(3)]
    ys = k.linear_fwd_group([dev(x) for x in xs], [dev(w) for w in ws], [dev(b) for b in bs])
    for y, x, w, b in zip(ys, xs, ws, bs):
        close(y, x.double() @ w.double().T + b.double(), rtol=TOL_FWD[mode])
    dxs = k.linear_bwd_data_group([dev(x) for x in xs], [dev(w) for w in ws])
    for dx, x, w in zip(dxs, xs, ws):
        close(dx, x.double() @ w.double(), rtol=TOL[mode])


def test_grouped_launch(mode):
    _grouped_case(mode)


def test_training_step_runs_split_and_restores_policy(cuda):
    """Model.forward_backward scopes the 'split' policy (fp16x3 forward, bf16x3 gradients) and restores the
    library default afterwards; the forward it produces stays within 1e-3 of the fp64 oracle and is at least
    as close to it as the exact-fp32 policy is on this case."""
    import numpy as np
    from boosted_detr_amd import kernels as k
    from oracle import detr_oracle as O
    from test_model_gpu import build_model
    cfg = O.CONFIG1
    batch = O.make_batch(cfg, 2, 20, seed=1234, num_objects=[3, 7])
    params = O.make_params(cfg, seed=0)
    out, _ = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    want = out.cat_preds.detach().numpy()
    errs = {}
    for mode in ("split", "fp32"):
        model = build_model(cfg, False)
        assert model.train_gemm_precision == "split"
        model.train_gemm_precision = mode
        model.forward_backward(batch)
        model.set_weights_dict(params)
        y = model.forward_backward(batch)
        assert k.get_gemm_precision() == "mixed"
        errs[mode] = np.abs(y[0].cpu().numpy().astype(np.float64) - want).max() / np.abs(want).max()
    assert errs["split"] < 1e-3 and errs["split"] < 2 * errs["fp32"] + 1e-6, errs


def test_fp16_split_range(cuda):
    """Split-fp16 forward products: fp32-grade inside [1.2e-4, 65504), graceful below (absolute error < 2^-35),
    NaN - not a silently wrong number - above."""
    from boosted_detr_amd import kernels as k
    x, w = rnd(256, 512, seed=1).clamp(-3, 3), rnd(128, 512, seed=2) / 16
    ref = x.double() @ w.double().T
    with k.gemm_precision("split"):
        e = lambda y, r: float((y.cpu().double() - r).norm() / r.norm())
        assert e(k.linear_fwd(dev(x), dev(w), None, 0), ref) < 4e-7
        assert e(k.linear_fwd(dev(x * 2e4), dev(w), None, 0), ref * 2e4) < 4e-7
        assert e(k.linear_fwd(dev(x * 1e-6), dev(w), None, 0), ref * 1e-6) < 1e-4
        big = x.clone(); big[3, 7] = 7e4
        y = k.linear_fwd(dev(big), dev(w), None, 0)
        assert not torch.isfinite(y[3]).any() and torch.isfinite(y[4]).all()


def test_frozen_batchnorm_demotes_the_fp16_forward(cuda):
    """Frozen BatchNormalization (moving statistics) need not normalise: with fresh moving statistics the config-1
    backbone activations exceed the f16 range.  The step must notice (ops._bn_forward -> kernels.demote_split_forward),
    finish its forward on the exact-fp32 path and still match the fp64 oracle; the policy is restored afterwards."""
    import numpy as np
    from boosted_detr_amd import kernels as k
    from oracle import detr_oracle as O
    from test_model_gpu import build_model
    cfg = O.CONFIG1
    batch = O.make_batch(cfg, 2, 20, seed=1234, num_objects=[3, 7])
    params = O.make_params(cfg, seed=0)
    model = build_model(cfg, False)
    model.forward_backward(batch)
    model.set_weights_dict(params)
    for layer in (model.EncoderBackbone, model.BackboneNeck, model.CategoryPredictionHead, model.AttributePredictionHead, model.BoxPredictionHead):
        layer.trainable = False                     # every BatchNormalization of the model on moving statistics, like the oracle's frozen_bn
    assert model.train_gemm_precision == "split"
    y = model.forward_backward(batch)
    assert k.get_gemm_precision() == "mixed"
    out, _ = O.train_step_grads(cfg, params, batch, dtype=torch.float64, frozen_bn=True)
    for got, want in zip(y, (out.cat_preds, out.attribute_preds, out.box_preds)):
        g, w = got.cpu().numpy().astype(np.float64), want.detach().numpy()
        assert np.isfinite(g).all() and np.abs(g - w).max() <= 1e-3 * np.abs(w).max()
