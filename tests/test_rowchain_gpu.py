"""The fused row-chain kernels (csrc/rowchain.hip: OutputProjection + Add/Dropout/LayerNorm [+ FeedForwardBlock] of a transformer layer in
one launch per direction) against (a) an fp64 PyTorch-CPU restatement of the same reference lines (transformers.py:101,135-137,174-180)
and (b) the unfused HIP path (separate GEMM / LayerNorm launches) with the SAME dropout masks.

Tolerances: forward products are f16 pairs (fp32-grade): 3e-5 x max|ref|; gradient products are bf16 pairs (2^-18 per product): 2e-4 x
max|ref| on tensors that went through up to three such products."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, dev, rnd

pytestmark = pytest.mark.gpu
D = 256


def chain_ref(ctx, resid, p, nstages, eps=1e-3):
    """fp64 reference; p: dict of torch fp64 tensors (requires_grad leaves).  Weights are [out][in]."""
    a = ctx @ p["Wo"].T + p["bo"]
    pre1 = resid + a
    x1 = F.layer_norm(pre1, (D,), p["g1"], p["be1"], eps)
    if nstages == 1:
        return {"pre1": pre1, "x1": x1, "out": x1}
    h = torch.relu(x1 @ p["W1"].T + p["b1"])
    f = h @ p["W2"].T + p["b2"]
    pre2 = x1 + f
    x2 = F.layer_norm(pre2, (D,), p["g2"], p["be2"], eps)
    return {"pre1": pre1, "x1": x1, "h": h, "pre2": pre2, "x2": x2, "out": x2}


def make_params(seed):
    names = {"Wo": (D, D), "bo": (D,), "g1": (D,), "be1": (D,), "W1": (D, D), "b1": (D,), "W2": (D, D), "b2": (D,), "g2": (D,), "be2": (D,)}
    p = {}
    for i, (k, shp) in enumerate(names.items()):
        t = rnd(*shp, seed=seed + i, scale=(D ** -0.5 if len(shp) == 2 else 0.3))
        if k in ("g1", "g2"):
            t = t + 1.0
        p[k] = t
    return p


@pytest.mark.parametrize("M,nstages", [(6400, 3), (1600, 3), (1600, 1), (98, 3), (33, 1), (1, 3)])
def test_rowchain_kernels_match_fp64(cuda, M, nstages):
    from boosted_detr_amd import _lib, kernels as k
    p = make_params(10)
    ctx, resid, dout = rnd(M, D, seed=1), rnd(M, D, seed=2), rnd(M, D, seed=3)
    n = int(_lib.lib().bdetr_rowchain_pack_elems())
    names = ["Wo", "W1", "W2"][:nstages]
    wdev = {w: dev(p[w]) for w in names}
    fwd = {w: torch.empty(n, device="cuda") for w in names}
    bwd = {w: torch.empty(n, device="cuda") for w in names}
    table = torch.tensor([[wdev[w].data_ptr(), fwd[w].data_ptr(), bwd[w].data_ptr()] for w in names], dtype=torch.int64).cuda()
    k.rowchain_pack_weights(table)
    biases = [dev(p[b]) for b in ["bo", "b1", "b2"][:nstages]]
    ln1, ln2 = (dev(p["g1"]), dev(p["be1"])), ((dev(p["g2"]), dev(p["be2"])) if nstages == 3 else None)
    saved = k.rowchain_fwd(dev(ctx), dev(resid), [fwd[w] for w in names], biases, ln1, ln2, 1e-3, 0.0, 0, 0, None)
    pd = {kk: v.double().requires_grad_(True) for kk, v in p.items()}
    cd, rd = ctx.double().requires_grad_(True), resid.double().requires_grad_(True)
    ref = chain_ref(cd, rd, pd, nstages)
    for key in ("pre1", "x1") + (("h", "pre2", "x2") if nstages == 3 else ()):
        close(saved[key], ref[key], rtol=3e-5)
    mu = ref["pre1"].mean(-1)
    close(saved["mean1"], mu, rtol=3e-5, atol=3e-5 * float(ref["pre1"].detach().abs().max()))
    close(saved["rstd1"], 1.0 / torch.sqrt(ref["pre1"].var(-1, unbiased=False) + 1e-3), rtol=3e-5)
    # backward
    ref["out"].backward(dout.double())
    gammas = (ln1[0],) + ((ln2[0],) if nstages == 3 else ())
    dctx, dres, G, partials, nparts = k.rowchain_bwd(dev(dout), saved, [bwd[w] for w in names], gammas, 0.0, 0, 0, None)
    close(dctx, cd.grad, rtol=2e-4)
    close(dres, rd.grad, rtol=2e-4)
    vec = [torch.zeros(D, device="cuda") for _ in range(7)]
    k.rowchain_reduce(partials, nparts, vec if nstages == 3 else [None] * 4 + vec[4:], [0] * 7)
    want = [pd["g2"].grad, pd["be2"].grad, pd["b2"].grad, pd["b1"].grad, pd["g1"].grad, pd["be1"].grad, pd["bo"].grad]
    for i in range(7):
        if nstages == 3 or i >= 4:
            close(vec[i], want[i], rtol=2e-4)
    # the Dense layers' output gradients reproduce the weight gradients: dW = G^T X
    xs = [ctx] + ([ref["x1"].detach(), ref["h"].detach()] if nstages == 3 else [])
    for g, x, w in zip(G, xs, names):
        close(g.double().cpu().T @ x.double(), pd[w].grad, rtol=2e-4)


def _blocks(seed, rate):
    from boosted_detr_amd import transformers as T
    T.AttentionBlock.dropout_rate = rate
    T.FeedForwardBlock.dropout_rate = rate
    attn = T.AttentionBlock(8, name="A", seed=seed)
    ffn = T.FeedForwardBlock(name="F", seed=seed)
    return attn, ffn


@pytest.mark.parametrize("rate", [0.0, 0.1])
@pytest.mark.parametrize("with_ffn", [True, False])
@pytest.mark.parametrize("nq,nk", [(100, 400), (49, 49)])
def test_fused_block_equals_the_unfused_block(cuda, rate, with_ffn, nq, nk):
    """attention_then() on the fused path against the layer-by-layer path: same weights, same inputs, same dropout seeds (the masks are a
    function of seed, site and element index in both): outputs and every gradient."""
    from boosted_detr_amd import kernels as k, ops, transformers as T
    from boosted_detr_amd.engine import Tape, recording
    keep = (T.AttentionBlock.dropout_rate, T.FeedForwardBlock.dropout_rate)
    try:
        attn, ffn = _blocks(5, rate)
        B = 3
        q, kv, v = dev(rnd(B, nq, D, seed=1)), dev(rnd(B, nk, D, seed=2)), dev(rnd(B, nk, D, seed=3))
        gout = dev(rnd(B, nq, D, seed=4))
        results = {}
        with k.gemm_precision("split"):
            attn([q, kv, v], training=True)                      # build
            if with_ffn:
                ffn([q], training=True)
            for var in attn.variables + ffn.variables:
                if var.value.dim() == 1:                         # biases / gamma / beta away from their 0 / 1 initial values
                    var.value.add_(dev(rnd(*var.value.shape, seed=hash(var.name) % 1000, scale=0.2)))
            from boosted_detr_amd.engine import bump_weights_version
            bump_weights_version()
            for fused in (False, True):
                ops.ROWCHAIN[0] = fused
                ops.set_dropout_seed(1234)
                for var in attn.variables + ffn.variables:
                    var.reset_grad()
                tape = Tape()
                with recording(tape):
                    out = T.attention_then(attn, ffn if with_ffn else None, [q, kv, v], True)
                grads = tape.backward({id(out): gout})
                from boosted_detr_amd.engine import join_side_stream
                join_side_stream()
                torch.cuda.synchronize()
                results[fused] = (out.clone(), {n: grads[id(t)].clone() for n, t in (("q", q), ("k", kv), ("v", v)) if id(t) in grads},
                                  {var.name: var.grad.clone() for var in attn.variables + (ffn.variables if with_ffn else []) if var.grad is not None})
        ops.ROWCHAIN[0] = True
        a, b = results[False], results[True]
        close(b[0], a[0], rtol=3e-5)
        assert set(a[1]) == set(b[1]) == {"q", "k", "v"} and set(a[2]) == set(b[2]) and len(a[2]) >= (14 if with_ffn else 10)
        for n in a[1]:
            close(b[1][n], a[1][n], rtol=3e-4)
        # (the key projection's bias gradient is zero up to round-off - softmax is shift invariant - so a tensor's own scale is not a
        # yardstick for it: at least 1 % of the largest parameter gradient)
        top = max(float(t.abs().max()) for t in a[2].values())
        for n in a[2]:
            close(b[2][n], a[2][n], atol=3e-4 * max(float(a[2][n].abs().max()), 1e-2 * top))
    finally:
        T.AttentionBlock.dropout_rate, T.FeedForwardBlock.dropout_rate = keep
        ops.ROWCHAIN[0] = True
