"""GPU parity of the set criterion (cost matrix -> on-GPU LSA -> loss + sparse backward).

Match indices are compared BIT-EXACT against the reference's real dependency
(scipy.optimize.linear_sum_assignment, called as losses_and_metrics.py:240-243 does) on the
same fp32 cost matrix; costs / losses / gradients against the CPU oracle within 1e-5 relative."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).contiguous().cuda()


def scipy_match(cost, nobj):
    B, M, N = cost.shape
    out = -np.ones((B, M), np.int32)
    for b in range(B):
        r, c = linear_sum_assignment(cost[b, :nobj[b], :])
        out[b, r] = c
    return out


def gpu_match(cost, nobj):
    from boosted_detr_amd import kernels as k
    m = k.lsa(dev(cost), dev(nobj, torch.int32))
    torch.cuda.synchronize()
    return m.cpu().numpy()


@pytest.mark.parametrize("M,N", [(20, 50), (100, 100), (93, 100), (30, 300), (120, 100)])
def test_lsa_random(cuda, M, N):
    rng = np.random.default_rng(M * 1000 + N)
    B = 24
    cost = (rng.random((B, M, N)) * 10).astype(np.float32)
    nobj = rng.integers(0, M + 1, size=B).astype(np.int32)
    nobj[0], nobj[1], nobj[2] = 0, 1, M
    got, want = gpu_match(cost, nobj), scipy_match(cost, nobj)
    assert np.array_equal(got, want)


def test_lsa_ties_and_constant(cuda):
    """adversarial ties: constant matrices, small-integer costs, duplicated columns/rows."""
    rng = np.random.default_rng(7)
    B, M, N = 32, 40, 64
    cost = np.zeros((B, M, N), np.float32)
    cost[0] = 0.0
    cost[1] = 3.5
    for b in range(2, 12):
        cost[b] = rng.integers(0, 3, size=(M, N))
    for b in range(12, 20):
        base = rng.random((M, 8)).astype(np.float32)
        cost[b] = np.tile(base, (1, N // 8))                      # duplicated columns
    for b in range(20, 26):
        base = rng.integers(0, 4, size=(5, N)).astype(np.float32)
        cost[b] = np.tile(base, (M // 5, 1))                      # duplicated rows
    for b in range(26, 32):
        cost[b] = np.round(rng.random((M, N)) * 4) / 4
    nobj = rng.integers(1, M + 1, size=B).astype(np.int32)
    nobj[:4] = M
    got, want = gpu_match(cost, nobj), scipy_match(cost, nobj)
    assert np.array_equal(got, want)


def test_lsa_tall_transposed(cuda):
    rng = np.random.default_rng(11)
    B, M, N = 8, 80, 30          # n_i > N -> scipy transposes
    cost = rng.random((B, M, N)).astype(np.float32)
    cost[3] = np.round(cost[3] * 3)
    nobj = np.array([80, 31, 30, 80, 45, 29, 1, 60], np.int32)
    got, want = gpu_match(cost, nobj), scipy_match(cost, nobj)
    assert np.array_equal(got, want)


def test_lsa_inf_and_invalid(cuda):
    B, M, N = 4, 6, 8
    rng = np.random.default_rng(3)
    cost = rng.random((B, M, N)).astype(np.float32)
    cost[0, 2, :5] = np.inf                  # feasible with +inf entries
    cost[1, 1, :] = np.inf                   # infeasible: scipy raises ValueError
    cost[2, 0, 0] = np.nan                   # invalid: scipy raises ValueError
    nobj = np.full(B, M, np.int32)
    got = gpu_match(cost, nobj)
    r, c = linear_sum_assignment(cost[0])
    assert np.array_equal(got[0][r], c)
    for b in (1, 2):
        with pytest.raises(ValueError):
            linear_sum_assignment(cost[b])
        assert (got[b] == -1).all()          # the host maps an all -1 row with n_i>0 to ValueError
    r, c = linear_sum_assignment(cost[3])
    assert np.array_equal(got[3][r], c)


def _oracle_loss(cfg, batch, cat, att, box, dtype=torch.float64):
    from oracle import detr_oracle as O
    cat_t, att_t = O.tokens_to_hot(batch["category"], batch["attribute"], cfg.num_categories, cfg.num_attributes, dtype)
    catp = torch.from_numpy(cat).to(dtype).requires_grad_(True)
    attp = torch.from_numpy(att).to(dtype).requires_grad_(True)
    boxp = torch.from_numpy(box).to(dtype).requires_grad_(True)
    lo = O.matching_loss(cat_t, att_t, torch.from_numpy(batch["bbox"]).to(dtype), batch["num_objects"], catp, attp, boxp,
                         attribute_weight=cfg.attribute_weight)
    lo.total.sum().backward()
    return lo, catp.grad, attp.grad, boxp.grad


@pytest.mark.parametrize("C,A,att_w,M,N", [(48, 296, 1.0, 20, 50), (82, 3, 0.0, 100, 100), (10, 6, 100.0, 12, 16)])
def test_cost_lsa_loss_vs_oracle(cuda, C, A, att_w, M, N):
    from boosted_detr_amd import kernels as k
    from oracle import detr_oracle as O
    cfg = O.Config(num_object_preds=N, num_categories=C, num_attributes=A, attribute_weight=att_w)
    B = 4
    batch = O.make_batch(cfg, B, M, seed=5, image_hw=(8, 8))
    rng = np.random.default_rng(1)
    logits = rng.standard_normal((B, N, C)).astype(np.float32) * 2
    cat = torch.softmax(torch.from_numpy(logits), -1).numpy()
    att = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, N, A)).astype(np.float32) * 3)).numpy()
    box = (rng.random((B, N, 4)) * np.array([0.7, 0.7, 0.5, 0.5])).astype(np.float32)
    box[:, ::7, 2] *= -1.0      # degenerate predicted widths exercise the max(0, .) branches
    lo, g_cat, g_att, g_box = _oracle_loss(cfg, batch, cat, att, box)

    att_hot = torch.nn.functional.one_hot(torch.from_numpy(batch["attribute"].astype(np.int64)), A).amax(2).float()
    d = k.loss_desc(B, M, N, C, A, 1000.0, att_w, 1.0, 100.0)
    args = (dev(cat), dev(att), dev(box), dev(batch["category"], torch.int32), att_hot.cuda(), dev(batch["bbox"]),
            dev(batch["num_objects"], torch.int32))
    cost, comps = k.cost_matrix(d, *args, components=True)
    nobj = batch["num_objects"]
    rowmask = (np.arange(M)[None, :] < nobj[:, None])[:, :, None]
    ref_cost = lo.cost_total.detach().numpy() * rowmask
    err = np.abs(cost.cpu().numpy() - ref_cost).max() / np.abs(ref_cost).max()
    assert err < 1e-5, err
    for got, want in zip(comps, lo.cost_components):
        w = want.detach().numpy() * rowmask
        assert np.abs(got.cpu().numpy() - w).max() <= 1e-5 * (np.abs(w).max() + 1e-12)

    match = k.lsa(cost, args[-1])
    want_match = scipy_match(cost.cpu().numpy(), nobj)
    assert np.array_equal(match.cpu().numpy(), want_match)                 # bit-exact vs scipy on the same costs
    oracle_match = -np.ones((B, M), np.int32)
    for b, (r, c) in enumerate(lo.matches):
        oracle_match[b, r] = c
    assert np.array_equal(match.cpu().numpy(), oracle_match)               # and equal to the oracle's assignment
    mask = k.match_to_mask(match, N).cpu().numpy()
    assert np.array_equal(mask, lo.mask.numpy())

    losses, d_cat, d_att, d_box = k.set_loss(d, *args, match)
    want = torch.stack([lo.total, lo.category, lo.attribute, lo.box, lo.exist, lo.iou]).detach().numpy()
    got = losses.cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (got, want)
    for gg, ww in ((d_cat, g_cat), (d_att, g_att), (d_box, g_box)):
        ww = ww.numpy()
        assert np.abs(gg.cpu().numpy() - ww).max() <= 5e-5 * (np.abs(ww).max() + 1e-12)
