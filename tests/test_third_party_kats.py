"""Published known-answer vectors of the third-party primitives behind the set criterion
(tests/golden/third_party_kats.json: TFA giou_loss, TFA SigmoidFocalCrossEntropy, Keras BinaryCrossentropy /
MeanSquaredError, scipy linear_sum_assignment docstring examples).

CPU: the oracle's restatements reproduce them.  GPU: bdetr_cost_matrix / bdetr_lsa / bdetr_set_loss reproduce them
through the C ABI (reference call sites: losses_and_metrics.py:14-23,44-72,133-150,242)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import detr_oracle as O
from oracle import lsap

KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "third_party_kats.json")))


def tf_to_coco(b):
    """[ymin, xmin, ymax, xmax] -> the reference's COCO format [xmin, ymin, w, h] (inverse of coco_to_tf, 59-66)."""
    b = np.asarray(b, np.float64)
    return np.stack([b[:, 1], b[:, 0], b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]], -1)


# --------------------------------------------------------------------------- CPU: the oracle
def test_oracle_giou_matches_tfa_docstring():
    k = KATS["tfa_giou_loss"]
    for dt, tol in ((torch.float32, 2e-7), (torch.float64, 5e-8)):
        b1, b2 = torch.tensor(k["boxes1"], dtype=dt), torch.tensor(k["boxes2"], dtype=dt)
        got = (1.0 - O.tfa_giou(b1, b2, "giou")).numpy()
        assert np.abs(got - np.array(k["giou_loss"])).max() <= tol * 2, got
    # and through coco_to_tf + the pairwise box cost (2 x giou_loss + 5 x mse(10 x boxes)): the diagonal
    t, p = torch.tensor(tf_to_coco(k["boxes1"])), torch.tensor(tf_to_coco(k["boxes2"]))
    cost = O.box_cost(t[None], p[None])[0]
    l2 = ((10 * O.coco_to_tf(t) - 10 * O.coco_to_tf(p)) ** 2).mean(-1)
    assert np.abs(((torch.diagonal(cost) - 5 * l2) / 2).numpy() - np.array(k["giou_loss"])).max() < 1e-6


def test_oracle_focal_matches_tfa_docstring():
    k = KATS["tfa_sigmoid_focal_crossentropy"]
    yt, yp = torch.tensor(k["y_true"], dtype=torch.float64), torch.tensor(k["y_pred"], dtype=torch.float64)
    # attribute_cost is pairwise [B,M,A] x [B,N,A] with A = 1 here; the docstring pairs are its diagonal
    got = torch.diagonal(O.attribute_cost(yt[None], yp[None])[0]).numpy()
    want = np.array(k["loss"])
    assert (np.abs(got - want) <= 1e-5 * want).all(), got      # the published triple is an fp32 result


def test_oracle_bce_and_mse_match_keras_docstrings():
    k = KATS["keras_binary_crossentropy"]
    got = O.keras_bce(torch.tensor(k["y_true"], dtype=torch.float64), torch.tensor(k["y_pred"], dtype=torch.float64)).numpy()
    assert np.array_equal(np.round(got, k["loss_decimals"]), np.array(k["loss"])), got
    m = KATS["keras_mean_squared_error"]
    yt, yp = torch.tensor(m["y_true"]), torch.tensor(m["y_pred"])
    assert np.array_equal(((yt - yp) ** 2).mean(-1).numpy(), np.array(m["loss"], np.float32))


def test_lsap_oracle_matches_scipy_docstring():
    k = KATS["scipy_linear_sum_assignment"]
    cost = np.array(k["cost"], np.float32)
    rows, cols = lsap.linear_sum_assignment_f32(cost)
    assert rows.tolist() == k["row_ind"] and cols.tolist() == k["col_ind"] and cost[rows, cols].sum() == k["total"]


# --------------------------------------------------------------------------- GPU: the C ABI
@pytest.mark.gpu
def test_gpu_cost_matrix_and_lsa_match_published_vectors(cuda):
    from boosted_detr_amd import kernels as kk
    dev = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).cuda()
    g, f = KATS["tfa_giou_loss"], KATS["tfa_sigmoid_focal_crossentropy"]
    # one image, M = 3 true objects, N = 3 predictions, C = 4 categories, A = 1 attribute: the diagonal of the
    # pairwise components carries the published pairs (the box example has two pairs; the third repeats pair 0)
    M = N = 3
    bt = np.concatenate([tf_to_coco(g["boxes1"]), tf_to_coco(g["boxes1"])[:1]])[None].astype(np.float32)
    bp = np.concatenate([tf_to_coco(g["boxes2"]), tf_to_coco(g["boxes2"])[:1]])[None].astype(np.float32)
    att_hot = np.array(f["y_true"], np.float32)[None]                  # [1,3,1]
    att_pred = np.array(f["y_pred"], np.float32)[None]                 # [1,3,1]
    cat_pred = np.full((1, N, 4), 0.25, np.float32)
    cat_ids = np.array([[2, 3, 2]], np.int32)
    d = kk.loss_desc(1, M, N, 4, 1, 1000.0, 1.0, 1.0, 100.0)
    cost, comps = kk.cost_matrix(d, dev(cat_pred), dev(att_pred), dev(bp), dev(cat_ids, torch.int32), dev(att_hot), dev(bt),
                                 dev(np.array([3], np.int32), torch.int32), components=True)
    a_comp = np.diagonal(comps[1].cpu().numpy()[0]).astype(np.float64)
    want = np.array(f["loss"])
    assert (np.abs(a_comp - want) <= 2e-5 * want).all(), a_comp                  # TFA focal docstring triple (fp32 kernel)
    b_comp = np.diagonal(comps[2].cpu().numpy()[0]).astype(np.float64)[:2]
    ct, cp = O.coco_to_tf(torch.tensor(bt[0, :2], dtype=torch.float64)), O.coco_to_tf(torch.tensor(bp[0, :2], dtype=torch.float64))
    l2 = ((10 * ct - 10 * cp) ** 2).mean(-1).numpy()
    giou_loss = (b_comp - 5 * l2) / 2
    # the fp32 cost carries the 5 * l2 term (up to ~2e4 for the far-apart pair): allow its fp32 round-off
    assert (np.abs(giou_loss - np.array(g["giou_loss"])) <= 2e-7 * np.abs(b_comp) + 1e-6).all(), giou_loss   # TFA giou_loss docstring pair
    # scipy docstring assignment through the on-GPU solver
    s = KATS["scipy_linear_sum_assignment"]
    match = kk.lsa(dev(np.array(s["cost"], np.float32)[None]), dev(np.array([3], np.int32), torch.int32)).cpu().numpy()[0]
    assert match.tolist() == s["col_ind"]


@pytest.mark.gpu
def test_gpu_existence_loss_matches_keras_bce_docstring(cuda):
    """exist = 100 * BCE(1 - assigned, clip(p[..., 0:1])) / (1 + N) (losses_and_metrics.py:139-150): the four (y, p)
    pairs of the Keras BinaryCrossentropy docstring as four queries of one image."""
    from boosted_detr_amd import kernels as kk
    dev = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).cuda()
    k = KATS["keras_binary_crossentropy"]
    y = np.array(k["y_true"]).ravel()                  # y = 1 - assigned
    p0 = np.array(k["y_pred"]).ravel()
    N, M, C = 4, 4, 3
    cat_pred = np.stack([p0, (1 - p0) * 0.5, (1 - p0) * 0.5], -1)[None].astype(np.float32)
    assigned = 1 - y                                   # queries 0, 2, 3 are matched
    match = -np.ones((1, M), np.int32)
    match[0, :3] = np.nonzero(assigned)[0]
    d = kk.loss_desc(1, M, N, C, 1, 1000.0, 0.0, 1.0, 100.0)
    box = np.full((1, N, 4), 0.25, np.float32)
    losses, _, _, _ = kk.set_loss(d, dev(cat_pred), dev(np.full((1, N, 1), 0.5, np.float32)), dev(box), dev(np.full((1, M), 2, np.int32), torch.int32),
                                  dev(np.zeros((1, M, 1), np.float32)), dev(np.full((1, M, 4), 0.25, np.float32)),
                                  dev(np.array([3], np.int32), torch.int32), dev(match, torch.int32))
    exist = float(losses.cpu().numpy()[4, 0])
    # docstring: per-row means [0.916, 0.714] -> mean over the four queries 0.815; x 100 / (1 + N)
    assert round(exist * (1 + N) / 100, 3) == round(float(np.mean(k["loss"])), 3), exist
    want = 100.0 * O.keras_bce(torch.tensor(y)[:, None], torch.tensor(p0)[:, None]).mean().item() / (1 + N)
    assert abs(exist - want) <= 2e-5 * want
