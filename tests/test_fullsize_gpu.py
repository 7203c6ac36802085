"""BASELINE.json configs[1] shapes on the GPU: 640x640, ResNet-50, 6 enc + 6 dec, d=256 h=8, 100
queries, COCO-80 (C=82, A=3, attribute_weight=0), M=100.

* batch 2: full comparison with the CPU oracle (the oracle needs ~2 s per step at this size).
* batch 16 (the benchmark's per-GPU batch): size-independent properties - the assignment is a valid
  minimum-cost matching (checked against scipy on the device's own cost matrix, bit-exact), losses
  and all gradients are finite, the step is deterministic, one optimizer step lowers the loss."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from _close import assert_elementwise, assert_logits, check_predictions

pytestmark = pytest.mark.gpu


def build(dropout=0.0):
    from boosted_detr_amd import parameters, transformers
    from boosted_detr_amd.model import DETR
    transformers.AttentionBlock.dropout_rate = dropout
    transformers.FeedForwardBlock.dropout_rate = dropout
    return DETR(num_object_preds=100, image_size=(640, 640), num_encoder_blocks=6, num_encoder_heads=8, encoder_dim=256,
                num_decoder_blocks=6, num_decoder_heads=8, decoder_dim=256, num_panoptic_heads=1, panoptic_dim=32,
                vocab_dict=parameters.COCO_VOCAB, attribute_weight=0.0)


def test_config2_batch2_matches_oracle(cuda):
    from oracle import detr_oracle as O
    cfg = O.CONFIG2
    batch = O.make_batch(cfg, 2, 100, seed=4321, num_objects=[9, 31])
    params = O.make_params(cfg, seed=0)
    model = build()
    model.forward_backward(batch)
    model.set_weights_dict(params)
    y = model.forward_backward(batch)
    torch.cuda.synchronize()
    out, grads = O.train_step_grads(cfg, params, batch, dtype=torch.float32)
    out64, g64 = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    cat = y[0].cpu().numpy()
    # probabilities, boxes and the three heads' pre-activation logits: every element within 1e-3 of the fp64 oracle (tests/_close.py)
    check_predictions((model.CategoryPredictionHead, model.AttributePredictionHead, model.BoxPredictionHead), y, out64)
    assert np.array_equal(cat.argmax(-1), out.cat_preds.detach().numpy().argmax(-1))       # class ids bit-exact
    match = model.loss_fn.last_match.cpu().numpy()
    want = -np.ones_like(match)
    for b, (r, c) in enumerate(out.loss.matches):
        want[b, r] = c
    assert np.array_equal(match, want)                                                      # match indices bit-exact
    logs = model.logs_to_host(model.step_logs())
    ref = float(out.loss_vector.detach().double().mean())
    assert abs(logs["loss"] - ref) <= 1e-3 * abs(ref)
    # gradient direction of the big tensors (element-wise comparison is not meaningful)
    for name in ("EncoderBackbone/resnet50/conv3_block2_2_conv/kernel", "ImageEncoderAttention/EncoderBlock_3/FeedForwardBlock/DenseRelu/kernel",
                 "DecoderBlock_5/JointAttentionBlock/AttentionLayer/ValueProjection/kernel", "CategoryPredictionHead/DenseLogits/kernel"):
        v = [x for x in model.variables if x.name == name][0]
        g, w = v.grad_numpy().astype(np.float64).ravel(), grads[name].astype(np.float64).ravel()
        cos = g @ w / (np.linalg.norm(g) * np.linalg.norm(w))
        assert cos > 0.995, (name, cos)
    # every trainable tensor against the fp64 oracle (relative L2, next to the fp32 oracle's own error)
    from test_model_gpu import check_grads
    check_grads(model, cfg, params, batch, g32=grads, g64=g64)
    # The headline's gradient arithmetic (bf16 pairs, 2^-18 per product) against the reference's (exact fp32 products), per tensor, against
    # the fp64 oracle: both backward passes start from the SAME saved forward (same activations, ReLU masks, match), so the difference is
    # the gradient products alone.  Bound: the split policy's error is at most twice the fp32 policy's (+ 1e-4 of the
    # tensor's norm: below that both are round-off of different summation orders).
    model.forward_backward(batch, keep_tape=True)
    model.replay_backward("split")
    g_split = {v.name: v.grad_numpy().astype(np.float64) for v in model.trainable_variables}
    model.replay_backward("fp32")
    g_fp32 = {v.name: v.grad_numpy().astype(np.float64) for v in model.trainable_variables}
    model.replay_backward("bf16x6")            # three bf16 terms, six products: the fp32-grade arithmetic on the 16-bit MFMA
    g_x6 = {v.name: v.grad_numpy().astype(np.float64) for v in model.trainable_variables}
    model._kept_tape = None
    gmax = max(np.abs(g).max() for g in g64.values())
    rows, rows6 = [], []
    for name, ref in g64.items():
        ref = np.asarray(ref, np.float64)
        if name not in g_split or np.abs(ref).max() < 1e-6 * gmax:
            continue
        nrm = np.linalg.norm(ref)
        e_s, e_f = np.linalg.norm(g_split[name].reshape(ref.shape) - ref) / nrm, np.linalg.norm(g_fp32[name].reshape(ref.shape) - ref) / nrm
        rows.append((e_s / max(e_f, 1e-12), e_s, e_f, name))
        assert e_s <= 2.0 * e_f + 1e-4, (name, e_s, e_f)
        e_6 = np.linalg.norm(g_x6[name].reshape(ref.shape) - ref) / nrm
        rows6.append((e_6 / max(e_f, 1e-12), e_6, e_f, name))
        assert e_6 <= 1.5 * e_f + 2e-6, (name, e_6, e_f)      # bf16x6: indistinguishable from exact fp32 products
    rows.sort(reverse=True)
    assert len(rows) > 250, len(rows)
    rows6.sort(reverse=True)
    print(f"bf16x6 / fp32 policy: worst ratio {rows6[0][0]:.2f} ({rows6[0][3]}: {rows6[0][1]:.2e} vs {rows6[0][2]:.2e}), median {rows6[len(rows6) // 2][0]:.2f}")
    over = [r for r in rows if r[0] > 2.0]
    print(f"gradient error vs fp64, split / fp32 policy, {len(rows)} tensors: worst ratio {rows[0][0]:.2f} ({rows[0][3]}: {rows[0][1]:.2e} vs {rows[0][2]:.2e}), "
          f"median ratio {rows[len(rows) // 2][0]:.2f}; {len(over)} tensors above 2x, their largest split error {max([r[1] for r in over], default=0.0):.2e}; "
          f"largest error overall: split {max(r[1] for r in rows):.2e}, fp32 {max(r[2] for r in rows):.2e}")
    # (measured, round 4: the tensors above 2x are all at errors below 1e-5 of their norm - the bound holds with a floor of 2e-5)
    assert all(r[1] <= 2.0 * r[2] + 2e-5 for r in rows), [r for r in rows if r[1] > 2.0 * r[2] + 2e-5][:3]


def test_config2_batch16_properties(cuda):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from boosted_detr_amd.training import SGD
    host = bench.make_batch(16, 640, 640, 100, 82, seed=1234)
    host["num_objects"][0] = 93                      # COCO's maximum object count
    host["category"][0, :93] = np.random.default_rng(0).integers(2, 82, 93)
    host["bbox"][0, :93] = np.random.default_rng(1).uniform(0.05, 0.5, (93, 4)).astype(np.float32)
    model = build(dropout=0.1)
    model.compile(optimizer=SGD(1e-3, momentum=0.9, nesterov=True, clipnorm=0.1))
    model.forward_backward(host)
    y1 = [t.cpu().numpy() for t in model.forward_backward(host)]
    cost = model.loss_fn.last_cost.cpu().numpy()
    match = model.loss_fn.last_match.cpu().numpy()
    nobj = host["num_objects"]
    for b in range(16):
        r, c = linear_sum_assignment(cost[b, :nobj[b], :])
        assert np.array_equal(match[b, r], c), b                       # bit-exact vs scipy on the same fp32 costs
        assert (match[b, nobj[b]:] == -1).all()
        assert len(set(c.tolist())) == nobj[b]                         # a matching: distinct predictions
    logs = model.logs_to_host(model.step_logs())
    assert all(np.isfinite(v) for v in logs.values()), logs
    for v in model.trainable_variables:
        assert v.grad is not None and bool(torch.isfinite(v.grad).all()), v.name
    y2 = [t.cpu().numpy() for t in model.forward_backward(host)]
    assert all(np.array_equal(a, b) for a, b in zip(y1, y2))           # deterministic forward (same dropout seed)
    ls = [model.logs_to_host(model.train_step(host))["loss"] for _ in range(10)]
    assert np.mean(ls[6:]) < np.mean(ls[:3]), ls                       # the optimizer descends (dropout noise: compare means, not single steps)


def test_config3_boosted_fashionpedia_batch2_matches_oracle(cuda):
    """BASELINE.json configs[2]: the config-2 shapes + the attribute head (Fashionpedia sizes: 46 categories,
    294 attributes, attribute_weight 1) and the BoostedDETR stack of 3 weak learners, against the CPU oracle."""
    from boosted_detr_amd import parameters, transformers
    from boosted_detr_amd.boosted_model import BoostedDETR
    from oracle import detr_oracle as O
    cfg = O.Config(image_size=(640, 640), num_object_preds=100, num_encoder_blocks=6, num_decoder_blocks=3, boosted=True)
    batch = O.make_batch(cfg, 2, 40, seed=99, num_objects=[11, 37])
    params = O.make_params(cfg, seed=1)
    transformers.AttentionBlock.dropout_rate = 0.0
    transformers.FeedForwardBlock.dropout_rate = 0.0
    model = BoostedDETR(num_object_preds=100, image_size=(640, 640), num_encoder_blocks=6, num_encoder_heads=8, encoder_dim=256,
                        num_decoder_blocks=3, num_decoder_heads=8, decoder_dim=256, num_panoptic_heads=1, panoptic_dim=32,
                        vocab_dict=parameters.synthetic_vocab(46, 294), attribute_weight=1.0)
    model.forward_backward(batch)
    model.set_weights_dict(params)
    y = model.forward_backward(batch)
    torch.cuda.synchronize()
    out, grads = O.train_step_grads(cfg, params, batch, dtype=torch.float32)
    out64, g64 = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    cat = y[0].cpu().numpy()
    for name, got, want in zip(("category", "attribute", "box"), y, (out64.cat_preds, out64.attribute_preds, out64.box_preds)):
        assert_elementwise(got.cpu().numpy(), want.detach().numpy(), name)               # cumulative predictions, every element, vs fp64
    for i in range(3):                                                                   # each weak learner's own logits
        for head, kind in zip((model.CategoryBlocks[i], model.AttributeBlocks[i], model.BoxBlocks[i]), ("Category", "Attribute", "Box")):
            key = f"{kind}PredictionHead_{i}/logits"
            assert_logits(head.last_logits.cpu().numpy(), out64.probes[key].detach().numpy(), key)
    assert np.array_equal(cat.argmax(-1), out.cat_preds.detach().numpy().argmax(-1))       # class ids bit-exact
    match = model.loss_fn.last_match.cpu().numpy()                                          # last learner's assignment
    want = -np.ones_like(match)
    for b, (r, c) in enumerate(out.learner_losses[-1].matches):
        want[b, r] = c
    assert np.array_equal(match, want)
    logs = model.logs_to_host(model.step_logs())
    ref = float(out.loss_vector.detach().double().mean())
    assert abs(logs["loss"] - ref) <= 1e-3 * abs(ref)
    # every trainable tensor against the fp64 oracle (relative L2 by grad_report's criterion)
    from test_model_gpu import check_grads
    check_grads(model, cfg, params, batch, g32=grads, g64=g64)


def test_config5_resnet101_1333x800_300_queries_batch1_matches_oracle(cuda):
    """BASELINE.json configs[4]'s detection path at full size: one 1333x800 image (odd feature-map sizes at every stride:
    667, 334, 167, 84, 42 x 400 ... 25), ResNet-101, 6 + 6 layers, 300 queries, COCO-80 - forward outputs within 1e-3 of the
    CPU oracle element by element (logits too), class ids and match indices bit-exact, loss within 1e-3, every gradient tensor
    by grad_report's relative-L2 criterion."""
    from boosted_detr_amd import parameters, transformers
    from boosted_detr_amd.backbone import RESNET101_STAGES
    from boosted_detr_amd.model import DETR
    from oracle import detr_oracle as O
    size = (1333, 800)
    cfg = O.Config(image_size=size, num_object_preds=300, num_encoder_blocks=6, num_decoder_blocks=6, num_categories=82, num_attributes=3,
                   attribute_weight=0.0, stages=RESNET101_STAGES)
    batch = O.make_batch(cfg, 1, 100, seed=77, num_objects=[23])
    params = O.make_params(cfg, seed=2)
    transformers.AttentionBlock.dropout_rate = 0.0
    transformers.FeedForwardBlock.dropout_rate = 0.0
    model = DETR(num_object_preds=300, image_size=size, num_encoder_blocks=6, num_encoder_heads=8, encoder_dim=256, num_decoder_blocks=6,
                 num_decoder_heads=8, decoder_dim=256, num_panoptic_heads=1, panoptic_dim=32, vocab_dict=parameters.COCO_VOCAB,
                 attribute_weight=0.0, backbone_name="ResNet101")
    ren = lambda k: k.replace("EncoderBackbone/resnet50/", "EncoderBackbone/resnet101/")
    model.forward_backward(batch)
    model.set_weights_dict({ren(k): v for k, v in params.items()})
    y = model.forward_backward(batch)
    torch.cuda.synchronize()
    out, grads = O.train_step_grads(cfg, params, batch, dtype=torch.float32)
    out64, g64 = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    check_predictions((model.CategoryPredictionHead, model.AttributePredictionHead, model.BoxPredictionHead), y, out64)
    assert np.array_equal(y[0].cpu().numpy().argmax(-1), out.cat_preds.detach().numpy().argmax(-1))
    match = model.loss_fn.last_match.cpu().numpy()
    want = -np.ones_like(match)
    for b, (r, c) in enumerate(out.loss.matches):
        want[b, r] = c
    assert np.array_equal(match, want)
    logs = model.logs_to_host(model.step_logs())
    ref = float(out.loss_vector.detach().double().mean())
    assert abs(logs["loss"] - ref) <= 1e-3 * abs(ref)
    # every trainable tensor against the fp64 oracle (the oracle keeps the scope name "resnet50" for every stage list)
    from test_model_gpu import check_grads
    check_grads(model, cfg, params, batch, rename=ren, g32=grads, g64=g64)
