"""GPU edge cases of the hot path that the reference's domain has: images without objects, more
objects than predictions (scipy transposes), inputs that need the bilinear resize, the Keras-style
surface (summary / get_config / parameter counts)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make(cfg, boosted=False):
    from boosted_detr_amd import parameters, transformers
    from boosted_detr_amd.boosted_model import BoostedDETR
    from boosted_detr_amd.model import DETR
    transformers.AttentionBlock.dropout_rate = 0.0
    transformers.FeedForwardBlock.dropout_rate = 0.0
    cls = BoostedDETR if boosted else DETR
    return cls(num_object_preds=cfg.num_object_preds, image_size=cfg.image_size, num_encoder_blocks=cfg.num_encoder_blocks,
               num_encoder_heads=8, encoder_dim=256, num_decoder_blocks=cfg.num_decoder_blocks, num_decoder_heads=8, decoder_dim=256,
               num_panoptic_heads=1, panoptic_dim=32, vocab_dict=parameters.synthetic_vocab(cfg.num_categories - 2, cfg.num_attributes - 2),
               attribute_weight=cfg.attribute_weight)


def compare(cfg, batch, boosted=False):
    from oracle import detr_oracle as O
    params = O.make_params(cfg, seed=5)
    model = make(cfg, boosted)
    model.forward_backward(batch)
    model.set_weights_dict(params)
    y = model.forward_backward(batch)
    out, grads = O.train_step_grads(cfg, params, batch, dtype=torch.float64)
    for got, want in zip(y, (out.cat_preds, out.attribute_preds, out.box_preds)):
        g, w = got.cpu().numpy().astype(np.float64), want.detach().numpy()
        assert np.abs(g - w).max() <= 1e-3 * np.abs(w).max()
    match = model.loss_fn.last_match.cpu().numpy()
    want = -np.ones_like(match)
    for b, (r, c) in enumerate(out.learner_losses[-1].matches):
        want[b, r] = c
    assert np.array_equal(match, want)
    logs = model.logs_to_host(model.step_logs())
    for k, w in (("loss", out.loss_vector), ("Existence_Loss", out.metrics["Existence_Loss"]), ("IOU", out.metrics["IOU"])):
        w = float(w.detach().double().mean())
        assert abs(logs[k] - w) <= 1e-3 * abs(w) + 1e-7, (k, logs[k], w)
    return model, out


def test_image_without_objects(cuda):
    from oracle import detr_oracle as O
    cfg = O.Config(image_size=(64, 64), num_object_preds=10, num_categories=12, num_attributes=6)
    batch = O.make_batch(cfg, 3, 5, seed=3, num_objects=[0, 4, 1])
    batch["bbox"][0] = -10.0
    batch["category"][0] = 0
    model, out = compare(cfg, batch)
    assert (model.loss_fn.last_match.cpu().numpy()[0] == -1).all()


def test_more_objects_than_predictions(cuda):
    """n_i > N: scipy transposes the cost matrix; only N objects get a prediction."""
    from oracle import detr_oracle as O
    cfg = O.Config(image_size=(64, 64), num_object_preds=6, num_categories=12, num_attributes=6)
    batch = O.make_batch(cfg, 2, 12, seed=4, num_objects=[12, 7])
    model, out = compare(cfg, batch)
    m = model.loss_fn.last_match.cpu().numpy()
    assert (m[0] >= 0).sum() == 6 and (m[1] >= 0).sum() == 6


def test_resized_input_and_inference_strings(cuda):
    from oracle import detr_oracle as O
    cfg = O.Config(image_size=(64, 96), num_object_preds=10, num_categories=12, num_attributes=6)
    params = O.make_params(cfg, seed=6)
    rng = np.random.default_rng(0)
    image = rng.random((2, 50, 70, 3), dtype=np.float32)         # any h,w is accepted and resized (backbone.py:44,54)
    model = make(cfg)
    category, attributes, boxes = model({"image": image}, training=False)     # build-by-first-call in inference mode
    model.set_weights_dict(params)
    category, attributes, boxes = model({"image": image}, training=False)
    ref = O.forward(O.Net(cfg, params, torch.float64), {"image": image}, training=False)
    ids, hot = O.decode_predictions(ref.cat_preds, ref.attribute_preds)
    vocab = ["<PAD>", "<OOV>"] + model.vocab_dict["category"]
    # the uint8 truncation after the resize may flip a pixel by one unit between implementations; the
    # decoded class ids must still agree wherever the top-2 margin is not microscopic
    p = ref.cat_preds.detach().numpy()
    top2 = np.sort(p, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-3
    got_ids = np.vectorize(vocab.index)(category[..., 0])
    assert np.array_equal(got_ids[clear], ids.numpy()[clear])
    assert np.abs(boxes.cpu().numpy() - ref.box_preds.detach().numpy()).max() < 2e-3
    assert category.shape == (2, 10, 1) and attributes.shape == (2, 10, 1) and isinstance(attributes[0, 0, 0], str)


def test_boosted_inference_and_surface(cuda):
    from oracle import detr_oracle as O
    cfg = O.Config(image_size=(64, 64), num_object_preds=10, num_decoder_blocks=2, num_categories=12, num_attributes=6, boosted=True)
    batch = O.make_batch(cfg, 2, 5, seed=8, num_objects=[2, 3])
    model, out = compare(cfg, batch, boosted=True)
    cat, att, box = model({"image": batch["image"]}, training=False)
    assert cat.shape == (2, 10, 1) and box.shape == (2, 10, 4)
    conf = model.get_config()
    assert conf["num_decoder_blocks"] == 2 and conf["image_size"] == (64, 64)
    assert len(model.EncoderTransformerBlocks) == 2 and len(model.CategoryBlocks) == 2     # public attributes the notebook freezes
    text = model.summary()
    assert "Total params" in text
    model.DecoderBlocks[0].FeedForwardBlock.show_summary()


def test_config2_parameter_count(cuda):
    """31,006,681 variables (30,944,345 trainable) at config 2."""
    from boosted_detr_amd import parameters
    from boosted_detr_amd.model import DETR
    import bench
    model = DETR(num_object_preds=100, image_size=(640, 640), num_encoder_blocks=6, num_encoder_heads=8, encoder_dim=256,
                 num_decoder_blocks=6, num_decoder_heads=8, decoder_dim=256, num_panoptic_heads=1, panoptic_dim=32,
                 vocab_dict=parameters.COCO_VOCAB, attribute_weight=0.0)
    host = bench.make_batch(1, 640, 640, 100, 82, seed=1)
    model(host, training=True)
    assert model.count_params() == 31_006_681
    assert sum(v.num_params for v in model.trainable_variables) == 30_944_345
    assert model.num_categories == 82 and model.num_attributes == 3


def test_operands_of_4GB_or_more_are_refused(cuda):
    """The loaders use 32-bit buffer offsets: a tensor that spans >= 4 GB must be rejected with an error, not
    read through wrapped offsets (the check runs before any memory is touched, so tiny dummy buffers suffice)."""
    import ctypes as C
    from boosted_detr_amd import _lib, kernels as k
    L = _lib.lib()
    t = torch.zeros(64, device="cuda")
    g = k.ConvGeom(700, 640, 640, 4, 64, 1, 1, 1, 0)             # 700*640*640*4 floats = 4.59 GB input
    d = g.desc()
    rc = L.bdetr_conv2d_fwd(t.data_ptr(), t.data_ptr(), None, t.data_ptr(), C.byref(d), 0, None, None, None)
    assert rc != 0 and b"4 GB" in L.bdetr_last_error()
    gd = _lib.GemmDesc(1 << 20, 64, 1024, 1, 1, t.data_ptr(), 1024, 0, 0, 1, t.data_ptr(), 1024, 0, 0, 1, t.data_ptr(), 64, 0, 0, None, 1.0, 0, 0, 1, 0)
    rc = L.bdetr_gemm(C.byref(gd), None)                          # A: 2^20 x 1024 floats = 4 GiB
    assert rc != 0 and b"4 GB" in L.bdetr_last_error()
