"""Loss trajectory of a few training steps with the pre-split (P16) conv path on and off, from identical weights."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from boosted_detr_amd import ops
from boosted_detr_amd.training import SGD
from oracle import detr_oracle as O
from test_training_gpu import small_batch, small_model

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg, batch = small_batch()
if size != 64:
    cfg = O.Config(image_size=(size, size), num_object_preds=10, num_decoder_blocks=2, num_categories=12, num_attributes=6)
    batch = O.make_batch(cfg, 2, 5, seed=9, num_objects=[2, 4])
params = O.make_params(cfg, seed=1)
for on in (False, True, False, True):
    ops.P16_ENABLED[0] = on
    m = small_model() if size == 64 else None
    if m is None:
        from boosted_detr_amd import parameters, transformers
        from boosted_detr_amd.model import DETR
        m = DETR(num_object_preds=10, image_size=(size, size), num_encoder_blocks=1, num_encoder_heads=8, encoder_dim=256, num_decoder_blocks=2,
                 num_decoder_heads=8, decoder_dim=256, num_panoptic_heads=1, panoptic_dim=32, vocab_dict=parameters.synthetic_vocab(10, 4), attribute_weight=1.0)
    m.compile(optimizer=SGD(1e-3, momentum=0.9, nesterov=True, clipnorm=0.1))
    m.forward_backward(batch)
    m.set_weights_dict(params)
    losses = []
    for step in range(8):
        losses.append(m.logs_to_host(m.train_step(batch))["loss"])
    print("P16", on, " ".join(f"{l:.4f}" for l in losses), flush=True)
