"""Generates tests/golden/config1.npz from the CPU oracle (oracle/detr_oracle.py).

The reference (TensorFlow) cannot run in the authoring container, so these vectors pin the
BUILD'S OWN restatement, not the reference ("parity unpinned", see the oracle's header): they
guard against silent drift of the oracle and give the GPU tests a committed target that does
not depend on the CPU model of the machine running them.  Matcher indices in the file were
produced by the reference's real dependency (scipy.optimize.linear_sum_assignment).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import detr_oracle as O  # noqa: E402

GRAD_PROBES = [
    "EncoderBackbone/resnet50/conv1_conv/kernel", "EncoderBackbone/resnet50/conv3_block2_2_conv/kernel",
    "EncoderBackbone/resnet50/conv5_block3_3_bn/gamma", "BackboneNeck/conv2d_downscaler/kernel",
    "ImageEncoderAttention/positional_encoding", "ImageEncoderAttention/EncoderBlock_0/SelfAttentionBlock/AttentionLayer/QueryProjection/kernel",
    "DecoderPrep/init_decoder_features", "DecoderBlock_0/JointAttentionBlock/AttentionLayer/ValueProjection/kernel",
    "DecoderBlock_0/FeedForwardBlock/LayerNorm/beta", "CategoryPredictionHead/DenseLogits/kernel",
    "AttributePredictionHead/DenseLinear/bias", "BoxPredictionHead/BoxCoords/kernel",
]


def probe_slice(t: torch.Tensor) -> np.ndarray:
    flat = t.detach().double().reshape(-1)
    idx = np.linspace(0, flat.numel() - 1, 64).astype(np.int64)
    return flat[idx].numpy()


def build(cfg, batch, tag, out):
    params = O.make_params(cfg, seed=0)
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode()); h.update(params[k].tobytes())
    out[f"{tag}/params_sha256"] = np.frombuffer(h.digest(), np.uint8)
    for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
        o, g = O.train_step_grads(cfg, params, batch, dtype=dt)
        out[f"{tag}/{name}/cat_preds"] = o.cat_preds.detach().numpy()
        out[f"{tag}/{name}/attribute_preds_slice"] = o.attribute_preds.detach().numpy()[:, :, ::16]
        out[f"{tag}/{name}/box_preds"] = o.box_preds.detach().numpy()
        out[f"{tag}/{name}/class_ids"] = o.cat_preds.detach().numpy().argmax(-1).astype(np.int64)
        out[f"{tag}/{name}/loss_vector"] = o.loss_vector.detach().numpy()
        for k, v in o.metrics.items():
            out[f"{tag}/{name}/metric/{k}"] = v.detach().numpy()
        match = -np.ones((len(o.loss.matches), batch["category"].shape[1]), np.int64)
        for b, (r, c) in enumerate(o.loss.matches):
            match[b, r] = c
        out[f"{tag}/{name}/match"] = match
        out[f"{tag}/{name}/cost_total"] = o.loss.cost_total.detach().numpy().astype(np.float32)
        for pk, pv in o.probes.items():
            out[f"{tag}/{name}/probe/{pk}"] = probe_slice(pv)
            out[f"{tag}/{name}/probe_sum/{pk}"] = np.array(float(pv.detach().double().sum()))
            if pk.endswith("/logits"):
                # the heads' pre-activation outputs in full (the attribute head's 296 columns: every 16th, like its predictions)
                a = pv.detach().numpy()
                out[f"{tag}/{name}/logits/{pk[:-7]}"] = a[:, :, ::16] if a.shape[-1] > 64 else a
        for pk in GRAD_PROBES:
            if pk in g:
                out[f"{tag}/{name}/grad_norm/{pk}"] = np.array(np.linalg.norm(g[pk].astype(np.float64)))
                out[f"{tag}/{name}/grad_slice/{pk}"] = probe_slice(torch.from_numpy(np.ascontiguousarray(g[pk])))
                if g[pk].size <= 16384:
                    # whole gradient tensors (small ones): the GPU test bounds the relative L2 error against these, not a norm
                    out[f"{tag}/{name}/grad_full/{pk}"] = g[pk].astype(np.float32)


def main():
    torch.set_num_threads(8)
    out = {}
    cfg = O.CONFIG1
    batch = O.make_batch(cfg, 2, 20, seed=1234, num_objects=[3, 7])
    build(cfg, batch, "config1", out)
    bcfg = O.Config(num_decoder_blocks=3, boosted=True)
    bbatch = O.make_batch(bcfg, 2, 20, seed=77, num_objects=[5, 2])
    build(bcfg, bbatch, "boosted3", out)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
