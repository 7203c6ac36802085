"""Forward throughput of the panoptic head (PanopticAttention + PanopticNeck) at BASELINE configs[4] shapes:
800x1333 input -> 25x42 feature map, d=256, 300 queries, num_panoptic_heads=1, panoptic_dim=32 (parameters.py:160-178).
The reference never wires the head into a model, so this is the head alone on synthetic encoder / decoder features.
Usage: python tools/panoptic_bench.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import panoptic_neck, transformers

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r, c, E, N = 25, 42, 256, 300
enc, pos = torch.randn(B, r, c, E, device="cuda"), torch.randn(B, r, c, E, device="cuda")
dec = torch.randn(B, N, 256, device="cuda")
att = transformers.PanopticAttention(num_attention_heads=1, hidden_dim=32)
neck = panoptic_neck.PanopticNeck()


def head():
    return neck([att([enc, dec, pos])])


out = head()
assert tuple(out.shape) == (B, N, 529) and bool(torch.isfinite(out).all())
for _ in range(2):
    head()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
e0.record()
for _ in range(iters):
    head()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"panoptic head forward, batch {B}, {r}x{c} map, {N} queries: {ms:.2f} ms = {B / ms * 1e3:.0f} images/s")
