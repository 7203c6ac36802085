"""Panoptic head (PanopticAttention transformers.py:460-559 + PanopticNeck panoptic_neck.py:8-186), forward
only, against the torch-fp64 restatement oracle/panoptic_oracle.py with the layers' own (seeded Keras-initialiser) weights.
Tolerance 1e-3 relative like the rest of the path; the kernels below run on the library default policy (exact-fp32 forward)."""
import numpy as np
import pytest
import torch

from test_kernels_gpu import close, dev, rnd

pytestmark = pytest.mark.gpu


def test_panoptic_pieces(cuda):
    """csrc/panoptic.hip kernel by kernel: bilinear resize (= torch F.interpolate, half-pixel centres), channel LayerNorm +
    leaky ReLU on an odd channel count with zero padding, column copies, NHWC -> NCHW."""
    import torch.nn.functional as F
    from boosted_detr_amd import kernels as k
    x = rnd(2, 7, 9, 8, seed=1)
    for (H, W) in ((96, 96), (5, 4), (7, 9)):
        ref = F.interpolate(x.double().permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        close(k.resize_bilinear(dev(x), H, W), ref, rtol=1e-5)
    C, ld = 29, 32
    t = torch.zeros(3, 5, 5, ld); t[..., :C] = rnd(3, 5, 5, C, seed=2)
    g, b = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    ref = F.leaky_relu(F.layer_norm(t[..., :C].double(), (C,), g.double(), b.double(), 1e-3), 0.01)
    out = k.layernorm_act(dev(t), C, dev(g), dev(b), 1e-3, 0.01)
    assert out.shape[-1] == 32 and float(out[..., C:].abs().max()) == 0.0
    close(out[..., :C], ref, rtol=1e-5)
    dst = torch.zeros(3, 5, 5, 44).cuda()
    k.copy_cols(out, C, dst, 10)
    assert torch.equal(dst[..., 10:10 + C], out[..., :C]) and float(dst[..., :10].abs().max()) == 0.0
    assert torch.equal(k.nhwc_to_nchw(out, C), out[..., :C].permute(0, 3, 1, 2).reshape(3, C, 25))


def test_panoptic_attention_and_neck_match_the_restatement(cuda):
    from boosted_detr_amd import panoptic_neck, transformers
    from oracle import panoptic_oracle as PO
    B, r, c, E, num_obj, heads, pdim = 2, 5, 6, 64, 48, 2, 32
    enc = rnd(B, r, c, E, seed=1)
    dec, pos = rnd(B, num_obj, 256, seed=2), rnd(B, r, c, E, seed=3)
    att = transformers.PanopticAttention(num_attention_heads=heads, hidden_dim=pdim, seed=5)
    maps = att([dev(enc), dev(dec), dev(pos)])
    assert tuple(maps.shape) == (B, r, c, num_obj, heads)
    w = {v.name.split("PanopticAttention/")[1]: torch.from_numpy(v.numpy()).double() for v in att.variables}
    want = PO.panoptic_attention(enc.double(), num_obj, heads, pdim, w)
    close(maps, want, rtol=1e-3)

    neck = panoptic_neck.PanopticNeck(seed=7)
    out = neck([maps])
    assert tuple(out.shape) == (B, num_obj, 529)                       # 23 x 23 masks per box
    wn = {v.name: torch.from_numpy(v.numpy()).double() for v in neck.variables}
    ref = PO.panoptic_neck(want, wn)
    close(out, ref, rtol=1e-3)
    # channel plan of the reference: x 2/3 down, x 3/2 up, integer division (96 -> 64 -> 42 -> 28,18 -> 12,8,5 -> 7,10,15 ...)
    assert [blk.out_channels for blk in (neck.DownscaleBlock_0, neck.DownscaleBlock_1, neck.DownscaleBlock_2, neck.DownscaleBlock_3,
                                         neck.UpscaleBlock_0, neck.UpscaleBlock_1, neck.UpscaleBlock_2, neck.UpscaleBlock_3,
                                         neck.DownscaleBlock_4)] == [64, 42, 18, 5, 15, 33, 49, 73, 75]


def test_panoptic_head_at_configs4_shapes(cuda):
    """BASELINE.json configs[4]'s mask head at ITS OWN shapes: one 800x1333 image -> 25 x 42 feature map (1,050 tokens), d = 256,
    300 queries, num_panoptic_heads = 1, panoptic_dim = 32 (parameters.py:160-178) -> a 96 x 96 x 300-channel U-Net -> 300 masks
    of 23 x 23.  Against the fp64 restatement, EVERY element within 1e-3 of itself or of the tensor's RMS (tests/_close.py:
    LayerNorm outputs and mask logits cross zero); channel plan 300 -> 200 -> 133 -> 88, 58 -> 38, 25, 16 -> 24, 36, 54 ..."""
    import time
    from _close import assert_logits
    from boosted_detr_amd import panoptic_neck, transformers
    from oracle import panoptic_oracle as PO
    B, r, c, E, num_obj, heads, pdim = 1, 25, 42, 256, 300, 1, 32
    enc = rnd(B, r, c, E, seed=11)
    dec, pos = rnd(B, num_obj, 256, seed=12), rnd(B, r, c, E, seed=13)
    att = transformers.PanopticAttention(num_attention_heads=heads, hidden_dim=pdim, seed=5)
    maps = att([dev(enc), dev(dec), dev(pos)])
    assert tuple(maps.shape) == (B, r, c, num_obj, heads)
    w = {v.name.split("PanopticAttention/")[1]: torch.from_numpy(v.numpy()).double() for v in att.variables}
    want = PO.panoptic_attention(enc.double(), num_obj, heads, pdim, w)
    rep_a = assert_logits(maps.cpu().numpy(), want.numpy(), "PanopticAttention maps")

    neck = panoptic_neck.PanopticNeck(seed=7)
    out = neck([maps])
    assert tuple(out.shape) == (B, num_obj, 529)
    wn = {v.name: torch.from_numpy(v.numpy()).double() for v in neck.variables}
    t0 = time.time()
    ref = PO.panoptic_neck(want, wn)
    rep_n = assert_logits(out.cpu().numpy(), ref.numpy(), "PanopticNeck masks")
    print(f"configs[4] panoptic head vs fp64: attention max|d| {rep_a['max_abs_err']:.2e}, masks max|d| {rep_n['max_abs_err']:.2e} "
          f"(mask RMS {float(ref.pow(2).mean().sqrt()):.3f}; oracle {time.time() - t0:.1f} s)")
    assert [blk.out_channels for blk in (neck.DownscaleBlock_0, neck.DownscaleBlock_1, neck.DownscaleBlock_2, neck.DownscaleBlock_3,
                                         neck.UpscaleBlock_0, neck.UpscaleBlock_1, neck.UpscaleBlock_2, neck.UpscaleBlock_3,
                                         neck.DownscaleBlock_4)] == [200, 133, 58, 16, 54, 121, 181, 252, 254]
    b2 = neck([att([dev(enc[:1].repeat(2, 1, 1, 1)), dev(dec[:1].repeat(2, 1, 1)), dev(pos[:1].repeat(2, 1, 1, 1))])])
    scale = float(out.abs().max())
    assert float((b2[0] - out[0]).abs().max()) <= 1e-6 * scale and float((b2[1] - out[0]).abs().max()) <= 1e-6 * scale    # batch rows are independent
