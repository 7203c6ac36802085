"""CPU restatement (torch fp64) of the reference's panoptic head - TEST INFRASTRUCTURE ONLY.

Follows /root/reference/ModelComponents/transformers.py:460-559 (PanopticAttention, including its quirk of projecting
keys and queries from the image encoding, lines 535-536) and panoptic_neck.py:8-186 (PanopticNeck, DownscaleBlock,
UpscaleBlock).  Third-party semantics assumed: Resizing = bilinear, half-pixel centres, no antialias
(S2); LayerNormalization over the last axis, eps 1e-3 (S6); ReLU(negative_slope=.01) = leaky ReLU; Conv2DTranspose
kernels are [kh, kw, out, in], stride 1, 'valid' (output = input + 1).  Parity unpinned: the reference never runs
these layers (model.py:4 has the import commented out) and TensorFlow is not installed here."""
import math

import torch
import torch.nn.functional as F


def panoptic_attention(image_encoding, num_obj, heads, hidden_dim, w):
    """image_encoding [B,r,c,E]; w: dict of Keras-layout arrays '<Layer>/kernel' [in,out], '<Layer>/bias', 'LayerNorm/gamma|beta'."""
    B, r, c, E = image_encoding.shape
    value = image_encoding.reshape(B, r * c, E)
    key_dim = max(1, hidden_dim // heads)
    dense = lambda x, n: x @ w[f"{n}/kernel"] + w[f"{n}/bias"]
    v, k, q = dense(value, "ValueProjection"), dense(value, "KeyProjection"), dense(value, "QueryProjection")
    s = q @ k.transpose(1, 2) / math.sqrt(float(key_dim))
    o = torch.softmax(s, -1) @ v
    o = F.layer_norm(o, (o.shape[-1],), w["LayerNorm/gamma"], w["LayerNorm/beta"], 1e-3)
    return o.reshape(B, r, c, num_obj, -1)


def _block(x, w, prefix, repeats, transpose):
    """x NCHW.  Conv2D / Conv2DTranspose (k=2) -> LayerNormalization over channels -> leaky ReLU, `repeats` times."""
    for i in range(repeats):
        k, b = w[f"{prefix}/Conv2D_{i}/kernel"], w[f"{prefix}/Conv2D_{i}/bias"]
        if transpose:
            x = F.conv_transpose2d(x, k.permute(3, 2, 0, 1), b)          # keras [kh,kw,out,in] -> torch [in,out,kh,kw]
        else:
            x = F.conv2d(x, k.permute(3, 2, 0, 1), b)                    # HWIO -> OIHW
        g, be = w[f"{prefix}/LayerNormalization_{i}/gamma"], w[f"{prefix}/LayerNormalization_{i}/beta"]
        x = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), g, be, 1e-3).permute(0, 3, 1, 2)
        x = F.leaky_relu(x, 0.01)
    return x


def panoptic_neck(features, w, scope="PanopticNeck"):
    """features [B,rows,cols,num_obj,dim] -> [B,num_obj,529]."""
    B, r, c, num_obj = features.shape[:4]
    x = features.reshape(B, r, c, -1).permute(0, 3, 1, 2)
    x = F.interpolate(x, size=(96, 96), mode="bilinear", align_corners=False, antialias=False)
    blk = lambda t, name, n, tr: _block(t, w, f"{scope}/{name}", n, tr)
    d0 = blk(x, "DownscaleBlock_0", 1, False)
    d1 = blk(d0, "DownscaleBlock_1", 1, False)
    d2 = blk(d1, "DownscaleBlock_2", 2, False)
    d3 = blk(d2, "DownscaleBlock_3", 3, False)
    u0 = blk(d3, "UpscaleBlock_0", 3, True)
    join_a = torch.cat([u0, d2], 1)
    u1 = blk(u0, "UpscaleBlock_1", 2, True)
    join_b = torch.cat([u1, d1], 1)
    u2 = blk(u1, "UpscaleBlock_2", 1, True)
    join_c = torch.cat([u2, d0], 1)
    join_a = blk(join_a, "UpscaleBlock_3", 2, True)
    join_c = blk(join_c, "DownscaleBlock_4", 1, False)
    f = torch.cat([join_a, join_b, join_c], 1)
    y = F.conv2d(f, w[f"{scope}/ConvOut/kernel"].permute(3, 2, 0, 1), w[f"{scope}/ConvOut/bias"], stride=4)
    return y.reshape(B, num_obj, -1)
