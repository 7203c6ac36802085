"""Panoptic head, forward only (drop-in for /root/reference/ModelComponents/panoptic_neck.py:8-186; its attention
input comes from transformers.PanopticAttention, transformers.py:460-559).

The reference never wires these layers into a model (the import is commented out, model.py:4) and never trains them,
so there is no loss or gradient to reproduce: this is BASELINE.json configs[4]'s "mask head" as a throughput path.
Channel counts shrink / grow by 2/3 and 3/2 (100 -> 66 -> 44 -> 29 ...), so tensors carry their TRUE channel count
next to a storage width padded to a multiple of 4 (zeros): ``(tensor [B,H,W,ld], C)``.  Convolutions run on the MFMA
implicit-GEMM kernels with zero-padded weights; Conv2DTranspose(k=2, stride 1, valid) is the full-padding convolution
with the taps flipped and the in/out axes of the Keras kernel swapped."""
from __future__ import annotations

import numpy as np
import torch

from . import kernels as K
from .engine import WEIGHTS_VERSION, Layer, to_device

LN_EPS = 1e-3            # tf.keras.layers.LayerNormalization default
LEAKY = 0.01             # ReLU(negative_slope=.01)


class _ConvLNBlock(Layer):
    """num_repeats x [Conv2D / Conv2DTranspose (k=2) -> LayerNormalization -> leaky ReLU]; filters *= 2/3 or 3/2."""

    transpose = False

    def __init__(self, num_repeats=2, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.num_repeats = num_repeats

    def get_config(self):
        c = super().get_config()
        c.update({"num_repeats": self.num_repeats})
        return c

    def next_filters(self, f: int) -> int:
        return 3 * f // 2 if self.transpose else 2 * f // 3

    def build(self, input_shape):
        self.features_shape = input_shape[0]
        f = self.in_channels
        self.convs = []
        for i in range(self.num_repeats):
            g = self.next_filters(f)
            keep, self.name = self.name, f"{self.name}/Conv2D_{i}"
            kshape = (2, 2, g, f) if self.transpose else (2, 2, f, g)          # Conv2DTranspose kernels are [kh, kw, out, in]
            kernel = self.add_weight("kernel", kshape, "glorot_uniform")
            bias = self.add_weight("bias", (g,), "zeros")
            self.name = f"{keep}/LayerNormalization_{i}"
            gamma, beta = self.add_weight("gamma", (g,), "ones"), self.add_weight("beta", (g,), "zeros")
            self.name = keep
            self.convs.append((kernel, bias, gamma, beta, f, g))
            f = g
        self.out_channels = f
        self._packed = None

    def _weights(self):
        """OHWI kernels with both channel axes zero-padded to multiples of 4 (rebuilt when the weights change)."""
        if self._packed is None or self._packed[0] != WEIGHTS_VERSION[0]:
            packed = []
            for kernel, bias, gamma, beta, f, g in self.convs:
                k = kernel.numpy()
                if self.transpose:
                    k = np.transpose(k[::-1, ::-1], (2, 0, 1, 3))      # [kh,kw,out,in] flipped -> [out,kh,kw,in]
                else:
                    k = np.transpose(k, (3, 0, 1, 2))                    # HWIO -> OHWI
                w = np.zeros((K.pad4(g), 2, 2, K.pad4(f)), np.float32)
                w[:g, :, :, :f] = k
                b = np.zeros(K.pad4(g), np.float32)
                b[:g] = bias.numpy()
                packed.append((to_device(w), to_device(b)))
            self._packed = (WEIGHTS_VERSION[0], packed)
        return self._packed[1]

    def __call__(self, inputs, training=False, **kw):
        x, C = inputs[0]
        if not self.built:
            self.in_channels = C
            self.build([tuple(x.shape)])
            self.built = True
        return self.call(inputs, training=training)

    def call(self, inputs, training=False):
        x, C = inputs[0]
        for (kernel, bias, gamma, beta, f, g), (w, b) in zip(self.convs, self._weights()):
            N, H, W, ld = x.shape
            geom = K.ConvGeom(N, H, W, ld, K.pad4(g), 2, 2, 1, 1 if self.transpose else 0)
            y, _ = K.conv2d_fwd(x, w, b, geom, K.ACT_NONE)
            x = K.layernorm_act(y, g, gamma.value, beta.value, LN_EPS, LEAKY)
        return x, self.out_channels


class DownscaleBlock(_ConvLNBlock):
    """panoptic_neck.py:91-135."""
    transpose = False


class UpscaleBlock(_ConvLNBlock):
    """panoptic_neck.py:138-186."""
    transpose = True


class PanopticNeck(Layer):
    """panoptic_neck.py:8-88.  call([features [B,rows,cols,num_obj,dim]]) -> [B, num_obj, 23*23]."""

    def __init__(self, name="PanopticNeck", **kwargs):
        super().__init__(name=name, **kwargs)

    def build(self, input_shapes):
        self.features_shape = input_shapes[0]
        self.num_obj = self.features_shape[3]
        p = f"{self.scope}/"
        mk = lambda cls, n, nm: cls(num_repeats=n, name=nm, scope_prefix=p)
        self.DownscaleBlock_0, self.DownscaleBlock_1 = mk(DownscaleBlock, 1, "DownscaleBlock_0"), mk(DownscaleBlock, 1, "DownscaleBlock_1")
        self.DownscaleBlock_2, self.DownscaleBlock_3 = mk(DownscaleBlock, 2, "DownscaleBlock_2"), mk(DownscaleBlock, 3, "DownscaleBlock_3")
        self.UpscaleBlock_0, self.UpscaleBlock_1 = mk(UpscaleBlock, 3, "UpscaleBlock_0"), mk(UpscaleBlock, 2, "UpscaleBlock_1")
        self.UpscaleBlock_2, self.UpscaleBlock_3 = mk(UpscaleBlock, 1, "UpscaleBlock_2"), mk(UpscaleBlock, 2, "UpscaleBlock_3")
        self.DownscaleBlock_4 = mk(DownscaleBlock, 1, "DownscaleBlock_4")
        self._conv_out = None

    @staticmethod
    def _concat(parts):
        """Concatenate (tensor, C) pairs along the TRUE channels into one padded tensor."""
        C = sum(c for _, c in parts)
        t0 = parts[0][0]
        out = torch.empty(tuple(t0.shape[:-1]) + (K.pad4(C),), dtype=torch.float32, device=t0.device)
        if K.pad4(C) != C:
            K.zero_(out)
        col = 0
        for t, c in parts:
            assert t.shape[:-1] == t0.shape[:-1], (t.shape, t0.shape)
            K.copy_cols(t, c, out, col)
            col += c
        return out, C

    def _conv_out_weights(self, cin: int):
        if self._conv_out is None or self._conv_out[0] != WEIGHTS_VERSION[0]:
            k = np.transpose(self.ConvOut_kernel.numpy(), (3, 0, 1, 2))
            w = np.zeros((K.pad4(self.num_obj), 3, 3, K.pad4(cin)), np.float32)
            w[:self.num_obj, :, :, :cin] = k
            b = np.zeros(K.pad4(self.num_obj), np.float32)
            b[:self.num_obj] = self.ConvOut_bias.numpy()
            self._conv_out = (WEIGHTS_VERSION[0], to_device(w), to_device(b))
        return self._conv_out[1], self._conv_out[2]

    def call(self, inputs, training=False):
        features = inputs[0]                                   # [B, rows, cols, num_obj, dim]
        B, r, c = features.shape[:3]
        C = int(np.prod(features.shape[3:]))
        x = features.reshape(B, r, c, C)                       # ReshapeInput
        if K.pad4(C) != C:
            x, _ = self._concat([(x.contiguous(), C)])
        orig = (K.resize_bilinear(x.contiguous(), 96, 96), C)  # Resize
        d0 = self.DownscaleBlock_0([orig])
        d1 = self.DownscaleBlock_1([d0])
        d2 = self.DownscaleBlock_2([d1])
        d3 = self.DownscaleBlock_3([d2])
        u0 = self.UpscaleBlock_0([d3])
        join_a = self._concat([u0, d2])
        u1 = self.UpscaleBlock_1([u0])
        join_b = self._concat([u1, d1])
        u2 = self.UpscaleBlock_2([u1])
        join_c = self._concat([u2, d0])
        join_a = self.UpscaleBlock_3([join_a])
        join_c = self.DownscaleBlock_4([join_c])
        feats, cin = self._concat([join_a, join_b, join_c])
        if not hasattr(self, "ConvOut_kernel"):
            keep, self.name = self.name, f"{self.name}/ConvOut"
            self.ConvOut_kernel = self.add_weight("kernel", (3, 3, cin, self.num_obj), "glorot_uniform")
            self.ConvOut_bias = self.add_weight("bias", (self.num_obj,), "zeros")
            self.name = keep
        w, b = self._conv_out_weights(cin)
        N, H, W, ld = feats.shape
        geom = K.ConvGeom(N, H, W, ld, K.pad4(self.num_obj), 3, 3, 4, 0)
        y, _ = K.conv2d_fwd(feats, w, b, geom, K.ACT_NONE)     # ConvOut: k=3, strides=4, valid
        return K.nhwc_to_nchw(y, self.num_obj)                 # TransposeOut + FlattenDim: [B, num_obj, OH*OW]
