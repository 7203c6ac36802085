"""EncoderBackbone / BackboneNeck on HIP kernels.

Mirrors /root/reference/ModelComponents/backbone.py (same class names, constructor arguments,
``call(list_of_tensors, training=)``); the arithmetic of the third-party pieces it delegates to
(``tf.keras.applications.resnet50.ResNet50`` + ``preprocess_input``, Keras Conv2D /
BatchNormalization) is implemented by csrc/igemm.hip, csrc/norm.hip and csrc/elementwise.hip.
"""
from __future__ import annotations

from . import kernels as K
from . import ops
from .engine import Layer

RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))     # keras.applications ResNet-50 v1
RESNET101_STAGES = ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))   # config 5 only (no reference counterpart)
RESNET_BN_EPS = 1.001e-5
KERAS_BN_EPS = 1e-3


class _ConvBN(Layer):
    """Conv2D(use_bias=True, glorot_uniform) + BatchNormalization as keras.applications builds them."""

    def __init__(self, prefix: str, conv_name: str, bn_name: str, cin: int, cout: int, ksize: int, stride: int, pad: int,
                 pad_in_channels: int = 0, **kw):
        super().__init__(name=conv_name, scope_prefix=prefix, **kw)
        self.stride, self.pad = stride, pad
        self.kernel = self.add_weight("kernel", (ksize, ksize, cin, cout), "glorot_uniform", kind="conv_kernel",
                                      pad_in_channels=pad_in_channels)
        self.bias = self.add_weight("bias", (cout,), "zeros")
        # BN variables live under the BN layer's own Keras name
        self.name = bn_name
        gamma = self.add_weight("gamma", (cout,), "ones")
        beta = self.add_weight("beta", (cout,), "zeros")
        mm = self.add_weight("moving_mean", (cout,), "zeros", trainable=False)
        mv = self.add_weight("moving_variance", (cout,), "ones", trainable=False)
        self.name = conv_name
        self.bn = ops.BNState(gamma, beta, mm, mv, RESNET_BN_EPS)
        self.built = True

    def call(self, inputs, training=False, relu=True, residual=None, x_needs_grad=True, want_fp32=True, want_p16=False, defer_apply=False,
             next_is_identity_unit=False, want_bf16=False):
        x = inputs[0]
        # S18: BN uses batch statistics only when training AND the layer is trainable
        return ops.conv_bn(x, self.kernel, self.bias, self.bn, self.stride, self.pad, relu, residual=residual,
                           training=training, bn_batch_stats=training and self.trainable, x_needs_grad=x_needs_grad,
                           want_fp32=want_fp32, want_p16=want_p16, defer_apply=defer_apply,
                           sole_consumer_is_identity_unit=next_is_identity_unit, want_bf16=want_bf16)


class ResNet(Layer):
    """Keras ResNet v1 topology, include_top=False: ZeroPad3 -> conv7x7/2 -> BN -> ReLU ->
    ZeroPad1 -> maxpool3x3/2 -> bottleneck stages with the stride on the first 1x1."""

    def __init__(self, stages=RESNET50_STAGES, name="resnet50", scope_prefix="", **kw):
        super().__init__(name=name, scope_prefix=scope_prefix, **kw)
        p = f"{self.scope}/"
        self.stem = _ConvBN(p, "conv1_conv", "conv1_bn", 3, 64, 7, 2, 3, pad_in_channels=1, **kw)
        self.blocks = []
        cin = 64
        for si, (f, nblocks, stride1) in enumerate(stages):
            for bi in range(nblocks):
                bp = f"conv{si + 2}_block{bi + 1}"
                s = stride1 if bi == 0 else 1
                blk = {
                    "short": _ConvBN(p, f"{bp}_0_conv", f"{bp}_0_bn", cin, 4 * f, 1, s, 0, **kw) if bi == 0 else None,
                    "c1": _ConvBN(p, f"{bp}_1_conv", f"{bp}_1_bn", cin, f, 1, s, 0, **kw),
                    "c2": _ConvBN(p, f"{bp}_2_conv", f"{bp}_2_bn", f, f, 3, 1, 1, **kw),
                    "c3": _ConvBN(p, f"{bp}_3_conv", f"{bp}_3_bn", f, 4 * f, 1, 1, 0, **kw),
                }
                for l in blk.values():
                    if l is not None:
                        self.track(l)
                self.blocks.append(blk)
                cin = 4 * f
        self.out_channels = cin
        self.built = True

    def call(self, inputs, training=False):
        x = inputs[0]                                    # [B,H,W,4] prepared image
        st = self.stem
        x = ops.conv_bn_relu_maxpool(x, st.kernel, st.bias, st.bn, st.stride, st.pad, training, training and st.trainable)
        # Inside a bottleneck the 1x1 -> 3x3 -> 1x1 links are consumed by convolutions only: on the pre-split operand
        # path (ops.conv_bn, 'split' policy) they exist as 16-bit pairs and never as fp32 tensors.  Block outputs are also
        # the next block's shortcut (read back from the f16 pair) and their own ReLU-mask source in the backward pass (the
        # hi halves of the bf16 pair); only the last one, which leaves the backbone, is written in fp32.
        for i, blk in enumerate(self.blocks):
            last = i + 1 == len(self.blocks)
            # projection shortcut: its BatchNorm is applied inside c3's pass (ops.conv_bn defer_apply)
            sc = blk["short"]([x], training=training, relu=False, defer_apply=True) if blk["short"] is not None else x
            # (c1's output feeds the 3x3: a quarter-width tensor whose bf16 pair copy is cheap and keeps the MFMA-bound 3x3 weight
            # gradient free of the in-register f16 -> bf16 conversion the wide 1x1 layers' weight gradients do instead, ops.conv_bn)
            y = blk["c1"]([x], training=training, relu=True, want_fp32=False, want_p16=True, want_bf16=True)
            y = blk["c2"]([y], training=training, relu=True, want_fp32=False, want_p16=True)
            nxt_identity = not last and self.blocks[i + 1]["short"] is None       # the next unit reads x through its c1 and its identity skip only
            x = blk["c3"]([y], training=training, relu=True, residual=sc, want_fp32=last, want_p16=not last,
                          next_is_identity_unit=nxt_identity)                                 # BN -> Add([shortcut, x]) -> ReLU
        return x


class EncoderBackbone(Layer):
    """backbone.py:15-58.  Only the ResNet branch (34-39) is built; the reference's constructor
    default ('EfficientNet', backbone.py:17) is outside the hot path named by BASELINE.json, so this
    class defaults to model_name='ResNet' and rejects anything else.  ImageNet weights cannot be
    downloaded offline: weights start from the Keras initialisers (load real ones with
    ``model.load_weights``/``set_weights``)."""

    def __init__(self, image_input_shape, model_name="ResNet", name="EncoderBackbone", **kwargs):
        super().__init__(name=name, **kwargs)
        if model_name not in ("ResNet", "ResNet50", "ResNet101"):
            raise NotImplementedError(f"model_name={model_name!r}: only the ResNet branch of backbone.py is on the hot path")
        self.image_input_shape = tuple(image_input_shape)
        stages = RESNET101_STAGES if model_name == "ResNet101" else RESNET50_STAGES
        self.ImageFeaturesExtractor = ResNet(stages, name="resnet50" if model_name != "ResNet101" else "resnet101",
                                             scope_prefix=f"{self.scope}/", **kwargs)
        self.built = True

    def config(self):
        return self.get_config()

    def get_config(self):
        c = super().get_config()
        c.update({"image_input_shape": self.image_input_shape})
        return c

    def call(self, inputs, training=False):
        image = inputs[0]                                  # [B,h,w,3] in [0,1]
        height, width = self.image_input_shape[:2]
        x = ops.image_prep(image, height, width)          # clip, Resize, uint8 round trip, preprocess_input
        return self.ImageFeaturesExtractor([x], training=training)


class BackboneNeck(Layer):
    """backbone.py:66-104: BatchNormalization -> Conv2D 1x1 (tanh, lecun_normal) -> BatchNormalization."""

    def __init__(self, encoder_dim, name="BackboneNeck", **kwargs):
        super().__init__(name=name, **kwargs)
        self.encoder_dim = encoder_dim

    def get_config(self):
        c = super().get_config()
        c.update({"encoder_dim": self.encoder_dim})
        return c

    def _bn(self, bn_name, c):
        keep, self.name = self.name, f"{self.name}/{bn_name}"
        st = ops.BNState(self.add_weight("gamma", (c,), "ones"), self.add_weight("beta", (c,), "zeros"),
                         self.add_weight("moving_mean", (c,), "zeros", trainable=False),
                         self.add_weight("moving_variance", (c,), "ones", trainable=False), KERAS_BN_EPS)
        self.name = keep
        return st

    def build(self, input_shape):
        self.features_shape = input_shape[0]
        cin = self.features_shape[-1]
        self.batch_norm1 = self._bn("batch_norm1", cin)
        keep, self.name = self.name, f"{self.name}/conv2d_downscaler"
        self.kernel = self.add_weight("kernel", (1, 1, cin, self.encoder_dim), "lecun_normal", kind="conv_kernel")
        self.bias = self.add_weight("bias", (self.encoder_dim,), "zeros")
        self.name = keep
        self.batch_norm2 = self._bn("batch_norm2", self.encoder_dim)

    def call(self, inputs, training=False):
        features = inputs[0]
        bstats = training and self.trainable
        features = ops.batchnorm(features, self.batch_norm1, bstats, bessel=True)
        features = ops.conv_act(features, self.kernel, self.bias, 1, 0, K.ACT_TANH)
        features = ops.batchnorm(features, self.batch_norm2, bstats, bessel=True)
        return features
