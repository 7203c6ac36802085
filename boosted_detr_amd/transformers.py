"""Encoder / decoder transformer layers on HIP kernels.

Mirrors /root/reference/ModelComponents/transformers.py class for class (MultiheadAttention 18-109,
AttentionBlock 112-158, FeedForwardBlock 161-198, EncoderBlock 200-241, ImageEncoderAttention
244-321, DecoderBlock_NoSelfAttention 324-353, DecoderBlock 356-394, DecoderPrep 397-456),
including the reference's quirks: the post-attention reshape without a head permute (line 100),
FFN hidden width == model width (174-177), position-parity positional table (282-292), no
positional term in the decoder self-attention (378-380).
"""
from __future__ import annotations

import math

import numpy as np

from . import kernels as K
from . import ops
from .engine import Layer

LN_EPS = 1e-3        # explicit at transformers.py:137; Keras default elsewhere (180)
DROPOUT_RATE = 0.1   # transformers.py:135,179


class _Dense:
    def __init__(self, layer: Layer, name: str, cin: int, cout: int, init: str):
        keep, layer.name = layer.name, f"{layer.name}/{name}"
        self.kernel = layer.add_weight("kernel", (cin, cout), init, kind="dense_kernel")
        self.bias = layer.add_weight("bias", (cout,), "zeros")
        layer.name = keep

    def __call__(self, x, act=K.ACT_NONE):
        return ops.dense(x, self.kernel, self.bias, act)


def _layer_norm_vars(layer: Layer, name: str, d: int):
    keep, layer.name = layer.name, f"{layer.name}/{name}"
    g, b = layer.add_weight("gamma", (d,), "ones"), layer.add_weight("beta", (d,), "zeros")
    layer.name = keep
    return g, b


class MultiheadAttention(Layer):
    def __init__(self, num_attention_heads, dim, name="MultiheadAttention", **kwargs):
        super().__init__(name=name, **kwargs)
        self.num_attention_heads = num_attention_heads
        self.dim = dim

    def get_config(self):
        c = super().get_config()
        c.update({"num_attention_heads": self.num_attention_heads, "dim": self.dim})
        return c

    def build(self, input_shape):
        self.query_shape, self.key_shape, self.value_shape = input_shape[0], input_shape[1], input_shape[2]
        query_dim = self.query_shape[-1]
        proj_dim = self.num_attention_heads * self.dim
        self.QueryProjection = _Dense(self, "QueryProjection", query_dim, proj_dim, "glorot_normal")
        self.KeyProjection = _Dense(self, "KeyProjection", self.key_shape[-1], proj_dim, "glorot_normal")
        self.ValueProjection = _Dense(self, "ValueProjection", self.value_shape[-1], proj_dim, "glorot_normal")
        self.OutputProjection = _Dense(self, "OutputProjection", proj_dim, query_dim, "glorot_normal")

    def call(self, inputs, attention_mask=None, training=False):
        if attention_mask is not None:
            raise NotImplementedError("the hot path always calls MultiheadAttention with attention_mask=None")
        return self.OutputProjection(self.context(inputs))

    def context(self, inputs):
        """Everything of the layer up to (not including) OutputProjection: the three projections, the attention core and the
        reference's reshape without a head permute.  [B, q, h*d]."""
        query, key, value = inputs
        q, k, v = ops.dense_group([query, key, value],
                                  [self.QueryProjection.kernel, self.KeyProjection.kernel, self.ValueProjection.kernel],
                                  [self.QueryProjection.bias, self.KeyProjection.bias, self.ValueProjection.bias])
        o = ops.attention_core(q, k, v, self.num_attention_heads)          # [B,h,q,d]
        B, h, nq, d = o.shape
        return ops.reshape(o, (B, nq, h * d))                               # ReshapePreOutput: no permute (line 100)


class AttentionBlock(Layer):
    def __init__(self, num_attention_heads, **kwargs):
        super().__init__(**kwargs)
        self.num_attention_heads = num_attention_heads

    def get_config(self):
        c = super().get_config()
        c.update({"num_attention_heads": self.num_attention_heads})
        return c

    def build(self, input_shape):
        self.query_shape, self.key_shape, self.value_shape = input_shape[0], input_shape[1], input_shape[2]
        query_dim = self.query_shape[-1]
        key_dim = query_dim // self.num_attention_heads
        self.AttentionLayer = MultiheadAttention(self.num_attention_heads, key_dim, name="AttentionLayer",
                                                 scope_prefix=f"{self.scope}/", seed=self._init_seed)
        self.ln_gamma, self.ln_beta = _layer_norm_vars(self, "LayerNorm", query_dim)

    def call(self, inputs, attention_mask=None, training=False):
        query, key, value = inputs
        att = self.AttentionLayer([query, key, value], attention_mask=attention_mask, training=training)
        return ops.add_dropout_layernorm(query, att, self.ln_gamma, self.ln_beta, LN_EPS, self.dropout_rate, training)

    dropout_rate = DROPOUT_RATE


class FeedForwardBlock(Layer):
    dropout_rate = DROPOUT_RATE

    def build(self, input_shape):
        self.features_shape = input_shape[0]
        d = self.features_shape[-1]
        self.DenseRelu = _Dense(self, "DenseRelu", d, d, "glorot_normal")
        self.DenseLinear = _Dense(self, "DenseLinear", d, d, "glorot_normal")
        self.ln_gamma, self.ln_beta = _layer_norm_vars(self, "LayerNorm", d)

    def call(self, inputs, training=False):
        features = inputs[0]
        y = self.DenseRelu(features, K.ACT_RELU)
        y = self.DenseLinear(y)
        return ops.add_dropout_layernorm(features, y, self.ln_gamma, self.ln_beta, LN_EPS, self.dropout_rate, training)


def attention_then(attn: AttentionBlock, ffn, inputs, training: bool):
    """AttentionBlock(inputs) followed by ``ffn`` (a FeedForwardBlock or None), exactly as the reference composes them
    (transformers.py:228-231, 348-351, 381-392) - on the fused row-chain kernels when they apply (csrc/rowchain.hip: a training step
    under the 'split' policy at model width 256, equal dropout rates, layers already built), otherwise layer by layer."""
    query, key, value = inputs
    fused = training and ops.rowchain_active(query.shape[-1]) and (ffn is None or ffn.dropout_rate == attn.dropout_rate)
    if fused:
        # build-by-first-call, as Layer.__call__ would do it: the very first step takes the same path as every later one
        shapes = [tuple(t.shape) for t in (query, key, value)]
        for layer, shp in ((attn, shapes), (getattr(attn, "AttentionLayer", None), shapes), (ffn, [shapes[0]])):
            if layer is not None and not layer.built:
                layer.build(shp)
                layer.built = True
        if not attn.AttentionLayer.built:                     # (created by attn.build just now)
            attn.AttentionLayer.build(shapes)
            attn.AttentionLayer.built = True
        fused = attn.AttentionLayer.num_attention_heads * attn.AttentionLayer.dim == query.shape[-1]
    if not fused:
        x = attn([query, key, value], training=training)
        return ffn([x], training=training) if ffn is not None else x
    mha = attn.AttentionLayer
    ctx = mha.context([query, key, value])
    ff = None if ffn is None else (ffn.DenseRelu.kernel, ffn.DenseRelu.bias, ffn.DenseLinear.kernel, ffn.DenseLinear.bias, ffn.ln_gamma, ffn.ln_beta)
    return ops.attention_out_chain(ctx, query, (mha.OutputProjection.kernel, mha.OutputProjection.bias), (attn.ln_gamma, attn.ln_beta), ff,
                                   LN_EPS, attn.dropout_rate, training)


class EncoderBlock(Layer):
    def __init__(self, num_attention_heads, **kwargs):
        super().__init__(**kwargs)
        self.num_attention_heads = num_attention_heads
        sp = f"{self.scope}/"
        self.SelfAttentionBlock = AttentionBlock(num_attention_heads, name="SelfAttentionBlock", scope_prefix=sp, seed=self._init_seed)
        self.FeedForwardBlock = FeedForwardBlock(name="FeedForwardBlock", scope_prefix=sp, seed=self._init_seed)

    def get_config(self):
        c = super().get_config()
        c.update({"num_attention_heads": self.num_attention_heads})
        return c

    def build(self, input_shape):
        self.encoder_features_shape, self.encoder_positional_shape = input_shape[0], input_shape[1]

    def call(self, inputs, training=False):
        encoder_features, encoder_positional = inputs           # positional is [T,D] (batch-broadcast)
        qk = ops.add_bcast(encoder_features, encoder_positional)  # Add1 and Add2 compute the same tensor (224-225)
        return attention_then(self.SelfAttentionBlock, self.FeedForwardBlock, [qk, qk, encoder_features], training)


def positional_init(r: int, c: int, D: int) -> np.ndarray:
    """transformers.py:282-292, evaluated in Python floats then stored as fp32."""
    k = np.arange(r * c, dtype=np.float64)[:, None]
    dim = np.arange(D, dtype=np.float64)[None, :]
    denom = 2 * (1 + dim) / D
    even = (np.arange(r * c) % 2)[:, None].astype(np.float64)
    odd = ((np.arange(r * c) + 1) % 2)[:, None].astype(np.float64)
    out = even * np.sin(k / denom) + odd * np.cos(k / denom)
    return out.reshape(r, c, D).astype(np.float32)


class ImageEncoderAttention(Layer):
    def __init__(self, num_blocks, num_attention_heads, name="ImageEncoderAttention", **kwargs):
        super().__init__(name=name, **kwargs)
        self.num_blocks = num_blocks
        self.num_attention_heads = num_attention_heads
        self.EncoderBlocks = [EncoderBlock(num_attention_heads, name=f"EncoderBlock_{i}", scope_prefix=f"{self.scope}/",
                                           seed=self._init_seed) for i in range(num_blocks)]
        for b in self.EncoderBlocks:
            self.track(b)

    def get_config(self):
        c = super().get_config()
        c.update({"num_blocks": self.num_blocks, "num_attention_heads": self.num_attention_heads})
        return c

    def build(self, input_shape):
        self.encoder_features_shape = input_shape[0]
        _, r, c, D = self.encoder_features_shape
        self.positional_encoding = self.add_weight("positional_encoding", (r, c, D), value=positional_init(r, c, D))

    def call(self, inputs, training=False):
        encoder_features = inputs[0]                       # [B,r,c,D]
        B, r, c, D = encoder_features.shape
        pos = self.positional_encoding            # Variable [r,c,D]; broadcast over the batch inside the add kernel
        x = ops.reshape(encoder_features, (B, r * c, D))
        for blk in self.EncoderBlocks:
            x = blk([x, pos], training=training)
        # the reference returns the batch-tiled positional tensor; the tile is kept implicit here
        return ops.reshape(x, (B, r, c, D)), self.positional_encoding


class DecoderBlock_NoSelfAttention(Layer):
    def __init__(self, num_attention_heads, **kwargs):
        super().__init__(**kwargs)
        self.num_attention_heads = num_attention_heads
        sp = f"{self.scope}/"
        self.JointAttentionBlock = AttentionBlock(num_attention_heads, name="JointAttentionBlock", scope_prefix=sp, seed=self._init_seed)
        self.FeedForwardBlock = FeedForwardBlock(name="FeedForwardBlock", scope_prefix=sp, seed=self._init_seed)

    def get_config(self):
        c = super().get_config()
        c.update({"num_attention_heads": self.num_attention_heads})
        return c

    def call(self, inputs, training=False):
        encoder_value, decoder_features, encoder_key, decoder_positional = inputs
        return attention_then(self.JointAttentionBlock, self.FeedForwardBlock, [decoder_features, encoder_key, encoder_value], training)


class DecoderBlock(Layer):
    def __init__(self, num_attention_heads, **kwargs):
        super().__init__(**kwargs)
        self.num_attention_heads = num_attention_heads
        sp = f"{self.scope}/"
        self.SelfAttentionBlock = AttentionBlock(num_attention_heads, name="SelfAttentionBlock", scope_prefix=sp, seed=self._init_seed)
        self.JointAttentionBlock = AttentionBlock(num_attention_heads, name="JointAttentionBlock", scope_prefix=sp, seed=self._init_seed)
        self.FeedForwardBlock = FeedForwardBlock(name="FeedForwardBlock", scope_prefix=sp, seed=self._init_seed)

    def get_config(self):
        c = super().get_config()
        c.update({"num_attention_heads": self.num_attention_heads})
        return c

    def call(self, inputs, training=False):
        encoder_value, decoder_features, encoder_key, decoder_positional = inputs
        x = attention_then(self.SelfAttentionBlock, None, [decoder_features, decoder_features, decoder_features], training)  # no positional (378-380)
        return attention_then(self.JointAttentionBlock, self.FeedForwardBlock, [x, encoder_key, encoder_value], training)


class DecoderPrep(Layer):
    def __init__(self, num_object_preds, decoder_dim, name="DecoderPrep", **kwargs):
        super().__init__(name=name, **kwargs)
        self.num_object_preds = num_object_preds
        self.decoder_dim = decoder_dim

    def get_config(self):
        c = super().get_config()
        c.update({"num_object_preds": self.num_object_preds, "decoder_dim": self.decoder_dim})
        return c

    def build(self, input_shape):
        self.encoder_features_shape = input_shape[0]
        self.encoder_positional_encoding_shape = input_shape[1]
        self.init_decoder_features = self.add_weight("init_decoder_features", (self.num_object_preds, self.decoder_dim), "zeros")

    def call(self, inputs, training=False):
        encoder_features, encoder_positional = inputs      # [B,r,c,D], positional Variable [r,c,D]
        B, r, c, D = encoder_features.shape
        encoder_value = ops.reshape(encoder_features, (B, r * c, D))
        encoder_key = ops.add_bcast(encoder_value, encoder_positional)     # value + positional (441)
        decoder_features = ops.tile_batch(self.init_decoder_features, B)
        decoder_positional = decoder_features
        return encoder_value, decoder_features, encoder_key, decoder_positional


class PanopticAttention(Layer):
    """transformers.py:460-559 ("not tested" in the reference, line 13), forward only.  Partial multi-head attention that
    emits one image-sized map per decoder box: value / key / query are ALL projected from the flattened image encoding
    (the reference feeds ``value`` to the key and query projections, lines 535-536 - reproduced), one softmax over the
    full num_heads*key_dim width (no head split), value width = num_heads * num_obj, LayerNormalization (eps 1e-3) of the
    result, reshaped to [B, rows, cols, num_obj, num_heads].  Dense layers use the Keras default glorot_uniform."""

    def __init__(self, num_attention_heads, hidden_dim, name="PanopticAttention", **kwargs):
        super().__init__(name=name, **kwargs)
        self.num_attention_heads = num_attention_heads
        self.hidden_dim = hidden_dim

    def get_config(self):
        c = super().get_config()
        c.update({"num_attention_heads": self.num_attention_heads, "hidden_dim": self.hidden_dim})
        return c

    def build(self, input_shape):
        self.image_encoding_shape, self.decoder_encoding_shape = input_shape[0], input_shape[1]
        enc = self.image_encoding_shape[3]
        self.num_obj = self.decoder_encoding_shape[1]
        self.key_dim = max(1, self.hidden_dim // self.num_attention_heads)
        h = self.num_attention_heads
        self.ValueProjection = _Dense(self, "ValueProjection", enc, h * self.num_obj, "glorot_uniform")
        self.KeyProjection = _Dense(self, "KeyProjection", enc, h * self.key_dim, "glorot_uniform")
        self.QueryProjection = _Dense(self, "QueryProjection", enc, h * self.key_dim, "glorot_uniform")
        self.ln_gamma, self.ln_beta = _layer_norm_vars(self, "LayerNorm", h * self.num_obj)

    def call(self, inputs, training=False, self_attention_mask=None):
        image_encoding, decoder_encoding, positional_encoding = inputs       # the last two only fix shapes (reference quirk)
        B, r, c, E = image_encoding.shape
        T = r * c
        value = image_encoding.reshape(B * T, E).contiguous()
        v = K.linear_fwd(value, self.ValueProjection.kernel.value, self.ValueProjection.bias.value)      # [B*T, h*num_obj]
        k = K.linear_fwd(value, self.KeyProjection.kernel.value, self.KeyProjection.bias.value)          # [B*T, h*kd]
        q = K.linear_fwd(value, self.QueryProjection.kernel.value, self.QueryProjection.bias.value)
        D, Dv = k.shape[1], v.shape[1]
        s = K.empty(B, T, T, like=value)
        K.gemm_raw(T, T, D, q, D, True, k, D, True, s, T, nb0=B, sa=(T * D, 0), sb=(T * D, 0), sc=(T * T, 0))       # MatMul_1
        p = K.softmax_rows_fwd(s.view(-1, T), 1.0 / math.sqrt(float(self.key_dim)), out=s.view(-1, T))               # Divide + Softmax
        o = K.empty(B, T, Dv, like=value)
        K.gemm_raw(T, Dv, T, p, T, True, v, Dv, False, o, Dv, nb0=B, sa=(T * T, 0), sb=(T * Dv, 0), sc=(T * Dv, 0))  # MatMul_2
        o = K.layernorm_act(o, Dv, self.ln_gamma.value, self.ln_beta.value, LN_EPS, 1.0, ld_out=Dv)
        return o.view(B, r, c, self.num_obj, Dv // self.num_obj)                                                       # ReshapeOutput
