"""Prediction heads on HIP kernels.

Mirrors /root/reference/ModelComponents/prediction_heads.py: BoxPredictionHead (13-69),
SingleClassPredictionHead (72-137), MultiClassPredictionHead (140-207):
Dense(hidden, relu, he_normal) -> BatchNormalization (stats over batch*queries) -> Dense ->
{3*sigmoid(x/100)-1, softmax, sigmoid}.  The Conv1D re-count branch (120-123) is only taken
when the incoming number of predictions differs from ``num_preds``; the hot path never does.
"""
from __future__ import annotations

from . import kernels as K
from . import ops
from .engine import Layer
from .transformers import _Dense

KERAS_BN_EPS = 1e-3


class _Head(Layer):
    dense_name = "Dense"
    out_name = "Out"

    def __init__(self, hidden_dim, num_preds, out_dim, name, **kwargs):
        super().__init__(name=name, **kwargs)
        self.hidden_dim, self.num_preds, self.out_dim = hidden_dim, num_preds, out_dim

    def build(self, input_shape):
        self.features_shape = input_shape[0]
        d = self.features_shape[-1]
        self.DenseHidden = _Dense(self, self.dense_name, d, self.hidden_dim, "he_normal")
        keep, self.name = self.name, f"{self.name}/BatchNorm"
        self.BatchNorm = ops.BNState(self.add_weight("gamma", (self.hidden_dim,), "ones"),
                                     self.add_weight("beta", (self.hidden_dim,), "zeros"),
                                     self.add_weight("moving_mean", (self.hidden_dim,), "zeros", trainable=False),
                                     self.add_weight("moving_variance", (self.hidden_dim,), "ones", trainable=False), KERAS_BN_EPS)
        self.name = keep
        self.DenseOut = _Dense(self, self.out_name, self.hidden_dim, self.out_dim, "glorot_normal")

    def trunk(self, inputs, training):
        features = inputs[0]                                  # [B, N, D]
        if features.shape[1] != self.num_preds:
            raise NotImplementedError("Conv1D re-count branch (prediction_heads.py:120-123) is off the hot path")
        x = self.DenseHidden(features, K.ACT_RELU)
        x = ops.batchnorm(x, self.BatchNorm, training and self.trainable, bessel=False)
        # the pre-activation output (softmax / sigmoid / box-sigmoid input: prediction_heads.py:111,180,44) stays reachable
        # as `last_logits`: the parity tests compare it with the oracle's, the north star's "logits within 1e-3"
        self.last_logits = self.DenseOut(x)
        return self.last_logits

    def get_config(self):
        c = super().get_config()
        c.update({"hidden_dim": self.hidden_dim, "num_preds": self.num_preds})
        return c


class BoxPredictionHead(_Head):
    dense_name, out_name = "Dense", "BoxCoords"

    def __init__(self, hidden_dim, num_preds, name="BoxPredictionHead", **kwargs):
        super().__init__(hidden_dim, num_preds, 4, name, **kwargs)

    def call(self, inputs, training=False):
        return ops.box_sigmoid(self.trunk(inputs, training))     # 3*sigmoid(x/100)-1 in (-1,2), COCO [xmin,ymin,w,h]


class SingleClassPredictionHead(_Head):
    dense_name, out_name = "DenseCateg", "DenseLogits"

    def __init__(self, num_classes, hidden_dim, num_preds, name="SingleClassPredictionHead", **kwargs):
        super().__init__(hidden_dim, num_preds, num_classes, name, **kwargs)
        self.num_classes = num_classes

    def get_config(self):
        c = super().get_config()
        c.update({"num_classes": self.num_classes})
        return c

    def call(self, inputs, training=False):
        return ops.softmax_lastdim(self.trunk(inputs, training))


class MultiClassPredictionHead(_Head):
    dense_name, out_name = "Dense", "DenseLinear"

    def __init__(self, num_classes, hidden_dim, num_preds, name="MultiClassPredictionHead", **kwargs):
        super().__init__(hidden_dim, num_preds, num_classes, name, **kwargs)
        self.num_classes = num_classes

    def get_config(self):
        c = super().get_config()
        c.update({"num_classes": self.num_classes})
        return c

    def call(self, inputs, training=False):
        return ops.sigmoid(self.trunk(inputs, training))
