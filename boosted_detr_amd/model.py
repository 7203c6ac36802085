"""DETR model (drop-in for /root/reference/ModelComponents/model.py:12-244).

Same constructor signature, same ``call(inputs: dict, training)`` contract, same public sub-layer
attributes (EncoderBackbone, BackboneNeck, ImageEncoderAttention, DecoderPrep, DecoderBlocks[i],
CategoryPredictionHead, AttributePredictionHead, BoxPredictionHead, loss_fn), losses built into the
model (``add_loss`` of a per-image [B] vector, 5 ``add_metric`` calls), ``test_step`` == ``train_step``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import backbone, losses_and_metrics, prediction_heads, tokenizers, transformers
from .engine import to_device
from .training import Model


def _prepare_targets(model, inputs):
    """model.py:148-158: tokenise strings (or accept ids) and move the targets to HBM."""
    category, attribute = model.Tokenization([inputs["category"], inputs["attribute"]])
    bbox = to_device(np.asarray(inputs["bbox"], np.float32) if not isinstance(inputs["bbox"], torch.Tensor) else inputs["bbox"])
    num_objects = inputs["num_objects"]
    num_objects = to_device(num_objects.reshape(-1) if isinstance(num_objects, torch.Tensor) else np.asarray(num_objects).reshape(-1), torch.int32)
    return [category, attribute, bbox, num_objects]


def _image(inputs):
    img = inputs["image"]
    return to_device(img if isinstance(img, torch.Tensor) else np.asarray(img, np.float32))


class DETR(Model):
    def __init__(self, num_object_preds, image_size, num_encoder_blocks, num_encoder_heads, encoder_dim,
                 num_decoder_blocks, num_decoder_heads, decoder_dim, num_panoptic_heads=1, panoptic_dim=32, vocab_dict=None,
                 classification_only=False, attribute_weight=1.0, name="DETR", **kwargs):
        seed = int(kwargs.pop("seed", 0))
        backbone_name = kwargs.pop("backbone_name", "ResNet")      # reference default is EfficientNet (out of scope)
        with_panoptic_head = bool(kwargs.pop("with_panoptic_head", False))
        super().__init__(name=name, seed=seed)                      # pad_value / oov_value etc. are swallowed like the reference's **kwargs
        category_weight = box_weight = exist_weight = None
        if classification_only:
            box_weight = 0.0
        self.num_object_preds = num_object_preds
        self.image_size = tuple(image_size)
        self.num_encoder_blocks, self.num_encoder_heads, self.encoder_dim = num_encoder_blocks, num_encoder_heads, encoder_dim
        self.num_decoder_blocks, self.num_decoder_heads, self.decoder_dim = num_decoder_blocks, num_decoder_heads, decoder_dim
        self.num_panoptic_heads, self.panoptic_dim = num_panoptic_heads, panoptic_dim
        self.vocab_dict = vocab_dict

        self.Tokenization = tokenizers.Tokenization(vocab_dict=vocab_dict, name="Tokenization")
        self.InverseTokenization = tokenizers.InverseTokenization(vocab_dict=vocab_dict)
        sizes = self.Tokenization.vocab_size_dict()
        self.num_categories, self.num_attributes = sizes["category"], sizes["attributes"]

        self.EncoderBackbone = backbone.EncoderBackbone(image_input_shape=self.image_size, model_name=backbone_name,
                                                        name="EncoderBackbone", seed=seed)
        self.BackboneNeck = backbone.BackboneNeck(encoder_dim=encoder_dim, name="BackboneNeck", seed=seed)
        self.ImageEncoderAttention = transformers.ImageEncoderAttention(num_blocks=num_encoder_blocks, num_attention_heads=num_encoder_heads,
                                                                        name="ImageEncoderAttention", seed=seed)
        self.DecoderPrep = transformers.DecoderPrep(num_object_preds, decoder_dim, name="DecoderPrep", seed=seed)
        self.DecoderBlocks = [transformers.DecoderBlock_NoSelfAttention(num_attention_heads=num_decoder_heads, name="DecoderBlock_0", seed=seed)]
        for i in range(1, num_decoder_blocks):
            self.DecoderBlocks.append(transformers.DecoderBlock(num_attention_heads=num_decoder_heads, name=f"DecoderBlock_{i}", seed=seed))
        for b in self.DecoderBlocks:
            self.track(b)
        self.CategoryPredictionHead = prediction_heads.SingleClassPredictionHead(num_classes=self.num_categories, hidden_dim=4 * decoder_dim,
                                                                                 num_preds=num_object_preds, name="CategoryPredictionHead", seed=seed)
        self.AttributePredictionHead = prediction_heads.MultiClassPredictionHead(num_classes=self.num_attributes, hidden_dim=4 * decoder_dim,
                                                                                 num_preds=num_object_preds, name="AttributePredictionHead", seed=seed)
        self.BoxPredictionHead = prediction_heads.BoxPredictionHead(hidden_dim=decoder_dim, num_preds=num_object_preds,
                                                                    name="BoxPredictionHead", seed=seed)
        self.loss_fn = losses_and_metrics.MatchingLoss(category_weight=category_weight, box_weight=box_weight,
                                                       attribute_weight=attribute_weight, exist_weight=exist_weight, name="MatchingLoss")
        # BASELINE.json configs[4]'s mask head.  The reference constructs neither layer (model.py:4 has the import commented out;
        # num_panoptic_heads / panoptic_dim are accepted and unused), so the head is opt-in, forward-only and frozen: it runs on the
        # features the last call left behind (`panoptic_masks`), outside the training step's tape and arithmetic policy.
        self.PanopticAttention = self.PanopticNeck = None
        if with_panoptic_head:
            from . import panoptic_neck
            self.PanopticAttention = transformers.PanopticAttention(num_attention_heads=num_panoptic_heads, hidden_dim=panoptic_dim, seed=seed)
            self.PanopticNeck = panoptic_neck.PanopticNeck(seed=seed)
            self.PanopticAttention.trainable = False
            self.PanopticNeck.trainable = False
        self._panoptic_inputs = None

    def panoptic_masks(self):
        """[B, num_object_preds, 23 * 23] mask logits of the last call's images (transformers.py:460-559 on the image encoding +
        panoptic_neck.py:8-88), or None before the first call."""
        if self.PanopticAttention is None:
            raise RuntimeError("construct the model with with_panoptic_head=True")
        if self._panoptic_inputs is None:
            return None
        enc, dec, pos = self._panoptic_inputs
        return self.PanopticNeck([self.PanopticAttention([enc, dec, pos])])

    def get_config(self):
        c = Model.get_config(self)            # explicit base: BoostedDETR reuses this function
        c.update({k: getattr(self, k) for k in ("num_object_preds", "image_size", "num_encoder_blocks", "num_encoder_heads", "encoder_dim",
                                                 "num_decoder_blocks", "num_decoder_heads", "decoder_dim", "num_panoptic_heads",
                                                 "panoptic_dim", "vocab_dict")})
        return c

    def call(self, inputs, training=False):
        image = _image(inputs)
        if training:
            y_true = _prepare_targets(self, inputs)

        encoder_features = self.EncoderBackbone([image], training=training)
        encoder_features = self.BackboneNeck([encoder_features], training=training)
        encoder_features, positional_encoding = self.ImageEncoderAttention([encoder_features], training=training)
        image_encoding = encoder_features                         # [B, r, c, D]
        encoder_features, decoder_features, encoder_key, decoder_positional = \
            self.DecoderPrep([encoder_features, positional_encoding], training=training)

        use_intermediate_losses = False       # hard-coded in the reference (model.py:179)
        loss_terms, metrics_i, y_pred_i = [], None, None
        for i in range(self.num_decoder_blocks):
            decoder_features = self.DecoderBlocks[i]([encoder_features, decoder_features, encoder_key, decoder_positional], training=training)
            if training and (use_intermediate_losses or i >= self.num_decoder_blocks - 1):
                cat_preds_i = self.CategoryPredictionHead([decoder_features], training=training)
                attribute_preds_i = self.AttributePredictionHead([decoder_features], training=training)
                box_coord_preds_i = self.BoxPredictionHead([decoder_features], training=training)
                y_pred_i = [cat_preds_i, attribute_preds_i, box_coord_preds_i]
                losses_i, metrics_i = self.loss_fn([y_true, y_pred_i])
                loss_terms.append(losses_i)
                self._loss_roots.append(self.loss_fn._losses_tensor)

        if self.PanopticAttention is not None:
            self._panoptic_inputs = (image_encoding, decoder_features, positional_encoding.value)
        if training:
            self._register(loss_terms, metrics_i)
            return y_pred_i

        cat_preds = self.CategoryPredictionHead([decoder_features], training=training)
        attribute_preds = self.AttributePredictionHead([decoder_features], training=training)
        box_coord_preds = self.BoxPredictionHead([decoder_features], training=training)
        category, attributes = self.InverseTokenization([cat_preds, attribute_preds], training=training)
        return category, attributes, box_coord_preds

    def _register(self, loss_terms, metrics_i):
        """model.py:206-221.  Per-learner loss vectors are kept as a list (summed on the host when
        logged) instead of being added on the device: the sum is never needed by the gradient."""
        for k, name in enumerate(["loss", "Category_Loss", "Attribute_Loss", "Box_Loss", "Existence_Loss"]):
            terms = [t[k] for t in loss_terms]
            if name == "loss":
                for t in terms:
                    self.add_loss(t)
            else:
                self.add_metric(terms, name)
        self.add_metric([metrics_i[0]], "IOU")

    def citation(self):
        print("DETR-like model for object detection and fine-grained classification, after 'End-to-end Object Detection "
              "with Transformers' (Carion et al.); MI355X-native re-implementation of the mvenouziou/Boosted_DETR training path.")
