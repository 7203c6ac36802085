"""BoostedDETR (drop-in for /root/reference/ModelComponents/boosted_model.py:12-282).

Per weak learner i: one encoder block with its own positional variable (EncoderTransformerBlocks[i]),
the shared DecoderPrep (queries re-tiled every learner, 210-211), decoder block i, three heads with
hidden = decoder_dim (113-137); predictions are accumulated with learner 0 counted twice (222-229) and
the matcher + loss run on the cumulative predictions of every learner (232-243).
"""
from __future__ import annotations

from . import backbone, losses_and_metrics, ops, prediction_heads, tokenizers, transformers
from .model import DETR, _image, _prepare_targets
from .training import Model


class BoostedDETR(Model):
    def __init__(self, num_object_preds, image_size, num_encoder_blocks, num_encoder_heads, encoder_dim,
                 num_decoder_blocks, num_decoder_heads, decoder_dim, num_panoptic_heads=1, panoptic_dim=32, vocab_dict=None,
                 classification_only=False, attribute_weight=1.0, name="DETR", use_intermediate_predictions=True, **kwargs):
        seed = int(kwargs.pop("seed", 0))
        backbone_name = kwargs.pop("backbone_name", "ResNet")
        super().__init__(name=name, seed=seed)
        category_weight = box_weight = exist_weight = None
        if classification_only:
            box_weight = 0.0
        self.num_object_preds = num_object_preds
        self.image_size = tuple(image_size)
        self.num_encoder_blocks, self.num_encoder_heads, self.encoder_dim = num_encoder_blocks, num_encoder_heads, encoder_dim   # num_encoder_blocks is ignored (86-92)
        self.num_decoder_blocks, self.num_decoder_heads, self.decoder_dim = num_decoder_blocks, num_decoder_heads, decoder_dim
        self.num_panoptic_heads, self.panoptic_dim = num_panoptic_heads, panoptic_dim
        self.vocab_dict = vocab_dict

        self.Tokenization = tokenizers.Tokenization(vocab_dict=vocab_dict, name="Tokenization")
        self.InverseTokenization = tokenizers.InverseTokenization(vocab_dict=vocab_dict)
        sizes = self.Tokenization.vocab_size_dict()
        self.num_categories, self.num_attributes = sizes["category"], sizes["attributes"]

        self.EncoderBackbone = backbone.EncoderBackbone(image_input_shape=self.image_size, model_name=backbone_name, name="EncoderBackbone", seed=seed)
        self.BackboneNeck = backbone.BackboneNeck(encoder_dim=encoder_dim, name="BackboneNeck", seed=seed)
        self.EncoderTransformerBlocks = [transformers.ImageEncoderAttention(num_blocks=1, num_attention_heads=num_encoder_heads,
                                                                            name=f"ImageEncoderAttention_{i}", seed=seed)
                                         for i in range(num_decoder_blocks)]
        self.DecoderPrep = transformers.DecoderPrep(num_object_preds, decoder_dim, name="DecoderPrep", seed=seed)
        self.DecoderBlocks = [transformers.DecoderBlock_NoSelfAttention(num_attention_heads=num_decoder_heads, name="DecoderBlock_0", seed=seed)]
        for i in range(1, num_decoder_blocks):
            self.DecoderBlocks.append(transformers.DecoderBlock(num_attention_heads=num_decoder_heads, name=f"DecoderBlock_{i}", seed=seed))
        self.CategoryBlocks, self.AttributeBlocks, self.BoxBlocks = [], [], []
        for i in range(num_decoder_blocks):
            self.CategoryBlocks.append(prediction_heads.SingleClassPredictionHead(num_classes=self.num_categories, hidden_dim=decoder_dim,
                                                                                  num_preds=num_object_preds, name=f"CategoryPredictionHead_{i}", seed=seed))
            self.AttributeBlocks.append(prediction_heads.MultiClassPredictionHead(num_classes=self.num_attributes, hidden_dim=decoder_dim,
                                                                                  num_preds=num_object_preds, name=f"AttributePredictionHead_{i}", seed=seed))
            self.BoxBlocks.append(prediction_heads.BoxPredictionHead(hidden_dim=decoder_dim, num_preds=num_object_preds,
                                                                     name=f"BoxPredictionHead_{i}", seed=seed))
        for group in (self.EncoderTransformerBlocks, self.DecoderBlocks, self.CategoryBlocks, self.AttributeBlocks, self.BoxBlocks):
            for l in group:
                self.track(l)
        self.loss_fn = losses_and_metrics.MatchingLoss(category_weight=category_weight, box_weight=box_weight,
                                                       attribute_weight=attribute_weight, exist_weight=exist_weight, name="MatchingLoss")

    get_config = DETR.get_config
    _register = DETR._register

    def call(self, inputs, training=False):
        focused_training_layer = None          # hard-coded in the reference (boosted_model.py:171)
        image = _image(inputs)
        if training:
            y_true = _prepare_targets(self, inputs)
        encoder_features = self.EncoderBackbone([image], training=training)
        encoder_features = self.BackboneNeck([encoder_features], training=training)

        loss_terms, metrics_i = [], None
        cat_preds = attribute_preds = box_coord_preds = None
        for i in range(self.num_decoder_blocks):
            # (the reference reshapes the carried features back to [B,r,c,D]; they already are)
            encoder_features, positional_encoding = self.EncoderTransformerBlocks[i]([encoder_features], training=training)
            enc_value, decoder_features, encoder_key, decoder_positional = \
                self.DecoderPrep([encoder_features, positional_encoding], training=training)
            decoder_features = self.DecoderBlocks[i]([enc_value, decoder_features, encoder_key, decoder_positional], training=training)
            cat_preds_i = self.CategoryBlocks[i]([decoder_features], training=training)
            attribute_preds_i = self.AttributeBlocks[i]([decoder_features], training=training)
            box_coord_preds_i = self.BoxBlocks[i]([decoder_features], training=training)
            if i == 0:                             # learner 0 is counted twice (222-229)
                cat_preds, attribute_preds, box_coord_preds = cat_preds_i, attribute_preds_i, box_coord_preds_i
            cat_preds = ops.add(cat_preds, cat_preds_i)
            attribute_preds = ops.add(attribute_preds, attribute_preds_i)
            box_coord_preds = ops.add(box_coord_preds, box_coord_preds_i)
            y_pred = [cat_preds, attribute_preds, box_coord_preds]
            if training and (i == focused_training_layer or focused_training_layer is None):
                losses_i, metrics_i = self.loss_fn([y_true, y_pred])
                loss_terms.append(losses_i)
                self._loss_roots.append(self.loss_fn._losses_tensor)
            if i == focused_training_layer:
                break

        if training:
            self._register(loss_terms, metrics_i)
            return y_pred
        category, attributes = self.InverseTokenization([cat_preds, attribute_preds], training=training)
        return category, attributes, box_coord_preds

    def citation(self):
        print("Boosted-ensemble adaptation of DETR for object detection and fine-grained classification; "
              "MI355X-native re-implementation of the mvenouziou/Boosted_DETR training path.")
