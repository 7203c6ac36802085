"""Default hyper-parameters and vocabularies (data carried from
/root/reference/ModelComponents/parameters.py:99-177; the code around them - Colab file paths,
tf.distribute strategy selection - is outside the hot path).

COCO: 80 categories / 1 attribute ('<none>'); Fashionpedia: 46 categories / 294 attributes.  The
Fashionpedia attribute strings are dataset metadata that is not needed for the synthetic benchmark;
``synthetic_vocab`` generates a vocabulary of the same sizes.
"""
from __future__ import annotations

PAD, OOV = "<PAD>", "<OOV>"

COCO_CATEGORIES = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant",
    "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe",
    "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat",
    "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl",
    "banana", "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
    "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave",
    "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush",
]
COCO_VOCAB = {"attribute": ["<none>"], "category": COCO_CATEGORIES}


def synthetic_vocab(num_categories: int, num_attributes: int) -> dict:
    return {"category": [f"category_{i:03d}" for i in range(num_categories)],
            "attribute": [f"attribute_{i:03d}" for i in range(num_attributes)]}


FASHIONPEDIA_SIZES = {"category": 46, "attribute": 294}


class ModelParameters:
    def __init__(self, dataset_name="COCO"):
        self._num_object_preds = 96
        self._image_size = (560, 560)
        self._pad, self._oov = PAD, OOV
        self._dataset_name = dataset_name

    def dataset_name(self):
        return self._dataset_name

    def vocab_dict(self, name=None):
        d = {"COCO": COCO_VOCAB, "Fashionpedia": synthetic_vocab(**{"num_categories": 46, "num_attributes": 294})}
        return d[name] if name else d

    def default_vocab(self):
        return self.vocab_dict(self._dataset_name)

    def default_params(self, value=None):
        parameters = {"image_size": self._image_size, "encoder_dim": 256, "num_encoder_blocks": 4, "num_encoder_heads": 8,
                      "num_decoder_blocks": 4, "num_decoder_heads": 8, "decoder_dim": 256, "num_panoptic_heads": 1,
                      "panoptic_dim": 32, "num_object_preds": self._num_object_preds, "vocab_dict": self.default_vocab(),
                      "pad_value": self._pad, "oov_value": self._oov}
        return parameters[value] if value is not None else parameters
