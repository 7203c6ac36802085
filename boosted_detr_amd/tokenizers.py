"""Host-side string <-> id mapping.

Mirrors /root/reference/ModelComponents/tokenizers.py: Tokenization (5-88) and
InverseTokenization (91-163).  ``tf.keras.layers.StringLookup`` semantics: index 0 is
the mask token '<PAD>', index 1 the OOV token '<OOV>', the vocabulary starts at 2.  Strings never
reach the GPU: the device kernels consume int32 category ids and a multi-hot attribute matrix.
"""
from __future__ import annotations

import re
from typing import Sequence

import numpy as np
import torch

from .engine import Layer, to_device

PAD, OOV = "<PAD>", "<OOV>"


def _as_str(x) -> str:
    if isinstance(x, bytes):
        return x.decode("utf-8", errors="replace")
    return str(x)


class Tokenization(Layer):
    def __init__(self, vocab_dict, name="Tokenization", **kwargs):
        super().__init__(name=name, **kwargs)
        self.vocab_dict = vocab_dict
        self.mask_token, self.out_of_vocab_token = PAD, OOV
        self._cat_vocab = [PAD, OOV] + list(vocab_dict["category"])
        self._att_vocab = [PAD, OOV] + list(vocab_dict["attribute"])
        self._cat_index = {w: i for i, w in reversed(list(enumerate(self._cat_vocab)))}
        self._att_index = {w: i for i, w in reversed(list(enumerate(self._att_vocab)))}
        self._vocab_size_category = len(self._cat_vocab)
        self._vocab_size_attributes = len(self._att_vocab)
        self.built = True

    def get_config(self):
        c = super().get_config()
        c.update({"vocab_dict": self.vocab_dict})
        return c

    def vocab_size_dict(self):
        return {"category": self._vocab_size_category, "attributes": self._vocab_size_attributes}

    def lookup(self, arr, index) -> np.ndarray:
        a = np.asarray(arr)
        if a.dtype.kind in "iu":                       # already tokenised
            if a.size and (a.min() < 0 or a.max() >= len(index)):
                # the device kernels index prediction rows with these ids: an id outside the vocabulary is a caller
                # error (tf.one_hot would silently give an all-zero row; StringLookup never produces one)
                raise ValueError(f"token id outside the vocabulary [0, {len(index)}): min {a.min()}, max {a.max()}")
            return a.astype(np.int32)
        flat = [index.get(_as_str(s), 1) for s in a.reshape(-1)]
        return np.asarray(flat, np.int32).reshape(a.shape)

    def call(self, inputs, training=False):
        """[category [B,M,1] or [B,M], attributes [B,M,Amax]] (strings or ids) ->
        (category ids int32 [B,M] on device, multi-hot attributes f32 [B,M,A] on device).
        The one-hot category matrix of the reference (tokenizers.py:72) is never materialised:
        the loss kernels gather by id."""
        category, attributes = inputs
        if isinstance(category, torch.Tensor) and category.is_cuda:
            # ids already resident in HBM: range check + multi-hot scatter on the device, no host hop
            from . import kernels as K
            return K.tokens_prepare(category, attributes, self._vocab_size_category, self._vocab_size_attributes)
        cat = self.lookup(category, self._cat_index)
        if cat.ndim == 3:
            cat = cat[..., 0]                           # tf.squeeze(axis=2)
        att = self.lookup(attributes, self._att_index)
        A = self._vocab_size_attributes
        hot = np.zeros(att.shape[:2] + (A,), np.float32)
        b, m, s = np.meshgrid(*[np.arange(n) for n in att.shape], indexing="ij")
        hot[b.reshape(-1), m.reshape(-1), att.reshape(-1)] = 1.0       # one_hot + reduce_max over slots (PAD sets bit 0)
        return to_device(cat, torch.int32), to_device(hot)


class InverseTokenization(Layer):
    def __init__(self, vocab_dict, name="Tokenization", **kwargs):
        super().__init__(name=name, **kwargs)
        self.vocab_dict = vocab_dict
        self.mask_token, self.out_of_vocab_token = PAD, OOV
        self._cat_vocab = [PAD, OOV] + list(vocab_dict["category"])
        self._att_vocab = [PAD, OOV] + list(vocab_dict["attribute"])
        self._vocab_size_category = len(self._cat_vocab)
        self._vocab_size_attributes = len(self._att_vocab)
        self.built = True

    def get_config(self):
        c = super().get_config()
        c.update({"vocab_dict": self.vocab_dict})
        return c

    def vocab_size_dict(self):
        return {"category": self._vocab_size_category, "attributes": self._vocab_size_attributes}

    def token_ids(self, cat_preds: torch.Tensor, attribute_preds: torch.Tensor):
        """argmax category ids (first max on ties) and the >= 0.5 attribute indicator, on the host."""
        cat = cat_preds.detach().cpu().numpy()
        att = attribute_preds.detach().cpu().numpy()
        return cat.argmax(-1).astype(np.int64), att >= 0.5

    def call(self, inputs, training=False):
        cat_preds, attribute_preds = inputs
        ids, hot = self.token_ids(cat_preds, attribute_preds)
        B, N = ids.shape
        category = np.empty((B, N, 1), dtype=object)
        attributes = np.empty((B, N, 1), dtype=object)
        for b in range(B):
            for n in range(N):
                category[b, n, 0] = self._cat_vocab[ids[b, n]]
                # tokens = multihot * range(A): index 0 (<PAD>) wherever the indicator is 0 (tokenizers.py:134-137)
                words = [self._att_vocab[a] if hot[b, n, a] else PAD for a in range(hot.shape[-1])]
                s = ", ".join(words)
                s = s.replace(PAD, "").replace(OOV, "")
                s = s.replace(" ,", "")
                s = re.sub(r"\A, ", "", s)
                attributes[b, n, 0] = s.strip()
        return category, attributes
