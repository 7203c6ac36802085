"""Input step in front of the hot path.

Mirrors the parts of /root/reference/ModelComponents/pipeline.py that shape the model's input dict:
``Pipeline.data_generator``'s padding contract (131-186: '<PAD>' strings, -10 boxes, num_objects) and
``Augmentations`` (260-341).  The reference builds a tf.data graph; here batches are plain dicts and
the image work runs in csrc/augment.hip.  Image decoding, TFRecords and the dataset downloaders stay
out of scope.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import kernels as K
from .engine import to_device

PAD = "<PAD>"
BOX_PAD = -10.0


def coco_records(coco: dict, normalise: bool = True) -> List[dict]:
    """COCO-format annotations (``images``, ``annotations``, ``categories``) -> one record per image:
    {'image_id', 'file_name', 'width', 'height', 'category': [[name], ...], 'bbox': [[x,y,w,h], ...]}
    with boxes divided by [W,H,W,H] (README.md:131-158: normalised COCO format) - the per-image
    grouping datasets.py:340-516 performs with pandas."""
    names = {c["id"]: c["name"] for c in coco["categories"]}
    by_image: Dict[int, dict] = {}
    for im in coco["images"]:
        by_image[im["id"]] = {"image_id": im["id"], "file_name": im.get("file_name", ""), "width": im["width"], "height": im["height"],
                              "category": [], "attribute": [], "bbox": []}
    for a in coco["annotations"]:
        rec = by_image[a["image_id"]]
        w, h = float(rec["width"]), float(rec["height"])
        x, y, bw, bh = a["bbox"]
        rec["bbox"].append([x / w, y / h, bw / w, bh / h] if normalise else [x, y, bw, bh])
        rec["category"].append([names[a["category_id"]]])
        rec["attribute"].append([str(t) for t in a.get("attribute_names", [])] or [PAD])
    return list(by_image.values())


def pad_annotations(records: Sequence[dict], max_objects: Optional[int] = None) -> Dict[str, np.ndarray]:
    """pipeline.py:139-181: ragged per-image lists -> uniform arrays.  category [B,M,1] (pad '<PAD>'),
    attribute [B,M,Amax] (pad '<PAD>'), bbox [B,M,4] (pad -10), num_objects [B]."""
    B = len(records)
    n = [len(r.get("bbox", [])) for r in records]
    M = max_objects if max_objects is not None else max(max(n), 1)
    amax = max([len(a) for r in records for a in r.get("attribute", [])] + [1])
    category = np.full((B, M, 1), PAD, dtype=object)
    attribute = np.full((B, M, amax), PAD, dtype=object)
    bbox = np.full((B, M, 4), BOX_PAD, np.float32)
    for b, r in enumerate(records):
        for m in range(min(n[b], M)):
            category[b, m, 0] = r["category"][m][0]
            atts = r.get("attribute", [[PAD]] * n[b])[m]
            attribute[b, m, :len(atts)] = atts
            bbox[b, m] = r["bbox"][m]
    return {"category": category, "attribute": attribute, "bbox": bbox, "num_objects": np.minimum(np.asarray(n, np.int32), M)}


class Augmentations:
    """pipeline.py:260-341 on the GPU.  ``apply_image_augmentations`` maps over an iterable of batch dicts
    exactly like the reference maps over a tf.data.Dataset (downsizer -> contrast -> brightness ->
    jpeg quality -> saturation); the per-image random draws come from a seeded NumPy generator (TF's
    stream is not reproducible) and can be injected for tests.  jpeg_quality=False leaves the JPEG
    round trip out (pipeline.py:319-325: quality uniform in [70, 100))."""

    image_key, bbox_key = "image", "bbox"

    def __init__(self, seed: int = 0, jpeg_quality: bool = True):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.jpeg_quality = jpeg_quality

    def draw(self, B: int, H: int, W: int) -> Dict[str, np.ndarray]:
        # rand_val = max(1, truncated_normal(mean .5, std .7)): mostly 1 (no down-size), up to ~1.9
        tn = self.rng.standard_normal((B, 2))
        bad = np.abs(tn) > 2.0
        while bad.any():
            tn[bad] = self.rng.standard_normal(int(bad.sum()))
            bad = np.abs(tn) > 2.0
        rand_val = np.maximum(1.0, 0.5 + 0.7 * tn).astype(np.float32)
        new_h = (np.float32(H) / rand_val[:, 0]).astype(np.int32)
        new_w = (np.float32(W) / rand_val[:, 1]).astype(np.int32)
        off_h = self.rng.integers(0, H - new_h + 1).astype(np.int32)
        off_w = self.rng.integers(0, W - new_w + 1).astype(np.int32)
        return {"rand_val": rand_val, "new_h": new_h, "new_w": new_w, "off_h": off_h, "off_w": off_w,
                "contrast": self.rng.uniform(0.8, 1.2, B).astype(np.float32),
                "brightness": self.rng.uniform(-0.1, 0.1, B).astype(np.float32),
                "saturation": self.rng.uniform(0.8, 1.2, B).astype(np.float32),
                "jpeg_quality": self.rng.integers(70, 100, B).astype(np.int32)}

    @staticmethod
    def adjust_boxes(bbox: np.ndarray, p: Dict[str, np.ndarray], H: int, W: int) -> np.ndarray:
        """pipeline.py:302-313, quirks included: the COCO [x,y,w,h] box is divided by [r_h, r_w, r_h, r_w]
        and the normalised [off_h, off_w, off_h, off_w] is added to all four entries (also to w and h,
        and also to the -10 padding rows)."""
        rv = p["rand_val"]
        denom = np.stack([rv[:, 0], rv[:, 1], rv[:, 0], rv[:, 1]], axis=-1)[:, None, :]
        oh = (p["off_h"] / np.float64(H)).astype(np.float32)
        ow = (p["off_w"] / np.float64(W)).astype(np.float32)
        shift = np.stack([oh, ow, oh, ow], axis=-1)[:, None, :]
        return (bbox / denom + shift).astype(np.float32)

    def apply(self, batch: dict, params: Optional[Dict[str, np.ndarray]] = None) -> dict:
        image = batch[self.image_key]
        image = to_device(image if isinstance(image, torch.Tensor) else np.asarray(image, np.float32))
        B, H, W, _ = image.shape
        p = params if params is not None else self.draw(B, H, W)
        ip = to_device(np.stack([p["new_h"], p["new_w"], p["off_h"], p["off_w"]], axis=-1).astype(np.int32), torch.int32)
        fp = to_device(np.stack([p["contrast"], p["brightness"], p["saturation"]], axis=-1).astype(np.float32))
        out = dict(batch)
        q = to_device(np.asarray(p["jpeg_quality"], np.int32), torch.int32) if (self.jpeg_quality and "jpeg_quality" in p) else None
        out[self.image_key] = K.augment(image, ip, fp, q)
        if self.bbox_key in batch:
            out[self.bbox_key] = self.adjust_boxes(np.asarray(batch[self.bbox_key], np.float32), p, H, W)
        return out

    def apply_image_augmentations(self, dataset: Iterable[dict]) -> Iterator[dict]:
        for batch in dataset:
            yield self.apply(batch)
