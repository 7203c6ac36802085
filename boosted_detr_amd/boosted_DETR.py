"""Alias: the reference's Boosted notebook imports ``boosted_DETR`` although the file is
``boosted_model.py`` (Boosted_DETR_COCO.ipynb cell 4).  Both names are provided."""
from .boosted_model import BoostedDETR  # noqa: F401
