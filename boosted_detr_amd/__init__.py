"""MI355X-native DETR training-step hot path behind the Keras-style surface of
mvenouziou/Boosted_DETR (DETR / BoostedDETR: build()/call()/compile()/fit()).

Importing the package does not touch the GPU; the first kernel call loads
``csrc/libbdetr.so`` (``python -m boosted_detr_amd.build`` builds it) and raises if it is missing.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "DETR":
        from .model import DETR
        return DETR
    if name == "BoostedDETR":
        from .boosted_model import BoostedDETR
        return BoostedDETR
    if name in ("SGD", "CosineDecayRestarts", "ModelCheckpoint", "TerminateOnNaN", "TensorBoard", "latest_checkpoint"):
        from . import training
        return getattr(training, name)
    raise AttributeError(name)
