"""MI355X-native DETR training-step hot path behind the Keras-style surface of
mvenouziou/Boosted_DETR (DETR / BoostedDETR: build()/call()/compile()/fit()).

Importing the package does not touch the GPU; the first kernel call loads
``csrc/libbdetr.so`` (``python -m boosted_detr_amd.build`` builds it) and raises if it is missing.
"""
__version__ = "0.1.0"

# Replaying the training step as a chain of hipGraphs (Model.use_graph) is only sound with the ROCm 7.2 runtime's pre-built AQL
# packet path switched off: with it, the second replay of a graph without a stream synchronisation in between handed NaN gradients
# to the optimizer (tools/graph_debug.py; clean and equally fast with the switch off).  The runtime reads the switch
# when it initialises, i.e. at the first HIP call of the process, so it is set here, at import - and `graph_replay_is_safe` says
# whether that was early enough (or the user exported it); Model falls back to eager steps otherwise.
import os as _os
import sys as _sys

_PACKET_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_exported = _os.environ.get(_PACKET_ENV)
_torch = _sys.modules.get("torch")
_hip_live = bool(_torch is not None and _torch.cuda.is_initialized())
if _exported is None and not _hip_live:
    _os.environ[_PACKET_ENV] = "0"
GRAPH_REPLAY_SAFE = (_exported == "0") or (_exported is None and not _hip_live)


def graph_replay_is_safe() -> bool:
    """True when the HIP runtime of this process runs hipGraph launches without pre-built packets (see above)."""
    return GRAPH_REPLAY_SAFE and _os.environ.get(_PACKET_ENV) == "0"



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
