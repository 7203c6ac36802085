"""MI355X-native DETR training-step hot path behind the Keras-style surface of
mvenouziou/Boosted_DETR (DETR / BoostedDETR: build()/call()/compile()/fit()).

Importing the package does not touch the GPU; the first kernel call loads
``csrc/libbdetr.so`` (``python -m boosted_detr_amd.build`` builds it) and raises if it is missing.
"""
__version__ = "0.1.0"

# Replaying the training step as a chain of hipGraphs (Model.use_graph) is only sound with the ROCm 7.2 runtime's pre-built AQL
# packet path switched off: with it, the second replay of a graph without a stream synchronisation in between handed NaN gradients
# to the optimizer (tools/graph_segment_checksums.py locates the first bad segment, tools/probes/graph_replay_repro.hip
# is the torch-free attempt).  The runtime reads the switch ONCE, when it initialises - at the first HIP call of the process, which
# may come from places Python cannot see (a profiler's preloaded tool library, another extension's ctypes load).  Hence:
#   * importing this package changes nothing in the environment (round 3 set the switch as a side effect of the import);
#   * `enable_graph_replay()` is the explicit opt-in: call it before anything touches the GPU (bench.py does, first thing; setting
#     `Model.use_graph = True` calls it too).  It sets the switch when it is unset and nothing suggests that HIP is already up;
#   * `graph_replay_is_safe()` is True only when the switch was exported before the process started, or when enable_graph_replay()
#     set it at a point where neither torch had initialised CUDA nor a profiler preload was present.  Anything else -> Model runs
#     eager steps (one warning); there is no silent replay on the unsound path.
import os as _os
import sys as _sys

_PACKET_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_EXPORTED_AT_START = _os.environ.get(_PACKET_ENV)          # what the process was started with
_SET_IN_TIME = [False]


def _hip_may_be_live() -> bool:
    """Anything that suggests the HIP runtime has read its flags already."""
    t = _sys.modules.get("torch")
    if t is not None and t.cuda.is_initialized():
        return True
    # rocprofv3 / rocprof-sys preload a tool library that initialises the GPU before the interpreter starts
    pre = " ".join(_os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB"))
    return any(x in pre for x in ("rocprof", "roctracer", "rocprofiler"))


def enable_graph_replay() -> bool:
    """Opt in to hipGraph replay of the training step: put DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in force for this process if that is still
    possible.  Returns graph_replay_is_safe().  Idempotent; never overrides a value the user exported."""
    if _EXPORTED_AT_START is None and _os.environ.get(_PACKET_ENV) is None and not _hip_may_be_live():
        _os.environ[_PACKET_ENV] = "0"
        _SET_IN_TIME[0] = True
    return graph_replay_is_safe()


def graph_replay_is_safe() -> bool:
    """True when the HIP runtime of this process runs hipGraph launches without pre-built packets (see above)."""
    if _os.environ.get("BDETR_GRAPH_UNSAFE") == "1":          # diagnostics only (tools/graph_segment_checksums.py); bench.py refuses it
        return True
    if _EXPORTED_AT_START is not None:
        return _EXPORTED_AT_START == "0" and _os.environ.get(_PACKET_ENV) == "0"
    return _SET_IN_TIME[0] and _os.environ.get(_PACKET_ENV) == "0"


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
