"""Minimal Keras-style host runtime for the hot path: Variable, Tape (reverse-mode at
layer granularity), Layer (build/call/get_config/show_summary) and initialisers.

PyTorch tensors are device containers only; every FLOP runs in csrc/ kernels via
``kernels.py``.  There is no autograd dependency: backward closures are recorded on a Tape
while ``call(training=True)`` runs and replayed in reverse.

Mirrors the layer surface the reference builds on (``tf.keras.layers.Layer``,
/root/reference/ModelComponents/*.py): ``build(input_shape)``, ``call(list_of_tensors,
training=)``, ``get_config()``, ``show_summary()``, ``.trainable``.
"""
from __future__ import annotations

import math
import zlib
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import kernels as K

_DEVICE = None


def device() -> torch.device:
    global _DEVICE
    if _DEVICE is None:
        if not torch.cuda.is_available():
            raise RuntimeError("boosted_detr_amd needs an AMD GPU (HIP device); there is no CPU execution path")
        _DEVICE = torch.device("cuda", torch.cuda.current_device())
    return _DEVICE


def to_device(a, dtype=torch.float32) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=device(), dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device(), dtype=dtype).contiguous()


# ----------------------------------------------------------------------------------------
# variables
# ----------------------------------------------------------------------------------------
# Bumped whenever weight values change (optimizer step, assign): derived copies of a weight (the pre-split P16
# operand copies of the conv kernels, ops.packed_weights) are valid for one version only.
WEIGHTS_VERSION = [0]


def bump_weights_version() -> None:
    WEIGHTS_VERSION[0] += 1


# Bumped whenever the layer tree, a variable list or a trainable flag changes: Layer.variables / trainable_variables are
# cached per layer against it (the training step asks for them several times; walking ~400 variables with duplicate checks
# costs milliseconds of host time per call).
STRUCT_VERSION = [0]


def bump_struct_version() -> None:
    STRUCT_VERSION[0] += 1


class Variable:
    """A model weight.  ``value`` is held in the kernels' layout (conv OHWI, dense [out][in]);
    ``numpy()`` / ``assign()`` speak the Keras layout (HWIO, [in][out])."""

    def __init__(self, name: str, keras_shape: Sequence[int], kind: str = "vector", trainable: bool = True,
                 pad_in_channels: int = 0):
        self.name = name
        self.keras_shape = tuple(int(s) for s in keras_shape)
        self.kind = kind
        self._trainable = bool(trainable)
        self.pad_in_channels = pad_in_channels      # conv1: 3 -> 4 input channels (zero weights)
        self.value: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.grad_buf: Optional[torch.Tensor] = None   # persistent slice of the optimizer's flat gradient buffer
        self._grad_flat: Optional[torch.Tensor] = None  # ... and the flat buffer it is a slice of (ops.GradSink checks it is live)
        self._grad_fresh = False                        # grad_buf already holds this step's gradient

    @property
    def trainable(self) -> bool:
        return self._trainable

    @trainable.setter
    def trainable(self, v: bool) -> None:
        self._trainable = bool(v)
        bump_struct_version()          # cached trainable_variables lists are stale

    # layout conversion -----------------------------------------------------------------
    def _to_internal(self, a: np.ndarray) -> np.ndarray:
        a = np.asarray(a, np.float32)
        assert tuple(a.shape) == self.keras_shape, (self.name, a.shape, self.keras_shape)
        if self.kind == "conv_kernel":
            a = np.transpose(a, (3, 0, 1, 2))            # HWIO -> OHWI
            if self.pad_in_channels:
                a = np.concatenate([a, np.zeros(a.shape[:3] + (self.pad_in_channels,), np.float32)], axis=3)
        elif self.kind == "dense_kernel":
            a = a.T
        return np.ascontiguousarray(a)

    def _to_keras(self, a: np.ndarray) -> np.ndarray:
        if self.kind == "conv_kernel":
            if self.pad_in_channels:
                a = a[..., : a.shape[3] - self.pad_in_channels]
            a = np.transpose(a, (1, 2, 3, 0))
        elif self.kind == "dense_kernel":
            a = a.T
        return np.ascontiguousarray(a)

    def assign(self, keras_array) -> None:
        t = to_device(self._to_internal(np.asarray(keras_array)))
        bump_weights_version()
        if self.value is None:
            self.value = t
        else:
            self.value.copy_(t)        # keep the storage: optimizer tables hold raw pointers

    def numpy(self) -> np.ndarray:
        return self._to_keras(self.value.detach().cpu().numpy())

    def grad_numpy(self) -> np.ndarray:
        return self._to_keras(self.grad.detach().cpu().numpy())

    @property
    def num_params(self) -> int:
        return int(np.prod(self.keras_shape))

    @property
    def needs_grad(self) -> bool:
        return self.trainable and getattr(self, "owner", None) is not None and self.owner.trainable

    def reset_grad(self) -> None:
        self.grad, self._grad_fresh = None, False


# Keras initialisers -- host-side numpy, seeded per variable name
def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def initializer(kind: str) -> Callable:
    def trunc_normal(rng, shape, std):
        out = rng.standard_normal(shape)
        bad = np.abs(out) > 2.0
        while bad.any():
            out[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(out) > 2.0
        return (out * std / 0.87962566103423978).astype(np.float32)

    def init(name, shape, seed=0):
        rng = np.random.Generator(np.random.PCG64(_seed_for(name, seed)))
        fan_in, fan_out = _fans(shape) if len(shape) >= 2 else (shape[0], shape[0])
        if kind == "zeros":
            return np.zeros(shape, np.float32)
        if kind == "ones":
            return np.ones(shape, np.float32)
        if kind == "glorot_normal":
            return trunc_normal(rng, shape, math.sqrt(2.0 / (fan_in + fan_out)))
        if kind == "he_normal":
            return trunc_normal(rng, shape, math.sqrt(2.0 / fan_in))
        if kind == "lecun_normal":
            return trunc_normal(rng, shape, math.sqrt(1.0 / fan_in))
        if kind == "glorot_uniform":
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, size=shape).astype(np.float32)
        raise ValueError(kind)

    return init


# ----------------------------------------------------------------------------------------
# tape
# ----------------------------------------------------------------------------------------
def materialise(g):
    """A gradient tensor tagged ``_lazy_mask`` stands for tensor * mask (the skip gradient of a residual unit, handed on
    without applying the unit's ReLU bit mask: ops.conv_bn).  Ops that know the tag fold the mask into their own kernel;
    for everybody else the mask is applied in place here."""
    c = getattr(g, "_compact_even", None) if g is not None else None
    if c is not None:
        # A gradient that exists at the pixels (2i, 2j) only, held as the compact [N, H/2, W/2, C] tensor the stride-2 backward-data
        # products wrote (ops.conv_bn).  Its two regular consumers read it through the pixel map; anybody else gets the dense
        # zero-filled tensor, built here (a torch strided copy: the fallback, not the step's path) - a NEW tensor: callers re-bind.
        N, H, W = c
        d = torch.empty(N, H, W, g.shape[-1], dtype=g.dtype, device=g.device)
        K.zero_(d)                                   # (a kernel, not torch.zeros: a hipMemset node must not enter a captured step - SegmentedCapture.census)
        d[:, ::2, ::2] = g.view(N, H // 2, W // 2, g.shape[-1])
        d._bdetr_owned = True
        d._even_pixels = (N, H, W)
        if getattr(g, "_lazy_mask", None) is not None:
            d._lazy_mask = g._lazy_mask
        g = d
    m = getattr(g, "_lazy_mask", None)
    if m is not None:
        K.relu_mask_apply_(g, m)
        del g._lazy_mask
    return g


class Tape:
    """Records (outputs, inputs, backward_fn) triples; ``backward`` replays them in reverse and
    sums gradients that reach the same tensor (residual branches, shared keys/values)."""

    def __init__(self):
        self.nodes: List[tuple] = []

    def record(self, outputs: Sequence[torch.Tensor], inputs: Sequence[Optional[torch.Tensor]], fn: Callable) -> None:
        self.nodes.append((tuple(outputs), tuple(inputs), fn))

    def backward(self, seeds: Dict[int, torch.Tensor]) -> Dict[int, torch.Tensor]:
        grads: Dict[int, torch.Tensor] = dict(seeds)
        for outputs, inputs, fn in reversed(self.nodes):
            gouts = [grads.pop(id(o), None) for o in outputs]
            if all(g is None for g in gouts):
                continue
            if not getattr(fn, "accepts_lazy", False):
                gouts = [materialise(g) for g in gouts]
            acc = None
            if getattr(fn, "wants_acc", False):
                # offer the op the gradient tensors already accumulated for its inputs (only when the
                # Tape is their sole owner) so it can add into them inside its own kernel epilogue
                acc = []
                for t in inputs:
                    have = grads.get(id(t)) if t is not None else None
                    acc.append(have if getattr(have, "_bdetr_owned", False) else None)
                gins = fn(*gouts, acc=acc)
            else:
                gins = fn(*gouts)
            if not isinstance(gins, (tuple, list)):
                gins = (gins,)
            assert len(gins) == len(inputs), (len(gins), len(inputs))
            for i, (t, g) in enumerate(zip(inputs, gins)):
                if t is None or g is None:
                    continue
                key = id(t)
                if key in grads:
                    have = grads[key]
                    if acc is not None and acc[i] is not None and g is acc[i]:
                        continue                                    # the op already accumulated into the offered tensor
                    if acc is not None and acc[i] is not None and getattr(g, "_replaces_acc", False):
                        del g._replaces_acc                         # ... or folded it into a fresh tensor (a compact even-pixel gradient merged into a dense one)
                        grads[key] = g
                        continue
                    g, have = materialise(g), materialise(have)
                    grads[key] = have                               # (materialise re-binds a compact even-pixel tensor to its dense form)
                    # partial sums that rode in with `have` (a BatchNorm-backward reduction fused into the epilogue that
                    # produced it: ops.conv_bn) describe the tensor BEFORE this contribution: drop them
                    if hasattr(have, "_bnb_parts"):
                        del have._bnb_parts
                    if hasattr(have, "_bnb_parts_shortcut"):
                        del have._bnb_parts_shortcut
                    if hasattr(have, "_even_pixels"):      # (ops.conv_bn's tag of a gradient that is zero off the even pixels: no longer true)
                        del have._even_pixels
                    if getattr(have, "_bdetr_owned", False):       # sole owner: accumulate in place
                        K.axpy_(1.0, g.view(have.shape), have)
                    else:
                        s = K.add(have, g.view(have.shape))
                        s._bdetr_owned = True
                        grads[key] = s
                else:
                    grads[key] = g
            if _CAPTURE[0] is not None:
                # segmented graph capture: cut between tape nodes once enough side tasks are pending
                _CAPTURE[0].maybe_cut(next((g for g in gins if isinstance(g, torch.Tensor)), None) if _DEBUG_LOG[0] is not None else None)
            elif _DEBUG_LOG[0] is not None and _DEBUG_LOG[0].eager_pending >= SegmentedCapture.SIDE_TASKS_PER_SEGMENT:
                _debug_cut(next((g for g in gins if isinstance(g, torch.Tensor)), None))
        return grads


_TAPE: Optional[Tape] = None
_CAPTURE = [None]            # the SegmentedCapture in progress (defined below), or None

# ----------------------------------------------------------------------------------------
# side stream: weight-gradient GEMMs run off the critical path
# ----------------------------------------------------------------------------------------
# In the backward pass only the activation-gradient chain (BN-bwd -> bwd-data) is sequential; the
# weight-gradient GEMMs are consumed by nobody until the optimizer runs.  They are launched on a
# second HIP stream so that (a) their MFMA-bound workgroups fill the tails of the main stream's
# kernels and (b) they overlap the HBM-bound BN/LayerNorm backward kernels of the earlier layers.
import os as _os

_SIDE = {"stream": None, "enabled": _os.environ.get("BDETR_SIDE_STREAM", "1") != "0", "used": False, "keep": []}


def side_stream() -> Optional["torch.cuda.Stream"]:
    if not _SIDE["enabled"]:
        return None
    if _SIDE["stream"] is None:
        # lowest priority class: the dispatcher serves the critical path first, the leaves fill the rest
        # (torch.cuda.Stream only offers normal / high, so the stream comes from the C ABI)
        if _os.environ.get("BDETR_SIDE_PRIORITY", "low") == "low":
            import ctypes as _C
            from . import _lib
            for _ in range(int(_os.environ.get("BDETR_SIDE_QUEUE_SKIP", "0"))):      # (probe: shift the stream onto the next hardware queue)
                d = _C.c_void_p()
                _lib.check(_lib.lib().bdetr_low_priority_stream_create(_C.byref(d)), "low_priority_stream_create")
                _SIDE.setdefault("dummies", []).append(d)
            ncand = int(_os.environ.get("BDETR_SIDE_CANDIDATES", "4"))
            if ncand > 1 and not torch.cuda.is_current_stream_capturing():
                # Which hardware queue the stream lands on relative to the critical path's decides how the two overlap (the same step at
                # 25.1 / 25.3 / 25.2 / 44.3 ms on the four a low-priority stream can get, and what else created streams earlier - an RCCL
                # process group, say - shifts which one is next).  So a new stream is MEASURED against the current stream
                # (csrc/elementwise.hip bdetr_side_stream_candidates: the critical path's small kernels under the candidate's load, 0.75 ms
                # on a good queue against 4.6 on the bad one and 0.5 with no load) and only kept when it is good; a bad one is
                # followed by the next candidate, up to `ncand`.  ~5 ms per candidate, once per process.
                _SIDE["placement"] = {"picked": None, "tick_ms": [], "unloaded_ms": None, "good": [], "limit": ncand}
                _SIDE["candidates"] = []
                _more_candidates(1)
                while not _SIDE["placement"]["good"] and len(_SIDE["placement"]["tick_ms"]) < ncand:
                    _more_candidates(1)
                pl = _SIDE["placement"]
                pick = pl["good"][0] if pl["good"] else min(range(len(pl["tick_ms"])), key=lambda c: pl["tick_ms"][c])
                pl["picked"] = pick
                _SIDE["stream"] = _SIDE["candidates"][pick]
                side_stream_release()
            else:
                h = _C.c_void_p()
                _lib.check(_lib.lib().bdetr_low_priority_stream_create(_C.byref(h)), "low_priority_stream_create")
                _SIDE["stream"] = torch.cuda.ExternalStream(h.value, device=device())
        else:
            _SIDE["stream"] = torch.cuda.Stream(device=device())
    return _SIDE["stream"]


SIDE_BAD_RATIO = 3.0        # a candidate is good when the critical path's ticks under its load take < 3 x their unloaded time (measured: 1.5 x / 9 x)


def _more_candidates(n: int) -> None:
    """Create and measure `n` more low-priority candidates against the current stream (not inside a capture)."""
    import ctypes as _C
    from . import _lib
    pl = _SIDE["placement"]
    hs, scores, base = (_C.c_void_p * n)(), (_C.c_float * n)(), _C.c_float(0.0)
    _lib.check(_lib.lib().bdetr_side_stream_candidates(_C.c_void_p(torch.cuda.current_stream().cuda_stream), n, 120, 200, hs, scores, _C.byref(base)),
               "side_stream_candidates")
    if pl["unloaded_ms"] is None:
        pl["unloaded_ms"] = round(float(base.value), 3)
    for c in range(n):
        _SIDE["candidates"].append(torch.cuda.ExternalStream(hs[c], device=device()))
        pl["tick_ms"].append(round(float(scores[c]), 3))
        if float(scores[c]) < SIDE_BAD_RATIO * max(pl["unloaded_ms"], 1e-3):
            pl["good"].append(len(pl["tick_ms"]) - 1)


def side_stream_expand() -> list:
    """Measure candidates up to the limit (BDETR_SIDE_CANDIDATES, 4) and return the indices of the good ones - for a caller that wants to
    choose between them by timing real steps (training.Model under data parallelism).  Between steps, not inside a capture."""
    pl = _SIDE.get("placement")
    if pl is None or _SIDE["used"] or torch.cuda.is_current_stream_capturing():
        return [] if pl is None else [c for c in pl["good"] if _SIDE["candidates"][c] is not None]
    while len(pl["tick_ms"]) < pl["limit"]:
        _more_candidates(1)
    return [c for c in pl["good"] if _SIDE["candidates"][c] is not None]


def side_stream_release() -> None:
    """Destroy every candidate but the one in use (idle low-priority queues are not free: three of them measured 0.5 % on the step)."""
    from . import _lib
    if _SIDE["used"]:
        raise RuntimeError("side_stream_release: side-stream work is pending (join_side_stream first)")
    for c, st in enumerate(_SIDE.get("candidates", [])):
        if st is not None and st is not _SIDE["stream"]:
            st.synchronize()
            _lib.check(_lib.lib().bdetr_stream_destroy(st.cuda_stream), "stream_destroy")
            _SIDE["candidates"][c] = None


def side_stream_placement() -> Optional[dict]:
    """What side_stream()'s placement measurement saw: {"picked": index, "tick_ms": [per candidate measured so far], "unloaded_ms", "good":
    [indices]} (+ "step_ms" once a data-parallel model has timed steps on the good ones); None before the stream exists or with
    BDETR_SIDE_CANDIDATES=1."""
    return _SIDE.get("placement")


def side_stream_select(index: int) -> None:
    """Run the side work on candidate `index` from now on.  Only between steps: nothing may be pending on the current side stream (the
    caller has joined it).  Captured steps are not tied to it: their side graphs replay on whatever side_stream() returns at replay time."""
    if _SIDE["used"]:
        raise RuntimeError("side_stream_select: side-stream work is pending (join_side_stream first)")
    if _SIDE["candidates"][index] is None:
        raise RuntimeError(f"side_stream_select: candidate {index} was released")
    _SIDE["stream"] = _SIDE["candidates"][index]
    _SIDE["placement"]["picked"] = index


def set_side_stream_enabled(on: bool) -> None:
    _SIDE["enabled"] = bool(on)


class on_side_stream:
    """``with on_side_stream(t1, t2, ...)``: run the body on the side stream after everything already
    queued on the current stream.  The listed tensors are kept alive until ``join_side_stream`` has made
    the main stream wait for the side stream: their memory returns to the caching allocator only after
    that join, so no main-stream allocation can reuse it while a side-stream kernel still reads it.
    (``Tensor.record_stream`` gives the same guarantee but parks every such block until the allocator
    polls its event: the round-1 soak grew to 57 GiB reserved for 13 GiB live.)"""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.side = side_stream()

    def __enter__(self):
        if self.side is None:
            return self
        ev = torch.cuda.Event()
        ev.record()
        self.side.wait_event(ev)
        _SIDE["keep"].extend(self.tensors)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        self.prev_handle = K.set_launch_stream(self.side.cuda_stream if K._LAUNCH_STREAM[0] is not None else None)
        _SIDE["used"] = True
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            K.set_launch_stream(self.prev_handle)
            self.ctx.__exit__(*exc)
        return False


# ----------------------------------------------------------------------------------------
# segmented capture: the step as a CHAIN of hipGraphs with the weight-gradient work in graphs of its own
# ----------------------------------------------------------------------------------------
# One hipGraph of the whole step cannot keep the side stream: captured as a forked branch its ~110 main -> side edges made the
# replay twice as slow (round 2), and captured in stream order the weight-gradient GEMMs no longer overlap anything.  Instead the
# step is captured as main segments M_0 .. M_k (the critical path, cut between tape nodes of the backward pass) and side segments
# S_0 .. S_k (the weight / bias gradient tasks whose operands M_i produced).  Replay: M_i on the main stream, an event, S_i on
# the low-priority side stream behind that event - S_i overlaps M_{i+1} - and one join before the optimizer segment.  Two private
# memory pools: graphs that replay concurrently must not share one (an allocation freed during the capture of S_i could be handed
# to M_{i+1}); tensors that cross (the operands of the side tasks) are kept alive by the deferred closures until the capture ends.


class DebugLog:
    """Diagnostic of the graph-replay path (tools/graph_segment_checksums.py): order-independent fingerprints of the live flat gradient
    buffer (and of the activation gradient that crosses a cut) appended to a device-resident log at every segment boundary of the
    backward pass - captured INTO the segments under use_graph, enqueued at the same points of an eager step (which counts side
    tasks the way SegmentedCapture.maybe_cut does).  Needs a stream-ordered step: BDETR_SIDE_STREAM=0 / BDETR_GRAPH_SIDE=0."""
    TAGS = {"A": 1, "M": 2, "S": 3, "L": 4, "F": 5}

    def __init__(self, cap: int = 1 << 14):
        dev = device()
        self.cap = cap
        self.scratch = torch.zeros(2, dtype=torch.int64, device=dev)
        self.scratch_side = torch.zeros(2, dtype=torch.int64, device=dev)     # "S" entries may run on the side stream: their own scratch words
        self.log = torch.zeros(cap * 3, dtype=torch.int64, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.eager_pending = 0

    def emit(self, kind: str, t: Optional[torch.Tensor]) -> None:
        if t is None or not isinstance(t, torch.Tensor) or t.numel() == 0 or t.dtype != torch.float32 or not t.is_contiguous():
            t = self.scratch.view(torch.float32)[:1]           # keep the entry count the same in both modes
        from . import _lib
        sc = self.scratch_side if kind == "S" else self.scratch
        _lib.check(_lib.lib().bdetr_debug_checksum(t.data_ptr(), t.numel(), sc.data_ptr(), self.log.data_ptr(), self.cursor.data_ptr(),
                                                   self.cap, self.TAGS[kind], K._stream()), "debug_checksum")

    def entries(self):
        torch.cuda.synchronize()
        n = min(int(self.cursor.item()), self.cap)
        a = self.log[: 3 * n].view(n, 3).cpu().numpy()
        inv = {v: k for k, v in self.TAGS.items()}
        return [(inv.get(int(r[2]), "?"), int(r[0]) & 0xFFFFFFFFFFFFFFFF, int(r[1])) for r in a]


_DEBUG_LOG: List[Optional[DebugLog]] = [None]


def set_debug_log(log: Optional[DebugLog]) -> None:
    _DEBUG_LOG[0] = log


def _debug_cut(extra) -> None:
    """Eager twin of SegmentedCapture.cut() for the debug log: the three entries a captured cut appends."""
    log = _DEBUG_LOG[0]
    from . import ops
    log.emit("F", K.overflow_flag().view(torch.float32))
    log.emit("A", extra)
    log.emit("M", ops._live_flat_grad[0])
    log.emit("S", ops._live_flat_grad[0])
    log.eager_pending = 0


class SegmentedCapture:
    SIDE_TASKS_PER_SEGMENT = int(_os.environ.get("BDETR_GRAPH_SEG", "10"))
    # hipStreamCaptureModeThreadLocal, not torch's default "global".  Under the global mode HIP refuses capture-unsafe calls from EVERY
    # thread of the process while a capture is open, and torch's ProcessGroupNCCL watchdog thread polls hipEventQuery on the end events of
    # eager collectives that are still on its list (it retires them on its own ~100 ms tick): a query that lands inside a segment's
    # capture returns hipErrorStreamCaptureUnsupported, the watchdog throws on its own thread and the process aborts (round 4:
    # profiles/r04_sigabrt_capture_vs_rccl_watchdog.log).  All launches of a step come from the one thread that opened the capture, so the
    # thread-local mode checks exactly what has to be checked; tests/test_dp_gpu.py holds the race open deterministically (a thread
    # that queries an event throughout a capture).
    CAPTURE_ERROR_MODE = _os.environ.get("BDETR_CAPTURE_MODE", "thread_local")

    def __init__(self):
        self.pool_main, self.pool_side = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        self.cap_main, self.cap_side = torch.cuda.Stream(device=device()), torch.cuda.Stream(device=device())
        self.mains, self.sides = [], []            # sides[i] (or None) runs behind mains[i]
        self.pending, self.done = [], []           # deferred side tasks of the open segment / closures kept alive until the end
        self._cur = None
        self.in_side = False

    # BDETR_GRAPH_CENSUS=1 (and tests): keep every captured hipGraph_t and count its nodes by type after the capture; anything but
    # kernel nodes (type 0) and empty / event-record / wait-event nodes of the capture's own joins (types 4, 7, 8) fails the capture.
    # A memset or memcpy node is what made a replay diverge in round 4 (bdetr_graph_node_census, include/bdetr.h).
    CENSUS = _os.environ.get("BDETR_GRAPH_CENSUS", "0") == "1"
    ALLOWED_NODE_TYPES = (0, 4, 7, 8)

    def _new_graph(self):
        return torch.cuda.CUDAGraph(keep_graph=True) if self.CENSUS else torch.cuda.CUDAGraph()

    def census(self) -> Dict[int, int]:
        """Node counts by hipGraphNodeType over every captured segment (needs CENSUS at capture time); raises on a disallowed type."""
        import ctypes
        from . import _lib
        total: Dict[int, int] = {}
        for g in list(self.mains) + [x for x in self.sides if x is not None]:
            counts = (ctypes.c_int64 * 16)()
            _lib.check(_lib.lib().bdetr_graph_node_census(ctypes.c_void_p(g.raw_cuda_graph()), counts, 16), "graph_node_census")
            for t, c in enumerate(counts):
                if c:
                    total[t] = total.get(t, 0) + int(c)
        bad = {t: c for t, c in total.items() if t not in self.ALLOWED_NODE_TYPES}
        if bad:
            raise RuntimeError(f"captured step holds non-kernel graph nodes {bad} (hipGraphNodeType: count); 1 = memcpy, 2 = memset: "
                               "these replay unsoundly on this runtime - find the torch op that lowered to them")
        return total

    def begin_main(self) -> None:
        g = self._new_graph()
        ctx = torch.cuda.graph(g, pool=self.pool_main, stream=self.cap_main, capture_error_mode=self.CAPTURE_ERROR_MODE)
        ctx.__enter__()
        self._cur = (g, ctx)
        K.set_launch_stream(self.cap_main.cuda_stream)

    def end_main(self) -> None:
        import warnings
        g, ctx = self._cur
        with warnings.catch_warnings():
            # a segment may be empty (the cut that closes the backward pass can directly follow a cut by count); an empty graph replays as a no-op
            warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
            ctx.__exit__(None, None, None)
        self.mains.append(g)
        self._cur = None

    def abort(self) -> None:
        """A step failed while a main segment was being captured: end that capture (its graph is dropped) so that the stream leaves
        capture mode; the exception that caused it propagates from the caller."""
        if self._cur is not None:
            import sys
            _, ctx = self._cur
            self._cur = None
            try:
                ctx.__exit__(*sys.exc_info())
            except Exception:
                pass
        self.pending, self.done = [], []

    def capture_side(self) -> None:
        if not self.pending:
            self.sides.append(None)
            return
        g = self._new_graph()
        with torch.cuda.graph(g, pool=self.pool_side, stream=self.cap_side, capture_error_mode=self.CAPTURE_ERROR_MODE):
            prev = K.set_launch_stream(self.cap_side.cuda_stream)
            self.in_side = True                   # (a data-parallel bucket completed by one of these tasks is captured here, inline)
            try:
                k = 0
                while k < len(self.pending):      # a task may append one (training.DataParallel._launch from a main-segment gradient)
                    self.pending[k]()
                    k += 1
                if _DEBUG_LOG[0] is not None:
                    from . import ops
                    _DEBUG_LOG[0].emit("S", ops._live_flat_grad[0])
            finally:
                self.in_side = False
                K.set_launch_stream(prev)
        self.sides.append(g)
        self.done.extend(self.pending)
        self.pending = []

    def cut(self, extra=None) -> None:
        """Close the open main segment, capture its side tasks, open the next main segment."""
        if _DEBUG_LOG[0] is not None:
            from . import ops
            _DEBUG_LOG[0].emit("F", K.overflow_flag().view(torch.float32))
            _DEBUG_LOG[0].emit("A", extra)
            _DEBUG_LOG[0].emit("M", ops._live_flat_grad[0])
            if not self.pending:
                _DEBUG_LOG[0].emit("S", ops._live_flat_grad[0])       # (no side graph for this segment: keep three entries per cut)
        self.end_main()
        self.capture_side()
        self.begin_main()

    def maybe_cut(self, extra=None) -> None:
        if len(self.pending) >= self.SIDE_TASKS_PER_SEGMENT:
            self.cut(extra)

    def replay(self, side) -> None:
        """mains[i] on the current stream, sides[i] on `side` behind an event; the LAST main segment (optimizer) waits for the side stream.
        Needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (graph_replay_is_safe below)."""
        main = torch.cuda.current_stream()
        last = len(self.mains) - 1
        used = False
        for i, g in enumerate(self.mains):
            if i == last and used:
                main.wait_stream(side)
            g.replay()
            sg = self.sides[i] if i < len(self.sides) else None
            if sg is None:
                continue
            if side is None:                       # side stream switched off: the same graphs, in stream order
                sg.replay()
                continue
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                sg.replay()
            used = True


def side_task(fn, *keep) -> None:
    """Run `fn` (weight / bias gradient launches that nobody on the critical path consumes) on the side stream - or, while a step is
    being captured in segments, defer it into the side graph of the open segment."""
    cap = _CAPTURE[0]
    if cap is not None:
        cap.pending.append(fn)
        return
    if _DEBUG_LOG[0] is not None:
        _DEBUG_LOG[0].eager_pending += 1
    with on_side_stream(*keep):
        fn()


def join_side_stream() -> None:
    """Make the current stream wait for all side-stream work (called once at the end of backward)."""
    if _SIDE["stream"] is not None and _SIDE["used"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["used"] = False
    _SIDE["keep"].clear()          # after the join: frees are ordered behind the side stream's work


def current_tape() -> Optional[Tape]:
    return _TAPE


class recording:
    def __init__(self, tape: Optional[Tape]):
        self.tape = tape

    def __enter__(self):
        global _TAPE
        self.prev = _TAPE
        _TAPE = self.tape
        return self.tape

    def __exit__(self, *exc):
        global _TAPE
        _TAPE = self.prev
        return False


# ----------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------
class Layer:
    def __init__(self, name: Optional[str] = None, **kwargs):
        self.name = name or type(self).__name__
        self.built = False
        self._trainable = True
        self._variables: List[Variable] = []
        self._sublayers: List["Layer"] = []
        self._init_seed = int(kwargs.get("seed", 0))
        self.scope_prefix = kwargs.get("scope_prefix", "")   # parent scope, e.g. "DecoderBlock_1/"

    # attribute tracking (Keras-style: assigning a Layer or a list of Layers registers it)
    def __setattr__(self, key, value):
        if isinstance(value, Layer) and key not in ("_parent",):
            self.__dict__.setdefault("_sublayers", []).append(value)
            bump_struct_version()
        elif isinstance(value, list) and value and all(isinstance(v, Layer) for v in value):
            self.__dict__.setdefault("_sublayers", []).extend(value)
            bump_struct_version()
        object.__setattr__(self, key, value)

    def track(self, layer: "Layer") -> "Layer":
        """Register a sub-layer appended to a list attribute after the attribute was assigned."""
        if layer not in self._sublayers:
            self._sublayers.append(layer)
            bump_struct_version()
        return layer

    @property
    def trainable(self) -> bool:
        return self._trainable

    @trainable.setter
    def trainable(self, v: bool) -> None:
        self._trainable = bool(v)
        bump_struct_version()
        for l in self._sublayers:
            l.trainable = v

    def add_weight(self, name: str, shape, initializer_name: str = "zeros", kind: str = "vector", trainable: bool = True,
                   value: Optional[np.ndarray] = None, pad_in_channels: int = 0) -> Variable:
        v = Variable(f"{self.scope}/{name}", shape, kind=kind, trainable=trainable, pad_in_channels=pad_in_channels)
        v.assign(value if value is not None else initializer(initializer_name)(v.name, tuple(shape), self._init_seed))
        v.owner = self
        self._variables.append(v)
        bump_struct_version()
        return v

    scope_prefix = ""

    @property
    def scope(self) -> str:
        return f"{self.scope_prefix}{self.name}"

    def layers(self) -> List["Layer"]:
        out = []
        for l in self._sublayers:
            if l not in out:
                out.append(l)
        return out

    @property
    def variables(self) -> List[Variable]:
        c = self.__dict__.get("_vars_cache")
        if c is not None and c[0] == STRUCT_VERSION[0]:
            return list(c[1])
        out, seen = [], set()
        for v in self._variables:
            if id(v) not in seen:
                seen.add(id(v)); out.append(v)
        for l in self.layers():
            for v in l.variables:
                if id(v) not in seen:
                    seen.add(id(v)); out.append(v)
        object.__setattr__(self, "_vars_cache", (STRUCT_VERSION[0], out))
        return list(out)

    @property
    def trainable_variables(self) -> List[Variable]:
        c = self.__dict__.get("_tvars_cache")
        if c is not None and c[0] == STRUCT_VERSION[0]:
            return list(c[1])
        out = [v for v in self.variables if v.trainable and getattr(v, "owner", self).trainable]
        object.__setattr__(self, "_tvars_cache", (STRUCT_VERSION[0], out))
        return list(out)

    def count_params(self) -> int:
        return sum(v.num_params for v in self.variables)

    def build(self, input_shape) -> None:
        pass

    def call(self, inputs, training: bool = False):
        raise NotImplementedError

    def __call__(self, inputs, training: bool = False, **kwargs):
        if not self.built:
            shapes = [tuple(t.value.shape if isinstance(t, Variable) else t.shape) for t in inputs] \
                if isinstance(inputs, (list, tuple)) else inputs
            self.build(shapes)
            self.built = True
        out = self.call(inputs, training=training, **kwargs)
        if _DEBUG_LOG[0] is not None and training:
            # diagnostic (DebugLog): fingerprint every layer's first output tensor and the range-guard word in forward order
            t = out[0] if isinstance(out, (list, tuple)) and out else out
            if isinstance(t, torch.Tensor):
                _DEBUG_LOG[0].emit("L", getattr(t, "_p16f", t) if getattr(t, "_p16_only", False) else t)
                _DEBUG_LOG[0].emit("F", K.overflow_flag().view(torch.float32))
        return out

    def get_config(self) -> dict:
        return {"name": self.name, "trainable": self.trainable}

    def show_summary(self) -> str:
        lines = [f'Layer "{self.name}" ({type(self).__name__})']
        for v in self.variables:
            lines.append(f"  {v.name:80s} {str(v.keras_shape):>22s} {v.num_params:>12,d}{'' if v.trainable else '  (non-trainable)'}")
        lines.append(f"  total params: {self.count_params():,d}")
        text = "\n".join(lines)
        print(text)
        return text
