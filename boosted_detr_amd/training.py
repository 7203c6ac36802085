"""The sliver of the Keras training runtime the reference's notebooks use: Model
(compile / fit / train_step / test_step / save_weights / load_weights / summary), the SGD
(Nesterov, per-tensor clipnorm) optimizer, CosineDecayRestarts, callbacks, and the
data-parallel gradient all-reduce (RCCL through torch.distributed).

Reference usage being mirrored: DETR_COCO.ipynb cells 26, 30, 35 (compile(optimizer=...),
fit(ds, epochs, validation_data, callbacks=[ModelCheckpoint, TerminateOnNaN, TensorBoard]),
load_weights(latest_checkpoint)); semantics SURVEY S14/S15.
"""
from __future__ import annotations

import glob
import math
import os
import sys
import time
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from . import _lib
from . import kernels as K
from . import ops
from .engine import Layer, Tape, Variable, bump_weights_version, device, join_side_stream, recording, to_device


# ----------------------------------------------------------------------------------------
# learning-rate schedules
# ----------------------------------------------------------------------------------------
class CosineDecayRestarts:
    """tf.keras.optimizers.schedules.CosineDecayRestarts (notebook cell 26:
    CosineDecayRestarts(1e-3, 4000, m_mul=.95, alpha=.1))."""

    def __init__(self, initial_learning_rate, first_decay_steps, t_mul=2.0, m_mul=1.0, alpha=0.0):
        self.initial_learning_rate = float(initial_learning_rate)
        self.first_decay_steps = float(first_decay_steps)
        self.t_mul, self.m_mul, self.alpha = float(t_mul), float(m_mul), float(alpha)

    def __call__(self, step: int) -> float:
        completed = step / self.first_decay_steps
        if self.t_mul == 1.0:
            i_restart = math.floor(completed)
            completed -= i_restart
        else:
            i_restart = math.floor(math.log(1.0 - completed * (1.0 - self.t_mul)) / math.log(self.t_mul))
            sum_r = (1.0 - self.t_mul ** i_restart) / (1.0 - self.t_mul)
            completed = (completed - sum_r) / self.t_mul ** i_restart
        m_fac = self.m_mul ** i_restart
        cosine = 0.5 * m_fac * (1.0 + math.cos(math.pi * completed))
        return self.initial_learning_rate * ((1 - self.alpha) * cosine + self.alpha)


# ----------------------------------------------------------------------------------------
# optimizer
# ----------------------------------------------------------------------------------------
class SGD:
    """Keras SGD(momentum, nesterov=True, clipnorm) as one fused multi-tensor HIP launch
    (csrc/optim.hip).  Gradients live in one flat HBM buffer so the data-parallel all-reduce
    moves a few large buckets."""

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, clipnorm=None, name="SGD"):
        self.learning_rate = learning_rate
        self.momentum = float(momentum)
        self.nesterov = bool(nesterov)
        self.clipnorm = float(clipnorm) if clipnorm else 0.0
        self.iterations = 0
        self._built_for = None
        if not self.nesterov and self.momentum != 0.0:
            raise NotImplementedError("the hot path's optimizer is SGD(momentum, nesterov=True); plain momentum is not built")

    def current_lr(self) -> float:
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def release(self) -> None:
        """Detach the variables of the previous build from this optimizer's flat buffer."""
        for v in getattr(self, "vars", []):
            if getattr(v, "_grad_flat", None) is getattr(self, "flat_grad", None):
                v.grad_buf, v._grad_flat = None, None

    def build(self, variables: List[Variable]) -> None:
        dev = device()
        # momentum survives a rebuild for every variable that stays trainable (Keras keeps one slot variable per weight:
        # freezing the backbone and unfreezing it later, Boosted_DETR_COCO.ipynb cell 30, does not reset the others' velocity)
        # The velocity also survives a freeze -> unfreeze cycle: a variable that leaves the trainable set parks a copy of its velocity on
        # itself (v._momentum) and gets it back when it re-enters (keyed by the Variable object, not by id(): ids are reused after garbage
        # collection).
        old_mom = {}
        for v, m in zip(getattr(self, "vars", []), getattr(self, "mom_views", [])):
            old_mom[id(v)] = (v, m)
        staying = {id(v) for v in variables}
        for k, (v, m) in old_mom.items():
            if k not in staying:
                v._momentum = m.clone()
        self.release()
        self.vars = list(variables)
        sizes = [v.value.numel() for v in self.vars]
        offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])])      # 16-byte aligned slots
        self.flat_grad = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self.flat_mom = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self.grad_views = [self.flat_grad[int(o): int(o) + s].view(v.value.shape) for o, s, v in zip(offs[:-1], sizes, self.vars)]
        mom_views = [self.flat_mom[int(o): int(o) + s] for o, s in zip(offs[:-1], sizes)]
        for v, m in zip(self.vars, mom_views):
            have = old_mom.get(id(v))
            if have is not None and have[0] is v and have[1].numel() == m.numel():
                m.copy_(have[1])
            elif getattr(v, "_momentum", None) is not None and v._momentum.numel() == m.numel():
                m.copy_(v._momentum)                 # re-entering the trainable set: Keras kept its slot variable all along
            v._momentum = None
        self.mom_views = mom_views
        ptrs = np.zeros((len(self.vars), 3), np.uint64)
        for i, v in enumerate(self.vars):
            ptrs[i] = (v.value.data_ptr(), self.grad_views[i].data_ptr(), mom_views[i].data_ptr())
        slab = _lib.lib().bdetr_sgd_slab_elems()
        slab_tensor, slab_first = [], [0]
        for i, s in enumerate(sizes):
            n = (s + slab - 1) // slab
            slab_tensor += [i] * n
            slab_first.append(slab_first[-1] + n)
        self.nslabs = len(slab_tensor)
        self.d_ptrs = to_device(ptrs.view(np.int64).reshape(-1), torch.int64)
        self.d_sizes = to_device(np.asarray(sizes, np.int64), torch.int64)
        self.d_slab_tensor = to_device(np.asarray(slab_tensor, np.int64), torch.int64)
        self.d_slab_first = to_device(np.asarray(slab_first, np.int64), torch.int64)
        self.d_partial = torch.empty(self.nslabs, dtype=torch.float32, device=dev)
        self.d_norms = torch.empty(len(self.vars), dtype=torch.float32, device=dev)
        self.d_lr = torch.zeros(1, dtype=torch.float32, device=dev)
        for v, gv in zip(self.vars, self.grad_views):
            v.grad_buf, v._grad_flat = gv, self.flat_grad
        self._built_for = [id(v) for v in self.vars]

    def stage_gradients(self, variables: List[Variable]) -> None:
        """Copy the per-variable gradients produced by the backward kernels into the flat buffer."""
        if self._built_for != [id(v) for v in variables]:
            self.build(variables)
        for v, gv in zip(self.vars, self.grad_views):
            if v.grad is gv:
                continue                                 # the backward kernels wrote straight into the flat buffer
            if v.grad is None:
                gv.zero_()
            else:
                gv.copy_(v.grad.view(gv.shape))          # D2D memcpy (plumbing): first step / shared-variable temporaries
            v.grad = gv

    def stage_lr(self) -> None:
        self.d_lr.fill_(self.current_lr())

    def apply_gradients(self, grad_scale: float = 1.0, skip_flag: Optional[torch.Tensor] = None, stage_lr: bool = True) -> None:
        """skip_flag: device int32; while it is non-zero the update is not applied (the step's range guard).
        stage_lr=False: the caller already wrote this iteration's learning rate to the device (graph replays)."""
        if stage_lr:
            self.stage_lr()
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().bdetr_sgd_nesterov_clipnorm(
            self.d_ptrs.data_ptr(), self.d_sizes.data_ptr(), len(self.vars), self.d_slab_tensor.data_ptr(),
            self.d_slab_first.data_ptr(), self.nslabs, self.d_partial.data_ptr(), self.d_norms.data_ptr(),
            self.d_lr.data_ptr(), self.momentum, self.clipnorm, float(grad_scale),
            skip_flag.data_ptr() if skip_flag is not None else None, st), "sgd")
        bump_weights_version()
        self.iterations += 1


# ----------------------------------------------------------------------------------------
# data parallelism (one process per GPU; RCCL all-reduce of the flat gradient buffer)
# ----------------------------------------------------------------------------------------
class DataParallel:
    """Replicas keep per-replica BN statistics, matcher and normaliser exactly like the reference under
    MirroredStrategy (parameters.py:74); the only collective is the gradient all-reduce (SUM; the loss is already
    scaled by 1/R), issued through torch.distributed - backend "nccl" is RCCL over xGMI on ROCm.

    The flat gradient buffer is cut into ~32 MB buckets (few, large collectives: xGMI is point-to-point and a ring
    is per-link bound).  Buckets are numbered from the END of the buffer, i.e. in the order the backward pass
    completes them; ``grad_ready`` (called by ops.GradSink when a parameter gradient has been written in place)
    counts a bucket down and launches its all-reduce as soon as its last gradient has been enqueued, so the
    collectives overlap the rest of the backward pass.  Whatever was not launched early (first step, shared or
    temporary gradients) goes out in ``finish``."""

    BUCKET_ELEMS = 8 * 1024 * 1024      # 32 MB fp32 buckets
    TAIL_ELEMS = int(os.environ.get("BDETR_DP_TAIL_ELEMS", str(1024 * 1024)))      # ... except the last one to complete (prepare): 4 MB

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.overlap = os.environ.get("BDETR_DP_OVERLAP", "1") != "0"
        # a one-rank process group still runs every collective when asked to (tests/test_dp_gpu.py: the RCCL branch - comm
        # stream, event waits, async handles - executes on the one-GPU box over a real "nccl" communicator of size 1)
        self.active = dist.is_initialized() and (self.world > 1 or os.environ.get("BDETR_DP_FORCE", "0") == "1")
        self._flat = None
        self._expected = None            # contributions per variable and step (learnt from the first step on a flat buffer)
        self._comm_stream = None
        self.profile = False             # bench.py: time every bucket's all-reduce with events on the communication stream
        self._prof_events: List[tuple] = []
        self.prof_steps: List[dict] = []

    # -- plain path (also the gloo CPU tests) ---------------------------------------------------------------
    def allreduce_(self, flat: torch.Tensor) -> None:
        if not self.active:
            return
        handles = []
        for o in range(0, flat.numel(), self.BUCKET_ELEMS):
            handles.append(self.dist.all_reduce(flat[o: o + self.BUCKET_ELEMS], op=self.dist.ReduceOp.SUM, async_op=True))
        for h in handles:
            h.wait()

    # -- overlapped path ---------------------------------------------------------------------------------------
    def prepare(self, optimizer: "SGD") -> None:
        """Bucket table for the optimizer's current flat buffer: bucket b covers [lo, hi) elements, counted from the end."""
        flat = optimizer.flat_grad
        if self._flat is flat:
            return
        self._flat = flat
        self._expected = None          # contributions per variable and step, learnt from the first (non-overlapped) step
        n = flat.numel()
        self._bounds = []
        hi = n
        while hi > 0:
            lo = max(0, hi - self.BUCKET_ELEMS)
            self._bounds.append((lo, hi))
            hi = lo
        # The bucket at the START of the buffer holds the first layers' gradients - the last ones the backward pass completes (the stem's
        # weight gradient is its final kernel) - so its all-reduce cannot overlap anything: keep it SMALL.  Without this cut it is whatever
        # the 32-MB grid leaves (28 MB at config 2: ~0.3 ms of exposed ring time per step at N = 8); with it the exposed collective is 4 MB
        # and the rest of that bucket goes out ~5 ms earlier, when the backward pass reaches stage 3.  (Round 5; never measured at N > 1.)
        lo, hi = self._bounds[-1]
        if lo == 0 and hi > 2 * self.TAIL_ELEMS:
            self._bounds[-1:] = [(self.TAIL_ELEMS, hi), (0, self.TAIL_ELEMS)]
        base = flat.data_ptr()
        self._var_bucket, self._bucket_size = {}, [0] * len(self._bounds)
        for v in optimizer.vars:
            off = (v.grad_buf.data_ptr() - base) // 4
            b = next(i for i, (lo, hi_) in enumerate(self._bounds) if lo <= off < hi_)
            # a tensor that straddles a boundary belongs to the LATER-finishing (lower) bucket as well: count it in both
            last = off + v.grad_buf.numel() - 1
            b2 = next(i for i, (lo, hi_) in enumerate(self._bounds) if lo <= last < hi_)
            self._var_bucket[id(v)] = (b, b2)
            for k in {b, b2}:
                self._bucket_size[k] += 1

    def begin_step(self, optimizer: "SGD", main_stream, side_stream) -> None:
        self._active = False
        if not self.active or not self.overlap or getattr(optimizer, "flat_grad", None) is None:
            return
        self.prepare(optimizer)
        self._seen = {}
        self._pending = list(self._bucket_size)
        self._launched = [False] * len(self._bounds)
        self._handles = []
        self._main, self._side = main_stream, side_stream
        from . import engine as _engine
        capturing = _engine._CAPTURE[0] is not None
        if self._comm_stream is None and self._flat.is_cuda and not capturing:
            self._comm_stream = torch.cuda.Stream(device=self._flat.device)
        self._prof_events = []
        if self.profile and self._flat.is_cuda and not capturing:
            self._prof_base = torch.cuda.Event(enable_timing=True)
            self._prof_base.record(self._comm_stream)
        self._active = True

    def _launch(self, b: int, inline: bool = False) -> None:
        lo, hi = self._bounds[b]
        self._launched[b] = True
        from . import engine as _engine
        cap = _engine._CAPTURE[0]
        if cap is not None:
            # The step is being captured as a chain of hipGraphs (Model._graph_step): the collective is captured too - RCCL kernels
            # are graph nodes like any other.  A bucket completed by a main-segment kernel goes into the side graph that replays
            # behind that segment (cap.pending: it then overlaps the next main segment like the weight-gradient GEMMs do); one
            # completed by a side task is captured right there, inside the side graph; what finish() still has to send goes into the
            # optimizer segment itself.  No communication stream, no handles: the graph's edges order everything.
            flat, op = self._flat, self.dist.ReduceOp.SUM
            fn = lambda: self.dist.all_reduce(flat[lo:hi], op=op)
            if inline or getattr(cap, "in_side", False):
                fn()
            else:
                cap.pending.append(fn)
            self._captured_buckets = getattr(self, "_captured_buckets", 0) + 1
            return
        if self._flat.is_cuda:
            # the bucket's gradients were written on the main stream (normalisation / bias gradients) and on the side
            # stream (weight-gradient GEMMs): the collective waits for both, neither of them waits for it
            comm = self._comm_stream
            for st in (self._main, self._side):
                if st is not None:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    comm.wait_event(ev)
            with torch.cuda.stream(comm):
                e0 = None
                if self.profile:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(comm)          # behind the waits above: the bucket's gradients are complete when this fires
                self._handles.append(self.dist.all_reduce(self._flat[lo:hi], op=self.dist.ReduceOp.SUM, async_op=True))
                if self.profile:
                    # the collective runs on the backend's own stream: its end is visible on `comm` only through the handle
                    self._handles[-1].wait()
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(comm)
                    self._prof_events.append((e0, e1, (hi - lo) * 4))
        else:
            self._handles.append(self.dist.all_reduce(self._flat[lo:hi], op=self.dist.ReduceOp.SUM, async_op=True))

    def grad_ready(self, var: Variable) -> None:
        """One gradient contribution to `var` has been enqueued.  A variable may receive several per step (shared layers:
        the re-tiled decoder queries of BoostedDETR); its bucket may only go out after the last one, so the first step
        after (re)building the buffer runs without early launches and records how many each variable gets."""
        if not getattr(self, "_active", False):
            return
        k = id(var)
        self._seen[k] = self._seen.get(k, 0) + 1
        if self._expected is None:
            return
        bs = self._var_bucket.get(k)
        if bs is None or getattr(var, "_grad_flat", None) is not self._flat:
            return
        if self._seen[k] > self._expected.get(k, 0):
            if any(self._launched[b] for b in set(bs)):
                raise RuntimeError(f"{var.name}: gradient contribution after its bucket's all-reduce was launched "
                                   "(the model's graph changed between steps; set BDETR_DP_OVERLAP=0)")
            return
        if self._seen[k] < self._expected[k] or var.grad is not var.grad_buf:
            return
        for b in set(bs):
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch(b)

    def finish(self, flat: torch.Tensor) -> None:
        """All-reduce whatever ``grad_ready`` did not launch, then make the current stream wait for every bucket."""
        if not self.active:
            return
        if getattr(self, "_active", False) and flat is not self._flat:
            # The step was armed on another buffer (the optimizer was rebuilt between begin_step and here).  Buckets already
            # in flight reduce a retired buffer: wait for them, then refuse - reducing `flat` again on top would double-count
            # whatever was copied out of the old one.  Model.forward_backward rebuilds the optimizer BEFORE arming the step, so
            # this is a programming error, not a state a training run reaches.
            launched = any(self._launched)
            for h in self._handles:
                h.wait()
            self._handles, self._active = [], False
            if launched:
                raise RuntimeError("data-parallel step: the optimizer's gradient buffer changed while bucket all-reduces were in flight")
        if not getattr(self, "_active", False) or flat is not self._flat:
            self.allreduce_(flat)
            return
        self._active = False
        if self._expected is None:
            self._expected = dict(self._seen)
        from . import engine as _engine
        capturing = _engine._CAPTURE[0] is not None
        for b in range(len(self._bounds)):
            if not self._launched[b]:
                self._launch(b, inline=True)
        for h in self._handles:
            h.wait()
        if flat.is_cuda and not capturing and self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._handles = []
        if self.profile and self._prof_events:
            self.prof_steps.append({"base": self._prof_base, "buckets": self._prof_events})
            self._prof_events = []

    def drain(self) -> None:
        """Retire every collective this process has enqueued eagerly: wait on the handles still held, then idle the device.  Called once
        before a data-parallel step is captured (Model._graph_step): a capture must not begin with eager collectives in flight - their
        completion events are polled by the backend's watchdog thread (see engine.SegmentedCapture.CAPTURE_ERROR_MODE) and their
        kernels would run concurrently with the capture's allocator warm-up."""
        for h in getattr(self, "_handles", []):
            h.wait()
        self._handles = []
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def profile_summary(self) -> Optional[dict]:
        """Per-step all-reduce time on the communication stream (union of the buckets' [start, end] intervals: buckets queue
        behind each other) and bandwidths, over the steps recorded while `profile` was on.  Synchronises the device."""
        if not self.prof_steps:
            return None
        torch.cuda.synchronize()
        busy_ms, nbytes, nbuckets = 0.0, 0, 0
        for st in self.prof_steps:
            iv = sorted((st["base"].elapsed_time(e0), st["base"].elapsed_time(e1)) for e0, e1, _ in st["buckets"])
            end = -1.0
            for a, b in iv:
                a = max(a, end)
                if b > a:
                    busy_ms += b - a
                    end = b
            nbytes += sum(n for _, _, n in st["buckets"])
            nbuckets += len(st["buckets"])
        n = len(self.prof_steps)
        self.prof_steps = []
        alg = nbytes / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0
        return {"steps": n, "allreduce_ms_per_step": round(busy_ms / n, 4), "bytes_per_step": nbytes // n, "buckets_per_step": nbuckets // n,
                "algbw_GBps": round(alg, 2), "busbw_GBps": round(alg * 2.0 * (self.world - 1) / max(self.world, 1), 2),
                "note": "events on the communication stream around each ~32 MB bucket (start = its gradients complete, end = the handle's "
                        "wait); busbw = algbw * 2 (R - 1) / R (ring all-reduce convention)"}

    def broadcast_variables(self, variables: List[Variable]) -> None:
        """Replicas start from rank 0's values (weights AND moving statistics), like MirroredStrategy's mirrored
        variables (parameters.py:74)."""
        if not self.active:
            return
        for v in variables:
            self.dist.broadcast(v.value, src=0)

    def barrier(self) -> None:
        if self.active:
            self.dist.barrier()

    def any_(self, flag: torch.Tensor) -> None:
        """flag <- max over replicas (the range guard: a replica that skips its update must make all of them skip)."""
        if self.active:
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)


# ----------------------------------------------------------------------------------------
# callbacks
# ----------------------------------------------------------------------------------------
class Callback:
    def set_model(self, model): self.model = model
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class TerminateOnNaN(Callback):
    def on_batch_end(self, batch, logs=None):
        loss = (logs or {}).get("loss")
        if loss is not None and not math.isfinite(loss):
            print(f"Batch {batch}: Invalid loss, terminating training")
            if getattr(self.model, "train_gemm_precision", None) == "split":
                print("  (policy 'split' multiplies forward operands as fp16 halves: a forward activation beyond 65504 turns into NaN; "
                      "model.train_gemm_precision = 'mixed' keeps the forward on the exact-fp32 MFMA)")
            self.model.stop_training = True


class ModelCheckpoint(Callback):
    def __init__(self, filepath, save_weights_only=True, **kwargs):
        self.filepath, self.save_weights_only = filepath, save_weights_only

    def on_epoch_end(self, epoch, logs=None):
        path = self.filepath.format(epoch=epoch + 1, **(logs or {}))
        dp = getattr(self.model, "_dp", None)
        if dp is None or dp.rank == 0:              # replicas hold identical weights: one writer, the others wait
            self.model.save_weights(path)
        if dp is not None:
            dp.barrier()


class TensorBoard(Callback):
    """Writes scalar logs as JSON lines (the TF event format needs TensorFlow, which is absent)."""

    def __init__(self, log_dir="logs", **kwargs):
        self.log_dir = log_dir

    def on_epoch_end(self, epoch, logs=None):
        import json
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "scalars.jsonl"), "a") as f:
            f.write(json.dumps({"epoch": epoch, **{k: float(v) for k, v in (logs or {}).items()}}) + "\n")


def latest_checkpoint(checkpoint_dir: str) -> Optional[str]:
    files = sorted(glob.glob(os.path.join(checkpoint_dir, "*.safetensors")), key=os.path.getmtime)
    return files[-1] if files else None


# ----------------------------------------------------------------------------------------
# Model
# ----------------------------------------------------------------------------------------
class Model(Layer):
    def __init__(self, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.optimizer: Optional[SGD] = None
        self.stop_training = False
        self._step_losses: List[torch.Tensor] = []      # [B] vectors handed to add_loss
        self._loss_roots: List[torch.Tensor] = []       # tape roots whose backward seeds the step
        self._step_metrics: Dict[str, torch.Tensor] = {}
        self._dp: Optional[DataParallel] = None
        self.steps_done = 0
        # Arithmetic of the conv/GEMM family during a training step (include/bdetr.h): 'split' = split-fp16
        # forward + split-bf16 gradient products.  The fp16 halves need |operand| < 65504, which the batch-
        # normalised / layer-normalised training forward guarantees; inference and anything outside
        # forward_backward run under the library default ('mixed': exact fp32 forward).  The environment
        # variable BDETR_GEMM_PRECISION, when set, wins (None = leave the library's mode alone).
        self.train_gemm_precision = None if os.environ.get("BDETR_GEMM_PRECISION") else "split"
        self.train_grad_precision = None      # None: the backward pass runs under train_gemm_precision too
        self.validate_matching = False      # fit() turns this on: it synchronises every step anyway (host logging)
        # Range guard of the 'split' policy: the f16 pairs of its forward products hold |x| < 65504, the reference's
        # fp32 does not overflow there.  Producers raise a device flag instead of feeding NaN downstream; while it is
        # up the optimizer applies nothing and moving statistics stay put.  The host learns of it WITHOUT synchronising:
        # every step ends with a one-lane kernel that logs the flag in pinned memory, and a later step looks at the
        # entries that have landed (`_guard_poll`).  Every batch from the one that raised the flag on is then
        # redone on the exact-fp32 forward ('mixed') and the step / learning-rate counters are rolled back for the
        # update-free attempts, so no batch is lost and the schedule does not run ahead.
        self.guard_check_every = 1       # 0 disables the host side of the guard
        self.range_redos = 0             # guarded steps that were redone on the exact-fp32 forward
        self.range_skipped = 0           # ... how many update-free attempts that covered (counters rolled back for each)
        self._guard_count = 0
        self._guard_pending: List[tuple] = []      # (step's batch, its ordinal in the pinned flag log, the event recorded behind it)
        self._guard_was = None
        self._graphs, self._graph_warm = {}, {}
        self.use_graph = os.environ.get("BDETR_GRAPH", "0") == "1"      # capture train_step as a hipGraph (see _graph_step)
        self.range_redo_streak = 0       # consecutive guarded steps that had to be redone (see _guard_redo: persistent demotion)

    @property
    def use_graph(self) -> bool:
        return self.__dict__.get("_use_graph", False)

    @use_graph.setter
    def use_graph(self, on: bool) -> None:
        # (no environment write here: since round 4 the replay is sound on the runtime's default packet path - the captured chain holds
        # kernel nodes only; boosted_detr_amd.enable_graph_replay() remains as an explicit opt-in to DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)
        self.__dict__["_use_graph"] = bool(on)

    # -- Keras bookkeeping -----------------------------------------------------------------
    def add_loss(self, loss) -> None:
        self._step_losses.append(loss)

    def add_metric(self, value, name) -> None:
        self._step_metrics[name] = value

    @property
    def losses(self):
        return list(self._step_losses)

    @property
    def metrics_names(self):
        return ["loss"] + list(self._step_metrics)

    def compile(self, optimizer=None, **kwargs) -> None:
        if kwargs.get("loss") is not None:
            raise NotImplementedError("the reference compiles without a loss (losses are built into the model, model.py:208)")
        if self.optimizer is not None and self.optimizer is not optimizer:
            self.optimizer.release()                # no variable keeps a slice of the retired optimizer's gradient buffer
        for v in self.variables:
            if optimizer is None or getattr(v, "_grad_flat", None) is not getattr(optimizer, "flat_grad", None):
                v.grad_buf, v._grad_flat = None, None
        self.optimizer = optimizer
        self._graphs, self._graph_warm = {}, {}      # captured steps update through the retired optimizer's buffers

    def distribute(self) -> "Model":
        """Enable data parallelism over the initialised torch.distributed (RCCL) process group."""
        self._dp = DataParallel()
        self.loss_fn.loss_scale = 1.0 / self._dp.world     # S14: per-replica loss scaled by 1/num_replicas
        self._dp_synced = False                            # variables are broadcast from rank 0 once they exist (first step)
        self._side_tune = None                             # the side stream's placement is settled again under the collectives (_side_tune_begin)
        return self

    # -- one step ----------------------------------------------------------------------------
    def _step_seed(self) -> int:
        # per-step, per-replica dropout masks: replicas draw independent masks (as under MirroredStrategy)
        rank = self._dp.rank if self._dp is not None else 0
        return 0x5EED + self.steps_done + 0x9E3779B1 * rank

    def forward_backward(self, data: dict, stage_seed: bool = True, keep_tape: bool = False):
        """forward + matcher + loss + backward.  Leaves gradients in Variable.grad.
        keep_tape: keep the recorded tape so that ``replay_backward`` can run the backward pass again."""
        self._step_losses, self._loss_roots, self._step_metrics = [], [], {}
        ops.set_dropout_seed(self._step_seed(), write=stage_seed)
        for v in self.variables:
            v.reset_grad()
        live = None
        if self.optimizer is not None and getattr(self.optimizer, "flat_grad", None) is not None:
            # The trainable set changed since the buffer was built (layer.trainable = False / True between steps): rebuild
            # NOW, before the buffer is zeroed and before the data-parallel step is armed on it - otherwise early bucket
            # all-reduces would run on a buffer that stage_gradients is about to retire (and be repeated on the new one).
            tv_ids = [id(v) for v in self.trainable_variables]
            if self.optimizer._built_for is not None and self.optimizer._built_for != tv_ids:
                self.optimizer.build(self.trainable_variables)
            live = self.optimizer.flat_grad
            live.zero_()                         # ONE memset for all gradients (split-K GEMMs accumulate into zeros)
        ops.set_live_flat_grad(live)             # in-place gradient sinks are valid for slices of THIS buffer only
        guarded = self._guarded()
        if self._guard_was is not None and self._guard_was != guarded:
            K.overflow_flag().zero_()            # a flag left up by a step under another policy must not freeze this one's statistics
            self._guard_pending = []
        self._guard_was = guarded
        K.set_guard_active(guarded)              # BatchNorm statistics watch the flag only while somebody reads and clears it
        if self._dp is not None:
            from .engine import side_stream
            self._dp.begin_step(self.optimizer, torch.cuda.current_stream(), side_stream())
            ops.set_grad_ready_hook(self._dp.grad_ready)
        tape = Tape()
        prev = K.set_launch_stream(torch.cuda.current_stream().cuda_stream)     # pin the launch stream for the step
        try:
            with K.gemm_precision(self.train_gemm_precision):
                with recording(tape):
                    y_pred = self(data, training=True)
                # train_grad_precision: the backward pass under its own policy ('fp32' after a 'split' forward = every product of the step
                # at 2^-22 or better; the backward closures pick their kernels by the policy in force when they run: ops.conv_bn)
                with K.gemm_precision(self.train_grad_precision):
                    tape.backward({id(t): t for t in self._loss_roots})     # parameter gradients land in Variable.grad (ops.GradSink)
                    from . import engine as _engine
                    if _engine._CAPTURE[0] is not None:
                        _engine._CAPTURE[0].cut()        # segmented capture: close the backward's last segment (its side tasks need this step's sinks)
                    elif _engine._DEBUG_LOG[0] is not None:
                        _engine._debug_cut(None)         # (diagnostic twin of that cut in an eager step)
            self._kept_tape = tape if keep_tape else None
        finally:
            K.set_launch_stream(prev)
            ops.set_live_flat_grad(None)
            ops.set_grad_ready_hook(None)
        join_side_stream()                                      # weight-gradient GEMMs ran on the side stream
        return y_pred

    def _guarded(self) -> bool:
        return (self.train_gemm_precision or K.get_gemm_precision()) == "split"

    def replay_backward(self, gemm_precision: str) -> None:
        """Diagnostic: run the backward pass of the last ``forward_backward(..., keep_tape=True)`` again from the SAME
        saved forward (same activations, same ReLU / dropout masks, same match) under another GEMM arithmetic policy.
        Differences between two replays are then the arithmetic of the gradient products alone."""
        tape = getattr(self, "_kept_tape", None)
        if tape is None:
            raise RuntimeError("replay_backward needs forward_backward(data, keep_tape=True) first")
        for v in self.variables:
            v.reset_grad()
        live = None
        if self.optimizer is not None and getattr(self.optimizer, "flat_grad", None) is not None:
            live = self.optimizer.flat_grad
            live.zero_()
        ops.set_live_flat_grad(live)
        prev = K.set_launch_stream(torch.cuda.current_stream().cuda_stream)
        try:
            with K.gemm_precision(gemm_precision):
                tape.backward({id(t): t for t in self._loss_roots})
        finally:
            K.set_launch_stream(prev)
            ops.set_live_flat_grad(None)
        join_side_stream()

    # -- the step as a chain of hipGraphs ------------------------------------------------------------------
    # ~1500 kernel launches, ~1300 allocator calls and the Python tape make up 20 ms of host work per step.  With
    # ``use_graph`` the third step on a given input signature is captured (torch.cuda.graph: the caching allocator
    # hands the capture private pools, so every intermediate of the step lives at a fixed address) and later steps
    # copy the batch into the captured input tensors, write the two per-step scalars (dropout seed, learning rate)
    # to HBM and replay.  The reference's equivalent is tf.function / XLA (DETR_COCO.ipynb cell 3 enables the JIT).
    # Round 2 captured ONE graph in stream order (the side-stream branch as a forked capture cost 2x: ~110 cross-stream
    # edges) and lost the overlap of the weight-gradient GEMMs: 447 images/s against 477-490 eager.  Round 3 captures a chain
    # of graphs instead (engine.SegmentedCapture): the overlap is kept at segment granularity with a dozen cross-stream
    # events per step, and the step stays host-free.
    def _graph_signature(self, data: dict):
        if not self.use_graph or self.validate_matching:
            return None
        if self._dp is not None and (not getattr(self, "_dp_synced", True) or (self._dp.active and self._dp.overlap and self._dp._expected is None)):
            return None          # replicas not yet broadcast / the bucket table not yet calibrated (its first eager step): not now
        if self.side_tuning_pending():
            return None          # the side stream's placement is still being settled by timing eager steps (_side_tune_begin)
        from . import graph_replay_is_safe
        if not graph_replay_is_safe():
            if not getattr(self, "_graph_refused", False):
                import sys
                self._graph_refused = True
                print("[boosted_detr_amd] use_graph: BDETR_ZERO_MEMSET=1 puts hipMemset nodes back into the captured step; those are only sound with "
                      "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in force when the HIP runtime initialised (export it) - running eager steps", file=sys.stderr)
            return None
        if not all(isinstance(v, torch.Tensor) and v.is_cuda for v in data.values()):
            return None
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(data.items())) + (self.train_gemm_precision, self.train_grad_precision) + self._graph_env()

    def _graph_env(self) -> tuple:
        """Everything a captured step bakes in besides its input shapes: the optimizer object and the variables its flat
        buffers were built for, the trainable set, dropout rates and loss weights.  A change makes a new signature, i.e. a
        fresh capture (after two eager steps) - never a replay that keeps training frozen layers or updates a retired buffer."""
        from . import transformers
        lf = getattr(self, "loss_fn", None)
        loss = tuple(getattr(lf, k, None) for k in ("category_weight", "attribute_weight", "box_weight", "exist_weight", "loss_scale"))
        opt = self.optimizer
        hyper = (opt.momentum, opt.nesterov, opt.clipnorm) if opt is not None else ()
        dp = self._dp
        return (id(opt), (id(dp), dp.world, dp.active, dp.overlap) if dp is not None else None,
                tuple(getattr(opt, "_built_for", None) or ()), tuple(id(v) for v in self.trainable_variables),
                bool(self.guard_check_every), transformers.AttentionBlock.dropout_rate, transformers.FeedForwardBlock.dropout_rate, loss, hyper)

    def _device_step(self, data: dict, stage_scalars: bool) -> Dict[str, list]:
        """Everything of a training step that runs on the device; no host synchronisation."""
        self.forward_backward(data, stage_seed=stage_scalars)
        tv = self.trainable_variables
        self.optimizer.stage_gradients(tv)
        guard = None
        if self._guarded():
            for root in self._loss_roots:
                K.flag_nonfinite(root)
            guard = K.overflow_flag()
        if self._dp is not None:
            self._dp.finish(self.optimizer.flat_grad)       # buckets not already in flight since the backward pass + join
            if guard is not None:
                self._dp.any_(guard)
        self.optimizer.apply_gradients(skip_flag=guard, stage_lr=stage_scalars)
        self._guard_snapshot()
        self.steps_done += 1
        return self.step_logs()

    def _graph_step(self, data: dict, sig) -> Optional[Dict[str, list]]:
        entry = self._graphs.get(sig)
        if entry is None:
            n = self._graph_warm.get(sig, 0)
            self._graph_warm[sig] = n + 1
            if n < 2 or getattr(self.optimizer, "flat_grad", None) is None:
                return None                                  # eager: build-by-first-call, allocator warm-up, flat buffers
            static = {k: v.clone() for k, v in data.items()}
            ops.set_dropout_seed(self._step_seed())          # the captured kernels read both scalars from HBM
            self.optimizer.stage_lr()
            keep = (self.steps_done, self.optimizer.iterations)
            # The step is captured as a chain of graphs (engine.SegmentedCapture): main segments cut between tape nodes of the
            # backward pass, the weight-gradient tasks of each in a side graph that replays on the low-priority stream while
            # the next main segment runs, the optimizer in the last main segment behind the join.
            from . import engine as _engine
            if getattr(self, "_dp", None) is not None:
                self._dp.drain()                             # no eager collective in flight when the first segment's capture opens
            cap = _engine.SegmentedCapture()
            _engine._CAPTURE[0] = cap
            prev_launch = K.set_launch_stream(None)
            try:
                cap.begin_main()
                logs = self._device_step(static, stage_scalars=False)
                cap.end_main()
                cap.sides.append(None)                       # (the optimizer segment has no side work)
            except BaseException:
                cap.abort()                                  # close the open capture: the stream must not stay in capture mode
                self.steps_done, self.optimizer.iterations = keep
                raise
            finally:
                _engine._CAPTURE[0] = None
                K.set_launch_stream(prev_launch)
            cap.done = []                                    # the deferred closures kept the crossing tensors alive during the capture
            if cap.CENSUS:
                self._graph_census = cap.census()            # raises if a memset / memcpy node entered the captured step
            self.steps_done, self.optimizer.iterations = keep         # capturing is not a step
            entry = (cap, static, logs, (list(self._step_losses), list(self._loss_roots), dict(self._step_metrics)))
            self._graphs[sig] = entry
        cap, static, logs, book = entry
        for k, v in data.items():
            if v is not static[k]:
                static[k].copy_(v)
        ops.set_dropout_seed(self._step_seed())
        self.optimizer.stage_lr()
        from .engine import side_stream
        cap.replay(side_stream() if os.environ.get("BDETR_GRAPH_SIDE", "1") != "0" else None)
        self._step_losses, self._loss_roots, self._step_metrics = list(book[0]), list(book[1]), dict(book[2])
        self.steps_done += 1
        self.optimizer.iterations += 1
        if self._guarded() and self.guard_check_every:
            self._guard_launched += 1                # the replayed optimizer segment ended with the snapshot kernel
        bump_weights_version()
        return logs

    def train_step(self, data: dict) -> Dict[str, torch.Tensor]:
        logs = self._train_step_once(data)
        if self._guarded() and self.guard_check_every:
            logs = self._guard_poll(data, logs, force=getattr(self, "_guard_force", False))
        return logs

    GUARD_LAG = 2        # steps between a snapshot and the host's look at it
    GUARD_DEMOTE_AFTER = 3   # consecutive redos after which the policy falls back to 'mixed' for good (_guard_redo)
    GUARD_RING = 8       # per-step entries of the pinned log (> GUARD_LAG + 1, the most that are ever pending)

    def _guard_snapshot(self) -> None:
        """Last launch of a guarded step: log the flag against the device-resident step ordinal in pinned memory (K.flag_snapshot).
        Inside the step - and so inside the captured optimizer segment under use_graph - rather than between steps: see
        bdetr_flag_snapshot (include/bdetr.h) for what an operation that reads the flag between graph launches did."""
        if not (self._guarded() and self.guard_check_every):
            return
        if self.__dict__.get("_guard_host") is None:
            self._guard_host = torch.zeros(1 + self.GUARD_RING, dtype=torch.int32).pin_memory()
            self._guard_ordinal = torch.zeros(1, dtype=torch.int32, device="cuda")
            self._guard_events = [torch.cuda.Event() for _ in range(self.GUARD_RING)]
            self._guard_launched = 0
        K.flag_snapshot(self._guard_ordinal, self._guard_host)
        from . import engine as _engine
        if _engine._CAPTURE[0] is None:
            self._guard_launched += 1              # (a capture only records the launch; each replay counts, in _graph_step)

    def _guard_poll(self, data: dict, logs, force: bool = False):
        """Host side of the range guard without stalling the device: every guarded step ends with a one-lane kernel that logs the
        flag in pinned memory (`_guard_snapshot`); step t looks at the entry of step t - GUARD_LAG, which has long landed (the
        wait on its event only bounds how far the host runs ahead).  A fixed lag, not a poll, so that data-parallel replicas -
        whose flags agree after the step's MAX all-reduce - take the same decision at the same step.  A raised entry names the
        step that left the fp16 range; that batch and the later ones ran without an update (the optimizer skips while the flag
        is up), so all of them are redone on the exact-fp32 forward and the counters are rolled back for the update-free
        attempts.  force: resolve every outstanding entry now (fit() reads the logs on the host anyway)."""
        if self.__dict__.get("_guard_host") is None:
            return logs                            # (no guarded step has run yet)
        ev = self._guard_events[self._guard_launched % self.GUARD_RING]
        ev.record()
        # The batch is kept BY REFERENCE until its snapshot has been examined (GUARD_LAG + 1 steps): a redo trains on these tensors
        # again, so an input pipeline must not overwrite them in place before then.  Their version counters are noted here and
        # checked in _guard_redo - a reused buffer is an error there, not a silently different batch.
        self._guard_pending.append((data, self._guard_launched, ev, tuple((k, v._version) for k, v in data.items() if isinstance(v, torch.Tensor))))
        return self._guard_resolve(logs, 0 if force else self.GUARD_LAG)

    def _guard_resolve(self, logs, keep: int):
        host = self._guard_host
        while len(self._guard_pending) > keep:
            _, k, e = self._guard_pending[0][:3]
            if int(host[0]) < k:                     # (not landed yet: normally it has, GUARD_LAG steps later)
                e.synchronize()
                if int(host[0]) < k:
                    raise RuntimeError(f"range guard: step ordinal {k} finished but its snapshot is missing (log at {int(host[0])})")
            if int(host[1 + k % self.GUARD_RING]) != 0:
                return self._guard_redo(0, logs)
            self._guard_pending.pop(0)
            self.range_redo_streak = 0               # a guarded step went through clean
        return logs

    def guard_flush(self):
        """Resolve the snapshots still in flight (end of a run / before reading counters).  Returns the redone step's logs or None."""
        if self._guarded() and self._guard_pending:
            return self._guard_resolve(None, 0)
        return None

    def _guard_redo(self, first_bad: int, logs):
        import sys
        torch.cuda.synchronize()                             # rare: every later attempt has finished (none of them applied an update)
        batches = [p[0] for p in self._guard_pending[first_bad:]]
        for p in self._guard_pending[first_bad:]:
            stale = [k for k, ver in (p[3] if len(p) > 3 else ()) if p[0][k]._version != ver]
            if stale:
                raise RuntimeError(f"range guard: the batch of step ordinal {p[1]} must be redone, but its tensors {stale} were modified in place since "
                                   f"(an input pipeline has to leave a batch untouched for GUARD_LAG + 1 = {self.GUARD_LAG + 1} steps under the 'split' policy)")
        self._guard_pending = []
        K.overflow_flag().zero_()
        n = len(batches)
        self.range_redos += 1
        self.range_skipped += n
        self.steps_done -= n                                 # update-free attempts are not steps: dropout seeds and the
        self.optimizer.iterations -= n                       # learning-rate schedule continue from the last applied update
        print(f"[boosted_detr_amd] step {self.steps_done}: the split-fp16 forward left its range (|x| >= 65504) or went non-finite; "
              f"no update was applied since - redoing {n} batch(es) on the exact-fp32 forward", file=sys.stderr)
        keep, self.train_gemm_precision = self.train_gemm_precision, "mixed"
        keep_graph, self.use_graph = self.use_graph, False   # the redone batches run eagerly: no fresh capture (and no second private pool) mid-training
        try:
            for d in batches:
                logs = self._train_step_once(d)
        finally:
            self.train_gemm_precision = keep
            self.use_graph = keep_graph
        K.overflow_flag().zero_()                            # (bn_stats may have re-raised it for a genuinely non-finite batch statistic)
        # A model whose weights or activations sit outside the f16 pair's range for good (e.g. a conv weight beyond 65504 / P16_W_SCALE)
        # would run, skip and redo EVERY step - three times the cost behind a stderr line.  After GUARD_DEMOTE_AFTER redos without a
        # clean guarded step in between, the exact-fp32 forward ('mixed') becomes the model's policy and says so once.
        self.range_redo_streak += 1
        if self.range_redo_streak >= self.GUARD_DEMOTE_AFTER and self.train_gemm_precision == "split":
            self.train_gemm_precision = "mixed"
            print(f"[boosted_detr_amd] {self.range_redo_streak} consecutive range-guard redos: train_gemm_precision is now 'mixed' (exact-fp32 forward) "
                  "for the rest of this model's life; set it back to 'split' by hand if the cause was transient", file=sys.stderr)
        return logs

    # -- placement of the side stream under data parallelism -------------------------------------------------------------------
    # engine.side_stream() measures its candidates against the critical path's stream and keeps the first good one; that settles the
    # single-process case (one of the four hardware queues a low-priority stream can land on costs 80 % of the step, the other three are equal).  With
    # collectives in flight two more of the four become 10 % slower (measured over a one-rank RCCL communicator: 24.9 / 28.0 / 28.0 ms on the
    # three "good" queues - presumably the ones that share a dispatch pipe with the communication stream and with RCCL's own stream, whose
    # barrier packets wait for the side stream's events), and which ones cannot be seen before the collectives run.  So a data-parallel model
    # times its first eager steps on each good candidate (SIDE_TUNE_STEPS per slot, the first of a slot discarded; GPU time between two events
    # on the step's stream: step begin -> the join of the side stream behind the backward pass, i.e. BEFORE the step waits for its
    # collectives - a rank times its own streams, not the slowest replica's) and keeps the fastest.  The schedule has a FIXED length - SIDE_TUNE_SLOTS slots from
    # step SIDE_TUNE_FROM on, whatever the number of good candidates (they are cycled) - so that every rank leaves it at the same step.
    SIDE_TUNE_FROM, SIDE_TUNE_SLOTS, SIDE_TUNE_STEPS = 3, 4, 3

    def side_tuning_pending(self) -> bool:
        return (self._dp is not None and self._dp.active and getattr(self, "_side_tune", None) != "done"
                and os.environ.get("BDETR_SIDE_TUNE", "1") != "0")

    def _side_tune_begin(self):
        if not self.side_tuning_pending() or self.steps_done < self.SIDE_TUNE_FROM:
            return None
        from . import engine as _engine
        st = getattr(self, "_side_tune", None)
        if st is None:
            good = _engine.side_stream_expand() if _engine.side_stream_placement() is not None else []      # measures the remaining candidates
            st = self._side_tune = {"good": good, "k": 0, "ms": {c: [] for c in good}}
        slot, rep = divmod(st["k"], self.SIDE_TUNE_STEPS)
        if slot >= self.SIDE_TUNE_SLOTS:
            return None
        if len(st["good"]) < 2:
            return ("idle", None, None, [None])       # nothing to choose from: the schedule still runs its length (rank-independent)
        c = st["good"][slot % len(st["good"])]
        if rep == 0:
            _engine.side_stream_select(c)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        return (c, rep, ev0, [None])

    def _side_tune_end(self, tok) -> None:
        if tok is None:
            return
        from . import engine as _engine
        st = self._side_tune
        c, rep, ev0, end = tok
        self._side_tune_tok = None
        if c != "idle":
            ev1 = end[0]
            if ev1 is None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
            ev1.synchronize()
            if rep > 0:
                st["ms"][c].append(ev0.elapsed_time(ev1))
        st["k"] += 1
        if st["k"] >= self.SIDE_TUNE_SLOTS * self.SIDE_TUNE_STEPS:
            if c != "idle":
                best = min((c for c in st["good"] if st["ms"][c]), key=lambda c: min(st["ms"][c]))
                _engine.side_stream_select(best)
                _engine.side_stream_placement()["step_ms"] = {c: round(min(v), 3) for c, v in st["ms"].items() if v}
            if _engine.side_stream_placement() is not None:
                _engine.side_stream_release()              # the candidates not chosen are destroyed (idle queues are not free)
            self._side_tune = "done"

    def _train_step_once(self, data: dict) -> Dict[str, torch.Tensor]:
        if self.optimizer is None:
            raise RuntimeError("call compile(optimizer=...) before fit/train_step")
        sig = self._graph_signature(data)
        if sig is not None:
            logs = self._graph_step(data, sig)
            if logs is not None:
                return logs
        tune = self._side_tune_tok = self._side_tune_begin()
        logs = self._eager_step(data)
        self._side_tune_end(tune)
        return logs

    def _eager_step(self, data: dict) -> Dict[str, torch.Tensor]:
        if self._dp is not None and not getattr(self, "_dp_synced", True) and self.built_variables():
            self._dp.broadcast_variables(self.variables)
            self._dp_synced = True
        self.forward_backward(data)
        tok = getattr(self, "_side_tune_tok", None)
        if tok is not None and tok[2] is not None:
            # placement tuning: the timed interval ends HERE - behind the join of the side stream, before the step waits for its collectives
            # (a rank must time its own streams, not the slowest replica's)
            tok[3][0] = torch.cuda.Event(enable_timing=True)
            tok[3][0].record()
        if self._dp is not None and not getattr(self, "_dp_synced", True):
            # build-by-first-call just created the variables: replicas adopt rank 0's initial values before any update
            self._dp.broadcast_variables(self.variables)
            self._dp_synced = True
        if self.validate_matching:
            # scipy raises ValueError on NaN/-inf or infeasible cost matrices (the reference's
            # tf.numpy_function then fails the step); the GPU solver leaves such rows at -1.
            from .losses_and_metrics import MatchingAssignment
            MatchingAssignment.validate(self.loss_fn.last_match, self.loss_fn.last_num_objects, self.num_object_preds)
        tv = self.trainable_variables
        self.optimizer.stage_gradients(tv)
        guard = None
        if self._guarded():
            for root in self._loss_roots:
                K.flag_nonfinite(root)
            guard = K.overflow_flag()
        if self._dp is not None:
            self._dp.finish(self.optimizer.flat_grad)       # buckets not already in flight since the backward pass + join
            if guard is not None:
                self._dp.any_(guard)
        self.optimizer.apply_gradients(skip_flag=guard)
        self._guard_snapshot()
        self.steps_done += 1
        return self.step_logs()

    def built_variables(self) -> bool:
        """True once build-by-first-call has created every variable (the loss layer is the last one to run)."""
        return bool(self.variables) and getattr(self.loss_fn, "last_match", None) is not None

    def test_step(self, data):
        return self.train_step(data)        # model.py:235-236: validation also trains (quirk kept)

    def step_logs(self) -> Dict[str, list]:
        """name -> list of per-image [B] device tensors (one per weak learner).  Nothing is copied to
        the host here; ``logs_to_host`` does the Keras-style reduction ('loss' = mean over the batch
        of the summed [B] vectors) when somebody actually wants to read the numbers."""
        logs = {"loss": list(self._step_losses)}
        logs.update({k: list(v) for k, v in self._step_metrics.items()})
        return logs

    @staticmethod
    def logs_to_host(logs: Dict[str, list]) -> Dict[str, float]:
        return {k: float(sum(t.detach().cpu().numpy().astype(np.float64) for t in v).mean()) for k, v in logs.items() if v}

    def fit(self, x: Iterable[dict], epochs: int = 1, validation_data: Optional[Iterable[dict]] = None,
            callbacks: Optional[list] = None, steps_per_epoch: Optional[int] = None, verbose: int = 1):
        callbacks = callbacks or []
        for cb in callbacks:
            cb.set_model(self)
        history = {"loss": []}
        self.stop_training = False
        self.validate_matching = True
        self.guard_check_every = 1
        self._guard_force = True              # every step ends in a host read of the logs anyway: resolve the guard snapshot at once
        for epoch in range(epochs):
            t0, n, sums = time.time(), 0, {}
            for step, batch in enumerate(x):
                if steps_per_epoch is not None and step >= steps_per_epoch:
                    break
                logs = self.logs_to_host(self.train_step(batch))                     # sync point: host logging
                for k, v in logs.items():
                    sums[k] = sums.get(k, 0.0) + v
                n += 1
                for cb in callbacks:
                    cb.on_batch_end(step, logs)
                if self.stop_training:
                    break
            epoch_logs = {k: v / max(n, 1) for k, v in sums.items()}
            if validation_data is not None and not self.stop_training:
                vs, vn = {}, 0
                for batch in validation_data:
                    for k, v in self.logs_to_host(self.test_step(batch)).items():
                        vs[k] = vs.get(k, 0.0) + v
                    vn += 1
                epoch_logs.update({f"val_{k}": v / max(vn, 1) for k, v in vs.items()})
            history["loss"].append(epoch_logs.get("loss"))
            if verbose:
                msg = " - ".join(f"{k}: {v:.4f}" for k, v in epoch_logs.items())
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - {n} steps - {msg}")
            for cb in callbacks:
                cb.on_epoch_end(epoch, epoch_logs)
            if self.stop_training:
                break
        return history

    # -- weights -----------------------------------------------------------------------------
    def get_weights_dict(self) -> Dict[str, np.ndarray]:
        return {v.name: v.numpy() for v in self.variables}

    def set_weights_dict(self, weights: Dict[str, np.ndarray], strict: bool = True) -> None:
        names = {v.name for v in self.variables}
        if strict:
            missing, extra = names - set(weights), set(weights) - names
            if missing or extra:
                raise KeyError(f"weight name mismatch: missing {sorted(missing)[:5]} extra {sorted(extra)[:5]}")
        for v in self.variables:
            if v.name in weights:
                v.assign(weights[v.name])

    def save_weights(self, filepath: str) -> None:
        from safetensors.numpy import save_file
        if not filepath.endswith(".safetensors"):
            filepath += ".safetensors"
        os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
        meta = {"steps_done": str(self.steps_done),
                "optimizer_iterations": str(self.optimizer.iterations if self.optimizer is not None else 0)}
        save_file({k: np.ascontiguousarray(v) for k, v in self.get_weights_dict().items()}, filepath, metadata=meta)

    def load_weights(self, filepath: str) -> None:
        from safetensors.numpy import load_file
        if not filepath.endswith(".safetensors"):
            filepath += ".safetensors"
        self.set_weights_dict(load_file(filepath))
        from safetensors import safe_open
        with safe_open(filepath, framework="np") as f:
            meta = f.metadata() or {}
        # the step counter seeds the dropout masks and drives the learning-rate schedule: a resumed run continues both
        self.steps_done = int(meta.get("steps_done", self.steps_done))
        if self.optimizer is not None and "optimizer_iterations" in meta:
            self.optimizer.iterations = int(meta["optimizer_iterations"])

    def summary(self) -> str:
        lines = [f'Model: "{self.name}"', "-" * 96]
        for l in self.layers():
            lines.append(f"{l.name:48s} {type(l).__name__:32s} {l.count_params():>12,d}")
        total = self.count_params()
        train = sum(v.num_params for v in self.trainable_variables)
        lines += ["-" * 96, f"Total params: {total:,d}", f"Trainable params: {train:,d}", f"Non-trainable params: {total - train:,d}"]
        text = "\n".join(lines)
        print(text)
        return text
