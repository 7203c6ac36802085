"""CPU tests of the host logic that needs no kernels: tokenizers, schedules, variable layout
conversion, tape accumulation order, bench helpers, and the 2-rank gloo gradient all-reduce."""
import os
import subprocess
import sys

import numpy as np
import pytest

def _free_port() -> str:
    """A port nobody listens on right now (fixed numbers collided with other jobs sharing the box's network namespace: one abort in
    RCCL's bootstrap in round 4)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokenization_string_lookup_semantics(monkeypatch):
    from boosted_detr_amd import engine, tokenizers
    import torch
    monkeypatch.setattr(engine, "to_device", lambda a, dtype=torch.float32: torch.as_tensor(np.asarray(a)).to(dtype))
    monkeypatch.setattr(tokenizers, "to_device", engine.to_device)
    vocab = {"category": ["cat", "dog"], "attribute": ["red", "big", "old"]}
    tok = tokenizers.Tokenization(vocab)
    assert tok.vocab_size_dict() == {"category": 4, "attributes": 5}
    cat = np.array([[["dog"], ["<PAD>"], ["zebra"]]], dtype=object)
    att = np.array([[["big", "red"], ["<PAD>", "<PAD>"], ["old", "<PAD>"]]], dtype=object)
    ids, hot = tok([cat, att])
    assert ids.tolist() == [[3, 0, 1]]                       # vocab from 2, PAD=0, OOV=1
    assert hot[0, 0].tolist() == [0, 0, 1, 1, 0]
    assert hot[0, 1].tolist() == [1, 0, 0, 0, 0]             # PAD slots set bit 0 (tokenizers.py:76)
    assert hot[0, 2].tolist() == [1, 0, 0, 0, 1]
    ids2, _ = tok([np.array([[3, 0, 1]], np.int32), np.zeros((1, 3, 2), np.int32)])
    assert ids2.tolist() == [[3, 0, 1]]                      # pre-tokenised ids pass through


def test_inverse_tokenization_strings():
    import torch
    from boosted_detr_amd import tokenizers
    vocab = {"category": ["cat", "dog"], "attribute": ["red", "big", "old"]}
    inv = tokenizers.InverseTokenization(vocab)
    cat = torch.tensor([[[0.1, 0.1, 0.2, 0.6], [0.7, 0.1, 0.1, 0.1], [0.25, 0.25, 0.25, 0.25]]])
    att = torch.tensor([[[0.1, 0.2, 0.9, 0.6, 0.1], [0.0, 0.0, 0.0, 0.0, 0.0], [0.9, 0.9, 0.1, 0.1, 0.5]]])
    c, a = inv([cat, att])
    assert c[0, :, 0].tolist() == ["dog", "<PAD>", "<PAD>"]  # argmax takes the first maximum on ties
    # the reference's regex clean-up (tokenizers.py:146-155) leaves a trailing comma when PAD slots follow: kept
    assert a[0, 0, 0] == "red, big," and a[0, 1, 0] == "" and a[0, 2, 0] == "old"


def test_cosine_decay_restarts_matches_oracle():
    from boosted_detr_amd.training import CosineDecayRestarts
    from oracle import detr_oracle as O
    s = CosineDecayRestarts(1e-3, 4000, m_mul=0.95, alpha=0.1)
    for step in (0, 1, 1999, 4000, 4001, 11999, 12000, 30000):
        assert abs(s(step) - O.cosine_decay_restarts(step, 1e-3, 4000, 2.0, 0.95, 0.1)) < 1e-15


def test_variable_layout_roundtrip(monkeypatch):
    import torch
    from boosted_detr_amd import engine
    monkeypatch.setattr(engine, "to_device", lambda a, dtype=torch.float32: torch.as_tensor(np.asarray(a)).to(dtype).contiguous())
    rng = np.random.default_rng(0)
    v = engine.Variable("c/kernel", (7, 7, 3, 64), kind="conv_kernel", pad_in_channels=1)
    k = rng.standard_normal((7, 7, 3, 64)).astype(np.float32)
    v.assign(k)
    assert tuple(v.value.shape) == (64, 7, 7, 4) and float(v.value[..., 3].abs().max()) == 0.0
    assert np.array_equal(v.numpy(), k)
    d = engine.Variable("d/kernel", (5, 9), kind="dense_kernel")
    w = rng.standard_normal((5, 9)).astype(np.float32)
    d.assign(w)
    assert tuple(d.value.shape) == (9, 5) and np.array_equal(d.numpy(), w)


def test_initializers_statistics():
    from boosted_detr_amd.engine import initializer
    w = initializer("glorot_normal")("x", (256, 256))
    assert abs(w.std() - np.sqrt(2.0 / 512)) < 2e-3 and np.abs(w).max() <= 2.0 * np.sqrt(2.0 / 512) / 0.8796 + 1e-6
    w = initializer("he_normal")("y", (256, 1024))
    assert abs(w.std() - np.sqrt(2.0 / 256)) < 2e-3
    assert initializer("zeros")("z", (4,)).sum() == 0 and initializer("ones")("o", (4,)).sum() == 4


def test_positional_init_matches_oracle():
    from boosted_detr_amd.transformers import positional_init
    from oracle import detr_oracle as O
    assert np.array_equal(positional_init(3, 4, 16), O.positional_init(3, 4, 16))


def test_bench_batch_generator_matches_oracle_generator():
    sys.path.insert(0, ROOT)
    import bench
    from oracle import detr_oracle as O
    a = bench.make_batch(3, 16, 16, 100, 82, seed=1234)
    b = O.make_batch(O.Config(image_size=(16, 16), num_categories=82, num_attributes=3), 3, 100, seed=1234)
    for k in ("image", "category", "bbox", "num_objects"):
        assert np.array_equal(a[k], b[k]), k
    assert 1 <= bench.usable_cores() <= 16


_DP_SCRIPT = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from boosted_detr_amd.training import DataParallel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
dp = DataParallel()
dp.BUCKET_ELEMS = 1000                      # several buckets
rng = np.random.default_rng(dp.rank)
flat = torch.from_numpy(rng.standard_normal(4321).astype(np.float32))
want = sum(np.random.default_rng(r).standard_normal(4321).astype(np.float32) for r in range(dp.world))
dp.allreduce_(flat)
assert np.allclose(flat.numpy(), want, atol=1e-6), "allreduce mismatch"
class V:  # broadcast path
    pass
v = V(); v.value = torch.full((5,), float(dp.rank))
dp.broadcast_variables([v])
assert float(v.value.sum()) == 0.0
# bucketed all-reduce launched from the backward pass (grad_ready) - buckets count from the END of the flat buffer
class Opt: pass
sizes = [1200, 800, 1500, 700, 796]
offs = np.concatenate([[0], np.cumsum(sizes)])
flat = torch.zeros(int(offs[-1]))
vs = []
for i, sz in enumerate(sizes):
    q = V(); q.name = "v%d" % i; q.grad_buf = flat[int(offs[i]): int(offs[i]) + sz]; q.grad = None; q._grad_flat = flat
    vs.append(q)
opt = Opt(); opt.flat_grad = flat; opt.vars = vs
dp2 = DataParallel(); dp2.BUCKET_ELEMS = 1000
for step in range(4):
    flat.zero_()
    dp2.begin_step(opt, None, None)
    early = 0
    for i in reversed(range(len(vs))):                      # backward order: last variable first
        q = vs[i]
        q.grad_buf.copy_(torch.full((sizes[i],), float((dp2.rank + 1) * (i + 1) + step)))
        q.grad = q.grad_buf
        dp2.grad_ready(q)
        if i == 2:                                          # a shared variable: second contribution lands later
            q.grad_buf.add_(0.5)
            dp2.grad_ready(q)
        early = max(early, sum(dp2._launched)) if dp2._expected is not None and dp2._active else early
    dp2.finish(flat)
    for i, sz in enumerate(sizes):
        want = sum((r + 1) * (i + 1) + step + (0.5 if i == 2 else 0.0) for r in range(dp2.world))
        got = flat[int(offs[i]): int(offs[i]) + sz]
        assert torch.all(got == want), (step, i, float(got[0]), want)
    assert (early > 0) == (step >= 1), (step, early)        # step 0 calibrates the contribution counts, later steps overlap
# The trainable set changed (freeze / unfreeze): Model.forward_backward rebuilds the optimizer BEFORE begin_step, so the step is
# armed on the new buffer, runs one calibration step without early launches, then overlaps again - every sum right throughout.
keep = [0, 2, 4]
sizes2 = [sizes[i] for i in keep]
offs2 = np.concatenate([[0], np.cumsum(sizes2)])
flat2 = torch.zeros(int(offs2[-1]))
vs2 = []
for j, i in enumerate(keep):
    q = vs[i]; q.grad_buf = flat2[int(offs2[j]): int(offs2[j]) + sizes2[j]]; q.grad = None; q._grad_flat = flat2
    vs2.append(q)
opt.flat_grad = flat2; opt.vars = vs2
for step in range(3):
    flat2.zero_()
    dp2.begin_step(opt, None, None)
    early = 0
    for j in reversed(range(len(vs2))):
        q = vs2[j]
        q.grad_buf.copy_(torch.full((sizes2[j],), float((dp2.rank + 1) * (j + 1) + step)))
        q.grad = q.grad_buf
        dp2.grad_ready(q)
        early = max(early, sum(dp2._launched)) if dp2._expected is not None and dp2._active else early
    dp2.finish(flat2)
    for j, sz in enumerate(sizes2):
        want = sum((r + 1) * (j + 1) + step for r in range(dp2.world))
        assert torch.all(flat2[int(offs2[j]): int(offs2[j]) + sz] == want), ("rebuilt", step, j)
    assert (early > 0) == (step >= 1), ("rebuilt", step, early)
# ... and the state the advisor described (armed on a buffer that is then retired, buckets already in flight) is refused
# loudly instead of reducing twice: every rank raises, after waiting for the handles it launched
flat2.zero_()
dp2.begin_step(opt, None, None)
for j in reversed(range(len(vs2))):
    vs2[j].grad = vs2[j].grad_buf
    dp2.grad_ready(vs2[j])
assert any(dp2._launched)
try:
    dp2.finish(torch.zeros(int(offs2[-1])))
    raise SystemExit("finish() accepted a retired buffer")
except RuntimeError as e:
    assert "gradient buffer changed" in str(e), e
dist.barrier(); dist.destroy_process_group()
print("rank", dp.rank, "ok")
'''


def test_data_parallel_allreduce_gloo_world2(tmp_path):
    """N>1 path on CPU: world_size-2 gloo process group through the same DataParallel class the GPU
    path uses with RCCL (bucketed async all-reduce of the flat gradient buffer + weight broadcast)."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_data_parallel_cuda_branch_with_fake_streams(monkeypatch):
    """The device branch of DataParallel._launch / finish (communication stream, event waits on the main and side streams,
    async handles, the join) cannot run here - there is no GPU - and on the one-GPU box it runs over a one-rank RCCL
    communicator (tests/test_dp_gpu.py).  This drives the same code with recording fakes for torch.cuda and
    torch.distributed and checks the ORDER of operations: a bucket's collective is enqueued on the communication stream only
    after that stream waits for events recorded on BOTH producer streams, nothing waits for the collective until finish(),
    and finish() waits for every handle before the current stream joins the communication stream."""
    import types
    import torch
    from boosted_detr_amd import training
    log = []

    class FakeStream:
        def __init__(self, name): self.name = name
        def wait_event(self, ev): log.append(("wait_event", self.name, ev.on))
        def wait_stream(self, other): log.append(("wait_stream", self.name, other.name))

    class FakeEvent:
        def __init__(self, enable_timing=False): self.on = None
        def record(self, st=None): self.on = (st or cur[0]).name; log.append(("record", self.on))

    class FakeCtx:
        def __init__(self, st): self.st = st
        def __enter__(self): self.prev, cur[0] = cur[0], self.st
        def __exit__(self, *a): cur[0] = self.prev

    main, side, comm = FakeStream("main"), FakeStream("side"), FakeStream("comm")
    cur = [main]
    fake_cuda = types.SimpleNamespace(Stream=lambda device=None: comm, Event=FakeEvent, stream=lambda st: FakeCtx(st),
                                      current_stream=lambda: cur[0], synchronize=lambda: None)
    monkeypatch.setattr(training.torch, "cuda", fake_cuda)

    class Handle:
        def __init__(self, n): self.n = n
        def wait(self): log.append(("handle_wait", self.n, cur[0].name))

    class FakeDist:
        class ReduceOp: SUM = "sum"; MAX = "max"
        @staticmethod
        def is_initialized(): return True
        @staticmethod
        def get_world_size(): return 2
        @staticmethod
        def get_rank(): return 0
        @staticmethod
        def all_reduce(t, op=None, async_op=False):
            log.append(("all_reduce", t.numel(), cur[0].name)); return Handle(t.numel())
    monkeypatch.setitem(sys.modules, "torch.distributed", FakeDist)
    monkeypatch.setattr(torch, "distributed", FakeDist, raising=False)

    class FlatOnDevice:                      # a flat buffer that says it lives in HBM
        is_cuda, device = True, "cuda:0"
        def __init__(self, n, base=4096): self.n, self.base = n, base
        def numel(self): return self.n
        def data_ptr(self): return self.base
        def __getitem__(self, sl): return FlatOnDevice(sl.stop - sl.start, self.base + 4 * sl.start)

    class V: pass
    flat = FlatOnDevice(2500)
    sizes, vs, off = [1200, 800, 500], [], 0
    for i, sz in enumerate(sizes):
        v = V(); v.name = f"v{i}"; v.grad_buf = flat[off: off + sz]; v.grad = v.grad_buf; v._grad_flat = flat
        vs.append(v); off += sz
    opt = types.SimpleNamespace(flat_grad=flat, vars=vs)
    dp = training.DataParallel()
    dp.BUCKET_ELEMS = 1000
    assert dp.active and dp.world == 2
    for step in range(2):                    # step 0 learns the contribution counts, step 1 launches from grad_ready
        log.clear()
        dp.begin_step(opt, main, side)
        for v in reversed(vs):
            dp.grad_ready(v)
        early = [e for e in log if e[0] == "all_reduce"]
        assert (len(early) > 0) == (step == 1), (step, log)
        assert not any(e[0] == "handle_wait" for e in log)                   # nothing waits before finish()
        dp.finish(flat)
        ar = [i for i, e in enumerate(log) if e[0] == "all_reduce"]
        assert len(ar) == 3 and all(log[i][2] == "comm" for i in ar)         # 3 buckets, all enqueued on the communication stream
        for i in ar:                                                         # ... each behind waits for main AND side
            pre = log[max(0, i - 4): i]
            assert ("wait_event", "comm", "main") in pre and ("wait_event", "comm", "side") in pre, (i, log)
        waits = [i for i, e in enumerate(log) if e[0] == "handle_wait"]
        join = log.index(("wait_stream", "main", "comm"))
        assert len(waits) == 3 and max(waits) < join and min(waits) > max(ar)
    flag = torch.zeros(1)
    dp.any_(flag)
    assert log[-1] == ("all_reduce", 1, "main")
    # bucket table at the bench's size (31.0 M gradients, 32-MB buckets): the bucket that completes LAST - the first layers', at the start
    # of the buffer - is cut to TAIL_ELEMS so that the all-reduce nothing can overlap is small (round 5)
    big = FlatOnDevice(31_006_681)
    vb = []
    for i, (o, sz) in enumerate(((0, 9408), (500_000, 147_456), (1_048_000, 4_000), (20_000_000, 2_359_296))):
        v = V(); v.name = f"b{i}"; v.grad_buf = big[o: o + sz]; v.grad = v.grad_buf; v._grad_flat = big
        vb.append(v)
    dp3 = training.DataParallel()
    dp3.prepare(types.SimpleNamespace(flat_grad=big, vars=vb))
    n, B, T = 31_006_681, dp3.BUCKET_ELEMS, dp3.TAIL_ELEMS
    assert dp3._bounds[0] == (n - B, n) and dp3._bounds[-2:] == [(T, n - 3 * B), (0, T)] and len(dp3._bounds) == 5, dp3._bounds
    assert all(a[0] == b[1] for a, b in zip(dp3._bounds[:-1], dp3._bounds[1:]))            # contiguous cover, high to low
    assert dp3._var_bucket[id(vb[0])] == (4, 4) and dp3._var_bucket[id(vb[2])] == (4, 3) and dp3._var_bucket[id(vb[3])][0] == 1      # vb[2] straddles the cut: counted in both


@pytest.mark.parametrize("good, step_ms, want", [([0, 1, 2], {0: 28.0, 1: 25.0, 2: 28.5}, 1), ([1, 3], {1: 27.0, 3: 24.9}, 3), ([2], {}, 2), ([], {}, None)])
def test_side_stream_tuning_schedule_has_a_fixed_length_and_keeps_the_fastest_candidate(monkeypatch, good, step_ms, want):
    """training.Model._side_tune_begin / _side_tune_end (the data-parallel model settles between the good side-stream candidates by timing its
    first eager steps) with fakes for the engine and the events: whatever the number of good candidates, tuning ends at the SAME step - every
    rank must leave it together: the captured step and bench.py's set-up loop key on it - the first step of a slot is not counted, the
    fastest candidate is selected and the others released."""
    import types
    from boosted_detr_amd import training, engine
    clock = [0.0]
    sel, released = [], []

    class FakeEvent:
        def __init__(self, enable_timing=False): self.t = None
        def record(self): self.t = clock[0]
        def synchronize(self): pass
        def elapsed_time(self, other): return other.t - self.t
    monkeypatch.setattr(training.torch, "cuda", types.SimpleNamespace(Event=FakeEvent))
    placement = {"picked": good[0], "good": list(good)} if good else None            # None: no measured side stream (BDETR_SIDE_CANDIDATES=1, side stream off)
    monkeypatch.setattr(engine, "side_stream_placement", lambda: placement)
    monkeypatch.setattr(engine, "side_stream_expand", lambda: list(good))
    monkeypatch.setattr(engine, "side_stream_select", lambda c: (sel.append(c), placement.__setitem__("picked", c)))
    monkeypatch.setattr(engine, "side_stream_release", lambda: released.append(True))
    m = training.Model.__new__(training.Model)
    m._dp = types.SimpleNamespace(active=True)
    m._side_tune, m.steps_done = None, 0
    first_done = None
    for step in range(30):
        pending = m.side_tuning_pending()
        tok = m._side_tune_tok = m._side_tune_begin()
        if tok is not None and tok[2] is not None:
            clock[0] += step_ms[tok[0]] + (7.0 if tok[1] == 0 else 0.0)       # the first step of a slot is slower (and must not count)
            tok[3][0] = FakeEvent(); tok[3][0].record()
        m._side_tune_end(tok)
        m.steps_done += 1
        if pending and not m.side_tuning_pending() and first_done is None:
            first_done = m.steps_done
    assert first_done == training.Model.SIDE_TUNE_FROM + training.Model.SIDE_TUNE_SLOTS * training.Model.SIDE_TUNE_STEPS, first_done
    if len(good) >= 2:
        assert placement["picked"] == want and sel[-1] == want and released, (placement, sel)
        assert placement["step_ms"] == {c: step_ms[c] for c in good}, placement
    else:
        assert not sel, sel
