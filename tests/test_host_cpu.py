"""CPU tests of the host logic that needs no kernels: tokenizers, schedules, variable layout
conversion, tape accumulation order, bench helpers, and the 2-rank gloo gradient all-reduce."""
import os
import subprocess
import sys

import numpy as np
import pytest

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
dist.barrier(); dist.destroy_process_group()
print("rank", dp.rank, "ok")
'''


def test_data_parallel_allreduce_gloo_world2(tmp_path):
    """N>1 path on CPU: world_size-2 gloo process group through the same DataParallel class the GPU
    path uses with RCCL (bucketed async all-reduce of the flat gradient buffer + weight broadcast)."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
