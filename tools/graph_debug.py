"""Segmented-graph replay against eager steps at config 2: losses and the range-guard flag per step.
BDETR_GRAPH_SIDE=0 replays the side graphs in stream order; BDETR_GRAPH_SEG=<n> sets the side tasks per segment;
GUARD=1 turns the host side of the range guard on; BETWEEN=clone|other enqueues a device read of the guard word | of another
tensor between steps (round 3: the former corrupted later replays - the guard's snapshot therefore lives inside the step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boosted_detr_amd import kernels as K
from boosted_detr_amd.engine import to_device
class A: pass
a = A(); a.queries = 100; a.image = 640; a.layers = 6; a.batch = 16; a.model = 'detr'; a.fashionpedia = False; a.image_w = 0; a.learners = 3; a.backbone = 'ResNet'; a.panoptic = False
class Tripped(Exception): pass
REF = {}
def state(m):
    out = {f"{i}:{getattr(v, 'name', '')}:{tuple(v.value.shape)}": v.value.detach().clone() for i, v in enumerate(m.variables)}
    opt = m.optimizer
    for k in ("flat_mom", "momentum_flat", "mom"):
        t = getattr(opt, k, None)
        if isinstance(t, torch.Tensor):
            out["opt." + k] = t.detach().clone()
    return out
def compare(m):
    cur = state(m)
    rows = []
    for k, a in cur.items():
        b = REF.get(k)
        if b is None or b.shape != a.shape:
            continue
        d = float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        rows.append((d, k, bool(torch.isfinite(a).all())))
    rows.sort(reverse=True)
    print("most deviating state tensors (relative L2 against the eager run after 8 steps):")
    for r in rows[:12]:
        print("   ", r)
    print("    median", rows[len(rows) // 2])
def run(graph, steps=8):
    torch.manual_seed(0)
    m = bench.build_model(a)
    host = bench.make_batch(16, 640, 640, 100, 82, 1234)
    batch = {"image": to_device(host["image"]), "category": to_device(host["category"], torch.int32), "attribute": to_device(host["attribute"], torch.int32), "bbox": to_device(host["bbox"]), "num_objects": to_device(host["num_objects"], torch.int32)}
    m.use_graph = graph
    m.guard_check_every = int(os.environ.get("GUARD", "0"))
    out = []
    for i in range(steps):
        logs = m.train_step(batch)
        if os.environ.get("BETWEEN") == "clone":          # a device operation that READS the guard word between two steps' graph launches
            keep = K.overflow_flag().clone()
        elif os.environ.get("BETWEEN") == "other":        # ... that reads another tensor the graphs touch (the learning rate)
            keep = m.optimizer.d_lr.clone()
        torch.cuda.synchronize()
        out.append((round(m.logs_to_host(logs)["loss"], 4), int(K.overflow_flag().item()), m.range_redos, [p[1] for p in m._guard_pending]))
        if not m.guard_check_every:
            K.overflow_flag().zero_()
    if os.environ.get("LAG"):
        m.GUARD_LAG = int(os.environ["LAG"])
    if not graph:
        REF.update(state(m))
    elif REF:
        def trip(*a, **k):
            raise Tripped()
        m._guard_redo = trip
    t0 = time.perf_counter()
    try:
        for i in range(10):
            m.train_step(batch)
            mode = os.environ.get("SYNC2", "")
            if mode == "1" or (mode == "every2" and i % 2 == 1) or (mode == "every4" and i % 4 == 3):
                torch.cuda.synchronize()
            elif mode == "stream":
                torch.cuda.current_stream().synchronize()
            elif mode == "logs":
                m.logs_to_host(logs_i) if False else [t.cpu() for t in m._step_losses]
    except Tripped:
        torch.cuda.synchronize()
        print("guard tripped in the unsynchronised loop at step", m.steps_done, "log", m._guard_host.tolist(), "pending", [p[1] for p in m._guard_pending])
        compare(m)
        return out, 0.0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    m.guard_flush()
    logs = m.train_step(batch)        # one more, read back: were the unsynchronised steps sound?
    torch.cuda.synchronize()
    out.append(("after", round(m.logs_to_host(logs)["loss"], 4), int(K.overflow_flag().item()), m.range_redos, m.steps_done))
    if graph:
        cap = list(m._graphs.values())[0][0]
        print("segments", len(cap.mains), "side graphs", sum(1 for s_ in cap.sides if s_ is not None))
    return out, dt
if os.environ.get("GRAPH_ONLY", "0") != "1":
    print("eager", *run(False))
print("graph", *run(True))
