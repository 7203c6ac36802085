"""Where does a replayed step first differ from the eagerly enqueued one?  (VERDICT r3 item 6 / ADVICE: the ROCm 7.2 replay defect.)

Runs the bench's config-2 step twice from identical weights under the DETERMINISTIC mode (no float atomics: an eager step and its
replay are bit-identical when the runtime is sound) with everything in stream order (BDETR_SIDE_STREAM=0, BDETR_GRAPH_SIDE=0):
once enqueued eagerly, once captured as the chain of hipGraphs and replayed WITHOUT synchronising between steps.  At every segment
boundary of the backward pass both runs append an order-independent fingerprint of (A) the activation gradient that crosses the cut,
(M) the flat gradient buffer behind the main segment, (S) the flat gradient buffer behind the segment's weight-gradient tasks to a
device-resident log (engine.DebugLog; the launches are captured INTO the segments).  The first entry that differs names the step
and the segment whose output went wrong; the non-finite counts say whether that is where the NaN is born.

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 BDETR_GRAPH_UNSAFE=1 python3 tools/graph_segment_checksums.py      # the failing runtime path
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python3 tools/graph_segment_checksums.py                           # the workaround: all equal

SMALL=1 uses the test-sized model instead of config 2.  STEPS=<n> replayed steps (default 8)."""
import json
import os
import sys

os.environ.setdefault("BDETR_DETERMINISTIC", "1")
os.environ.setdefault("BDETR_SIDE_STREAM", "0")
os.environ.setdefault("BDETR_GRAPH_SIDE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import boosted_detr_amd  # noqa: E402

boosted_detr_amd.enable_graph_replay()
import torch  # noqa: E402

import bench  # noqa: E402
from boosted_detr_amd import engine, kernels as K  # noqa: E402
from boosted_detr_amd.engine import to_device  # noqa: E402


class A:
    queries, image, layers, batch, model, fashionpedia, image_w, learners, backbone, panoptic = 100, 640, 6, 16, "detr", False, 0, 3, "ResNet", False


a = A()
if os.environ.get("SMALL") == "1":
    a.image, a.layers, a.batch, a.queries = 224, 1, 2, 50
STEPS = int(os.environ.get("STEPS", "8"))


def run(graph: bool):
    torch.manual_seed(0)
    m = bench.build_model(a)
    host = bench.make_batch(a.batch, a.image, a.image, 100, 82, 1234)
    batch = {"image": to_device(host["image"]), "category": to_device(host["category"], torch.int32), "attribute": to_device(host["attribute"], torch.int32),
             "bbox": to_device(host["bbox"]), "num_objects": to_device(host["num_objects"], torch.int32)}
    m.guard_check_every = 0                       # no redo: the run must show what the replay did, not repair it
    m.use_graph = graph
    log = engine.DebugLog()
    engine.set_debug_log(log)
    for i in range(3 + STEPS):
        m.train_step(batch)
        if i < 3:
            torch.cuda.synchronize()                             # set-up steps (2 eager + the capture and its first replay) are synchronised
    torch.cuda.synchronize()
    engine.set_debug_log(None)
    ent = log.entries()
    weights_finite = all(bool(torch.isfinite(v.value).all()) for v in m.variables)
    K.overflow_flag().zero_()
    return ent, weights_finite, (len(m._graphs) if graph else 0)


def main():
    print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), " graph_replay_is_safe:", boosted_detr_amd.graph_replay_is_safe(), flush=True)
    e_ent, e_fin, _ = run(False)
    g_ent, g_fin, ngraphs = run(True)
    print(f"eager: {len(e_ent)} entries, weights finite {e_fin};  graph: {len(g_ent)} entries, weights finite {g_fin}, captured signatures {ngraphs}")
    total_steps = 3 + STEPS
    out = {"entries_eager": len(e_ent), "entries_graph": len(g_ent), "weights_finite": {"eager": e_fin, "graph": g_fin}}
    if len(e_ent) != len(g_ent) or len(e_ent) % total_steps:
        print("entry counts differ or do not divide by the step count: cut points are not the same in both modes", len(e_ent), len(g_ent))
    per = len(e_ent) // total_steps if total_steps else 0
    first = None
    nonfinite_first = None
    for i, (x, y) in enumerate(zip(e_ent, g_ent)):
        step, cut, kind = i // per, (i % per) // 3, x[0]
        if nonfinite_first is None and y[2] > 0:
            nonfinite_first = (step, cut, y[0], y[2])
        if kind == "M":
            continue                                             # (M is taken before the side tasks in graph mode, after them in the eager twin)
        if first is None and (x[1] != y[1] or x[2] != y[2]):
            first = (step, cut, kind, x, y)
    out["entries_per_step"] = per
    out["first_difference"] = None if first is None else {"step": first[0], "cut": first[1], "kind": first[2], "eager": first[3], "graph": first[4]}
    out["first_nonfinite_in_graph_run"] = nonfinite_first
    print("entries per step:", per, " first difference (A / S entries):", out["first_difference"], " first non-finite entry in the graph run:", nonfinite_first)
    if first is not None:
        lo = max(0, (first[0] * per + first[1] * 3) - 6)
        for i in range(lo, min(len(e_ent), lo + 18)):
            print(f"   step {i // per} cut {(i % per) // 3} {e_ent[i][0]}: eager {e_ent[i][1]:016x}/{e_ent[i][2]}  graph {g_ent[i][1]:016x}/{g_ent[i][2]}", "  <--" if e_ent[i][1:] != g_ent[i][1:] and e_ent[i][0] != "M" else "")
    print("GRAPH_SEGMENT_CHECKSUMS " + json.dumps(out, default=str))


if __name__ == "__main__":
    main()
