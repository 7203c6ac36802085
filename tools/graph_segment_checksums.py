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
    K.overflow_flag().zero_()
    m.use_graph = graph
    log = engine.DebugLog()
    engine.set_debug_log(log)
    for i in range(3 + STEPS):
        m.train_step(batch)
        if i < int(os.environ.get("SYNC_STEPS", "3")):
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
    out = {"entries_eager": len(e_ent), "entries_graph": len(g_ent), "weights_finite": {"eager": e_fin, "graph": g_fin}, "steps": total_steps,
           "sync_steps": int(os.environ.get("SYNC_STEPS", "3"))}
    # entries of the main and the side stream may interleave: compare the sequences per tag (M is taken before the side tasks in graph
    # mode and after them in the eager twin: not compared)
    for kind in ("L", "F", "A", "S"):
        xs, ys = [e for e in e_ent if e[0] == kind], [e for e in g_ent if e[0] == kind]
        per = len(xs) // total_steps if total_steps else 0
        first = next((i for i, (x, y) in enumerate(zip(xs, ys)) if x[1:] != y[1:]), None)
        bad = next((i for i, y in enumerate(ys) if y[2] > 0 or (kind == "F" and y[1] != 0)), None)
        out[kind] = {"per_step": per, "n": (len(xs), len(ys)),
                     "first_difference": None if first is None else {"step": first // max(per, 1), "index_in_step": first % max(per, 1), "eager": xs[first][1:], "graph": ys[first][1:]},
                     "first_nonfinite_or_flag_in_graph_run": None if bad is None else {"step": bad // max(per, 1), "index_in_step": bad % max(per, 1), "entry": ys[bad][1:]}}
        print(kind, out[kind], flush=True)
    print("GRAPH_SEGMENT_CHECKSUMS " + json.dumps(out, default=str))


if __name__ == "__main__":
    main()
