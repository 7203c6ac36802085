"""Why is the data-parallel step (one-rank RCCL rehearsal) 20 ms slower than the single-process step in tools/soak.py when bench.py measures them 1 ms apart?
Hypothesis: the cyclic garbage collector (bench.py's timed regions run with it off).  Counts and times the collector's passes per generation over blocks of
50 eager steps: collector on, after gc.freeze(), collector off.   usage: [BDETR_DP_FORCE=1] python tools/dp_gc_probe.py"""
import gc, os, sys, time
sys.path.insert(0, '.')
import boosted_detr_amd
import torch
import bench
class A: pass
a = A(); a.queries = 100; a.image = 640; a.layers = 6; a.batch = 16; a.model = 'detr'; a.fashionpedia = False; a.image_w = 0; a.learners = 3; a.backbone = 'ResNet'; a.panoptic = False
from boosted_detr_amd.engine import to_device
DP = os.environ.get("BDETR_DP_FORCE", "0") == "1"
INIT_FIRST = os.environ.get("PROBE_INIT_FIRST", "0") == "1"
def init_pg():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist
if DP and INIT_FIRST:
    torch.cuda.set_device(0); dist = init_pg()
m = bench.build_model(a)
if DP:
    if not INIT_FIRST: dist = init_pg()
    m.distribute()
host = bench.make_batch(16, 640, 640, 100, 82, 1234)
batch = {"image": to_device(host["image"]), "category": to_device(host["category"], torch.int32), "attribute": to_device(host["attribute"], torch.int32),
         "bbox": to_device(host["bbox"]), "num_objects": to_device(host["num_objects"], torch.int32)}
stat = {"n": [0, 0, 0], "t": [0.0, 0.0, 0.0], "found": [0, 0, 0], "t0": 0.0}
def cb(phase, info):
    if phase == "start": stat["t0"] = time.perf_counter()
    else:
        g = info["generation"]; stat["n"][g] += 1; stat["t"][g] += time.perf_counter() - stat["t0"]; stat["found"][g] += info["collected"]
gc.callbacks.append(cb)
def block(tag, n=50):
    for k in ("n", "found"): stat[k] = [0, 0, 0]
    stat["t"] = [0.0, 0.0, 0.0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.train_step(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag:28s} {1e3*(t2-t0)/n:6.2f} ms/step (host enqueue {1e3*(t1-t0)/n:6.2f})  gc passes gen0/1/2 {stat['n']}  gc ms/step {[round(1e3*x/n,2) for x in stat['t']]}  collected {stat['found']}  tracked objects {len(gc.get_objects())}", flush=True)
print(("data-parallel (one-rank RCCL)" + (", process group before the model" if INIT_FIRST else ", process group after the model")) if DP else "single process", flush=True)
block("warm-up"); block("collector on"); block("collector on")
from boosted_detr_amd import engine as _e
print("side stream placement:", _e.side_stream_placement(), flush=True)
if os.environ.get("PROBE_SHORT", "0") == "1":
    if DP:
        torch.cuda.synchronize(); dist.destroy_process_group()
    sys.exit(0)
if os.environ.get("PROBE_CPROFILE", "0") == "1":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): m.train_step(batch)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(25)
gc.collect(); gc.freeze(); block("after gc.freeze()"); block("after gc.freeze()"); gc.unfreeze()
gc.collect(); gc.disable(); block("collector off"); gc.enable()
if DP:
    torch.cuda.synchronize(); dist.destroy_process_group()
