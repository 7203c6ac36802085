"""Soak: N training steps at config 2, reporting step time, loss and allocator high-water marks (leak check).
usage: python tools/soak.py [steps] [policy] [graph]   - run once per arithmetic policy to compare the loss trajectories; `graph`:
device-resident int32 targets and the step replayed as a chain of hipGraphs (Model.use_graph), as bench.py runs it.
BDETR_DP_FORCE=1: the data-parallel step over a one-rank RCCL communicator."""
import sys, time
sys.path.insert(0, '.')
import boosted_detr_amd          # (before the first CUDA call: sets the hipGraph runtime switch)
import torch
import bench
class A: pass
a = A(); a.queries = 100; a.image = 640; a.layers = 6; a.batch = 16; a.model = (sys.argv[4] if len(sys.argv) > 4 else 'detr'); a.fashionpedia = a.model == 'boosted'; a.image_w = 0; a.learners = 3; a.backbone = 'ResNet'; a.panoptic = False
from boosted_detr_amd.engine import to_device
m = bench.build_model(a)
import os
if os.environ.get("BDETR_DP_FORCE", "0") == "1":
    # the data-parallel step over a ONE-rank RCCL communicator (the multi-rank code path: bucket table, per-bucket all-reduces from the backward pass)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    m.distribute()
    print("data-parallel over a one-rank RCCL communicator", flush=True)
host = bench.make_batch(16, 640, 640, 100, 48 if a.fashionpedia else 82, 1234, **({'A': 296} if a.fashionpedia else {}))
batch = {"image": to_device(host["image"]), "category": host["category"], "attribute": host["attribute"], "bbox": to_device(host["bbox"]), "num_objects": to_device(host["num_objects"], torch.int32)}
GRAPH = len(sys.argv) > 3 and sys.argv[3] == "graph"
if GRAPH:
    batch["category"], batch["attribute"] = to_device(host["category"], torch.int32), to_device(host["attribute"], torch.int32)
    m.use_graph = True
else:
    c, h = m.Tokenization([host["category"], host["attribute"]]); m.Tokenization.call = lambda i, training=False: (c, h)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:
    m.train_gemm_precision = sys.argv[2]          # split | mixed | fp32 | bf16x3
print('policy', m.train_gemm_precision, 'graph replay' if GRAPH else 'eager', flush=True)
for blk in range(N // 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        logs = m.train_step(batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"steps {blk*50:4d}-{blk*50+49:4d}: {dt/50*1e3:6.2f} ms/step  loss {m.logs_to_host(logs)['loss']:.4f}  alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB  guard redos {m.range_redos}", flush=True)
if os.environ.get("BDETR_DP_FORCE", "0") == "1":
    torch.cuda.synchronize()
    dist.destroy_process_group()
