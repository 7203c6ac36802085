"""Soak: N training steps at config 2, reporting step time, loss and allocator high-water marks (leak check).
usage: python tools/soak.py [steps] [policy]   - run once per arithmetic policy to compare the loss trajectories."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
class A: pass
a = A(); a.queries = 100; a.image = 640; a.layers = 6; a.batch = 16; a.model = 'detr'; a.fashionpedia = False; a.image_w = 0; a.learners = 3; a.backbone = 'ResNet'
from boosted_detr_amd.engine import to_device
m = bench.build_model(a)
host = bench.make_batch(16, 640, 640, 100, 82, 1234)
batch = {"image": to_device(host["image"]), "category": host["category"], "attribute": host["attribute"], "bbox": to_device(host["bbox"]), "num_objects": to_device(host["num_objects"], torch.int32)}
c, h = m.Tokenization([host["category"], host["attribute"]]); m.Tokenization.call = lambda i, training=False: (c, h)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:
    m.train_gemm_precision = sys.argv[2]          # split | mixed | fp32 | bf16x3
print('policy', m.train_gemm_precision, flush=True)
for blk in range(N // 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        logs = m.train_step(batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"steps {blk*50:4d}-{blk*50+49:4d}: {dt/50*1e3:6.2f} ms/step  loss {m.logs_to_host(logs)['loss']:.4f}  alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
