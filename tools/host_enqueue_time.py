"""Host enqueue time vs GPU time of one training step (is the step GPU-bound?)."""
import sys, time, torch
sys.path.insert(0,'.')
import bench
class A: pass
a=A(); a.queries=100; a.image=640; a.layers=6; a.batch=16; a.model='detr'; a.fashionpedia=False; a.image_w=0; a.learners=3; a.backbone='ResNet'
from boosted_detr_amd.engine import to_device
m=bench.build_model(a)
host=bench.make_batch(16,640,640,100,82,1234)
batch={"image":to_device(host["image"]),"category":to_device(host["category"],torch.int32),"attribute":to_device(host["attribute"],torch.int32),"bbox":to_device(host["bbox"]),"num_objects":to_device(host["num_objects"],torch.int32)}
for _ in range(3): m.train_step(batch)
torch.cuda.synchronize()
for _ in range(3):
    t0=time.perf_counter(); m.train_step(batch); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
import cProfile, pstats
pr=cProfile.Profile(); pr.enable(); m.train_step(batch); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
