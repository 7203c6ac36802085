"""Which PyTorch ops (copies, fills, elementwise) still run inside one training step (they are plumbing, not kernels of ours)."""
import sys, torch
sys.path.insert(0, '.')
import bench
from torch.profiler import profile, ProfilerActivity
class A: pass
a = A(); a.queries = 100; a.image = 640; a.layers = 6; a.batch = 16; a.model = 'detr'; a.fashionpedia = False; a.image_w = 0; a.learners = 3; a.backbone = 'ResNet'
from boosted_detr_amd.engine import to_device
m = bench.build_model(a)
host = bench.make_batch(16, 640, 640, 100, 82, 1234)
batch = {"image": to_device(host["image"]), "category": to_device(host["category"], torch.int32), "attribute": to_device(host["attribute"], torch.int32), "bbox": to_device(host["bbox"]), "num_objects": to_device(host["num_objects"], torch.int32)}
for _ in range(3): m.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    m.train_step(batch)
torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=40))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=30, max_name_column_width=40, max_src_column_width=90))
