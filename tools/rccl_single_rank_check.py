import os, torch, torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
flat = torch.ones(31_000_000, device="cuda")
hs = [dist.all_reduce(flat[o:o + 8 * 1024 * 1024], op=dist.ReduceOp.SUM, async_op=True) for o in range(0, flat.numel(), 8 * 1024 * 1024)]
for h in hs: h.wait()
dist.broadcast(flat[:100], src=0)
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl single-rank ok", float(flat.sum()), float(t))
dist.destroy_process_group()
