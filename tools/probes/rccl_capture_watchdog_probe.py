"""Which event query of ProcessGroupNCCL's watchdog thread fails while a hipGraph capture is open?  (round 5; the round-4 abort:
profiles/r04_sigabrt_capture_vs_rccl_watchdog.log)

One-rank RCCL communicator, one variant per process (an abort on the watchdog thread kills the process: the caller reads the exit code).
    python rccl_capture_watchdog_probe.py <variant> <capture_error_mode>

variants
    eager_then_capture      64 eager async all-reduces, no host wait, then a capture held open for 1.5 s
    captured_then_capture   a capture that contains an all-reduce, ended; then a second capture held open for 1.5 s
    captured_held_open      a capture that contains an all-reduce and stays open for 1.5 s after it
    captured_eager_capture  a capture with an all-reduce, ended; 64 eager all-reduces (event objects reused); a capture held open 1.5 s
"""
import os
import sys
import time

import torch
import torch.distributed as dist

variant, mode = sys.argv[1], sys.argv[2]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
y = torch.ones(1 << 16, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
time.sleep(0.3)
pool = torch.cuda.graph_pool_handle()
stream = torch.cuda.Stream()


def hold_open(seconds, collective=False):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode=mode):
        if collective:
            dist.all_reduce(x)
        t0 = time.time()
        while time.time() - t0 < seconds:
            y.mul_(1.0)
            time.sleep(5e-3)
    return g


if variant == "eager_then_capture":
    hs = [dist.all_reduce(x, async_op=True) for _ in range(64)]
    hold_open(1.5)
elif variant == "captured_then_capture":
    g1 = hold_open(0.0, collective=True)
    g2 = hold_open(1.5)
elif variant == "captured_held_open":
    g1 = hold_open(1.5, collective=True)
elif variant == "captured_eager_capture":
    g1 = hold_open(0.0, collective=True)
    for _ in range(64):
        dist.all_reduce(x)
    hs = [dist.all_reduce(x, async_op=True) for _ in range(64)]
    g2 = hold_open(1.5)
else:
    raise SystemExit("unknown variant")
torch.cuda.synchronize()
time.sleep(0.3)
print("PROBE_OK", variant, mode, flush=True)
dist.destroy_process_group()
