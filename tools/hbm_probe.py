"""HBM ceilings of this box as plain PyTorch fills / copies see them (what a write-dominated epilogue can hope for)."""
import torch


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for mb in (105, 419, 1678):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device="cuda"); b = torch.randn(n, device="cuda")
    t = timeit(lambda: a.zero_()); print(f"{mb:5d} MB zero_  {t:.3f} ms  {mb / t / 1e3:.2f} TB/s written")
    t = timeit(lambda: a.fill_(1.5)); print(f"{mb:5d} MB fill_  {t:.3f} ms  {mb / t / 1e3:.2f} TB/s written")
    t = timeit(lambda: a.copy_(b)); print(f"{mb:5d} MB copy_  {t:.3f} ms  {2 * mb / t / 1e3:.2f} TB/s read+written")
    t = timeit(lambda: b.sum()); print(f"{mb:5d} MB sum    {t:.3f} ms  {mb / t / 1e3:.2f} TB/s read")
