"""HBM efficiency of the BatchNorm kernels on the model's activation shapes (rows x C)."""
import sys, torch
sys.path.insert(0, '.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
def bench(name, fn, nbytes, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:44s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:7.0f} GB/s", flush=True)
    return ms
tot = {"apply": 0.0, "apply_res": 0.0, "bwd": 0.0}
# (rows, C, count per step, count with residual)
shapes = [(1638400, 64, 1, 0), (409600, 64, 6, 0), (409600, 256, 4, 3), (102400, 128, 8, 0), (102400, 512, 5, 4), (25600, 256, 12, 0), (25600, 1024, 7, 6), (6400, 512, 6, 0), (6400, 2048, 4, 3)]
for rows, C, n, nres in shapes:
    x = torch.randn(rows, C, device='cuda'); dy = torch.randn(rows, C, device='cuda'); res = torch.randn(rows, C, device='cuda')
    mean, rstd, gamma, beta = [torch.rand(C, device='cuda') + 0.5 for _ in range(4)]
    T = rows * C * 4
    a = bench(f"bn_apply relu        {rows}x{C}", lambda: k.bn_apply(x, mean, rstd, gamma, beta, None, True), 2 * T)
    b = bench(f"bn_apply +res relu   {rows}x{C}", lambda: k.bn_apply(x, mean, rstd, gamma, beta, res, True), 3 * T)
    c = bench(f"bn_bwd (stats+apply) {rows}x{C}", lambda: k.bn_bwd(dy, None, x, mean, rstd, gamma, True, False, beta=beta), 5 * T)
    tot["apply"] += a * (n - nres); tot["apply_res"] += b * nres; tot["bwd"] += c * n
print({k_: round(v, 3) for k_, v in tot.items()}, "ms per step (model mix, isolated kernels)")
