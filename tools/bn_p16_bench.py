"""HBM rate of the P16 BatchNorm producers on the model's activation shapes (rows x C), isolated."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k


def bench(name, fn, nbytes, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:46s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:6.0f} GB/s", flush=True)
    return ms


tot = 0.0
# (rows, C, launches per step without residual, with residual)
for rows, C, n, nres in [(409600, 64, 6, 0), (409600, 256, 1, 3), (102400, 128, 8, 0), (102400, 512, 1, 4), (25600, 256, 12, 0), (25600, 1024, 1, 6), (6400, 512, 6, 0), (6400, 2048, 1, 2)]:
    x, dy = torch.randn(rows, C, device="cuda"), torch.randn(rows, C, device="cuda")
    mean, rstd, gamma, beta = [torch.rand(C, device="cuda") + 0.5 for _ in range(4)]
    _, resf, _ = k.bn_apply_p16(torch.randn(rows, C, device="cuda"), mean, rstd, gamma, beta, None, True, want_fp32=False)
    T = rows * C * 4
    a = bench(f"apply          -> f16+bf16      {rows}x{C}", lambda: k.bn_apply_p16(x, mean, rstd, gamma, beta, None, True, want_fp32=False), 3 * T)
    b = bench(f"apply +res f16 -> f16+bf16+bits {rows}x{C}", lambda: k.bn_apply_p16(x, mean, rstd, gamma, beta, resf, True, want_fp32=False, residual_p16=True, want_mask=True), 4 * T)
    _, _, _, bits = k.bn_apply_p16(x, mean, rstd, gamma, beta, resf, True, want_fp32=False, residual_p16=True, want_mask=True)
    c = bench(f"bwd (reduce + apply) recompute   {rows}x{C}", lambda: k.bn_bwd_p16(dy, None, x, mean, rstd, gamma, True, False, beta=beta), 5 * T)
    d = bench(f"bwd (reduce + apply) bit mask    {rows}x{C}", lambda: k.bn_bwd_p16(dy, bits, x, mean, rstd, gamma, True, False, beta=beta, out_p16=2), 5 * T)
    tot += a * n + b * nres + c * n + d * nres
print("model mix, isolated kernels: %.2f ms per step" % tot)
