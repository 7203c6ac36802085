import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boosted_detr_amd import kernels as k
def bench(name, fn, nbytes, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:52s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:6.0f} GB/s", flush=True)
for rows, C in [(409600, 64), (409600, 256), (102400, 512), (25600, 1024)]:
    x, dy = torch.randn(rows, C, device="cuda"), torch.randn(rows, C, device="cuda")
    mean, rstd, gamma, beta = [torch.rand(C, device="cuda") + 0.5 for _ in range(4)]
    _, resf, _ = k.bn_apply_p16(torch.randn(rows, C, device="cuda"), mean, rstd, gamma, beta, None, True, want_fp32=False, want_bf16=False)
    raw = torch.randn(rows, C, device="cuda")
    T = rows * C * 4
    bench(f"apply -> f16                      {rows}x{C}", lambda: k.bn_apply_p16(x, mean, rstd, gamma, beta, None, True, want_fp32=False, want_bf16=False), 2 * T)
    bench(f"apply -> f16+bf16                 {rows}x{C}", lambda: k.bn_apply_p16(x, mean, rstd, gamma, beta, None, True, want_fp32=False), 3 * T)
    bench(f"apply +res f16 -> f16+bits        {rows}x{C}", lambda: k.bn_apply_p16(x, mean, rstd, gamma, beta, resf, True, want_fp32=False, want_bf16=False, residual_p16=True, want_mask=True), 3 * T)
    bench(f"apply +res raw+bn -> f16+bits     {rows}x{C}", lambda: k.bn_apply_p16(x, mean, rstd, gamma, beta, raw, True, want_fp32=False, want_bf16=False, want_mask=True, residual_bn=(mean, rstd, gamma, beta)), 3 * T)
    _, _, _, bits = k.bn_apply_p16(x, mean, rstd, gamma, beta, resf, True, want_fp32=False, want_bf16=False, residual_p16=True, want_mask=True)
    ws_pre = None
    bench(f"bwd apply only (pre sums) recompute {rows}x{C}", lambda: k.bn_bwd_p16(dy, None, x, mean, rstd, gamma, True, False, beta=beta, pre=(torch.zeros(8, C, device='cuda'), torch.zeros(8, C, device='cuda'), 8)), 3 * T)
    bench(f"bwd apply only (pre sums) bit mask  {rows}x{C}", lambda: k.bn_bwd_p16(dy, bits, x, mean, rstd, gamma, True, False, beta=beta, out_p16=2, pre=(torch.zeros(8, C, device='cuda'), torch.zeros(8, C, device='cuda'), 8)), 3 * T)
    a = torch.empty(rows * C, device="cuda"); b = torch.empty(rows * C, device="cuda")
    bench(f"torch copy                         {rows}x{C}", lambda: b.copy_(a), 2 * T)
