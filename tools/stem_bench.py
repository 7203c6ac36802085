"""Times the fused stem tail (csrc/norm.hip stem_*) at configs[1]'s shape (16 x 320 x 320 x 64 raw conv1 output) next to the unfused
kernels it replaces; prints one JSON object (microseconds per launch, algorithmic GB/s)."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from boosted_detr_amd import kernels as k  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    N, H, W, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 320, 320, 64)
    torch.manual_seed(0)
    y = torch.randn(N, H, W, C, device="cuda") * 1.5 + 0.3
    mean, rstd = y.mean((0, 1, 2)), 1.0 / torch.sqrt(y.var((0, 1, 2), unbiased=False) + 1e-5)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    _, of, tap = k.stem_pool_fwd(y, mean, rstd, gamma, beta)
    dpool = torch.randn(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, device="cuda")
    full, pooled = y.numel() * 4, dpool.numel() * 4
    out = {"shape": [N, H, W, C]}
    t = timed(lambda: k.stem_pool_fwd(y, mean, rstd, gamma, beta))
    out["stem_pool_fwd_us"] = round(t, 1); out["stem_pool_fwd_gbs"] = round((full + pooled + pooled / 4) / t / 1e3, 0)
    t = timed(lambda: k.stem_pool_bwd(dpool, tap, y, mean, rstd, gamma, beta))
    out["stem_pool_bwd_us"] = round(t, 1); out["stem_pool_bwd_gbs"] = round((3 * full + 2 * (pooled + pooled / 4)) / t / 1e3, 0)
    # the unfused chain
    y2 = y.view(-1, C)
    a = k.bn_apply(y2, mean, rstd, gamma, beta, None, True).view(N, H, W, C)
    p = k.maxpool_fwd(a)
    t = timed(lambda: (k.bn_apply(y2, mean, rstd, gamma, beta, None, True), k.maxpool_fwd(a), k.p16_pack(p, want_f16=True, want_bf16=False)))
    out["unfused_fwd_us"] = round(t, 1)
    da = k.maxpool_bwd(a, p, dpool)
    t = timed(lambda: (k.maxpool_bwd(a, p, dpool), k.bn_bwd(da.view(-1, C), None, y2, mean, rstd, gamma, True, False, beta=beta)))
    out["unfused_bwd_us"] = round(t, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
