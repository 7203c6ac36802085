"""1x1 backward-data at the ResNet-50 shapes (batch 16, 640x640) under each epilogue the training step uses: plain, fused
BatchNorm-backward sums (conv3 -> bn2), masked accumulate (conv1 of an identity unit: the skip gradient merged in), masked
accumulate + sums; each next to the bytes it has to move at 6.3 TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k

B = 16


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print(f"{'layer (input C -> K)':28s}  plain   +bn sums   masked accum   masked accum + sums   (ms; HBM floor of each in brackets)")
with k.gemm_precision("split"):
    for (H, Cc, K_) in [(160, 256, 64), (80, 512, 128), (40, 1024, 256), (20, 2048, 512), (160, 64, 256), (80, 128, 512), (40, 256, 1024), (20, 512, 2048)]:
        g = k.ConvGeom(B, H, H, Cc, K_, 1, 1, 1, 0)
        M = B * H * H
        w = torch.randn(K_, 1, 1, Cc, device="cuda") * Cc ** -0.5
        dy = torch.randn(B, H, H, K_, device="cuda")
        _, wt = k.p16_pack_conv_weights(w)
        _, dyb = k.p16_pack(dy, want_f16=False)
        y_prev = torch.randn(B, H, H, Cc, device="cuda")
        mean, rstd, gamma, beta = (torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda"), torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda"))
        bits = torch.randint(-2 ** 62, 2 ** 62, ((M * Cc + 255) // 256 * 4,), device="cuda", dtype=torch.int64)
        dx = torch.randn(B, H, H, Cc, device="cuda")
        t = [timeit(lambda: k.p16_conv2d_bwd_data(dyb, wt, g)),
             timeit(lambda: k.p16_conv2d_bwd_data_bnstats(dyb, wt, g, y_prev, mean, rstd, gamma, beta, True)),
             timeit(lambda: k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dx, bits)),
             timeit(lambda: k.p16_conv2d_bwd_data_masked_accum(dyb, wt, g, dx, bits, (y_prev, mean, rstd, gamma, beta, bits)))]
        rd, wr = M * K_ * 4, M * Cc * 4
        fl = [(rd + wr), (rd + 2 * wr), (rd + 2 * wr + wr / 32), (rd + 3 * wr + wr / 16)]     # + y read; + old read + mask; old, y (for xhat) and both masks
        print(f"{H:3d}x{H:<3d} dy K{K_:<4d} -> dx C{Cc:<4d}   " + "   ".join(f"{a:6.3f} [{b / 6.3e12 * 1e3:5.3f}]" for a, b in zip(t, fl)), flush=True)
