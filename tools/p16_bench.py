"""Per-layer timing of the ResNet-50 convolutions at the bench's batch (16 x 640 x 640): the in-kernel split path
(igemm.hip under the 'split' policy) next to the pre-split P16 path (sgemm.hip).  Usage: python tools/p16_bench.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ONLY_WGRAD = len(sys.argv) > 2 and sys.argv[2] == "wgrad"
ONLY_P16 = len(sys.argv) > 2 and sys.argv[2] == "p16"          # skip the in-kernel split path (halves the run)
ITERS = 10
# (H, C, K, R, stride) of the distinct conv shapes of ResNet-50 at 640x640 (input side H x H), with their multiplicity
LAYERS = [(160, 64, 64, 1, 1, 1), (160, 64, 64, 3, 1, 3), (160, 64, 256, 1, 1, 4), (160, 256, 64, 1, 1, 2),
          (160, 256, 128, 1, 2, 1), (160, 256, 512, 1, 2, 1), (80, 128, 128, 3, 1, 4), (80, 128, 512, 1, 1, 4), (80, 512, 128, 1, 1, 3),
          (80, 512, 256, 1, 2, 1), (80, 512, 1024, 1, 2, 1), (40, 256, 256, 3, 1, 6), (40, 256, 1024, 1, 1, 6), (40, 1024, 256, 1, 1, 5),
          (40, 1024, 512, 1, 2, 1), (40, 1024, 2048, 1, 2, 1), (20, 512, 512, 3, 1, 3), (20, 512, 2048, 1, 1, 3), (20, 2048, 512, 1, 1, 2)]


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


tot = {"old": [0.0, 0.0, 0.0], "p16": [0.0, 0.0, 0.0]}
print(f"{'layer':34s} {'GFLOP':>7s} | old fwd/dgrad/wgrad ms (TF/s)            | p16 fwd/dgrad/wgrad ms (TF/s)")
with k.gemm_precision("split"):
    for (H, Cc, K_, R, s, mult) in LAYERS:
        g = k.ConvGeom(B, H, H, Cc, K_, R, R, s, R // 2)
        x = torch.randn(B, H, H, Cc, device="cuda")
        w = torch.randn(K_, R, R, Cc, device="cuda") * (R * R * Cc) ** -0.5
        bias = torch.zeros(K_, device="cuda")
        dy = torch.randn(B, g.OH, g.OW, K_, device="cuda")
        dw = torch.zeros_like(w)
        gf = 2.0 * g.M * K_ * R * R * Cc / 1e9
        xf, xb = k.p16_pack(x)
        wf, wt = k.p16_pack_conv_weights(w)
        _, dyb = k.p16_pack(dy, want_f16=False)
        if ONLY_WGRAD:
            t_old = [1e-9, 1e-9, 1e-9]
            t_new = [1e-9, 1e-9, timeit(lambda: k.p16_conv2d_bwd_weight(xb, dyb, g, dw=dw, prezeroed=True))]
        else:
            t_old = [1e-9, 1e-9, 1e-9] if ONLY_P16 else [timeit(lambda: k.conv2d_fwd(x, w, bias, g, 0, want_stats=True)), timeit(lambda: k.conv2d_bwd_data(dy, w, g)),
                                                         timeit(lambda: k.conv2d_bwd_weight(x, dy, g, dw=dw, prezeroed=True))]
            t_new = [timeit(lambda: k.p16_conv2d_fwd(xf, wf, bias, g, 0, want_stats=True)), timeit(lambda: k.p16_conv2d_bwd_data(dyb, wt, g)),
                     timeit(lambda: k.p16_conv2d_bwd_weight(xb, dyb, g, dw=dw, prezeroed=True))]
        for i in range(3):
            tot["old"][i] += mult * t_old[i]; tot["p16"][i] += mult * t_new[i]
        f = lambda ts: " ".join(f"{t:6.3f}({gf / t:5.0f})" for t in ts)
        # HBM floor of one pass: both activation tensors once at 4 bytes per element (P16 pair in, fp32 out), at the 6.3 TB/s a copy reaches
        hbm_ms = (B * H * H * Cc + g.M * K_) * 4 / 6.3e12 * 1e3
        print(f"{H:3d}x{H:<3d} C{Cc:<4d} K{K_:<4d} {R}x{R} s{s} x{mult:<2d}     {gf:7.1f} | {f(t_old)} | {f(t_new)} | hbm floor {hbm_ms:6.3f} ms, fwd/dgrad/wgrad at "
              + "/".join(f"{hbm_ms / t:4.2f}" for t in t_new), flush=True)
print("per step (weighted by multiplicity): old fwd %.2f dgrad %.2f wgrad %.2f = %.2f ms | p16 fwd %.2f dgrad %.2f wgrad %.2f = %.2f ms"
      % (*tot["old"], sum(tot["old"]), *tot["p16"], sum(tot["p16"])))
