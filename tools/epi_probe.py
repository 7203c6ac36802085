"""Where does a short-K 1x1 layer spend its time?  BDETR_SGEMM_DBG bits: 1 = no C stores, 2 = no K loop (no loads / MFMAs),
4 = per-element stores from the accumulators (no LDS transposition), 8 = no fragment reads / MFMAs, 16 = no staging loads.
Diagnostic builds only: BDETR_CXXFLAGS=-DBDETR_SGEMM_DIAG python -m boosted_detr_amd.build --force (the production library
never reads BDETR_SGEMM_DBG)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k

LAYERS = [(160, 64, 256, 1, 1), (160, 256, 64, 1, 1), (40, 256, 1024, 1, 1), (40, 1024, 256, 1, 1), (160, 64, 64, 3, 1), (40, 256, 256, 3, 1)]


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


print("dbg", os.environ.get("BDETR_SGEMM_DBG", "0"), "tile", os.environ.get("BDETR_STILE", "auto"))
for (H, Cc, K_, R, s) in LAYERS:
    g = k.ConvGeom(16, H, H, Cc, K_, R, R, s, R // 2)
    x = torch.randn(16, H, H, Cc, device="cuda")
    w = torch.randn(K_, R, R, Cc, device="cuda") * (R * R * Cc) ** -0.5
    dy = torch.randn(16, g.OH, g.OW, K_, device="cuda")
    dw = torch.zeros_like(w)
    xf, xb = k.p16_pack(x)
    wf, wt = k.p16_pack_conv_weights(w)
    _, dyb = k.p16_pack(dy, want_f16=False)
    ts = [timeit(lambda: k.p16_conv2d_fwd(xf, wf, None, g, 0, want_stats=True)), timeit(lambda: k.p16_conv2d_fwd(xf, wf, None, g, 0, want_stats=False)),
          timeit(lambda: k.p16_conv2d_bwd_data(dyb, wt, g)), timeit(lambda: k.p16_conv2d_bwd_weight(xb, dyb, g, dw=dw, prezeroed=True))]
    print(f"{H:3d}x{H:<3d} C{Cc:<4d} K{K_:<4d} {R}x{R}  fwd+stats {ts[0]:.3f}  fwd {ts[1]:.3f}  dgrad {ts[2]:.3f}  wgrad {ts[3]:.3f} ms", flush=True)
