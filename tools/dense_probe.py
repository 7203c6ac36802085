"""Transformer-sized Dense products (M x 256 x 256, M = 6400 / 1600) on the two GEMM paths: igemm.hip (fp32 operands split in the kernel,
what the transformer layers launch today) and sgemm.hip (operands pre-split by their producer) - is moving LayerNorm / attention outputs to
P16 producers worth it?  Usage: python tools/dense_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


for M, Cin, Cout in ((6400, 256, 256), (1600, 256, 256), (1600, 256, 1024), (1600, 1024, 82 - 2)):
    Cout8 = (Cout + 7) // 8 * 8
    x = torch.randn(M, Cin, device="cuda")
    w = torch.randn(Cout8, Cin, device="cuda") * Cin ** -0.5
    b = torch.zeros(Cout8, device="cuda")
    dy = torch.randn(M, Cout8, device="cuda")
    dw = torch.zeros_like(w)
    with k.gemm_precision("split"):
        t_old = [timeit(lambda: k.linear_fwd(x, w, b, 1)), timeit(lambda: k.linear_bwd_data(dy, w)), timeit(lambda: k.linear_bwd_weight(dy, x, dw=dw, prezeroed=True))]
        g = k.ConvGeom(1, M, 1, Cin, Cout8, 1, 1, 1, 0)
        xf, xb = k.p16_pack(x.view(1, M, 1, Cin))
        wf, wt = k.p16_pack_conv_weights(w.view(Cout8, 1, 1, Cin))
        _, dyb = k.p16_pack(dy.view(1, M, 1, Cout8), want_f16=False)
        dw4 = dw.view(Cout8, 1, 1, Cin)
        t_new = [timeit(lambda: k.p16_conv2d_fwd(xf, wf, b, g, 1)), timeit(lambda: k.p16_conv2d_bwd_data(dyb, wt, g)),
                 timeit(lambda: k.p16_conv2d_bwd_weight(xb, dyb, g, dw=dw4, prezeroed=True))]
    print(f"M {M:5d} {Cin:4d} -> {Cout8:4d}: igemm fwd/dgrad/wgrad {t_old[0]:6.1f} {t_old[1]:6.1f} {t_old[2]:6.1f} us | sgemm (P16) {t_new[0]:6.1f} {t_new[1]:6.1f} {t_new[2]:6.1f} us", flush=True)
