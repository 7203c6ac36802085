"""The stem's weight gradient at the bench's size (16 x 640 x 640): igemm.hip's in-kernel-split kernel against the space-to-depth path on the
pre-split XX kernel (pack + zero + product + unpack), per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k
N = 16
x = torch.randn(N, 640, 640, 4, device="cuda"); x[..., 3] = 0
dy = torch.randn(N, 320, 320, 64, device="cuda")
g = k.ConvGeom(N, 640, 640, 4, 64, 7, 7, 2, 3)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
with k.gemm_precision("split"):
    _, dyb = k.p16_pack(dy, want_f16=False)
    dw = torch.zeros(64, 7, 7, 4, device="cuda")
    print("igemm (in-kernel split, split-K 1024): %.1f us" % timeit(lambda: k.conv2d_bwd_weight(x, dy, g, dw=dw, prezeroed=True)))
    print("space-to-depth on the pre-split XX kernel (pack + zero + product + unpack): %.1f us" % timeit(lambda: k.stem_bwd_weight_s2d(x, dyb, dw)))
