"""1x1 and 3x3 weight gradients at the ResNet-50 shapes (batch 16, 640x640) on pre-split operands: time per launch next to the bytes
both operands hold (read once) at 8 TB/s, and - with a diagnostic build (BDETR_CXXFLAGS="-DBDETR_SGEMM_DIAG -DBDETR_SGEMM_STAMPS", BDETR_LIB
pointing at it) and STAMPS=1 - the cycle stamps of one workgroup's first 12 K-steps (wait for loads / barrier / issue next stage / MFMAs).
Usage: python tools/wgrad_probe.py [xf16]"""
import ctypes as C
import os
import sys

STAMPS = os.environ.get("STAMPS", "0") == "1"
if STAMPS:
    os.environ["BDETR_SGEMM_DBG"] = "32"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import _lib, kernels as k

XF16 = len(sys.argv) > 1 and sys.argv[1] == "xf16"
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


shapes = [(160, 64, 256, 1), (160, 256, 64, 1), (80, 128, 512, 1), (80, 512, 128, 1), (40, 256, 1024, 1), (40, 1024, 256, 1), (20, 512, 2048, 1), (20, 2048, 512, 1),
          (160, 64, 64, 3), (80, 128, 128, 3), (40, 256, 256, 3), (20, 512, 512, 3)]
print(f"{'H':>4} {'C':>5} {'K':>5} {'RxS':>3}   ms/launch   operand bytes at 8 TB/s (ms)   frac   split-K")
with k.gemm_precision("split"):
    for (H, Cc, K_, R) in shapes:
        g = k.ConvGeom(B, H, H, Cc, K_, R, R, 1, R // 2)
        x = torch.randn(B, H, H, Cc, device="cuda")
        dy = torch.randn(B, H, H, K_, device="cuda")
        xf, xb = k.p16_pack(x, want_f16=XF16 and R == 1, want_bf16=not (XF16 and R == 1))
        _, dyb = k.p16_pack(dy, want_f16=False)
        dw = torch.zeros(K_, R, R, Cc, device="cuda")
        xin, isf16 = (xf, True) if (XF16 and R == 1) else (xb, False)
        fn = lambda: k.p16_conv2d_bwd_weight(xin, dyb, g, dw=dw, prezeroed=True, x_f16=isf16)
        t = timeit(fn)
        byts = B * H * H * (Cc + K_) * 4
        sk = _lib.lib().bdetr_p16_conv2d_bwd_weight_splitk(C.byref(g.desc()))
        print(f"{H:4d} {Cc:5d} {K_:5d} {R}x{R}   {t:8.4f}   {byts / 8e12 * 1e3:8.4f}   {byts / 8e12 * 1e3 / t:5.2f}   {sk}", flush=True)
        if STAMPS:
            buf = (C.c_uint64 * 48)()
            f2 = _lib.lib().bdetr_sgemm_debug_stamps
            f2.restype, f2.argtypes = C.c_int, [C.c_void_p, C.c_int]
            _lib.check(f2(buf, 48), "stamps")
            tt = list(buf)
            rows = []
            for kt in range(12):
                a, b, c, d = tt[4 * kt: 4 * kt + 4]
                prev = tt[4 * kt - 1] if kt else a
                rows.append((a - prev, b - a, c - b, d - c))
            med = lambda i: sorted(r[i] for r in rows[2:])[len(rows[2:]) // 2]
            print(f"      K-step cycles (median of steps 2-11): wait_loads {med(0)}  barrier {med(1)}  issue_next {med(2)}  mfma {med(3)}   12 steps: {tt[47] - tt[0]}", flush=True)
