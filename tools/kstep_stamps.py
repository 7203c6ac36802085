"""Cycle stamps of one workgroup's first 12 K-steps of a pre-split 3x3 forward launch (BDETR_SGEMM_DBG=32 is set here):
per step the time spent waiting for the stage's loads, in the barrier, issuing the next stage, and issuing the MFMAs.
Needs a diagnostic build of the library: BDETR_CXXFLAGS="-DBDETR_SGEMM_DIAG -DBDETR_SGEMM_STAMPS" python -m boosted_detr_amd.build --force
(the stamp array, its entry point and every BDETR_SGEMM_DBG switch are compiled out of the production library).  Usage: python tools/kstep_stamps.py [H C K R]"""
import ctypes as C
import os
import sys

os.environ["BDETR_SGEMM_DBG"] = "32"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import _lib, kernels as k

H, Cc, K_, R = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (40, 256, 256, 3)
g = k.ConvGeom(16, H, H, Cc, K_, R, R, 1, R // 2)
x = torch.randn(16, H, H, Cc, device="cuda")
w = torch.randn(K_, R, R, Cc, device="cuda") * (R * R * Cc) ** -0.5
xf, _ = k.p16_pack(x, want_bf16=False)
wf, _ = k.p16_pack_conv_weights(w)
for _ in range(3):
    k.p16_conv2d_fwd(xf, wf, None, g, 0, want_stats=False)
torch.cuda.synchronize()
buf = (C.c_uint64 * 48)()
fn = _lib.lib().bdetr_sgemm_debug_stamps          # diagnostic builds only: not part of include/bdetr.h
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
_lib.check(fn(buf, 48), "stamps")
t = list(buf)
print("step  wait_loads  barrier  issue_next  mfma_issue   (cycles; the step's total is the sum + the next step's wait)")
for kt in range(12):
    a, b, c, d = t[4 * kt: 4 * kt + 4]
    prev = t[4 * kt - 1] if kt else a
    print(f"{kt:3d}  {a - prev:10d} {b - a:8d} {c - b:11d} {d - c:11d}")
print("12 steps:", t[47] - t[0], "cycles")
