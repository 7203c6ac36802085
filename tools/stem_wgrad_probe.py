import ctypes as C, os, sys, torch
sys.path.insert(0, '/root/repo')
from boosted_detr_amd import _lib, kernels as k
from boosted_detr_amd.kernels import _p, _stream, check
L = _lib.lib()
N = 16
g = k.ConvGeom(N, 640, 640, 4, 64, 7, 7, 2, 3)
x = torch.randn(N, 640, 640, 4, device="cuda"); dy = torch.randn(N, 320, 320, 64, device="cuda")
dw = torch.zeros(64, 7, 7, 4, device="cuda")
d = g.desc()
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
with k.gemm_precision("split"):
    print("default splitk", L.bdetr_conv2d_bwd_weight_splitk(C.byref(d)))
    for sk in (0, 64, 128, 256, 512, 1024, 2048):
        def fn():
            check(L.bdetr_conv2d_bwd_weight_ws(_p(x), _p(dy), _p(dw), C.byref(d), sk, None, 0, _stream()), "w")
        try:
            print(sk, round(timeit(fn) * 1e3, 1), "us")
        except Exception as e:
            print(sk, "ERR", str(e)[:100])
