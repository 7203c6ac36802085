"""GEMM / conv micro-benchmark over the model's shapes, both arithmetic modes, with the error of each
against an fp64 product.  Usage: python tools/gemm_bench.py [fp32|bf16x3|both]   (BDETR_TILE=<bm>x<bn> forces a tile)"""
import sys, torch
sys.path.insert(0, '.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
modes = ["fp32", "bf16x3"] if len(sys.argv) < 2 or sys.argv[1] == "both" else [sys.argv[1]]

def bench(name, fn, flops, ref=None, iters=20):
    out = fn()
    err = ""
    if ref is not None:
        r = ref()
        err = f" relL2 {float((out.double() - r).norm() / r.norm()):.2e} max {float((out.double() - r).abs().max() / r.abs().max()):.2e}"
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:46s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF/s{err}", flush=True)

shapes = [(25600,1024,1024),(25600,256,1024),(102400,512,128),(409600,256,64),(409600,64,256),(6400,2048,512),(6400,512,2048),(6400,256,256),(1600,2048,256),(8192,8192,1024)]
for mode in modes:
    k.set_gemm_precision(mode)
    print("==", mode, flush=True)
    for (M, O, K) in shapes:
        x = torch.randn(M, K, device='cuda'); w = torch.randn(O, K, device='cuda'); b = torch.randn(O, device='cuda'); dy = torch.randn(M, O, device='cuda')
        chk = M * O * K < 3e10
        bench(f"fwd  RCxRC M={M} O={O} K={K}", lambda: k.linear_fwd(x, w, b, 0), 2*M*O*K, (lambda: x.double() @ w.double().t() + b.double()) if chk else None)
        bench(f"bwdD RCxXC M={M} O={O} K={K}", lambda: k.linear_bwd_data(dy, w), 2*M*O*K, (lambda: dy.double() @ w.double()) if chk else None)
        bench(f"bwdW XCxXC M={M} O={O} K={K}", lambda: k.linear_bwd_weight(dy, x), 2*M*O*K, (lambda: dy.double().t() @ x.double()) if chk else None)
    g = k.ConvGeom(16, 40, 40, 256, 256, 3, 3, 1, 1)
    x = torch.randn(16, 40, 40, 256, device='cuda'); w = torch.randn(256, 3, 3, 256, device='cuda'); b = torch.randn(256, device='cuda'); dy = torch.randn(16, 40, 40, 256, device='cuda')
    fl = 2 * g.M * 256 * 2304
    import torch.nn.functional as F
    xd, wd = x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)
    bench("conv3x3 fwd 16x40x40x256", lambda: k.conv2d_fwd(x, w, b, g, 0, True)[0], fl,
          lambda: (F.conv2d(xd, wd, b.double(), padding=1)).permute(0, 2, 3, 1))
    bench("conv3x3 bwd-data", lambda: k.conv2d_bwd_data(dy, w, g), fl,
          lambda: torch.nn.grad.conv2d_input(xd.shape, wd, dy.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1))
    bench("conv3x3 bwd-weight", lambda: k.conv2d_bwd_weight(x, dy, g), fl,
          lambda: torch.nn.grad.conv2d_weight(xd, wd.shape, dy.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1))
