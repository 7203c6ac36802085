"""Accuracy of the GEMM arithmetic modes against fp64 on benign and on hostile operand ranges."""
import sys, torch
sys.path.insert(0, '.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(0)
def logu(shape, lo, hi):
    e = torch.rand(shape, generator=g) * (hi - lo) + lo
    return (10.0 ** e) * (torch.randint(0, 2, shape, generator=g) * 2 - 1)
M, K, O = 2048, 1024, 512
cases = {
    "normal x, w~1/32": (torch.randn(M, K, generator=g), torch.randn(O, K, generator=g) / 32),
    "x in 1e-8..1e3 log-uniform, w~1e-3": (logu((M, K), -8, 3), torch.randn(O, K, generator=g) * 1e-3),
    "x~1, w in 1e-9..1e-1 log-uniform": (torch.randn(M, K, generator=g), logu((O, K), -9, -1)),
    "x~1e-6, w~1e-5 (all tiny)": (torch.randn(M, K, generator=g) * 1e-6, torch.randn(O, K, generator=g) * 1e-5),
    "x~3e4 (near fp16 max), w~1": (torch.randn(M, K, generator=g).clamp(-2, 2) * 1.5e4, torch.randn(O, K, generator=g)),
    "relu-sparse x (half zeros), w~1/32": (torch.randn(M, K, generator=g).relu(), torch.randn(O, K, generator=g) / 32),
}
for name, (x, w) in cases.items():
    ref = x.double() @ w.double().T
    absdot = x.double().abs() @ w.double().abs().T          # condition-free scale of each output
    print(name)
    for mode in ["fp32", "bf16x3", "split"]:
        k.set_gemm_precision(mode)
        y = k.linear_fwd(x.cuda(), w.cuda(), None, 0).cpu().double()
        print(f"   {mode:7s} relL2 {float((y-ref).norm()/ref.norm()):.2e}   max |err| / sum|a||b| {float(((y-ref).abs()/absdot).max()):.2e}   finite {bool(torch.isfinite(y).all())}")
k.set_gemm_precision("mixed")
