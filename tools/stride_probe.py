"""Does the row pitch of the operands (power-of-two K -> every tile row on the same L2 channel) limit the staging rate?"""
import sys, torch
sys.path.insert(0, '.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
k.set_gemm_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, O = 25600, 1024
for K in (1024, 1056, 1088, 2048, 2080, 512, 544, 256, 288):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(O, K, device='cuda'); dy = torch.randn(M, O, device='cuda')
    f = bench(lambda: k.linear_fwd(x, w, None, 0)); d = bench(lambda: k.linear_bwd_data(dy, w)); g = bench(lambda: k.linear_bwd_weight(dy, x))
    fl = 2 * M * O * K
    print(f"K={K:5d}  fwd {fl/f/1e9:6.1f}  bwdD {fl/d/1e9:6.1f}  bwdW {fl/g/1e9:6.1f} TF/s", flush=True)
