"""A handful of launches of one GEMM flavour for PMC collection:
  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python3 tools/gemm_pmc_probe.py [mode] [M O K] [fwd|bwdD|bwdW]
(mode: fp32 | bf16x3 | split; BDETR_TILE forces a tile)."""
import sys, torch
sys.path.insert(0, '.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
M, O, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (25600, 1024, 1024)
what = sys.argv[5] if len(sys.argv) > 5 else "fwd"
k.set_gemm_precision(mode)
x = torch.randn(M, K, device='cuda'); w = torch.randn(O, K, device='cuda'); b = torch.randn(O, device='cuda'); dy = torch.randn(M, O, device='cuda')
fn = {"fwd": lambda: k.linear_fwd(x, w, b, 0), "bwdD": lambda: k.linear_bwd_data(dy, w), "bwdW": lambda: k.linear_bwd_weight(dy, x)}[what]
for _ in range(10): fn()
torch.cuda.synchronize()
