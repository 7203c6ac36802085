"""A few launches of ONE pre-split conv product for `rocprofv3 --pmc` (counters of sgemm_kernel on a fixed shape).
Usage: python tools/p16_pmc_probe.py <fwd|dgrad|wgrad|wgradf16> H C K R [batch]   e.g.  fwd 40 256 256 3"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosted_detr_amd import kernels as k

which, H, Cc, K_, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 16
g = k.ConvGeom(B, H, H, Cc, K_, R, R, 1, R // 2)
x = torch.randn(B, H, H, Cc, device="cuda")
w = torch.randn(K_, R, R, Cc, device="cuda") * (R * R * Cc) ** -0.5
dy = torch.randn(B, g.OH, g.OW, K_, device="cuda")
bias = torch.zeros(K_, device="cuda")
dw = torch.zeros_like(w)
xf, xb = k.p16_pack(x)
wf, wt = k.p16_pack_conv_weights(w)
_, dyb = k.p16_pack(dy, want_f16=False)
fn = {"fwd": lambda: k.p16_conv2d_fwd(xf, wf, bias, g, 0, want_stats=True), "dgrad": lambda: k.p16_conv2d_bwd_data(dyb, wt, g),
      "wgrad": lambda: k.p16_conv2d_bwd_weight(xb, dyb, g, dw=dw, prezeroed=True),
      "wgradf16": lambda: k.p16_conv2d_bwd_weight(xf, dyb, g, dw=dw, prezeroed=True, x_f16=True)}[which]      # x as the forward's f16 pair (the 1x1 layers of the step)
for _ in range(6):
    fn()
torch.cuda.synchronize()
