"""Per-launch time of the fused attention kernels (csrc/attention.hip) at the step's shapes: encoder self-attention (16 x 8 heads,
400 x 400), decoder cross-attention (100 x 400) and decoder self-attention (100 x 100).  `python tools/attn_bench.py [fwd|bwd|all]
[enc|cross|self|all] [reps]` - a single kind / shape for the --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boosted_detr_amd import kernels as k

what = sys.argv[1] if len(sys.argv) > 1 else "all"
which = sys.argv[2] if len(sys.argv) > 2 else "all"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50


def t(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
B, H, D = 16, 8, 256
with k.gemm_precision("split"):
    for name, nq, nk in (("enc", 400, 400), ("cross", 100, 400), ("self", 100, 100)):
        if which not in ("all", name):
            continue
        q, kk, v = (torch.randn(B, n, D, device="cuda", generator=g) for n in (nq, nk, nk))
        o, lse = k.attention_fwd(q, kk, v, H, 32 ** -0.5)
        do = torch.randn_like(o)
        line = f"{name} nq={nq} nk={nk}:"
        if what in ("all", "fwd"):
            line += f" fwd {t(lambda: k.attention_fwd(q, kk, v, H, 32 ** -0.5)):.1f} us"
        if what in ("all", "bwd"):
            line += f" bwd (dvec + dQ + dK/dV) {t(lambda: k.attention_bwd(q, kk, v, o, do, lse, H, 32 ** -0.5)):.1f} us"
        flops = 4.0 * B * H * nq * nk * 32
        print(line, f"| forward FLOPs {flops / 1e9:.2f} G (x3 split products)")
