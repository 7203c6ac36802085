"""Per-launch time of the row-chain kernels (csrc/rowchain.hip) at the transformer's shapes: M = 6400 (encoder, batch 16 x 400 tokens)
and M = 1600 (decoder, 16 x 100 queries), 1 and 3 stages, against the launches they replace (3 Dense + 2 add_dropout_layernorm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boosted_detr_amd import _lib, kernels as k

D = 256
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n = int(_lib.lib().bdetr_rowchain_pack_elems())
g = torch.Generator(device="cuda").manual_seed(0)
W = [torch.randn(D, D, device="cuda", generator=g) * D ** -0.5 for _ in range(3)]
fw = [torch.empty(n, device="cuda") for _ in range(3)]; bw = [torch.empty(n, device="cuda") for _ in range(3)]
table = torch.tensor([[W[i].data_ptr(), fw[i].data_ptr(), bw[i].data_ptr()] for i in range(3)], dtype=torch.int64).cuda()
k.rowchain_pack_weights(table)
print("pack (3 matrices): %.1f us" % t(lambda: k.rowchain_pack_weights(table)))
vec = lambda: torch.randn(D, device="cuda", generator=g) * 0.1
b = [vec() for _ in range(3)]; ln1 = (vec() + 1, vec()); ln2 = (vec() + 1, vec())
seedb = torch.full((1,), 77, dtype=torch.int64, device="cuda")
for M in (6400, 1600):
    ctx, res, dout = (torch.randn(M, D, device="cuda", generator=g) for _ in range(3))
    for ns in (3, 1):
        for rate in (0.0, 0.1):
            saved = k.rowchain_fwd(ctx, res, fw[:ns], b[:ns], ln1, ln2 if ns == 3 else None, 1e-3, rate, 11, 12, seedb)
            tf = t(lambda: k.rowchain_fwd(ctx, res, fw[:ns], b[:ns], ln1, ln2 if ns == 3 else None, 1e-3, rate, 11, 12, seedb))
            tb = t(lambda: k.rowchain_bwd(dout, saved, bw[:ns], (ln1[0], ln2[0])[:1 if ns == 1 else 2], rate, 11, 12, seedb))
            print(f"M={M} stages={ns} rate={rate}: fwd {tf:.1f} us  bwd {tb:.1f} us")
    with k.gemm_precision("split"):
        x = ctx
        def unfused():
            a = k.linear_fwd(x, W[0], b[0]); o, m, r = k.add_dropout_layernorm_fwd(res, a, ln1[0], ln1[1], 1e-3, 0.1, 11, seed_base=seedb)
            h = k.linear_fwd(o, W[1], b[1], 1); f = k.linear_fwd(h, W[2], b[2]); return k.add_dropout_layernorm_fwd(o, f, ln2[0], ln2[1], 1e-3, 0.1, 12, seed_base=seedb)
        print(f"M={M} unfused forward (3 Dense + 2 LN launches): {t(unfused):.1f} us")
