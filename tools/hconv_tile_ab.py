"""Tile-shape A/B of the halo-resident 3x3 kernel (csrc/hconv.hip) per ResNet-50 layer shape at batch 16: the tile quantisation
question of VERDICT r3 item 3 (25 * 2^k-pixel feature maps fill 78 % of 256 CUs with any power-of-two tile) answered with the tiles
that exist - would a different split of the same work fill the chip better?  One child process per forced tile (the switch is read once)."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAYERS = [(80, 128), (40, 256), (20, 512)]
if len(sys.argv) > 1:
    import torch
    from boosted_detr_amd import kernels as k
    out = {}
    with k.gemm_precision("split"):
        for H, C in LAYERS:
            g = k.ConvGeom(16, H, H, C, C, 3, 3, 1, 1)
            x, w, dy = torch.randn(16, H, H, C, device="cuda"), torch.randn(C, 3, 3, C, device="cuda") * (9 * C) ** -0.5, torch.randn(16, H, H, C, device="cuda")
            xf, _ = k.p16_pack(x, want_bf16=False); wf, wt = k.p16_pack_conv_weights(w); _, dyb = k.p16_pack(dy, want_f16=False)
            bias = torch.zeros(C, device="cuda")
            def t(fn):
                for _ in range(3): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 20 * 1e3
            gf = 2.0 * g.M * C * 9 * C / 1e9
            tf, tb = t(lambda: k.p16_conv2d_fwd(xf, wf, bias, g, 0, want_stats=True)), t(lambda: k.p16_conv2d_bwd_data(dyb, wt, g))
            out[f"{H}x{H}x{C}"] = {"fwd_us": round(tf, 1), "fwd_frac_of_roof": round(gf / tf / 1e3 / (2500 / 3) * 1e3 / 1e3, 3), "dgrad_us": round(tb, 1),
                                   "dgrad_frac_of_roof": round(gf / tb / 1e3 / (2500 / 3), 3), "tiles_256x128": -(-g.M // 256) * (C // 128), "tiles_128x128": -(-g.M // 128) * (C // 128)}
    print("HCONV_AB " + json.dumps(out))
else:
    res = {}
    for tile in ("0", "256128", "128128", "256064"):
        env = dict(os.environ, BDETR_HCONV_TILE=tile)
        if tile == "im2col":
            env["BDETR_HCONV"] = "0"
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("HCONV_AB ")]
        res["heuristic" if tile == "0" else tile] = json.loads(line[0][9:]) if line else r.stderr[-500:]
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, BDETR_HCONV="0"), capture_output=True, text=True)
    res["im2col (BDETR_HCONV=0)"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("HCONV_AB ")][0][9:])
    print(json.dumps(res, indent=1))
