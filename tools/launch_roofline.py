"""Per-launch roofline of the conv/GEMM family from a bdetr_prof_dump CSV (BDETR_PROF_DUMP=<file> python bench.py ...): every
distinct (shape, tile, kind) with its time per step, the time its own roof allows (the larger of algorithmic FLOPs / the 3-product
MFMA roof and algorithmic bytes / HBM peak) and the gap between them, sorted by the gap - where the family's time is lost."""
import csv
import sys
from collections import defaultdict

MFMA = 2500.0 / 3          # TFLOP/s, 3-product split
FP32 = 157.3
HBM = 8000.0               # GB/s


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for r in rows:
        I, J, R, z, bm, bn, kind, ms, gf = int(r["I"]), int(r["J"]), int(r["R"]), int(r["z"]), int(r["bm"]), int(r["bn"]), int(r["kind"]), float(r["ms"]), float(r["gflop"])
        if ms <= 0:
            continue
        arith, lk = kind // 10000, kind % 10000
        if arith >= 3:
            a_patch, b_patch = lk in (1000, 4000), lk == 3000
        else:
            a_patch, b_patch = lk // 1000 == 1, lk // 10 % 10 == 1
        dim = R if a_patch else J
        taps = (49 if dim % 49 == 0 else 9 if dim % 9 == 0 else 1) if (a_patch or b_patch) else 1
        mult = max(1, round(gf * 1e9 / (2.0 * I * J * R)))
        nbytes = 4.0 * mult * (I * R / (taps if a_patch else 1) + J * R / (taps if b_patch else 1) + I * J)
        t_mfma = gf / (FP32 if arith == 0 else MFMA)                # GFLOP / (TFLOP/s) = ms
        t_hbm = nbytes / (HBM * 1e6)                              # ms
        a = agg[(I, J, R, z, bm, bn, kind)]
        a[0] += 1; a[1] += ms; a[2] += max(t_mfma, t_hbm); a[3] += (1 if t_mfma >= t_hbm else 0)
    tot = sum(a[1] for a in agg.values()) / steps
    roof = sum(a[2] for a in agg.values()) / steps
    print(f"family: {tot:.2f} ms/step, own-roof time {roof:.2f} ms/step ({roof / tot:.2f})")
    print(f"{'I':>8} {'J':>6} {'R':>6} {'z':>3} {'tile':>8} {'kind':>6} {'n/step':>6} {'ms/step':>8} {'roof ms':>8} {'frac':>5} {'gap ms':>7} bound")
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
        I, J, R, z, bm, bn, kind = k
        print(f"{I:8d} {J:6d} {R:6d} {z:3d} {bm:4d}x{bn:<3d} {kind:6d} {a[0] / steps:6.1f} {a[1] / steps:8.3f} {a[2] / steps:8.3f} {a[2] / a[1]:5.2f} {(a[1] - a[2]) / steps:7.3f} {'mfma' if a[3] else 'hbm'}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
