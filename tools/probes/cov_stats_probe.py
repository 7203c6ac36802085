"""Feasibility of DESIGN §9 item 0 (BatchNorm statistics of a 1x1 convolution's output from its INPUT's moments) - numerics only, on the CPU.

a = relu(standardised noise * gamma + beta) rows (the c2 activation's shape), W = He-initialised [f, 4f]; operands rounded to the bf16 hi/lo pair the GEMMs use
(16 mantissa bits), moment product S = a^T a accumulated in fp32 over the pixels in 128-row tiles (the GEMM's split-K partials), then summed in fp64
(what bn_finalize does with partials).  Compares mean / variance of y = aW per channel against the fp64 statistics of the fp32 y the product path forms.

    python tools/probes/cov_stats_probe.py
"""
import numpy as np


def pair16(x):
    """bf16 hi + bf16 lo of an fp32 array, returned as their fp32 sum (the value the split GEMM multiplies)"""
    def bf16(v):
        u = v.astype(np.float32).view(np.uint32)
        r = ((u >> 16) & 1) + 0x7FFF
        return ((u + r) & 0xFFFF0000).view(np.float32)
    hi = bf16(x)
    return hi + bf16(x - hi)


rng = np.random.default_rng(0)
print(f"{'M':>8s} {'f':>5s}  max rel err of var   max |mean err|/sigma   (direct-from-y in fp32 for scale)")
for M, f in ((409600, 64), (102400, 128), (25600, 256), (6400, 512)):
    z = rng.standard_normal((M, f), dtype=np.float32)
    g = rng.uniform(0.5, 1.5, f).astype(np.float32); b = rng.uniform(-0.5, 0.5, f).astype(np.float32)
    z = z + 0.3 * rng.standard_normal((M, 1), dtype=np.float32)          # correlated channels
    a = pair16(np.maximum(z * g + b, 0.0))
    W = pair16((rng.standard_normal((f, 4 * f)) * np.sqrt(2.0 / f)).astype(np.float32))
    y32 = a @ W                                                           # fp32 accumulate (MFMA fp32 accumulators)
    y64 = a.astype(np.float64) @ W.astype(np.float64)
    mu_ref = y64.mean(0); var_ref = y64.var(0)
    # today's path: per-tile fp32 partial sums of y and y^2, fp64 across tiles
    T = 128
    ps = y32.reshape(-1, T, 4 * f).sum(1, dtype=np.float32).astype(np.float64).sum(0)
    pq = (y32 * y32).reshape(-1, T, 4 * f).sum(1, dtype=np.float32).astype(np.float64).sum(0)
    mu_d = ps / M; var_d = pq / M - mu_d ** 2
    # moments path: S = sum over tiles of fp32 (a_t^T a_t), s = column sums; fp64 after the tiles
    S = np.zeros((f, f)); s = np.zeros(f)
    for t0 in range(0, M, 4096):                                          # 4096-row split-K partials in fp32
        at = a[t0:t0 + 4096]
        S += (at.T @ at).astype(np.float64); s += at.sum(0, dtype=np.float32).astype(np.float64)
    mean_a = s / M; cov = S / M - np.outer(mean_a, mean_a)
    W64 = W.astype(np.float64)
    mu_m = mean_a @ W64; var_m = np.einsum("ic,ij,jc->c", W64, cov, W64)
    # the same with the small algebra in fp32 (cov rounded to fp32 first)
    cov32 = cov.astype(np.float32); var_m32 = np.einsum("ic,ij,jc->c", W, cov32, W)
    sig = np.sqrt(var_ref)
    print(f"{M:8d} {f:5d}  moments {np.max(np.abs(var_m / var_ref - 1)):.2e} (fp32 algebra {np.max(np.abs(var_m32 / var_ref - 1)):.2e})   "
          f"{np.max(np.abs(mu_m - mu_ref) / sig):.2e}      direct {np.max(np.abs(var_d / var_ref - 1)):.2e} {np.max(np.abs(mu_d - mu_ref) / sig):.2e}")
