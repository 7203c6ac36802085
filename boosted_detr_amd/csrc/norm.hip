// norm.hip - HBM-bound normalisation kernels for gfx950: BatchNorm (training + frozen),
// residual+dropout+LayerNorm, row softmax.  All fp32, 16-byte-per-lane coalesced accesses,
// 64-lane wave reductions, deterministic two-level column reductions (no atomics).
//
// Replaces: keras BatchNormalization (backbone.py:79-80,92-94; ResNet-50 internals;
// prediction_heads.py:42,108,177), keras LayerNormalization + Dropout + Add
// (transformers.py:135-137,178-180), softmax (transformers.py:89, prediction_heads.py:111).
#include "common.h"
#include "p16.h"

namespace {

// ------------------------------------------------------------------------------------
// column reductions over [rows][C] (C % 4 == 0): per-chunk partial sums of two quantities
// ------------------------------------------------------------------------------------
struct ColGeom { int tx, ty, gx; };
inline ColGeom col_geom(int C) {
    int c4 = C / 4;
    int tx = 1; while (tx < c4 && tx < 64) tx <<= 1;     // threads along columns (power of two <= 64)
    ColGeom g; g.tx = tx; g.ty = 256 / tx; g.gx = (c4 + tx - 1) / tx; return g;
}

// functor(row, c, &a4, &b4): the two float4 quantities to accumulate for columns c..c+3 of `row`
template <class F>
__global__ __launch_bounds__(256) void colreduce2_kernel(F f, int64_t rows, int C, int tx, int64_t rows_per_chunk,
                                                         float* __restrict__ pa, float* __restrict__ pb) {
    __shared__ f32x4 sa[256], sb[256];
    const int tid = threadIdx.x;
    const int cx = tid % tx, ry = tid / tx, ty = 256 / tx;
    const int c = (blockIdx.x * tx + cx) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rows_per_chunk);
    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (c < C) {
        // two rows per trip: both rows' loads are in flight before the first is consumed (same summation order)
        int64_t r = r0 + ry;
        for (; r + ty < r1; r += 2 * ty) {
            f32x4 qa0, qb0, qa1, qb1;
            f(r, c, qa0, qb0); f(r + ty, c, qa1, qb1);
            a += qa0; b += qb0; a += qa1; b += qb1;
        }
        if (r < r1) { f32x4 qa, qb; f(r, c, qa, qb); a += qa; b += qb; }
    }
    sa[tid] = a; sb[tid] = b;
    __syncthreads();
    for (int s = ty >> 1; s > 0; s >>= 1) {
        if (ry < s) { sa[tid] += sa[tid + s * tx]; sb[tid] += sb[tid + s * tx]; }
        __syncthreads();
    }
    if (ry == 0 && c < C) {
        *reinterpret_cast<f32x4*>(pa + (int64_t)blockIdx.y * C + c) = sa[tid];
        *reinterpret_cast<f32x4*>(pb + (int64_t)blockIdx.y * C + c) = sb[tid];
    }
}

// sum partials [nparts][C] in fp64, fixed order: 32 columns x 8 phases per block
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psq, int nparts,
                                                          int64_t rows, int C, float eps, float momentum, int bessel,
                                                          float* mean, float* rstd, float* mmean, float* mvar, int* guard) {
    __shared__ double s1[256], s2[256];
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double a = 0, b = 0;
    if (c < C) {
        int p = py;
        for (; p + 3 * 8 < nparts; p += 4 * 8) {                      // 4 rows' loads in flight per thread, fixed order
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { va[u] = psum[(int64_t)(p + 8 * u) * C + c]; vb[u] = psq[(int64_t)(p + 8 * u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += va[u]; b += vb[u]; }
        }
        for (; p < nparts; p += 8) { a += psum[(int64_t)p * C + c]; b += psq[(int64_t)p * C + c]; }
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    if (py == 0 && c < C) {
        for (int k = 1; k < 8; ++k) { a += s1[cx + 32 * k]; b += s2[cx + 32 * k]; }
        double m = a / (double)rows;
        double var = b / (double)rows - m * m;
        if (var < 0) var = 0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        // guard: a forward that left the f16 pair's range (flag set by an upstream producer) or produced non-finite
        // statistics must not poison the moving statistics - the host redoes such a step on the exact-fp32 forward
        bool ok = true;
        if (guard != nullptr) {
            if (!(fabs(m) <= 3.0e38) || !(var <= 3.0e38)) *guard = 1;      // catches NaN too
            ok = *guard == 0;
        }
        if (mmean != nullptr && ok) {
            double vm = (bessel && rows > 1) ? var * ((double)rows / (double)(rows - 1)) : var;
            mmean[c] = mmean[c] * momentum + (float)m * (1.f - momentum);
            mvar[c] = mvar[c] * momentum + (float)vm * (1.f - momentum);
        }
    }
}

// [nparts][C] -> [nout][C]: output row o sums input rows o, o+nout, o+2*nout, ... (fixed order)
__global__ __launch_bounds__(256) void fold_partials2_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nparts, int C, int nout,
                                                             float* __restrict__ oa, float* __restrict__ ob) {
    __shared__ float s1[256], s2[256];
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx, o = blockIdx.y;
    float a = 0.f, b = 0.f;
    if (c < C) {
        int p = o + py * nout;
        for (; p + 3 * 8 * nout < nparts; p += 4 * 8 * nout) {        // 4 rows' loads in flight per thread, fixed order
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { va[u] = pa[(int64_t)(p + 8 * nout * u) * C + c]; vb[u] = pb[(int64_t)(p + 8 * nout * u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += va[u]; b += vb[u]; }
        }
        for (; p < nparts; p += 8 * nout) { a += pa[(int64_t)p * C + c]; b += pb[(int64_t)p * C + c]; }
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    if (py == 0 && c < C) {
        for (int k = 1; k < 8; ++k) { a += s1[cx + 32 * k]; b += s2[cx + 32 * k]; }
        oa[(int64_t)o * C + c] = a; ob[(int64_t)o * C + c] = b;
    }
}

__global__ void bn_frozen_kernel(const float* mm, const float* mv, int C, float eps, float* mean, float* rstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = mm[c]; rstd[c] = 1.0f / sqrtf(mv[c] + eps); }
}

// sum partials [nparts][C] -> out1[C], out2[C] (fp32 inputs, fp64 accumulation, fixed order): 8 columns x 32 row
// phases per block, so a thread walks nparts / 32 rows (the 32 columns x 8 phases shape took 12.9 us on 512 partial
// rows: a 64-deep dependent load + add chain per thread)
__global__ __launch_bounds__(256) void sum_partials2_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nparts, int C,
                                                            float* oa, float* ob) {
    __shared__ double s1[256], s2[256];
    const int cx = threadIdx.x & 7, py = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    double a = 0, b = 0;
    if (c < C) {
        // 8 rows' loads in flight per thread (the rolled loop waited for each pair of loads before issuing the next:
        // 200 round trips on the 6,400 partial rows of a stage-2 layer); the summation order stays fixed
        int p = py;
        for (; p + 7 * 32 < nparts; p += 8 * 32) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { va[u] = pa[(int64_t)(p + 32 * u) * C + c]; vb[u] = pb[(int64_t)(p + 32 * u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
        }
        for (; p < nparts; p += 32) { a += pa[(int64_t)p * C + c]; b += pb[(int64_t)p * C + c]; }
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int s = 16; s > 0; s >>= 1) {          // fixed-order tree over the 32 phases
        if (py < s) { s1[threadIdx.x] += s1[threadIdx.x + 8 * s]; s2[threadIdx.x] += s2[threadIdx.x + 8 * s]; }
        __syncthreads();
    }
    if (py == 0 && c < C) { oa[c] = (float)s1[cx]; ob[c] = (float)s2[cx]; }
}

// Both reductions above in ONE launch for any number of partial rows (round 4: fold_partials2 + bn_finalize were two launches behind
// every convolution with more than 256 epilogue rows, sum_partials2 walked up to 200 rows per thread): a block owns FOUR columns - one
// 16-byte load per partial row and quantity - and its 256 threads are 256 row phases with four rows' loads in flight each, then a
// fixed-order fp64 tree over the phases.  FINALIZE: the totals become mean / rstd / moving statistics (bn_finalize_kernel's tail),
// else they are stored (oa <- sum pa, ob <- sum pb).  C % 4 == 0.
struct BnFinalizeArgs { int64_t rows; float eps, momentum; int bessel; float* mean; float* rstd; float* mmean; float* mvar; int* guard; };

template <bool FINALIZE>
__global__ __launch_bounds__(256) void reduce_partials2_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nparts, int C,
                                                               float* __restrict__ oa, float* __restrict__ ob, BnFinalizeArgs fin) {
    __shared__ double sh[256][8];
    const int t = threadIdx.x, c = blockIdx.x * 4;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    int p = t;
    for (; p + 3 * 256 < nparts; p += 4 * 256) {
        f32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = *reinterpret_cast<const f32x4*>(pa + (int64_t)(p + 256 * u) * C + c);
            vb[u] = *reinterpret_cast<const f32x4*>(pb + (int64_t)(p + 256 * u) * C + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += va[u][e]; b[e] += vb[u][e]; }
    }
    for (; p < nparts; p += 256) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(pa + (int64_t)p * C + c), vb = *reinterpret_cast<const f32x4*>(pb + (int64_t)p * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] += va[e]; b[e] += vb[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[t][e] = a[e]; sh[t][4 + e] = b[e]; }
    __syncthreads();
    // tree over the phases: 256 -> 8 rows with all threads busy (thread = (row, value)), then the last 8 rows serially
    for (int half = 128; half >= 8; half >>= 1) {
        const int r = t >> 3, v = t & 7;
        for (int rr = r; rr < half; rr += 32) sh[rr][v] += sh[rr + half][v];
        __syncthreads();
    }
    if (t < 4) {
        double ta = 0, tb = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) { ta += sh[r][t]; tb += sh[r][4 + t]; }
        if (!FINALIZE) { oa[c + t] = (float)ta; ob[c + t] = (float)tb; return; }
        const double m = ta / (double)fin.rows;
        double var = tb / (double)fin.rows - m * m;
        if (var < 0) var = 0;
        fin.mean[c + t] = (float)m;
        fin.rstd[c + t] = (float)(1.0 / sqrt(var + (double)fin.eps));
        bool ok = true;                          // (the guard: see bn_finalize_kernel)
        if (fin.guard != nullptr) {
            if (!(fabs(m) <= 3.0e38) || !(var <= 3.0e38)) *fin.guard = 1;
            ok = *fin.guard == 0;
        }
        if (fin.mmean != nullptr && ok) {
            const double vm = (fin.bessel && fin.rows > 1) ? var * ((double)fin.rows / (double)(fin.rows - 1)) : var;
            fin.mmean[c + t] = fin.mmean[c + t] * fin.momentum + (float)m * (1.f - fin.momentum);
            fin.mvar[c + t] = fin.mvar[c + t] * fin.momentum + (float)vm * (1.f - fin.momentum);
        }
    }
}

// fewest partial rows for which the one-launch reduction is used: BDETR_BN_WIDE_REDUCE=<rows>; unset or 0 = never, the default.
// Measured (round 4, configs[1], 120 graph-replayed steps, three alternating runs each, images/s):
//     never (fold_partials2 + bn_finalize / sum_partials2)   594.1  593.2  592.2
//     from 257 rows (every case that needed the fold)         583.5  588.0  586.0
//     from 1025 rows                                          587.9  588.1  586.1
//     always                                                  585.1  584.2  584.9
// One launch of C/4 workgroups that each walk every partial row is slower than two launches that spread the same walk over
// 64 x C/32 workgroups: inside a replayed graph a launch costs ~1-2 us, less than the latency of the longer dependent walk.  The
// three-kernel chain stays; this kernel is kept as the measured alternative.
int wide_reduce_min() {
    static const int n = [] { const char* e = getenv("BDETR_BN_WIDE_REDUCE"); int v = e ? atoi(e) : 0; return v <= 0 ? 0x7fffffff : v; }();
    return n;
}

// oa <- column sums of pa, ob <- of pb
void sum_partials2(const float* pa, const float* pb, int nparts, int C, float* oa, float* ob, hipStream_t st) {
    if (nparts >= wide_reduce_min() && C % 4 == 0 && (reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) % 16 == 0)
        hipLaunchKernelGGL(reduce_partials2_kernel<false>, dim3(C / 4), dim3(256), 0, st, pa, pb, nparts, C, oa, ob, BnFinalizeArgs{});
    else
        hipLaunchKernelGGL(sum_partials2_kernel, dim3((C + 7) / 8), dim3(256), 0, st, pa, pb, nparts, C, oa, ob);
}

struct StatFn {
    const float* x; int C;
    __device__ __forceinline__ void operator()(int64_t r, int c, f32x4& a, f32x4& b) const {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c);
        a = v; b = v * v;
    }
};

// the BN affine map, written once with an explicit fma so that the forward (bn_apply) and the
// backward's recomputed ReLU mask round identically
__device__ __forceinline__ float bn_affine(float v, float m, float rs, float g, float b) { return __builtin_fmaf(v - m, rs * g, b); }
// ... and BatchNorm's input gradient, (g - dbeta / rows - xhat * dgamma / rows) * rstd * gamma, with its fused multiply-adds spelled out:
// the fp32 kernel, the P16 kernel (which hoists the per-channel terms out of its loop) and the stem's fused backward must round
// identically whatever the compiler would contract in each of them (their tests compare with torch.equal)
__device__ __forceinline__ float bn_bwd_dx(float g, float xv, float m, float rs, float gm, float dg, float db, float inv_rows) {
    const float xh = (xv - m) * rs;
    const float t = __builtin_fmaf(-db, inv_rows, g);
    return __builtin_fmaf(-xh, dg * inv_rows, t) * (rs * gm);
}

// ReLU bit mask written by bn_apply_p16 (1 bit per element instead of re-reading a 4-byte-per-element tensor in both
// backward passes): element e of float4 index i lives in word (i >> 6) * 4 + e, bit i & 63
__device__ __forceinline__ unsigned relu_mask_bits4(const void* mask, int64_t i) {
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(mask) + (i >> 6) * 4;
    const int b = (int)(i & 63);
    return (unsigned)((w[0] >> b) & 1ull) | ((unsigned)((w[1] >> b) & 1ull) << 1) | ((unsigned)((w[2] >> b) & 1ull) << 2) | ((unsigned)((w[3] >> b) & 1ull) << 3);
}

struct BnBwdFn {   // a = g (masked dout), b = g * xhat
    const float* dout; const float* out; const float* x; const float* mean; const float* rstd; const float* gamma; const float* beta;
    int C; int relu;
    int out_p16 = 0;      // mask source `out`: 0 fp32 forward output, 1 its bf16 pair copy (P16 layout), 2 bn_apply_p16's ReLU bit mask
    // even_h > 0: dout is zero outside the pixels (2i, 2j) of an [N, even_h, even_w] grid (it came out of the backward-data of stride-2
    // 1x1 convolutions): the reduction then runs over the N * ceil(H/2) * ceil(W/2) rows that can be non-zero, `r` counts those
    int even_h = 0, even_w = 0;
    // dout_compact (with even_h > 0; round 5): dout is the COMPACT [N, ceil(H/2), ceil(W/2), C] tensor of those pixels alone - the stride-2
    // backward-data products wrote it densely, no zero-filled [N, H, W, C] tensor exists - and `r` IS its row index
    int dout_compact = 0;
    __device__ __forceinline__ void operator()(int64_t r, int c, f32x4& a, f32x4& b) const {
        const int64_t rg = r;
        if (even_h > 0) {
            const unsigned w2 = (unsigned)(even_w + 1) >> 1, h2 = (unsigned)(even_h + 1) >> 1;
            const unsigned q = (unsigned)r, j2 = q % w2, t = q / w2, i2 = t % h2, n = t / h2;
            r = ((int64_t)n * even_h + 2 * i2) * even_w + 2 * j2;
        }
        f32x4 g = *reinterpret_cast<const f32x4*>(dout + (dout_compact ? rg : r) * C + c);
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * C + c);
        f32x4 m = *reinterpret_cast<const f32x4*>(mean + c);
        f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + c);
        if (relu) {
            if (out != nullptr && out_p16) {
                const unsigned pm = out_p16 == 2 ? relu_mask_bits4(out, (r * C + c) >> 2) : p16_positive4_bf16(out, (r * C + c) >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!((pm >> e) & 1u)) g[e] = 0.f;
            } else if (out != nullptr) {
                f32x4 o = *reinterpret_cast<const f32x4*>(out + r * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(o[e] > 0.f)) g[e] = 0.f;
            } else {     // no residual: the mask is a function of x alone - recompute it, do not re-read `out`
                f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(bn_affine(xv[e], m[e], rs[e], gm[e], bt[e]) > 0.f)) g[e] = 0.f;
            }
        }
        a = g; b = g * ((xv - m) * rs);
    }
};

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, int relu, float* __restrict__ out,
                                                       int64_t n4, int c4n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 m = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bn_affine(v[e], m[e], rs[e], g[e], b[e]);
        if (residual != nullptr) o += reinterpret_cast<const f32x4*>(residual)[i];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu, int frozen,
                                                           float* __restrict__ dx, float* __restrict__ dres, int64_t n4, int c4n, float inv_rows) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        f32x4 g = reinterpret_cast<const f32x4*>(dout)[i];
        f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + c), gm = *reinterpret_cast<const f32x4*>(gamma + c);
        f32x4 m = *reinterpret_cast<const f32x4*>(mean + c);
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        const bool recompute = relu && out == nullptr;
        if (!frozen || recompute) xv = reinterpret_cast<const f32x4*>(x)[i];
        if (relu) {
            if (!recompute) {
                f32x4 o = reinterpret_cast<const f32x4*>(out)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(o[e] > 0.f)) g[e] = 0.f;
            } else {
                f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(bn_affine(xv[e], m[e], rs[e], gm[e], bt[e]) > 0.f)) g[e] = 0.f;
            }
        }
        if (dres != nullptr) reinterpret_cast<f32x4*>(dres)[i] = g;
        f32x4 r;
        if (frozen) {
            r = g * (rs * gm);
        } else {
            f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + c), db = *reinterpret_cast<const f32x4*>(dbeta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = bn_bwd_dx(g[e], xv[e], m[e], rs[e], gm[e], dg[e], db[e], inv_rows);
        }
        reinterpret_cast<f32x4*>(dx)[i] = r;
    }
}

// ---- the ResNet stem's tail: BatchNorm -> ReLU -> ZeroPadding2D(1) -> MaxPool 3x3 / 2 in one pass each way -------------------------
// (keras ResNet50 conv1_bn / conv1_relu / pool1_pad / pool1_pool, reached from the reference's backbone.py:79-80.)  The 64-channel
// 320x320 tensor between conv1 and the pool is the largest activation of the network (420 MB at 32 images); unfused it is written by
// bn_apply, read by the pool, and in the backward pass written by the pool's backward and read twice by BatchNorm's - here it exists only
// as the RAW convolution output y: the forward pass normalises inside the pooling window and writes the pooled tensor as the f16 pair
// its two consumers read plus one byte per element naming the window tap that held the maximum; the backward pass rebuilds the
// gradient of the normalised tensor from those bytes on the fly, in BatchNorm's reduction pass and again in its apply pass.
// A tie goes to the first tap in (row, column) order.  Ties between positive values need bit-equal fp32 numbers; ties at zero get
// no gradient either way (the ReLU mask).
struct StemGeom { int N, H, W, C, PH, PW; };

__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, StemGeom s,
                                                            float* __restrict__ out32, void* __restrict__ out_f16, unsigned* __restrict__ tap,
                                                            int* __restrict__ overflow_flag) {
    const int c4n = s.C / 4;
    const int64_t n4 = (int64_t)s.N * s.PH * s.PW * c4n;      // even, and the stride is even: the two lanes of an f16 pair group run together
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n); unsigned t = (unsigned)(i / c4n);
        const int pw = (int)(t % (unsigned)s.PW); t /= (unsigned)s.PW; const int ph = (int)(t % (unsigned)s.PH); const int n = (int)(t / (unsigned)s.PH);
        const int c = c4 * 4;
        const f32x4 m = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 v[9];
        bool in[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {                        // all nine loads in flight before the first compare
            const int ih = ph * 2 - 1 + k / 3, iw = pw * 2 - 1 + k % 3;
            in[k] = (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (in[k]) v[k] = reinterpret_cast<const f32x4*>(y)[(((int64_t)n * s.H + ih) * s.W + iw) * c4n + c4];
        }
        // (the padding ring holds zeros and every window has a real tap, whose value is >= 0 after the ReLU: skipping the ring's
        // taps gives the same maximum as comparing against their zeros)
        f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned arg = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (!in[k]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = fmaxf(bn_affine(v[k][e], m[e], rs[e], g[e], b[e]), 0.f);
                if (a > mx[e]) { mx[e] = a; arg = (arg & ~(0xFFu << (8 * e))) | ((unsigned)k << (8 * e)); }
            }
        }
        tap[i] = arg;
        if (out32 != nullptr) reinterpret_cast<f32x4*>(out32)[i] = mx;
        if (out_f16 != nullptr) {
            p16_store4<true>(out_f16, i, mx[0], mx[1], mx[2], mx[3]);
            if (overflow_flag != nullptr) {
                const float top = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
                if (!(top < P16_F16_LIMIT)) *overflow_flag = 1;       // (-inf survives only where every tap was NaN: caught here too)
            }
        }
    }
}

// gradient of the NORMALISED, rectified tensor at input pixel `px` (flat n*H*W + ih*W + iw), channels c..c+3: the pooled gradient of
// every window whose recorded tap is this pixel, masked by the ReLU (recomputed from y with the forward's bn_affine).  Also returns xhat.
__device__ __forceinline__ f32x4 stem_pixel_grad(const float* __restrict__ dpool, const unsigned* __restrict__ tap, const StemGeom& s, unsigned px, int c,
                                                 const f32x4 yv, const f32x4 m, const f32x4 rs, const f32x4 gm, const f32x4 bt, f32x4& xhat) {
    const int c4n = s.C / 4, c4 = c / 4;
    const int iw = (int)(px % (unsigned)s.W); const unsigned t = px / (unsigned)s.W;
    const int ih = (int)(t % (unsigned)s.H), n = (int)(t / (unsigned)s.H);
    // windows ph with 2 ph - 1 <= ih <= 2 ph + 1: ih / 2 and, for odd ih, (ih + 1) / 2 - always FOUR candidate windows, all eight
    // loads issued before the first compare (a candidate that does not exist re-reads a real window and is given a tap id no byte holds;
    // the data-dependent 1..2 x 1..2 loop this replaces left one load in flight per lane: 454 us for the reduction pass, round 4)
    const int ph0 = ih >> 1, pw0 = iw >> 1;
    const bool two_h = (ih & 1) & (int)(ph0 + 1 < s.PH), two_w = (iw & 1) & (int)(pw0 + 1 < s.PW);
    unsigned word[4];
    f32x4 dv[4];
    unsigned mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = k >> 1, b = k & 1;
        const bool real = (a == 0 || two_h) && (b == 0 || two_w);
        const int ph = ph0 + (a & (int)two_h), pw = pw0 + (b & (int)two_w);
        const int64_t o = (((int64_t)n * s.PH + ph) * s.PW + pw) * c4n + c4;
        mine[k] = real ? (unsigned)((ih - (ph * 2 - 1)) * 3 + (iw - (pw * 2 - 1))) : 0xFFu;
        word[k] = tap[o];
        dv[k] = reinterpret_cast<const f32x4*>(dpool)[o];
    }
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)           // same order as the loop it replaces: (ph0, pw0), (ph0, pw0 + 1), (ph0 + 1, pw0), (ph0 + 1, pw0 + 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) if (((word[k] >> (8 * e)) & 0xFFu) == mine[k]) g[e] += dv[k][e];
    xhat = (yv - m) * rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (!(bn_affine(yv[e], m[e], rs[e], gm[e], bt[e]) > 0.f)) g[e] = 0.f;
    return g;
}

struct StemBwdFn {   // a = g, b = g * xhat  (colreduce2_kernel functor; `r` is the flat input pixel)
    const float* dpool; const unsigned* tap; const float* y; const float* mean; const float* rstd; const float* gamma; const float* beta; StemGeom s;
    __device__ __forceinline__ void operator()(int64_t r, int c, f32x4& a, f32x4& b) const {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * s.C + c);
        const f32x4 m = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 xh;
        a = stem_pixel_grad(dpool, tap, s, (unsigned)r, c, yv, m, rs, gm, bt, xh);
        b = a * xh;
    }
};

__global__ __launch_bounds__(256) void stem_bwd_apply_kernel(const float* __restrict__ dpool, const unsigned* __restrict__ tap, const float* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                             StemGeom s, float* __restrict__ dy, int dy_p16, int64_t n4, float inv_rows) {
    // dy_p16 (round 5): dy is written as the bf16 pair the stem's pre-split weight gradient reads (bdetr_p16_stem_bwd_weight) - the same
    // 4 bytes per element, no fp32 copy (n4 and the stride are even: the two lanes of a pair share every trip, p16_store4)
    const int c4n = s.C / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto one = [&](int64_t i, int c, const f32x4 m, const f32x4 rs, const f32x4 gm, const f32x4 bt, const f32x4 dg, const f32x4 db) {
        const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 xh;
        const f32x4 g = stem_pixel_grad(dpool, tap, s, (unsigned)(i / c4n), c, yv, m, rs, gm, bt, xh);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = bn_bwd_dx(g[e], yv[e], m[e], rs[e], gm[e], dg[e], db[e], inv_rows);      // bn_bwd_apply_kernel's expression
        if (dy_p16) p16_store4<false>(dy, i, r[0], r[1], r[2], r[3]);
        else reinterpret_cast<f32x4*>(dy)[i] = r;
    };
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (stride % c4n == 0) {                                   // (uniform; always for the stem's 64 channels: a thread keeps its channel group)
        const int c = (int)(i % c4n) * 4;
        const f32x4 m = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
        const f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + c), db = *reinterpret_cast<const f32x4*>(dbeta + c);
        for (; i < n4; i += stride) one(i, c, m, rs, gm, bt, dg, db);
        return;
    }
    for (; i < n4; i += stride) {
        const int c = (int)(i % c4n) * 4;
        one(i, c, *reinterpret_cast<const f32x4*>(mean + c), *reinterpret_cast<const f32x4*>(rstd + c), *reinterpret_cast<const f32x4*>(gamma + c),
            *reinterpret_cast<const f32x4*>(beta + c), *reinterpret_cast<const f32x4*>(dgamma + c), *reinterpret_cast<const f32x4*>(dbeta + c));
    }
}

// ---- P16 producers (sgemm.hip operand layout): the same maps as bn_apply / bn_bwd_apply, 8 channels per thread,
// writing the f16 pair (forward operand of the consumer conv), the bf16 pair (its weight-gradient operand) and / or
// the fp32 tensor.  The fp32 arithmetic is identical to the kernels above (bn_affine), so a ReLU mask recomputed in
// the backward pass agrees with what the forward wrote.
__global__ __launch_bounds__(256) void bn_apply_p16_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const void* __restrict__ residual, int residual_p16, bdetr_bn_affine rbn, int relu, float* __restrict__ out32,
                                                           void* __restrict__ out_f16, void* __restrict__ out_bf16, unsigned long long* __restrict__ relu_mask,
                                                           int* __restrict__ overflow_flag, int64_t n4, int c4n) {
    // n4 is even and the stride is even: the two lanes of a pair (one 8-element group) always run together
    struct Affine { f32x4 m, rs, g, b; };
    auto affine_of = [&](const float* pm, const float* prs, const float* pg, const float* pb, int c) {
        return Affine{*reinterpret_cast<const f32x4*>(pm + c), *reinterpret_cast<const f32x4*>(prs + c), *reinterpret_cast<const f32x4*>(pg + c), *reinterpret_cast<const f32x4*>(pb + c)};
    };
    // the residual's four values, decoded (f16 pair) or raw (fp32 / the shortcut's un-normalised convolution output)
    auto load_res = [&](int64_t i) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (residual != nullptr) {
            if (residual_p16 == 1) { float r4[4]; p16_load4_f16(residual, i, r4); r = f32x4{r4[0], r4[1], r4[2], r4[3]}; }
            else r = reinterpret_cast<const f32x4*>(residual)[i];
        }
        return r;
    };
    auto body = [&](int64_t i, const f32x4 v, const f32x4 rv, const Affine& p, const Affine& p2) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bn_affine(v[e], p.m[e], p.rs[e], p.g[e], p.b[e]);
        if (residual != nullptr) {
            if (residual_p16 == 2) {
                // the projection shortcut's BatchNorm, applied here: same bn_affine, same fp32 add as the two-pass form
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += bn_affine(rv[e], p2.m[e], p2.rs[e], p2.g[e], p2.b[e]);
            } else {
                o += rv;
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        if (relu_mask != nullptr) {
            // a wave covers 64 consecutive float4 indices (256-thread blocks, strides that are multiples of 256)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long m = __ballot(o[e] > 0.f);
                if ((i & 63) == 0) relu_mask[(i >> 6) * 4 + e] = m;
            }
        }
        if (out32 != nullptr) reinterpret_cast<f32x4*>(out32)[i] = o;
        if (out_f16 != nullptr) {
            p16_store4<true>(out_f16, i, o[0], o[1], o[2], o[3]);
            if (overflow_flag != nullptr) {
                const float mx = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
                if (!(mx < P16_F16_LIMIT) || o[0] != o[0] || o[1] != o[1] || o[2] != o[2] || o[3] != o[3]) *overflow_flag = 1;
            }
        }
        if (out_bf16 != nullptr) p16_store4<false>(out_bf16, i, o[0], o[1], o[2], o[3]);
    };
    // Two row groups per trip with EVERY load of the trip issued first: both x groups and both residual groups.  The per-channel
    // parameters are loaded once per thread when the grid stride is a multiple of the channel count (the launch makes it one: a thread's
    // channel then never changes).  Round 4: the loop used to reload the parameters per group and wait for each residual load on the spot
    // - five dependent round trips per trip (ISA) - and ran at 4.2-4.7 TB/s on the residual forms against 5.3 for a plain copy.
    // The trip count is decided per aligned 64-index group = per wave, so that the __ballot of a group always sees all of its lanes together.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool fixed_c = stride % c4n == 0;                    // (uniform)
    const bool rb = residual != nullptr && residual_p16 == 2;
    if (fixed_c) {
        const int c = (int)(i % c4n) * 4;
        const Affine p = affine_of(mean, rstd, gamma, beta, c);
        const Affine p2 = rb ? affine_of(rbn.mean, rbn.rstd, rbn.gamma, rbn.beta, c) : p;
        for (; (i | 63) + stride < n4; i += 2 * stride) {
            const f32x4 v0 = reinterpret_cast<const f32x4*>(x)[i], v1 = reinterpret_cast<const f32x4*>(x)[i + stride];
            const f32x4 r0 = load_res(i), r1 = load_res(i + stride);
            body(i, v0, r0, p, p2);
            body(i + stride, v1, r1, p, p2);
        }
        for (; i < n4; i += stride) { const f32x4 v0 = reinterpret_cast<const f32x4*>(x)[i]; const f32x4 r0 = load_res(i); body(i, v0, r0, p, p2); }
        return;
    }
    for (; i < n4; i += stride) {
        const int c = (int)(i % c4n) * 4;
        const Affine p = affine_of(mean, rstd, gamma, beta, c);
        const Affine p2 = rb ? affine_of(rbn.mean, rbn.rstd, rbn.gamma, rbn.beta, c) : p;
        const f32x4 v0 = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 r0 = load_res(i);
        body(i, v0, r0, p, p2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_p16_kernel(const float* __restrict__ dout, const void* __restrict__ out, int out_p16, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu, int frozen,
                                                               float* __restrict__ dx32, void* __restrict__ dx_bf16, float* __restrict__ dres,
                                                               int64_t n4, int c4n, float inv_rows, int cH, int cW) {
    // cH > 0 (round 5): dout is the compact [N, cH/2, cW/2, C] gradient of the even pixels of an [N, cH, cW] map (BnBwdFn::dout_compact);
    // every other pixel's dout is zero - its dx is not (the mean terms), so the pass still visits every element
    // Same shape as bn_apply_p16_kernel: per-channel terms once per thread when the grid stride is a multiple of the channel count, two
    // row groups per trip with all their loads issued first.  The arithmetic is the unhoisted kernel's, operation for operation.
    const bool recompute = relu && out == nullptr;
    const bool need_x = !frozen || recompute;
    struct Chan { f32x4 m, rs, gm, bt, dg, db; };
    auto chan_of = [&](int c) {
        Chan p;
        p.rs = *reinterpret_cast<const f32x4*>(rstd + c); p.gm = *reinterpret_cast<const f32x4*>(gamma + c); p.m = *reinterpret_cast<const f32x4*>(mean + c);
        p.bt = recompute ? *reinterpret_cast<const f32x4*>(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        p.dg = f32x4{0.f, 0.f, 0.f, 0.f}; p.db = p.dg;
        if (!frozen) { p.dg = *reinterpret_cast<const f32x4*>(dgamma + c); p.db = *reinterpret_cast<const f32x4*>(dbeta + c); }
        return p;
    };
    struct In { f32x4 g, xv, o; unsigned pm; };
    auto load_in = [&](int64_t i) {
        In q;
        if (cH > 0) {
            const unsigned iu = (unsigned)i, row = iu / (unsigned)c4n, cc = iu - row * (unsigned)c4n;      // (host: n4 < 2^32 in this mode)
            const unsigned w = row % (unsigned)cW, t = row / (unsigned)cW, h = t % (unsigned)cH, n = t / (unsigned)cH;
            q.g = f32x4{0.f, 0.f, 0.f, 0.f};
            if (((h | w) & 1u) == 0u) q.g = reinterpret_cast<const f32x4*>(dout)[((int64_t)(n * (unsigned)(cH >> 1) + (h >> 1)) * (cW >> 1) + (w >> 1)) * c4n + cc];
        } else {
            q.g = reinterpret_cast<const f32x4*>(dout)[i];
        }
        q.xv = need_x ? reinterpret_cast<const f32x4*>(x)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        q.o = f32x4{0.f, 0.f, 0.f, 0.f}; q.pm = 0u;
        if (relu && !recompute) {
            if (out_p16) q.pm = out_p16 == 2 ? relu_mask_bits4(out, i) : p16_positive4_bf16(out, i);
            else q.o = reinterpret_cast<const f32x4*>(out)[i];
        }
        return q;
    };
    auto body = [&](int64_t i, const In& q, const Chan& p) {
        f32x4 g = q.g;
        if (relu) {
            if (!recompute && out_p16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!((q.pm >> e) & 1u)) g[e] = 0.f;
            } else if (!recompute) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(q.o[e] > 0.f)) g[e] = 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(bn_affine(q.xv[e], p.m[e], p.rs[e], p.gm[e], p.bt[e]) > 0.f)) g[e] = 0.f;
            }
        }
        if (dres != nullptr) reinterpret_cast<f32x4*>(dres)[i] = g;
        f32x4 r;
        if (frozen) {
            r = g * (p.rs * p.gm);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = bn_bwd_dx(g[e], q.xv[e], p.m[e], p.rs[e], p.gm[e], p.dg[e], p.db[e], inv_rows);
        }
        if (dx32 != nullptr) reinterpret_cast<f32x4*>(dx32)[i] = r;
        p16_store4<false>(dx_bf16, i, r[0], r[1], r[2], r[3]);
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (stride % c4n == 0) {                                   // (uniform)
        const Chan p = chan_of((int)(i % c4n) * 4);
        for (; i + stride < n4; i += 2 * stride) {            // (n4 and the stride are even: the two lanes of a pair share every trip)
            const In q0 = load_in(i), q1 = load_in(i + stride);
            body(i, q0, p);
            body(i + stride, q1, p);
        }
        if (i < n4) body(i, load_in(i), p);
        return;
    }
    for (; i < n4; i += stride) body(i, load_in(i), chan_of((int)(i % c4n) * 4));
}

// ------------------------------------------------------------------------------------
// LayerNorm: one wave per row, D % 4 == 0, D <= 64*4*LN_MAXV
// ------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;   // float4 per lane -> D <= 1024

__device__ __forceinline__ uint32_t hash_u64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
    return hash_u64(seed ^ (idx * 0xD6E8FEB86659FD93ull)) >= thresh ? inv_keep : 0.f;
}

__global__ __launch_bounds__(256) void add_drop_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ out, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                              int64_t rows, int D, float eps, float rate, uint64_t seed,
                                                              const uint64_t* __restrict__ seed_base) {
    if (seed_base != nullptr) seed ^= *seed_base * 0x100000001B3ull;       // per-step seed kept in HBM (graph replays)
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int d4 = D / 4;
    const uint32_t thresh = rate > 0.f ? (uint32_t)fminf(rate * 4294967296.0f, 4294967295.0f) : 0u;
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    f32x4 h[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c4 = lane + 64 * k;
        h[k] = f32x4{0, 0, 0, 0};
        if (c4 < d4) {
            f32x4 xv = reinterpret_cast<const f32x4*>(x + row * D)[c4];
            f32x4 yv = reinterpret_cast<const f32x4*>(y + row * D)[c4];
            if (rate > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[e] *= keep_scale(seed, (uint64_t)(row * D + c4 * 4 + e), thresh, inv_keep);
            }
            h[k] = xv + yv;
            s += h[k][0] + h[k][1] + h[k][2] + h[k][3];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        if (lane + 64 * k < d4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float t = h[k][e] - mean; q += t * t; }
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c4 = lane + 64 * k;
        if (c4 < d4) {
            f32x4 g = reinterpret_cast<const f32x4*>(gamma)[c4], b = reinterpret_cast<const f32x4*>(beta)[c4];
            reinterpret_cast<f32x4*>(out + row * D)[c4] = (h[k] - mean) * rstd * g + b;
        }
    }
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// one block = one chunk of rows; 4 waves, each row handled by one wave; column partials of
// dgamma/dbeta kept in registers, combined through LDS at the end.
__global__ __launch_bounds__(256) void add_drop_ln_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                              float* __restrict__ dx, float* __restrict__ dy, float* __restrict__ pdg, float* __restrict__ pdb,
                                                              int64_t rows, int D, int64_t rows_per_chunk, float rate, uint64_t seed, int accumulate_dx,
                                                              const uint64_t* __restrict__ seed_base) {
    if (seed_base != nullptr) seed ^= *seed_base * 0x100000001B3ull;
    __shared__ f32x4 sg[4][64 * LN_MAXV], sb[4][64 * LN_MAXV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d4 = D / 4;
    const uint32_t thresh = rate > 0.f ? (uint32_t)fminf(rate * 4294967296.0f, 4294967295.0f) : 0u;
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    f32x4 adg[LN_MAXV], adb[LN_MAXV], gm[LN_MAXV];
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        adg[k] = f32x4{0, 0, 0, 0}; adb[k] = f32x4{0, 0, 0, 0}; gm[k] = f32x4{0, 0, 0, 0};
        if (lane + 64 * k < d4) gm[k] = reinterpret_cast<const f32x4*>(gamma)[lane + 64 * k];
    }
    for (int64_t row = r0 + wave; row < r1; row += 4) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        f32x4 xh[LN_MAXV], g[LN_MAXV], keep[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c4 = lane + 64 * k;
            xh[k] = f32x4{0, 0, 0, 0}; g[k] = f32x4{0, 0, 0, 0}; keep[k] = f32x4{1, 1, 1, 1};
            if (c4 < d4) {
                f32x4 xv = reinterpret_cast<const f32x4*>(x + row * D)[c4];
                f32x4 yv = reinterpret_cast<const f32x4*>(y + row * D)[c4];
                if (rate > 0.f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) keep[k][e] = keep_scale(seed, (uint64_t)(row * D + c4 * 4 + e), thresh, inv_keep);
                    yv *= keep[k];
                }
                f32x4 dov = reinterpret_cast<const f32x4*>(dout + row * D)[c4];
                xh[k] = (xv + yv - mean) * rstd;
                g[k] = dov * gm[k];
                adg[k] += dov * xh[k]; adb[k] += dov;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += g[k][e]; s2 += g[k][e] * xh[k][e]; }
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c4 = lane + 64 * k;
            if (c4 < d4) {
                f32x4 dh = (g[k] - m1 - xh[k] * m2) * rstd;
                f32x4* dxp = reinterpret_cast<f32x4*>(dx + row * D) + c4;
                if (accumulate_dx) *dxp += dh; else *dxp = dh;
                reinterpret_cast<f32x4*>(dy + row * D)[c4] = dh * keep[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) { sg[wave][lane + 64 * k] = adg[k]; sb[wave][lane + 64 * k] = adb[k]; }
    __syncthreads();
    for (int c4 = threadIdx.x; c4 < d4; c4 += 256) {
        f32x4 a = sg[0][c4] + sg[1][c4] + sg[2][c4] + sg[3][c4];
        f32x4 b = sb[0][c4] + sb[1][c4] + sb[2][c4] + sb[3][c4];
        reinterpret_cast<f32x4*>(pdg + (int64_t)blockIdx.x * D)[c4] = a;
        reinterpret_cast<f32x4*>(pdb + (int64_t)blockIdx.x * D)[c4] = b;
    }
}

// ------------------------------------------------------------------------------------
// row softmax: one wave per row, up to 64*NPER columns held in registers
// ------------------------------------------------------------------------------------
template <int NPER>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, float* __restrict__ p, int64_t rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NPER];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NPER; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < cols ? s[row * cols + c] * scale : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NPER; ++k) { v[k] = lane + 64 * k < cols ? expf(v[k] - mx) : 0.f; sum += v[k]; }
    sum = wave_sum(sum);
#pragma unroll
    for (int k = 0; k < NPER; ++k) { const int c = lane + 64 * k; if (c < cols) p[row * cols + c] = v[k] / sum; }
}

template <int NPER>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ ds,
                                                          int64_t rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float pv[NPER], dv[NPER];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NPER; ++k) {
        const int c = lane + 64 * k;
        pv[k] = c < cols ? p[row * cols + c] : 0.f;
        dv[k] = c < cols ? dp[row * cols + c] : 0.f;
        dot += pv[k] * dv[k];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < NPER; ++k) { const int c = lane + 64 * k; if (c < cols) ds[row * cols + c] = scale * pv[k] * (dv[k] - dot); }
}

// row chunks so that (column blocks x chunks) ~ 2048 workgroups: wide tensors need few chunks, which keeps the
// fixed-order fp64 fold of the partials short
int64_t chunks_for_width(int gx) { int64_t c = 2048 / (gx > 0 ? gx : 1); return c < 32 ? 32 : (c > 512 ? 512 : c); }

int64_t chunk_rows_for(int64_t rows, int64_t maxchunks, int64_t minrows) {
    int64_t rpc = cdiv64(rows, maxchunks);
    if (rpc < minrows) rpc = minrows;
    return rpc;
}

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
extern "C" int bdetr_bn_bwd_chunks(int64_t rows) { return (int)cdiv64(rows, chunk_rows_for(rows, 512, 64)); }
extern "C" int bdetr_ln_bwd_chunks(int64_t rows) { return (int)cdiv64(rows, chunk_rows_for(rows, 512, 16)); }

constexpr int BN_FOLD = 64;
extern "C" int bdetr_bn_stats_fold_rows(void) { return BN_FOLD; }

extern "C" int bdetr_bn_stats(const float* x, int64_t rows, int C, const float* part_sum, const float* part_sq,
                              int nparts, float eps, float momentum, int bessel,
                              float* mean, float* rstd, float* moving_mean, float* moving_var, float* fold_ws, int* guard_flag, void* stream) {
    BDETR_CHECK_ARG(mean && rstd && rows > 0 && C > 0, "bdetr_bn_stats: bad arguments");
    BDETR_CHECK_ARG(part_sum != nullptr && part_sq != nullptr && nparts > 0,
                    "bdetr_bn_stats: partial sums required (use bdetr_colstats to produce them from x)");
    (void)x;
    hipStream_t st = (hipStream_t)stream;
    if (nparts >= wide_reduce_min() && C % 4 == 0 && (reinterpret_cast<uintptr_t>(part_sum) | reinterpret_cast<uintptr_t>(part_sq)) % 16 == 0) {
        (void)fold_ws;
        hipLaunchKernelGGL(reduce_partials2_kernel<true>, dim3(C / 4), dim3(256), 0, st, part_sum, part_sq, nparts, C, (float*)nullptr, (float*)nullptr,
                           BnFinalizeArgs{rows, eps, momentum, bessel, mean, rstd, moving_mean, moving_var, guard_flag});
        return bdetr_launch_status("bn_finalize");
    }
    if (nparts > 4 * BN_FOLD && fold_ws != nullptr) {
        // many epilogue partials (one per 32/64 output rows): fold them to BN_FOLD rows with a wide grid first
        float* fa = fold_ws; float* fb = fold_ws + (int64_t)BN_FOLD * C;
        hipLaunchKernelGGL(fold_partials2_kernel, dim3((C + 31) / 32, BN_FOLD), dim3(256), 0, st, part_sum, part_sq, nparts, C, BN_FOLD, fa, fb);
        part_sum = fa; part_sq = fb; nparts = BN_FOLD;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, part_sum, part_sq, nparts, rows, C, eps, momentum, bessel,
                       mean, rstd, moving_mean, moving_var, guard_flag);
    return bdetr_launch_status("bn_finalize");
}

// partial column sums of x and x*x: part_* are [bdetr_bn_bwd_chunks(rows)][C]
extern "C" int bdetr_colstats(const float* x, int64_t rows, int C, float* part_sum, float* part_sq, void* stream) {
    BDETR_CHECK_ARG(x && part_sum && part_sq && rows > 0 && C > 0 && C % 4 == 0, "bdetr_colstats: bad arguments (C %% 4 == 0 required)");
    ColGeom g = col_geom(C);
    int64_t rpc = chunk_rows_for(rows, 512, 64);     // must match bdetr_bn_bwd_chunks(rows): the caller sizes part_* with it
    int nch = (int)cdiv64(rows, rpc);
    StatFn f{x, C};
    hipLaunchKernelGGL((colreduce2_kernel<StatFn>), dim3(g.gx, nch), dim3(256), 0, (hipStream_t)stream, f, rows, C, g.tx, rpc, part_sum, part_sq);
    return bdetr_launch_status("colstats");
}

extern "C" int bdetr_bn_stats_frozen(const float* moving_mean, const float* moving_var, int C, float eps,
                                     float* mean, float* rstd, void* stream) {
    BDETR_CHECK_ARG(moving_mean && moving_var && mean && rstd && C > 0, "bdetr_bn_stats_frozen: bad arguments");
    hipLaunchKernelGGL(bn_frozen_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, moving_mean, moving_var, C, eps, mean, rstd);
    return bdetr_launch_status("bn_frozen");
}

extern "C" int bdetr_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, const float* residual, int relu, float* out,
                              int64_t rows, int C, void* stream) {
    BDETR_CHECK_ARG(x && mean && rstd && gamma && beta && out && rows > 0 && C > 0 && C % 4 == 0, "bdetr_bn_apply: bad arguments (C %% 4 == 0 required)");
    int64_t n4 = rows * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4, 256, 2)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, residual, relu, out, n4, C / 4);
    return bdetr_launch_status("bn_apply");
}

extern "C" int bdetr_bn_bwd(const float* dout, const float* out, const float* x, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                            float* dx, float* dgamma, float* dbeta, float* dresidual,
                            float* ws, int64_t rows, int C, void* stream) {
    BDETR_CHECK_ARG(dout && x && mean && rstd && gamma && dx && dgamma && dbeta && ws && rows > 0 && C > 0 && C % 4 == 0,
                    "bdetr_bn_bwd: bad arguments (C %% 4 == 0 required)");
    BDETR_CHECK_ARG(!relu || out || beta, "bdetr_bn_bwd: relu needs the forward output, or beta to recompute the mask from x");
    hipStream_t st = (hipStream_t)stream;
    ColGeom g = col_geom(C);
    int64_t rpc = chunk_rows_for(rows, chunks_for_width(g.gx), 64);   // <= bdetr_bn_bwd_chunks(rows), which sizes ws
    int nch = (int)cdiv64(rows, rpc);
    float* pa = ws; float* pb = ws + (int64_t)nch * C;
    BnBwdFn f{dout, out, x, mean, rstd, gamma, beta, C, relu};
    hipLaunchKernelGGL((colreduce2_kernel<BnBwdFn>), dim3(g.gx, nch), dim3(256), 0, st, f, rows, C, g.tx, rpc, pa, pb);
    sum_partials2(pa, pb, nch, C, dbeta, dgamma, st);
    int64_t n4 = rows * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4, 256, 2)), dim3(256), 0, st, dout, out, x, mean, rstd, gamma, beta, dgamma, dbeta,
                       relu, frozen, dx, dresidual, n4, C / 4, 1.0f / (float)rows);
    return bdetr_launch_status("bn_bwd");
}

extern "C" int bdetr_stem_pool_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                   int N, int H, int W, int C, float* out32, void* out_f16, uint8_t* tap, int* overflow_flag, void* stream) {
    BDETR_CHECK_ARG(y && mean && rstd && gamma && beta && tap && (out32 || out_f16) && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0,
                    "bdetr_stem_pool_fwd: bad arguments (C %% 8 == 0 required)");
    BDETR_CHECK_ARG((int64_t)N * H * W < (int64_t)1 << 31, "bdetr_stem_pool_fwd: more than 2^31 pixels");
    const StemGeom s{N, H, W, C, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1};
    const int64_t n4 = (int64_t)N * s.PH * s.PW * (C / 4);
    hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3(ew_grid(n4, 256, 1)), dim3(256), 0, (hipStream_t)stream, y, mean, rstd, gamma, beta, s, out32, out_f16,
                       reinterpret_cast<unsigned*>(tap), overflow_flag);
    return bdetr_launch_status("stem_pool_fwd");
}

constexpr int STEM_BWD_CHUNKS = 2048;
extern "C" int bdetr_stem_pool_bwd_chunks(int64_t rows) { return (int)cdiv64(rows, chunk_rows_for(rows, STEM_BWD_CHUNKS, 64)); }

extern "C" int bdetr_stem_pool_bwd(const float* dpool, const uint8_t* tap, const float* y, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int N, int H, int W, int C, float* dy, int dy_p16, float* dgamma, float* dbeta, float* ws, void* stream) {
    BDETR_CHECK_ARG(dpool && tap && y && mean && rstd && gamma && beta && dy && dgamma && dbeta && ws && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0,
                    "bdetr_stem_pool_bwd: bad arguments (C %% 8 == 0 required)");
    BDETR_CHECK_ARG((int64_t)N * H * W < (int64_t)1 << 31, "bdetr_stem_pool_bwd: more than 2^31 pixels");
    hipStream_t st = (hipStream_t)stream;
    const StemGeom s{N, H, W, C, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1};
    const int64_t rows = (int64_t)N * H * W;
    ColGeom g = col_geom(C);
    // (64 channels = ONE column block: 2048 row chunks instead of bn_bwd's 512, or a quarter of the chip's wave slots stay empty)
    int64_t rpc = chunk_rows_for(rows, STEM_BWD_CHUNKS, 64);
    int nch = (int)cdiv64(rows, rpc);
    float* pa = ws; float* pb = ws + (int64_t)nch * C;
    StemBwdFn f{dpool, reinterpret_cast<const unsigned*>(tap), y, mean, rstd, gamma, beta, s};
    hipLaunchKernelGGL((colreduce2_kernel<StemBwdFn>), dim3(g.gx, nch), dim3(256), 0, st, f, rows, C, g.tx, rpc, pa, pb);
    sum_partials2(pa, pb, nch, C, dbeta, dgamma, st);
    const int64_t n4 = rows * C / 4;
    hipLaunchKernelGGL(stem_bwd_apply_kernel, dim3(ew_grid(n4, 256, 2)), dim3(256), 0, st, dpool, reinterpret_cast<const unsigned*>(tap), y, mean, rstd, gamma, beta,
                       dgamma, dbeta, s, dy, dy_p16, n4, 1.0f / (float)rows);
    return bdetr_launch_status("stem_pool_bwd");
}

// element-wise grid whose stride (256 threads per workgroup) is a multiple of the channel-group count C / 4 where that costs at most a
// few extra workgroups: a thread then keeps one channel group for the whole launch and loads its per-channel terms once
// (bn_apply_p16_kernel, bn_bwd_apply_p16_kernel)
static int channel_aligned_grid(int64_t n4, int C) {
    int grid = ew_grid(n4, 256, 2);
    int a = C / 4, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }              // a = gcd(C / 4, 256)
    const int mult = (C / 4) / a;
    if (mult > 1 && mult <= 8) grid = (grid + mult - 1) / mult * mult;
    return grid;
}

extern "C" int bdetr_bn_apply_p16(const float* x, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, const void* residual, int residual_p16, const bdetr_bn_affine* residual_bn, int relu,
                                  float* out32, void* out_f16, void* out_bf16,
                                  uint64_t* relu_mask, int* overflow_flag, int64_t rows, int C, void* stream) {
    BDETR_CHECK_ARG(x && mean && rstd && gamma && beta && (out32 || out_f16 || out_bf16) && rows > 0 && C > 0 && C % 8 == 0,
                    "bdetr_bn_apply_p16: bad arguments (C %% 8 == 0 required)");
    BDETR_CHECK_ARG(residual_p16 != 2 || (residual && residual_bn && residual_bn->mean && residual_bn->rstd && residual_bn->gamma && residual_bn->beta),
                    "bdetr_bn_apply_p16: residual_p16 = 2 needs the raw shortcut tensor and its BatchNorm");
    const bdetr_bn_affine rbn = residual_p16 == 2 ? *residual_bn : bdetr_bn_affine{nullptr, nullptr, nullptr, nullptr};
    const int64_t n4 = rows * C / 4;
    const int grid = channel_aligned_grid(n4, C);
    hipLaunchKernelGGL(bn_apply_p16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, residual, residual_p16, rbn, relu,
                       out32, out_f16, out_bf16, reinterpret_cast<unsigned long long*>(relu_mask), overflow_flag, n4, C / 4);
    return bdetr_launch_status("bn_apply_p16");
}

namespace {
// x <- x * mask, mask = bn_apply_p16's 1-bit ReLU mask (the fallback for a lazily masked skip gradient, see ops.py)
__global__ __launch_bounds__(256) void relu_mask_apply_kernel(float* __restrict__ x, const unsigned long long* __restrict__ mask, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned pm = relu_mask_bits4(mask, i);
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!((pm >> e) & 1u)) v[e] = 0.f;
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}
}  // namespace

extern "C" int bdetr_relu_mask_apply(float* x, const uint64_t* relu_mask, int64_t n, void* stream) {
    BDETR_CHECK_ARG(x && relu_mask && n > 0 && n % 4 == 0, "bdetr_relu_mask_apply: bad arguments (n %% 4 == 0 required)");
    hipLaunchKernelGGL(relu_mask_apply_kernel, dim3(ew_grid(n / 4, 256, 2)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<const unsigned long long*>(relu_mask), n / 4);
    return bdetr_launch_status("relu_mask_apply");
}

static int bn_bwd_p16_impl(const float* dout, const void* out, int out_p16, const float* x, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                           float* dx32, void* dx_bf16, float* dgamma, float* dbeta, float* dresidual,
                           float* ws, const float* pre_g, const float* pre_gx, int pre_n, int64_t rows, int C, int even_h, int even_w, void* stream,
                           int dout_compact = 0) {
    BDETR_CHECK_ARG(!dout_compact || (even_h > 0 && even_h % 2 == 0 && even_w % 2 == 0 && pre_g == nullptr && dresidual == nullptr),
                    "bdetr_bn_bwd_p16_even_pixels: a compact dout needs an even map, the even-pixel reduction and no residual-gradient output");
    BDETR_CHECK_ARG(!dout_compact || rows * C / 4 < ((int64_t)1 << 32), "bdetr_bn_bwd_p16_even_pixels: compact dout: more than 2^32 float4s");
    BDETR_CHECK_ARG(dout && x && mean && rstd && gamma && dx_bf16 && dgamma && dbeta && (ws || pre_g) && rows > 0 && C > 0 && C % 8 == 0,
                    "bdetr_bn_bwd_p16: bad arguments (C %% 8 == 0 required)");
    BDETR_CHECK_ARG((pre_g == nullptr) == (pre_gx == nullptr) && (pre_g == nullptr || pre_n > 0), "bdetr_bn_bwd_p16: pre_g / pre_gx / pre_n inconsistent");
    BDETR_CHECK_ARG(!relu || out || beta, "bdetr_bn_bwd_p16: relu needs the forward output, or beta to recompute the mask from x");
    hipStream_t st = (hipStream_t)stream;
    ColGeom g = col_geom(C);
    int64_t rpc = chunk_rows_for(rows, chunks_for_width(g.gx), 64);   // <= bdetr_bn_bwd_chunks(rows), which sizes ws
    int nch = (int)cdiv64(rows, rpc);
    float* pa = ws; float* pb = ws + (int64_t)nch * C;
    if (pre_g != nullptr) {
        sum_partials2(pre_g, pre_gx, pre_n, C, dbeta, dgamma, st);
    } else {
        BnBwdFn f{dout, reinterpret_cast<const float*>(out), x, mean, rstd, gamma, beta, C, relu, out_p16, even_h, even_w, dout_compact};
        int64_t red_rows = rows;
        if (even_h > 0) {
            red_rows = rows / ((int64_t)even_h * even_w) * ((even_h + 1) / 2) * ((even_w + 1) / 2);
            rpc = chunk_rows_for(red_rows, chunks_for_width(g.gx), 64);
            nch = (int)cdiv64(red_rows, rpc);
            pb = ws + (int64_t)nch * C;
        }
        hipLaunchKernelGGL((colreduce2_kernel<BnBwdFn>), dim3(g.gx, nch), dim3(256), 0, st, f, red_rows, C, g.tx, rpc, pa, pb);
        sum_partials2(pa, pb, nch, C, dbeta, dgamma, st);
    }
    const int64_t n4 = rows * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_p16_kernel, dim3(channel_aligned_grid(n4, C)), dim3(256), 0, st, dout, out, out_p16, x, mean, rstd, gamma, beta, dgamma, dbeta,
                       relu, frozen, dx32, dx_bf16, dresidual, n4, C / 4, 1.0f / (float)rows, dout_compact ? even_h : 0, dout_compact ? even_w : 0);
    return bdetr_launch_status("bn_bwd_p16");
}

extern "C" int bdetr_bn_bwd_p16(const float* dout, const void* out, int out_p16, const float* x, const float* mean,
                                const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                                float* dx32, void* dx_bf16, float* dgamma, float* dbeta, float* dresidual,
                                float* ws, const float* pre_g, const float* pre_gx, int pre_n, int64_t rows, int C, void* stream) {
    return bn_bwd_p16_impl(dout, out, out_p16, x, mean, rstd, gamma, beta, relu, frozen, dx32, dx_bf16, dgamma, dbeta, dresidual, ws, pre_g, pre_gx, pre_n,
                           rows, C, 0, 0, stream);
}

extern "C" int bdetr_bn_bwd_p16_even_pixels(const float* dout, const void* out, int out_p16, const float* x, const float* mean,
                                            const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                                            float* dx32, void* dx_bf16, float* dgamma, float* dbeta, float* dresidual,
                                            float* ws, int N, int H, int W, int C, int dout_compact, void* stream) {
    BDETR_CHECK_ARG(N > 0 && H > 0 && W > 0 && ws && (int64_t)N * H * W < (int64_t)1 << 31, "bdetr_bn_bwd_p16_even_pixels: bad geometry");
    return bn_bwd_p16_impl(dout, out, out_p16, x, mean, rstd, gamma, beta, relu, frozen, dx32, dx_bf16, dgamma, dbeta, dresidual, ws, nullptr, nullptr, 0,
                           (int64_t)N * H * W, C, H, W, stream, dout_compact);
}

extern "C" int bdetr_add_dropout_layernorm_fwd(const float* x, const float* y, const float* gamma, const float* beta,
                                               float* out, float* mean, float* rstd, int64_t rows, int D,
                                               float eps, float rate, uint64_t seed, const uint64_t* seed_base, void* stream) {
    BDETR_CHECK_ARG(x && y && gamma && beta && out && mean && rstd && rows > 0, "bdetr_add_dropout_layernorm_fwd: null/empty argument");
    BDETR_CHECK_ARG(D % 4 == 0 && D > 0 && D <= 256 * LN_MAXV, "bdetr_add_dropout_layernorm_fwd: D=%d unsupported (multiple of 4, <= %d)", D, 256 * LN_MAXV);
    BDETR_CHECK_ARG(rate >= 0.f && rate < 1.f, "bdetr_add_dropout_layernorm_fwd: dropout rate must be in [0,1)");
    hipLaunchKernelGGL(add_drop_ln_fwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       x, y, gamma, beta, out, mean, rstd, rows, D, eps, rate, seed, seed_base);
    return bdetr_launch_status("add_dropout_layernorm_fwd");
}

extern "C" int bdetr_add_dropout_layernorm_bwd(const float* dout, const float* x, const float* y, const float* gamma,
                                               const float* mean, const float* rstd, float* dx, float* dy,
                                               float* dgamma, float* dbeta, float* ws, int64_t rows, int D,
                                               float rate, uint64_t seed, const uint64_t* seed_base, int accumulate_dx, void* stream) {
    BDETR_CHECK_ARG(dout && x && y && gamma && mean && rstd && dx && dy && dgamma && dbeta && ws && rows > 0,
                    "bdetr_add_dropout_layernorm_bwd: null/empty argument");
    BDETR_CHECK_ARG(D % 4 == 0 && D > 0 && D <= 256 * LN_MAXV, "bdetr_add_dropout_layernorm_bwd: D=%d unsupported", D);
    hipStream_t st = (hipStream_t)stream;
    int64_t rpc = chunk_rows_for(rows, 512, 16);
    int nch = (int)cdiv64(rows, rpc);
    float* pg = ws; float* pb = ws + (int64_t)nch * D;
    hipLaunchKernelGGL(add_drop_ln_bwd_kernel, dim3(nch), dim3(256), 0, st, dout, x, y, gamma, mean, rstd, dx, dy, pg, pb,
                       rows, D, rpc, rate, seed, accumulate_dx, seed_base);
    sum_partials2(pg, pb, nch, D, dgamma, dbeta, st);
    return bdetr_launch_status("add_dropout_layernorm_bwd");
}

static int softmax_dispatch(bool bwd, const float* a, const float* b, float* o, int64_t rows, int cols, float scale, hipStream_t st) {
    dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
    const int nper = (cols + 63) / 64;
#define SM_LAUNCH(N)                                                                                          \
    do {                                                                                                      \
        if (bwd) hipLaunchKernelGGL((softmax_bwd_kernel<N>), grid, block, 0, st, a, b, o, rows, cols, scale); \
        else hipLaunchKernelGGL((softmax_fwd_kernel<N>), grid, block, 0, st, a, o, rows, cols, scale);        \
    } while (0)
    if (nper <= 2) SM_LAUNCH(2);
    else if (nper <= 8) SM_LAUNCH(8);
    else if (nper <= 32) SM_LAUNCH(32);
    else { bdetr_set_error("softmax: cols=%d exceeds 2048", cols); return -1; }
#undef SM_LAUNCH
    return bdetr_launch_status("softmax");
}

extern "C" int bdetr_softmax_rows_fwd(const float* s, float* p, int64_t rows, int cols, float scale, void* stream) {
    BDETR_CHECK_ARG(s && p && rows > 0 && cols > 0, "bdetr_softmax_rows_fwd: bad arguments");
    return softmax_dispatch(false, s, nullptr, p, rows, cols, scale, (hipStream_t)stream);
}
extern "C" int bdetr_softmax_rows_bwd(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale, void* stream) {
    BDETR_CHECK_ARG(p && dp && ds && rows > 0 && cols > 0, "bdetr_softmax_rows_bwd: bad arguments");
    return softmax_dispatch(true, p, dp, ds, rows, cols, scale, (hipStream_t)stream);
}
extern "C" int bdetr_softmax_lastdim_fwd(const float* logits, float* p, int64_t rows, int cols, void* stream) {
    return bdetr_softmax_rows_fwd(logits, p, rows, cols, 1.0f, stream);
}
extern "C" int bdetr_softmax_lastdim_bwd(const float* p, const float* dp, float* dlogits, int64_t rows, int cols, void* stream) {
    return bdetr_softmax_rows_bwd(p, dp, dlogits, rows, cols, 1.0f, stream);
}
