// rowchain.hip - the row-local part of a transformer layer as ONE launch per direction (gfx950).
//
// Between two attention cores every operation of the reference's encoder / decoder layer acts on one token row at a time
// (transformers.py:101 OutputProjection, 135-137 Add + Dropout + LayerNorm, 174-180 Dense-ReLU, Dense, Add + Dropout + LayerNorm):
//   a  = ctx Wo^T + bo;   x1 = LN1(resid + drop(a));   h = relu(x1 W1^T + b1);   f = h W2^T + b2;   x2 = LN2(x1 + drop(f))
// Round 3 ran that as five launches forward (three 12.5-us GEMM launches and two LayerNorm launches, all latency bound: the model
// width is 256) and about twenty backward.  Here a workgroup owns 32 token rows and walks the whole chain with the activations in
// LDS / registers: forward = rowchain_fwd (1 or 3 GEMM stages), backward = rowchain_bwd (LayerNorm backward, the data gradients of
// the three Dense layers, the ReLU mask, the dropout masks, and per-workgroup partial sums of every bias / gamma / beta gradient),
// plus rowchain_reduce for those partial sums.  The weight gradients stay GEMMs of their own (reduction over all tokens).
//
// Arithmetic: the policy of the training step ('split', include/bdetr.h): forward products on f16 pairs (activations split in
// registers, weights pre-packed as the pair of 2^8 w - p16.h), gradient products on bf16 pairs, three v_mfma_f32_32x32x16 products
// each, fp32 accumulate; LayerNorm, dropout, ReLU in fp32 exactly like norm.hip's add_drop_ln kernels (same dropout hash: the masks
// of the fused and the unfused path are identical).
//
// Layout.  TRANSPOSED products: the MFMA's rows are output FEATURES (A operand = the weight tile), its columns are TOKENS (B
// operand = the activations), so that the accumulator of one stage - a lane owns ONE token and 16 features per 32-feature tile -
// has the LayerNorm axis inside the lane (+ one cross-wave exchange) and goes back to LDS as the next stage's B operand without a
// transpose.  Wave w of the 8 owns features [32 w, 32 w + 32) (one 32-feature tile; -DBDETR_RC_WAVES=4: two tiles per wave, the
// first form of the kernel - 256 VGPRs and AGPR spills; eight waves need 190-211 and measured +0.6-0.7 % on the step with any
// look-ahead depth from 4 to 8, round 4).  Weights never touch LDS: every wave reads a DIFFERENT eighth of a
// matrix, pre-packed in fragment order ([tile][k-step][hi | lo][lane] x 16 bytes: one fully coalesced 1-KiB load per fragment),
// prefetched PF (8) k-steps ahead across stage boundaries (weights do not depend on data).  What bounds a stage is the 256 KB of
// weights each workgroup streams from L2 (~70 GB/s per CU): ~3.7 us per stage, not the 96 MFMAs per wave.
#include "gemm_common.h"
#include "p16.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rf16x8 __attribute__((ext_vector_type(8)));

constexpr int RD = 256;                    // model width (the only one built: encoder_dim = decoder_dim = 256 in every BASELINE config)
constexpr int RBM = 32;                    // token rows per workgroup
#ifndef BDETR_RC_WAVES
#define BDETR_RC_WAVES 8
#endif
constexpr int RNW = BDETR_RC_WAVES, RNT = 64 * RNW;      // waves / threads per workgroup (4 or 8)
constexpr int TPW = 8 / RNW;               // 32-feature tiles per wave
static_assert(RNW == 4 || RNW == 8, "the eight feature tiles split over 4 or 8 waves");
constexpr int ROWB = 2 * RD + 16;          // bytes per token row of one LDS plane: +16 makes the 16-byte fragment reads of 16 lanes hit 64 distinct banks
constexpr int PLANE = RBM * ROWB;          // hi plane, then lo plane
#ifndef BDETR_RC_PF
#define BDETR_RC_PF 8
#endif
constexpr int PF = BDETR_RC_PF;           // k-steps of weight fragments in flight
constexpr int KSTEPS = RD / 16;
constexpr int W_PIECES = 8 * KSTEPS * 2 * 64;      // 16-byte pieces of one packed matrix (256 KB)
constexpr int NVEC = 7;                    // partial-sum vectors of the backward: dgamma2, dbeta2, dbias2, dbias1, dgamma1, dbeta1, dbias0

__device__ __forceinline__ uint32_t rc_hash(uint64_t z) {      // = norm.hip hash_u64
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float rc_keep(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
    return rc_hash(seed ^ (idx * 0xD6E8FEB86659FD93ull)) >= thresh ? inv_keep : 0.f;
}

struct Lane {                              // who this lane is inside the workgroup's tile
    int j, lh, wave, fbase;                // token row inside the tile, half (0 / 1), wave, first feature of the lane's 4-feature groups
    int m0, M;
    bool ok;                               // the lane's token exists
};
// registers: v[t][q] = features fbase + 32 t + 8 q + {0,1,2,3} of the lane's token (t = tile, q = accumulator quad: e = 4 q + c)
typedef f32x4 Tile[TPW][4];

__device__ __forceinline__ void tile_from_acc(const f32x16 (&acc)[TPW], Tile& v) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float x = acc[t][4 * q + c]; v[t][q][c] = x; }
}
__device__ __forceinline__ void tile_load(const float* __restrict__ base, const Lane& L, Tile& v) {
    const float* row = base + (int64_t)(L.m0 + L.j) * RD + L.fbase;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[t][q] = L.ok ? *reinterpret_cast<const f32x4*>(row + 32 * t + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void tile_store(float* __restrict__ base, const Lane& L, const Tile& v) {
    if (!L.ok) return;
    float* row = base + (int64_t)(L.m0 + L.j) * RD + L.fbase;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(row + 32 * t + 8 * q) = v[t][q];
}
__device__ __forceinline__ void vec_load(const float* __restrict__ vec, const Lane& L, Tile& v) {     // a per-feature parameter in the lane's layout
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[t][q] = *reinterpret_cast<const f32x4*>(vec + L.fbase + 32 * t + 8 * q);
}
// the tile as 16-bit pairs into the LDS planes (the next stage's B operand): 8 bytes hi + 8 bytes lo per 4 features
template <bool F16>
__device__ __forceinline__ void tile_to_lds(unsigned char* lds, const Lane& L, const Tile& v) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x2 hi, lo;
            unsigned h0, l0, h1, l1;
            if (F16) { p16_split2_f16(v[t][q][0], v[t][q][1], h0, l0); p16_split2_f16(v[t][q][2], v[t][q][3], h1, l1); }
            else     { p16_split2_bf16(v[t][q][0], v[t][q][1], h0, l0); p16_split2_bf16(v[t][q][2], v[t][q][3], h1, l1); }
            hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
            const int off = L.j * ROWB + (L.fbase + 32 * t + 8 * q) * 2;
            *reinterpret_cast<u32x2*>(lds + off) = hi;
            *reinterpret_cast<u32x2*>(lds + PLANE + off) = lo;
        }
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x16 acc) {
    if constexpr (F16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rf16x8, al), __builtin_bit_cast(rf16x8, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rf16x8, ah), __builtin_bit_cast(rf16x8, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rf16x8, ah), __builtin_bit_cast(rf16x8, bh), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rbf16x8, al), __builtin_bit_cast(rbf16x8, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rbf16x8, ah), __builtin_bit_cast(rbf16x8, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rbf16x8, ah), __builtin_bit_cast(rbf16x8, bh), acc, 0, 0, 0);
    }
    return acc;
}

// The weight fragments of k-step s (global step = 16 * stage + s) for the wave's two feature tiles: [tile][plane].
struct WFrag { u32x4 a[TPW][2]; };
__device__ __forceinline__ void wfrag_load(WFrag& f, const u32x4* __restrict__ wp, int wave, int lane, int s) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) f.a[t][p] = wp[(((TPW * wave + t) * KSTEPS + s) * 2 + p) * 64 + lane];
}

// One GEMM stage: acc[t] = W[features of tile t][:] . X[token][:] over the 256-deep reduction, X from the LDS planes.  `ring`
// holds the fragments of this stage's first PF k-steps on entry and of `wnext`'s first PF k-steps on exit.
template <bool F16>
__device__ __forceinline__ void gemm_stage(const u32x4* __restrict__ wp, const u32x4* __restrict__ wnext, const unsigned char* lds,
                                           const Lane& L, int lane, WFrag (&ring)[PF], f32x16 (&acc)[TPW]) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const unsigned char* brow = lds + L.j * ROWB + L.lh * 16;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const u32x4 bh = *reinterpret_cast<const u32x4*>(brow + 32 * s);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(brow + PLANE + 32 * s);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = mfma3<F16>(ring[s % PF].a[t][0], ring[s % PF].a[t][1], bh, bl, acc[t]);
        // The slot's MFMAs have issued: refill it for k-step s + PF.  The scheduling barriers keep the loads HERE - without them the
        // machine scheduler sinks every load to just in front of its first use (vmcnt(1) before each MFMA: no prefetch at all, measured
        // 11 us per stage instead of ~3).
        __builtin_amdgcn_sched_barrier(0);
        if (s + PF < KSTEPS) wfrag_load(ring[s % PF], wp, L.wave, lane, s + PF);
        else wfrag_load(ring[s % PF], wnext, L.wave, lane, s + PF - KSTEPS);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// sums over the 256 features of every token: a lane's 16 * TPW values, its half-wave partner, then the RNW waves through LDS.
// `red` region: [2][RNW][RBM] floats, one region per call site so that a single barrier per reduction is enough.
__device__ __forceinline__ void row_reduce2(float& a, float& b, float* red, const Lane& L) {
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (L.lh == 0) { red[L.wave * RBM + L.j] = a; red[RNW * RBM + L.wave * RBM + L.j] = b; }
    __syncthreads();
    float sa = red[L.j], sb = red[RNW * RBM + L.j];
#pragma unroll
    for (int w = 1; w < RNW; ++w) { sa += red[w * RBM + L.j]; sb += red[RNW * RBM + w * RBM + L.j]; }      // (4 waves: the same left-to-right order as before)
    a = sa; b = sb;
}
__device__ __forceinline__ float tile_sum(const Tile& v) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += (v[t][q][0] + v[t][q][1]) + (v[t][q][2] + v[t][q][3]);
    return s;
}

// x = LN(r + keep * y): y holds the Dense output on entry and the normalised row on exit, `s_out` the pre-norm sum
__device__ __forceinline__ void add_drop_ln(Tile& y, const Tile& r, Tile& s_out, const Tile& g, const Tile& b, float eps,
                                            float rate, uint64_t seed, float* red, const Lane& L, float& mean, float& rstd) {
    const uint32_t thresh = rate > 0.f ? (uint32_t)fminf(rate * 4294967296.0f, 4294967295.0f) : 0u;
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    const uint64_t row0 = (uint64_t)(L.m0 + L.j) * RD + L.fbase;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float yv = y[t][q][c];
                if (rate > 0.f) yv *= rc_keep(seed, row0 + 32 * t + 8 * q + c, thresh, inv_keep);
                s_out[t][q][c] = r[t][q][c] + yv;
            }
    float s = tile_sum(s_out), dummy = 0.f;
    row_reduce2(s, dummy, red, L);
    mean = s * (1.0f / RD);
    float qv = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float d = s_out[t][q][c] - mean; qv += d * d; }
    dummy = 0.f;
    row_reduce2(qv, dummy, red + 2 * RNW * RBM, L);
    rstd = 1.0f / sqrtf(qv * (1.0f / RD) + eps);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[t][q] = (s_out[t][q] - mean) * rstd * g[t][q] + b[t][q];
}

}  // namespace

struct bdetr_rowchain_fwd_args {
    int M, nstages; float eps, rate;
    const float* ctx; const float* resid;
    const void* w[3]; const float* bias[3];
    const float* g1; const float* b1; const float* g2; const float* b2;
    float* pre1; float* x1; float* mean1; float* rstd1; float* h; float* pre2; float* x2; float* mean2; float* rstd2;
    uint64_t seed1, seed2; const uint64_t* seed_base;
};

namespace {

__global__ __launch_bounds__(RNT) void rowchain_fwd_kernel(bdetr_rowchain_fwd_args a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PLANE];
    __shared__ float red[4][2 * RNW * RBM];
    const int tid = threadIdx.x, lane = tid & 63;
    Lane L;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.j = lane & 31; L.lh = lane >> 5; L.fbase = 32 * TPW * L.wave + 4 * L.lh;
    L.m0 = blockIdx.x * RBM; L.M = a.M; L.ok = L.m0 + L.j < a.M;
    uint64_t seed1 = a.seed1, seed2 = a.seed2;
    if (a.seed_base != nullptr) { const uint64_t sb = *a.seed_base * 0x100000001B3ull; seed1 ^= sb; seed2 ^= sb; }
    const u32x4* w0 = reinterpret_cast<const u32x4*>(a.w[0]);
    const u32x4* w1 = reinterpret_cast<const u32x4*>(a.nstages > 1 ? a.w[1] : a.w[0]);
    const u32x4* w2 = reinterpret_cast<const u32x4*>(a.nstages > 1 ? a.w[2] : a.w[0]);

    // weights of the first k-steps: in flight while the context tile is staged
    WFrag ring[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) wfrag_load(ring[s], w0, L.wave, lane, s);

    // stage the attention context [32 tokens][256] as f16 pairs: row-major coalesced reads, 4 features per thread and pass
    {
        const float inv = 1.f; (void)inv;
#pragma unroll
        for (int p = 0; p < RBM * RD / 4 / RNT; ++p) {
            const int v = tid + RNT * p, r = v >> 6, c4 = v & 63;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (L.m0 + r < a.M) x = *reinterpret_cast<const f32x4*>(a.ctx + (int64_t)(L.m0 + r) * RD + 4 * c4);
            unsigned h0, l0, h1, l1;
            p16_split2_f16(x[0], x[1], h0, l0); p16_split2_f16(x[2], x[3], h1, l1);
            u32x2 hi, lo; hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
            *reinterpret_cast<u32x2*>(lds + r * ROWB + 8 * c4) = hi;
            *reinterpret_cast<u32x2*>(lds + PLANE + r * ROWB + 8 * c4) = lo;
        }
    }
    __syncthreads();

    f32x16 acc[TPW];
    Tile y, r, s;
    float mean, rstd;
    constexpr float WS = 1.0f / P16_W_SCALE;
    // ---- stage 1: output projection + Add + Dropout + LayerNorm ----
    // (the epilogue's operands are requested BEFORE the K loop and pinned there: one workgroup per CU has nobody to hide a load
    // behind, every round trip left in an epilogue is ~1 us of the launch)
    Tile bia, gam, bet;
    tile_load(a.resid, L, r);
    vec_load(a.bias[0], L, bia); vec_load(a.g1, L, gam); vec_load(a.b1, L, bet);
    __builtin_amdgcn_sched_barrier(0);
    gemm_stage<true>(w0, w1, lds, L, lane, ring, acc);
    tile_from_acc(acc, y);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) y[t][q] = y[t][q] * WS + bia[t][q];
    add_drop_ln(y, r, s, gam, bet, a.eps, a.rate, seed1, red[0], L, mean, rstd);
    tile_store(a.pre1, L, s);
    tile_store(a.x1, L, y);
    if (L.wave == 0 && L.lh == 0 && L.ok) { a.mean1[L.m0 + L.j] = mean; a.rstd1[L.m0 + L.j] = rstd; }
    if (a.nstages == 1) return;
    // (every wave has left the K loop: add_drop_ln's barriers) -> x1 becomes the next B operand
    tile_to_lds<true>(lds, L, y);
    __syncthreads();
    // ---- stage 2: Dense + ReLU (x1 stays in `y`: the residual of stage 3) ----
    vec_load(a.bias[1], L, bia);
    __builtin_amdgcn_sched_barrier(0);
    gemm_stage<true>(w1, w2, lds, L, lane, ring, acc);
    Tile hh;
    tile_from_acc(acc, hh);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) hh[t][q][c] = fmaxf(hh[t][q][c] * WS + bia[t][q][c], 0.f);
    tile_store(a.h, L, hh);
    __syncthreads();                        // all waves are done reading x1's planes
    tile_to_lds<true>(lds, L, hh);
    __syncthreads();
    // ---- stage 3: Dense + Add + Dropout + LayerNorm ----
    vec_load(a.bias[2], L, bia); vec_load(a.g2, L, gam); vec_load(a.b2, L, bet);
    __builtin_amdgcn_sched_barrier(0);
    gemm_stage<true>(w2, w2, lds, L, lane, ring, acc);
    Tile f;
    tile_from_acc(acc, f);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) f[t][q] = f[t][q] * WS + bia[t][q];
    add_drop_ln(f, y, s, gam, bet, a.eps, a.rate, seed2, red[2], L, mean, rstd);
    tile_store(a.pre2, L, s);
    tile_store(a.x2, L, f);
    if (L.wave == 0 && L.lh == 0 && L.ok) { a.mean2[L.m0 + L.j] = mean; a.rstd2[L.m0 + L.j] = rstd; }
}

}  // namespace

struct bdetr_rowchain_bwd_args {
    int M, nstages; float rate;
    const float* dout;                       // gradient of x2 (3 stages) or of x1 (1 stage)
    const float* pre2; const float* mean2; const float* rstd2; const float* g2;
    const float* h;
    const float* pre1; const float* mean1; const float* rstd1; const float* g1;
    const void* wt[3];                       // packed backward copies (bf16 pairs of W^T) of Wo, W1, W2
    float* G2; float* G1; float* G0;         // dropout-masked LayerNorm input gradients / ReLU-masked hidden gradient: the Dense layers' output gradients (operands of their weight gradients)
    float* dresid; float* dctx;
    float* partials;                         // [gridDim.x][NVEC][256]
    uint64_t seed1, seed2; const uint64_t* seed_base;
};

namespace {

// one butterfly step: exchange with lane ^ N; the N values whose index bit matches the lane's token bit survive (compile-time indices only)
template <int N, int LEN>
__device__ __forceinline__ void col_butterfly(float (&x)[LEN], int j) {
    const bool up = (j & N) != 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float mine = up ? x[i + N] : x[i], give = up ? x[i] : x[i + N];
        x[i] = mine + __shfl_xor(give, N, 64);
    }
}

// Sum over the workgroup's 32 tokens of every register of `v` (lanes of one half hold different tokens, the same features):
// butterfly over the five token bits with halving - after step k a lane keeps only the values whose index has bit (4 - k) equal to its
// own token bit - so 31 exchanges instead of 160.  Lane (j, lh) ends up with the total of value number j of its half:
// value n <-> (t = n >> 4, e = n & 15) <-> feature fbase + 32 t + 8 (e >> 2) + (e & 3).  Invalid tokens must hold zeros.
__device__ __forceinline__ void col_reduce_store(const Tile& v, float* __restrict__ dst /* [256] */, const Lane& L) {
    float x[16 * TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) x[16 * t + 4 * q + c] = v[t][q][c];
    if constexpr (TPW == 2) {
        col_butterfly<16>(x, L.j);
    } else {
        // one tile per wave (eight waves): 16 values for 32 tokens - token bit 4 folds without halving, lanes j and j ^ 16 end up equal
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] += __shfl_xor(x[i], 16, 64);
    }
    col_butterfly<8>(x, L.j);
    col_butterfly<4>(x, L.j);
    col_butterfly<2>(x, L.j);
    col_butterfly<1>(x, L.j);
    const int n = L.j, t = TPW == 2 ? n >> 4 : 0, e = n & 15;
    if (TPW == 2 || n < 16) dst[L.fbase + 32 * t + 8 * (e >> 2) + (e & 3)] = x[0];
}

// LayerNorm backward of one row set: g = dout * gamma, xhat = (s - mean) * rstd, dh = (g - mean_f(g) - xhat * mean_f(g xhat)) * rstd.
// Returns dh in `d` (in place), accumulates the column partial sums of dgamma / dbeta into dst.
__device__ __forceinline__ void ln_bwd(Tile& d, const Tile& s, Tile& g, float mean, float rstd, float* red, float* __restrict__ dgamma_part,
                                       float* __restrict__ dbeta_part, const Lane& L) {
    Tile xh, dg;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xh[t][q] = (s[t][q] - mean) * rstd;
            dg[t][q] = d[t][q] * xh[t][q];                  // dgamma contribution (dout * xhat)
            g[t][q] = d[t][q] * g[t][q];
#pragma unroll
            for (int c = 0; c < 4; ++c) { s1 += g[t][q][c]; s2 += g[t][q][c] * xh[t][q][c]; }
        }
    col_reduce_store(dg, dgamma_part, L);
    col_reduce_store(d, dbeta_part, L);
    row_reduce2(s1, s2, red, L);
    const float m1 = s1 * (1.0f / RD), m2 = s2 * (1.0f / RD);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) d[t][q] = (g[t][q] - m1 - xh[t][q] * m2) * rstd;
}
__device__ __forceinline__ void apply_keep(Tile& v, float rate, uint64_t seed, const Lane& L) {
    if (!(rate > 0.f)) return;
    const uint32_t thresh = (uint32_t)fminf(rate * 4294967296.0f, 4294967295.0f);
    const float inv_keep = 1.0f / (1.0f - rate);
    const uint64_t row0 = (uint64_t)(L.m0 + L.j) * RD + L.fbase;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[t][q][c] *= rc_keep(seed, row0 + 32 * t + 8 * q + c, thresh, inv_keep);
}

__global__ __launch_bounds__(RNT) void rowchain_bwd_kernel(bdetr_rowchain_bwd_args a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PLANE];
    __shared__ float red[4][2 * RNW * RBM];
    const int tid = threadIdx.x, lane = tid & 63;
    Lane L;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.j = lane & 31; L.lh = lane >> 5; L.fbase = 32 * TPW * L.wave + 4 * L.lh;
    L.m0 = blockIdx.x * RBM; L.M = a.M; L.ok = L.m0 + L.j < a.M;
    uint64_t seed1 = a.seed1, seed2 = a.seed2;
    if (a.seed_base != nullptr) { const uint64_t sb = *a.seed_base * 0x100000001B3ull; seed1 ^= sb; seed2 ^= sb; }
    const u32x4* wo = reinterpret_cast<const u32x4*>(a.wt[0]);
    const u32x4* w1 = reinterpret_cast<const u32x4*>(a.nstages > 1 ? a.wt[1] : a.wt[0]);
    const u32x4* w2 = reinterpret_cast<const u32x4*>(a.nstages > 1 ? a.wt[2] : a.wt[0]);
    float* part = a.partials + (int64_t)blockIdx.x * NVEC * RD;

    WFrag ring[PF];
    {
        const u32x4* first = a.nstages > 1 ? w2 : wo;
#pragma unroll
        for (int s = 0; s < PF; ++s) wfrag_load(ring[s], first, L.wave, lane, s);
    }
    f32x16 acc[TPW];
    Tile d;                                  // the running gradient, in the lane's [feature][token] layout
    tile_load(a.dout, L, d);                 // (zeros for tokens beyond M: every partial sum below relies on that)
    // saved tensors of the FIRST LayerNorm and the hidden activations: requested now, consumed after one or two K loops
    Tile s1t, g1t, hh;
    tile_load(a.pre1, L, s1t);
    vec_load(a.g1, L, g1t);
    const float mean1 = L.ok ? a.mean1[L.m0 + L.j] : 0.f, rstd1 = L.ok ? a.rstd1[L.m0 + L.j] : 0.f;
    if (a.nstages > 1) {
        tile_load(a.h, L, hh);
        Tile s2t, g2t;
        tile_load(a.pre2, L, s2t);
        vec_load(a.g2, L, g2t);
        const float mean2 = L.ok ? a.mean2[L.m0 + L.j] : 0.f, rstd2 = L.ok ? a.rstd2[L.m0 + L.j] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        // ---- LayerNorm 2 backward: dh2 -> residual branch (kept in `dres`) and, dropout-masked, the gradient of the second Dense ----
        ln_bwd(d, s2t, g2t, mean2, rstd2, red[0], part + 0 * RD, part + 1 * RD, L);
        Tile dres;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) dres[t][q] = d[t][q];
        apply_keep(d, a.rate, seed2, L);
        tile_store(a.G2, L, d);
        col_reduce_store(d, part + 2 * RD, L);
        tile_to_lds<false>(lds, L, d);
        __syncthreads();
        // ---- dH = dF W2 (A = packed W2^T), ReLU mask from the saved hidden activations ----
        gemm_stage<false>(w2, w1, lds, L, lane, ring, acc);
        tile_from_acc(acc, d);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[t][q][c] = hh[t][q][c] > 0.f ? d[t][q][c] : 0.f;
        tile_store(a.G1, L, d);
        col_reduce_store(d, part + 3 * RD, L);
        __syncthreads();                     // every wave has left the K loop
        tile_to_lds<false>(lds, L, d);
        __syncthreads();
        // ---- dx1 = dPre1 W1 + residual gradient ----
        gemm_stage<false>(w1, wo, lds, L, lane, ring, acc);
        tile_from_acc(acc, d);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) d[t][q] += dres[t][q];
        __syncthreads();                     // (K loop done before the planes are rewritten below)
    }
    // ---- LayerNorm 1 backward: dh1 -> gradient of the residual input and, dropout-masked, of the output projection ----
    ln_bwd(d, s1t, g1t, mean1, rstd1, red[2], part + 4 * RD, part + 5 * RD, L);
    tile_store(a.dresid, L, d);
    apply_keep(d, a.rate, seed1, L);
    tile_store(a.G0, L, d);
    col_reduce_store(d, part + 6 * RD, L);
    tile_to_lds<false>(lds, L, d);
    __syncthreads();
    // ---- dctx = dA Wo ----
    gemm_stage<false>(wo, wo, lds, L, lane, ring, acc);
    tile_from_acc(acc, d);
    tile_store(a.dctx, L, d);
}

// dst[v][c] (+)= sum over the workgroups' partial rows, fixed order (deterministic)
struct RcReduce { const float* partials; int nparts; float* dst[NVEC]; int accumulate[NVEC]; };
__global__ __launch_bounds__(256) void rowchain_reduce_kernel(RcReduce r) {
    const int v = blockIdx.x, c = threadIdx.x;
    float* dst = v == 0 ? r.dst[0] : v == 1 ? r.dst[1] : v == 2 ? r.dst[2] : v == 3 ? r.dst[3] : v == 4 ? r.dst[4] : v == 5 ? r.dst[5] : r.dst[6];
    const int accum = v == 0 ? r.accumulate[0] : v == 1 ? r.accumulate[1] : v == 2 ? r.accumulate[2] : v == 3 ? r.accumulate[3] : v == 4 ? r.accumulate[4] : v == 5 ? r.accumulate[5] : r.accumulate[6];
    if (dst == nullptr) return;
    const float* p = r.partials + (int64_t)v * RD + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 4 <= r.nparts; i += 4) {
        s0 += p[(int64_t)(i + 0) * NVEC * RD]; s1 += p[(int64_t)(i + 1) * NVEC * RD];
        s2 += p[(int64_t)(i + 2) * NVEC * RD]; s3 += p[(int64_t)(i + 3) * NVEC * RD];
    }
    for (; i < r.nparts; ++i) s0 += p[(int64_t)i * NVEC * RD];
    const float s = (s0 + s1) + (s2 + s3);
    dst[c] = accum ? dst[c] + s : s;
}

// Packs: one thread per 16-byte piece pair (hi + lo) of both copies.
//   forward  copy: A operand of  Y^T = W X^T : piece (T, s, lane) = 2^8 W[32 T + (lane & 31)][16 s + 8 (lane >> 5) + 0..7] as an f16 pair
//   backward copy: A operand of dX^T = W^T dY^T: piece (T, s, lane) = W[16 s + 8 (lane >> 5) + 0..7][32 T + (lane & 31)] as a bf16 pair
__global__ __launch_bounds__(256) void rowchain_pack_kernel(const int64_t* __restrict__ table, int* __restrict__ overflow) {
    const int64_t* row = table + (int64_t)blockIdx.y * 3;
    const float* w = reinterpret_cast<const float*>(row[0]);
    u32x4* fwd = reinterpret_cast<u32x4*>(row[1]);
    u32x4* bwd = reinterpret_cast<u32x4*>(row[2]);
    const int p = blockIdx.x * 256 + threadIdx.x;            // (T, s, lane)
    if (p >= 8 * KSTEPS * 64) return;
    const int lane = p & 63, s = (p >> 6) % KSTEPS, T = p / (64 * KSTEPS);
    const int i = 32 * T + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
    float a[8], b[8];
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(w + (int64_t)i * RD + k0), hi4 = *reinterpret_cast<const f32x4*>(w + (int64_t)i * RD + k0 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = lo4[e] * P16_W_SCALE; a[4 + e] = hi4[e] * P16_W_SCALE; }
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = w[(int64_t)(k0 + e) * RD + i];
    if (overflow != nullptr && p16_f16_overflow(a)) *overflow = 1;
    u32x4 h, l;
    const int64_t base = ((int64_t)(T * KSTEPS + s) * 2) * 64 + lane;
    if (fwd != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsigned hh, ll; p16_split2_f16(a[2 * e], a[2 * e + 1], hh, ll); h[e] = hh; l[e] = ll; }
        fwd[base] = h; fwd[base + 64] = l;
    }
    if (bwd != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsigned hh, ll; p16_split2_bf16(b[2 * e], b[2 * e + 1], hh, ll); h[e] = hh; l[e] = ll; }
        bwd[base] = h; bwd[base + 64] = l;
    }
}

}  // namespace

extern "C" int bdetr_rowchain_width(void) { return RD; }
extern "C" int64_t bdetr_rowchain_pack_elems(void) { return (int64_t)W_PIECES * 4; }          // fp32-sized elements of one packed copy (256 KB)
extern "C" int bdetr_rowchain_partial_rows(int64_t M) { return (int)cdiv64(M, RBM); }

extern "C" int bdetr_rowchain_pack_weights(const int64_t* table, int n, int* overflow_flag, void* stream) {
    BDETR_CHECK_ARG(table && n > 0, "bdetr_rowchain_pack_weights: bad arguments");
    hipLaunchKernelGGL(rowchain_pack_kernel, dim3(8 * KSTEPS * 64 / 256, n), dim3(256), 0, (hipStream_t)stream, table, overflow_flag);
    return bdetr_launch_status("rowchain_pack_weights");
}

extern "C" int bdetr_rowchain_fwd(const bdetr_rowchain_fwd_desc* d, void* stream) {
    BDETR_CHECK_ARG(d && d->M > 0 && (d->nstages == 1 || d->nstages == 3), "bdetr_rowchain_fwd: M > 0 and 1 or 3 stages");
    BDETR_CHECK_ARG(d->ctx && d->resid && d->w[0] && d->bias[0] && d->g1 && d->b1 && d->pre1 && d->x1 && d->mean1 && d->rstd1, "bdetr_rowchain_fwd: null pointer (stage 1)");
    BDETR_CHECK_ARG(d->nstages == 1 || (d->w[1] && d->w[2] && d->bias[1] && d->bias[2] && d->g2 && d->b2 && d->h && d->pre2 && d->x2 && d->mean2 && d->rstd2),
                    "bdetr_rowchain_fwd: null pointer (stages 2-3)");
    BDETR_CHECK_ARG(d->rate >= 0.f && d->rate < 1.f, "bdetr_rowchain_fwd: dropout rate must be in [0,1)");
    bdetr_rowchain_fwd_args a;
    a.M = (int)d->M; a.nstages = d->nstages; a.eps = d->eps; a.rate = d->rate; a.ctx = d->ctx; a.resid = d->resid;
    for (int k = 0; k < 3; ++k) { a.w[k] = d->w[k]; a.bias[k] = d->bias[k]; }
    a.g1 = d->g1; a.b1 = d->b1; a.g2 = d->g2; a.b2 = d->b2;
    a.pre1 = d->pre1; a.x1 = d->x1; a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.h = d->h; a.pre2 = d->pre2; a.x2 = d->x2; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.seed1 = d->seed1; a.seed2 = d->seed2; a.seed_base = d->seed_base;
    // live profiling (bench.py's roofline leg): the chain's GEMM stages count as one launch of nstages x 2 M 256^2 algorithmic FLOPs
    const bool prof = bdgemm::g_prof_on;
    if (prof) bdgemm::prof_begin((hipStream_t)stream, 2.0 * (double)d->M * RD * RD * d->nstages, (int)d->M, RD, RD * d->nstages, 1, RBM, RD * 10 + 7, bdgemm::AR_FP16X3 * 10000 + 6000);
    hipLaunchKernelGGL(rowchain_fwd_kernel, dim3((unsigned)cdiv64(d->M, RBM)), dim3(RNT), 0, (hipStream_t)stream, a);
    if (prof) bdgemm::prof_end((hipStream_t)stream);
    return bdetr_launch_status("rowchain_fwd");
}

extern "C" int bdetr_rowchain_bwd(const bdetr_rowchain_bwd_desc* d, void* stream) {
    BDETR_CHECK_ARG(d && d->M > 0 && (d->nstages == 1 || d->nstages == 3), "bdetr_rowchain_bwd: M > 0 and 1 or 3 stages");
    BDETR_CHECK_ARG(d->dout && d->pre1 && d->mean1 && d->rstd1 && d->g1 && d->wt[0] && d->G0 && d->dresid && d->dctx && d->partials, "bdetr_rowchain_bwd: null pointer (stage 1)");
    BDETR_CHECK_ARG(d->nstages == 1 || (d->pre2 && d->mean2 && d->rstd2 && d->g2 && d->h && d->wt[1] && d->wt[2] && d->G1 && d->G2), "bdetr_rowchain_bwd: null pointer (stages 2-3)");
    bdetr_rowchain_bwd_args a;
    a.M = (int)d->M; a.nstages = d->nstages; a.rate = d->rate; a.dout = d->dout;
    a.pre2 = d->pre2; a.mean2 = d->mean2; a.rstd2 = d->rstd2; a.g2 = d->g2; a.h = d->h; a.pre1 = d->pre1; a.mean1 = d->mean1; a.rstd1 = d->rstd1; a.g1 = d->g1;
    for (int k = 0; k < 3; ++k) a.wt[k] = d->wt[k];
    a.G2 = d->G2; a.G1 = d->G1; a.G0 = d->G0; a.dresid = d->dresid; a.dctx = d->dctx; a.partials = d->partials;
    a.seed1 = d->seed1; a.seed2 = d->seed2; a.seed_base = d->seed_base;
    const bool prof = bdgemm::g_prof_on;
    if (prof) bdgemm::prof_begin((hipStream_t)stream, 2.0 * (double)d->M * RD * RD * d->nstages, (int)d->M, RD, RD * d->nstages, 1, RBM, RD * 10 + 7, bdgemm::AR_BF16X3 * 10000 + 6000);
    hipLaunchKernelGGL(rowchain_bwd_kernel, dim3((unsigned)cdiv64(d->M, RBM)), dim3(RNT), 0, (hipStream_t)stream, a);
    if (prof) bdgemm::prof_end((hipStream_t)stream);
    return bdetr_launch_status("rowchain_bwd");
}

extern "C" int bdetr_rowchain_reduce(const float* partials, int nparts, float* const* dst7, const int* accumulate7, void* stream) {
    BDETR_CHECK_ARG(partials && nparts > 0 && dst7 && accumulate7, "bdetr_rowchain_reduce: bad arguments");
    RcReduce r;
    r.partials = partials; r.nparts = nparts;
    for (int v = 0; v < NVEC; ++v) { r.dst[v] = dst7[v]; r.accumulate[v] = accumulate7[v]; }
    hipLaunchKernelGGL(rowchain_reduce_kernel, dim3(NVEC), dim3(256), 0, (hipStream_t)stream, r);
    return bdetr_launch_status("rowchain_reduce");
}
