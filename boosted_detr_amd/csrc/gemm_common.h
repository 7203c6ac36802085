// gemm_common.h - pieces shared by the two MFMA GEMM translation units (gfx950):
//   igemm.hip  fp32 operands in HBM, split (or not) in registers on the way to LDS
//   sgemm.hip  operands pre-split by their producers ("P16" layout), HBM -> LDS by buffer_load ... lds
// Both compute  C[i][j] (+)= act(alpha * sum_r A(i,r) * B(j,r) + bias[j])  with 32x32 MFMA tiles held in
// f32x16 accumulators and share the problem description, the epilogue and the live-profiling hooks.
#pragma once
#include "common.h"
#include <type_traits>

namespace bdgemm {

enum { ST_STORE = 0, ST_ACCUM = 1, ST_ATOMIC = 2 };
// arithmetic codes (profiling `kind` = arith * 10000 + ...): 0-2 are igemm.hip's, 3-4 sgemm.hip's
enum { AR_FP32 = 0, AR_BF16X3 = 1, AR_FP16X3 = 2, AR_P16_F16 = 3, AR_P16_BF16 = 4, AR_BF16X6 = 5 };      // AR_BF16X6: three bf16 terms, six products (igemm.hip)

struct GemmParams {
    int I, J, R;
    int nb1;                 // batch index z = b0*nb1 + b1 (when splitk == 1)
    int splitk, r_chunk;     // split of the r range over gridDim.z (r_chunk multiple of the K-step)
    int tiles_i, tiles_j;
    float* c; int64_t ldc, sc0, sc1;
    const float* bias; float alpha; int act; int mode;
    float* stat_sum; float* stat_sq;    // [tiles_i*WM][J] partial column sums (may be null)
    int vec_store;                      // C rows are 16-byte aligned and J % 4 == 0: LDS-transposed float4 stores
    int rowmap;                         // scatter C rows through a strided-pixel map (conv s>1 bwd-data)
    int rm_OW, rm_OHOW, rm_H, rm_W, rm_stride;
    // grouped launch (dense operands only): blockIdx.z selects one of up to 4 independent problems that
    // share J, R and the epilogue flags but have their own pointers and row count (Q/K/V projections)
    int ngroups;
    const float* ga[4]; const float* gb[4]; float* gc[4]; const float* gbias[4]; int gI[4];
    // Fused BatchNorm-backward reduction (backward-data launches whose output IS the gradient of a BatchNorm(+ReLU) output
    // y' = relu?(gamma * (y - mean) * rstd + beta) with no residual): while the C tile passes through the epilogue, its
    // columns' partial sums of g = C * [y' > 0] and g * xhat are written to bnb_sum_g / bnb_sum_gx [tiles_i][J] - the pass
    // that would re-read C and y (colreduce2<BnBwdFn>) disappears.  bnb_y has C's shape and leading dimension.
    const float* bnb_y; const float* bnb_mean; const float* bnb_rstd; const float* bnb_gamma; const float* bnb_beta;
    int bnb_relu; float* bnb_sum_g; float* bnb_sum_gx;
    const unsigned long long* bnb_mask;      // (sgemm's dense 1x1 kernels, with acc_mask) the ReLU decision comes from these bits, not from recomputing it
    // A SECOND BatchNorm that shares the gradient g (the projection shortcut's: out = relu(bn3(y3) + bn0(y0)), round 4): its sum of g is the
    // first one's, only sum(g * xhat0) is new - one more read of y0 here instead of a reduction pass over (g, bits, y0) of its own.
    // Masked-accumulate + masked-sums launches only.
    const float* bnb2_y; const float* bnb2_mean; const float* bnb2_rstd; float* bnb2_sum_gx;
    // ST_ACCUM with acc_mask set (sgemm's dense 1x1 kernels only): C = product + C * bit, bit = bn_apply_p16's 1-bit
    // ReLU mask of the element (norm.hip relu_mask_bits4 layout) - the skip branch of a residual unit merged without ever
    // materialising its masked gradient
    const unsigned long long* acc_mask;
    // (round 5, with acc_mask; sgemm's EPI_COMPACT instantiation) the OLD gradient is not in C: it is the compact [N, acc_H/2, acc_W/2, J]
    // tensor of the even pixels of an [N, acc_H, acc_W] map (what the stride-2 1x1 backward-data products of the next stage wrote) and
    // zero everywhere else; C = product + old * bit is written to a fresh dense tensor
    const float* acc_src; int acc_H, acc_W;
    // (sgemm's XX weight-gradient kernels) the B operand is stored as f16 pairs - the forward operand of the same convolution - and
    // is converted to bf16 pairs in registers after the fragment read, so that its producer need not write a bf16 copy at all
    int b_f16;
#ifdef BDETR_SGEMM_DIAG
    int dbg;                            // diagnostic builds only (tools/epi_probe.py): BDETR_SGEMM_DBG bits 1 = no C stores, 2 = no K loop, 4 = per-element stores, 8 = no fragment reads / MFMAs, 16 = no staging loads
#endif
};

// Diagnostic switches that compile work OUT of a launch exist only in builds made with BDETR_CXXFLAGS=-DBDETR_SGEMM_DIAG
// (tools/epi_probe.py); in the production library the test is the constant false and the environment variable is never read.
#ifdef BDETR_SGEMM_DIAG
#define BDETR_DBG(g, bits) (((g).dbg & (bits)) != 0)
#else
#define BDETR_DBG(g, bits) false
#endif

inline void init_params(GemmParams& g) {
    g = GemmParams{};
    g.nb1 = 1; g.splitk = 1; g.alpha = 1.f; g.mode = ST_STORE;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == BDETR_ACT_RELU) return fmaxf(v, 0.f);
    if (act == BDETR_ACT_TANH) return tanhf(v);
    return v;
}

// XCD-aware tile order: blocks that are dispatched to the same XCD (blockIdx % 8) get consecutive logical
// tiles (bijective for any tile count), so one XCD's L2 sees neighbouring tiles' shared operand rows.
__device__ __forceinline__ int xcd_tile(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// First half of every epilogue, in registers: bias + activation, then per-column partial sums of the output
// (BatchNorm statistics ride the conv epilogue).  The activation is a compile-time constant of each copy of the loop
// nest: a run-time switch per element compiled to a chain of scalar branches per accumulator register (64 of them per
// lane on a 128x128 tile - about 3 us per workgroup with nothing else going on).
// bias_pre (may be null): the tile's bias values, one per 32-column block of this wave, loaded by the caller before its K
// loop so that the epilogue does not start with an exposed memory round trip (gemm_load_bias).
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_load_bias(const GemmParams& g, int j0, float (&bias_pre)[BN / WN / 32]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave % WN;
#pragma unroll
    for (int b = 0; b < BN / WN / 32; ++b) {
        const int j = j0 + wn * (BN / WN) + b * 32 + (lane & 31);
        bias_pre[b] = (g.bias != nullptr && j < g.J) ? g.bias[j] : 0.f;
    }
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_bias_act_stats(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const GemmParams& g, int tile_i, int i0, int j0,
                                                    const float* bias_pre = nullptr) {
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // recompute the lane-derived indices here (see sgemm.hip's direct_epilogue)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    auto bias_act_stats = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int j = j0 + wn * WTN + b * 32 + li;
            const bool jok = j < g.J;
            const float bias = bias_pre != nullptr ? bias_pre[b] : ((g.bias != nullptr && jok) ? g.bias[j] : 0.f);
            float csum = 0.f, csq = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = apply_act(g.alpha * acc[a][b][e] + bias, ACT);
                    acc[a][b][e] = v;
                    const float vs = (i < g.I && jok) ? v : 0.f;
                    csum += vs; csq += vs * vs;
                }
            }
            if (g.stat_sum != nullptr) {
                csum += __shfl_xor(csum, 32, 64);
                csq += __shfl_xor(csq, 32, 64);
                if (lh == 0 && jok) {
                    const int64_t chunk = (int64_t)tile_i * WM + wm;
                    g.stat_sum[chunk * g.J + j] = csum;
                    g.stat_sq[chunk * g.J + j] = csq;
                }
            }
        }
    };
    if (g.act == BDETR_ACT_NONE) {
        if (g.bias != nullptr || g.alpha != 1.f || g.stat_sum != nullptr) bias_act_stats(std::integral_constant<int, BDETR_ACT_NONE>{});
    } else if (g.act == BDETR_ACT_RELU) bias_act_stats(std::integral_constant<int, BDETR_ACT_RELU>{});
    else bias_act_stats(std::integral_constant<int, BDETR_ACT_TANH>{});
}

// bn_apply_p16's ReLU bit mask (norm.hip relu_mask_bits4): component e of float4 index i lives in 64-bit word (i >> 6) * 4 + e, bit i & 63
__device__ __forceinline__ unsigned mask_bits4(const unsigned long long* mask, int64_t i) {
    const unsigned long long* w = mask + (i >> 6) * 4;
    const int b = (int)(i & 63);
    return (unsigned)((w[0] >> b) & 1ull) | ((unsigned)((w[1] >> b) & 1ull) << 1) | ((unsigned)((w[2] >> b) & 1ull) << 2) | ((unsigned)((w[3] >> b) & 1ull) << 3);
}

// Epilogue of a BM x BN workgroup tile held as TM x TN 32x32 accumulators per wave (C/D layout of the
// 32x32 MFMAs: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)): bias + activation and the
// statistics (above), then either LDS-transposed 16-byte row stores (store / accumulate) or per-element stores /
// atomics (split-K).  `lds` must hold LDS_FLOATS >= BM * BN floats and be free to overwrite once every wave has
// passed the barrier this function starts with.
template <int BM, int BN, int WM, int WN, int NT, int LDS_FLOATS, bool MASKED_VARIANTS = false, bool BNB2_ONLY = false, bool COMPACT_ONLY = false>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const GemmParams& g, float* lds,
                                              int tile_i, int i0, int j0, float* cbase, const float* bias_pre = nullptr) {
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    auto out_row = [&](int i) -> int64_t {
        if (!g.rowmap) return i;
        int n = i / g.rm_OHOW; int rem = i - n * g.rm_OHOW;
        int oh = rem / g.rm_OW; int ow = rem - oh * g.rm_OW;
        return ((int64_t)n * g.rm_H + (int64_t)oh * g.rm_stride) * g.rm_W + (int64_t)ow * g.rm_stride;
    };
    gemm_bias_act_stats<BM, BN, WM, WN>(acc, g, tile_i, i0, j0, bias_pre);

    if (g.vec_store && g.mode != ST_ATOMIC) {
        // Transpose the tile through LDS (the staging buffers are free once every wave is past the barrier) so
        // that every lane stores 16 contiguous bytes: whole rows per 16 lanes instead of 4-byte stores in 128-byte
        // segments - 4x fewer store instructions, full-line writes.
        constexpr bool PAD = BM * (BN + 4) <= LDS_FLOATS;
        constexpr int CLD = PAD ? BN + 4 : BN;
        static_assert(BM * CLD <= LDS_FLOATS, "epilogue tile must fit in the staging LDS");
        // Without room for the row padding the two half-waves of an accumulator register (rows r and r + 4) would hit the
        // same 32 banks: XOR bit 5 of the column with bit 2 of the row instead (64 banks x 4 B).
        constexpr bool SWZ = !PAD && BN % 64 == 0;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    lds[(wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CLD + ((wn * WTN + b * 32 + li) ^ (SWZ ? lh << 5 : 0))] = acc[a][b][e];
        __syncthreads();
        constexpr int V_PER_ROW = BN / 4, RSTEP = NT / V_PER_ROW, ITERS = BM / RSTEP;
        static_assert(NT % V_PER_ROW == 0 && BM % RSTEP == 0, "a thread keeps one column group and a fixed row stride");
        static_assert(!SWZ || RSTEP % 8 == 0, "the swizzle bit of a thread's rows must not change between iterations");
        const int c4 = tid % V_PER_ROW, r0 = tid / V_PER_ROW;
        const int jc = j0 + 4 * c4;                              // this thread's 4 columns (the same in every iteration)
        const bool jok = jc < g.J;
        const float* lrow = lds + r0 * CLD + ((4 * c4) ^ (SWZ ? ((r0 >> 2) & 1) << 5 : 0));
        // Every uniform switch (row map, accumulate, fused BatchNorm-backward sums) selects a fully unrolled copy of the
        // loop: all LDS reads, then all global loads, then the stores - one memory round trip per tile instead of one
        // per row group (the run-time switches inside a rolled loop waited on every load and store in turn).
        // MASKED (1x1 stride-1 backward-data of a residual unit's first convolution): the old value is the unit's output gradient
        // still without its ReLU mask - C = product + old * bit (g.acc_mask) - and, with BNB, the ReLU decision of the fused sums
        // comes from g.bnb_mask instead of being recomputed.  Accumulate + sums together keep three float4 per row group alive
        // (product, old, y): that combination runs the tile in two halves so that it stays inside 256 VGPRs.
        auto vec_out = [&](auto rowmap_c, auto accum_c, auto bnb_c, auto masked_c, auto bnb2_c, auto compact_c) {
            constexpr bool ROWMAP = decltype(rowmap_c)::value, ACCUM = decltype(accum_c)::value, BNB = decltype(bnb_c)::value, MASKED = decltype(masked_c)::value;
            constexpr bool BNB2 = decltype(bnb2_c)::value;         // + sum(g * xhat) of a second BatchNorm over the same g (bnb2_*)
            constexpr bool COMPACT = decltype(compact_c)::value;   // the old gradient comes from g.acc_src (even pixels only), not from C
            static_assert(!COMPACT || (ACCUM && MASKED && !ROWMAP && !BNB2), "the compact old gradient rides the masked accumulate");
            static_assert(!BNB2 || (BNB && ACCUM && MASKED), "the second BatchNorm rides the masked accumulate with masked sums");
            constexpr int NPART = (ACCUM && BNB) ? (BNB2 && ITERS % 4 == 0 ? 4 : 2) : 1, PIT = ITERS / NPART;
            static_assert(ITERS % NPART == 0, "the tile splits into equal parts");
            f32x4 bnv[BNB ? 6 : 1];                       // mean, rstd, gamma, beta of the thread's 4 columns; running sums of g, g * xhat
            f32x4 bn2[BNB2 ? 3 : 1];                      // mean, rstd of the second BatchNorm; running sum of g * xhat2
            if constexpr (BNB2) {
                bn2[0] = f32x4{0, 0, 0, 0}; bn2[1] = f32x4{0, 0, 0, 0}; bn2[2] = f32x4{0, 0, 0, 0};
                if (jok) { bn2[0] = *reinterpret_cast<const f32x4*>(g.bnb2_mean + jc); bn2[1] = *reinterpret_cast<const f32x4*>(g.bnb2_rstd + jc); }
            }
            if constexpr (BNB) {
#pragma unroll
                for (int q = 0; q < 6; ++q) bnv[q] = f32x4{0, 0, 0, 0};
                if (jok) {
                    bnv[0] = *reinterpret_cast<const f32x4*>(g.bnb_mean + jc); bnv[1] = *reinterpret_cast<const f32x4*>(g.bnb_rstd + jc);
                    bnv[2] = *reinterpret_cast<const f32x4*>(g.bnb_gamma + jc); bnv[3] = *reinterpret_cast<const f32x4*>(g.bnb_beta + jc);
                }
            }
#pragma unroll
            for (int part = 0; part < NPART; ++part) {
                f32x4 val[PIT], old[ACCUM ? PIT : 1], yv[BNB ? PIT : 1], y2v[BNB2 ? PIT : 1];
                unsigned abits[(ACCUM && MASKED) ? PIT : 1], bbits[(BNB && MASKED) ? PIT : 1];
                float* dst[PIT];
                bool ok[PIT];
#pragma unroll
                for (int k = 0; k < PIT; ++k) val[k] = *reinterpret_cast<const f32x4*>(lrow + (part * PIT + k) * RSTEP * CLD);
#pragma unroll
                for (int k = 0; k < PIT; ++k) {
                    const int i = i0 + r0 + (part * PIT + k) * RSTEP;
                    ok[k] = jok && i < g.I;
                    int64_t row = i;
                    if constexpr (ROWMAP) row = out_row(i);
                    dst[k] = cbase + row * g.ldc + jc;
                    if constexpr (ACCUM && COMPACT) {
                        old[k] = f32x4{0, 0, 0, 0};
                        const unsigned iu = (unsigned)i, w = iu % (unsigned)g.acc_W, t = iu / (unsigned)g.acc_W, h = t % (unsigned)g.acc_H, n = t / (unsigned)g.acc_H;
                        if (ok[k] && ((h | w) & 1u) == 0u)
                            old[k] = *reinterpret_cast<const f32x4*>(g.acc_src + ((int64_t)(n * (unsigned)(g.acc_H >> 1) + (h >> 1)) * (g.acc_W >> 1) + (w >> 1)) * g.ldc + jc);
                    } else if constexpr (ACCUM) { old[k] = f32x4{0, 0, 0, 0}; if (ok[k]) old[k] = *reinterpret_cast<const f32x4*>(dst[k]); }
                    if constexpr (ACCUM && MASKED) { abits[k] = 0u; if (ok[k]) abits[k] = mask_bits4(g.acc_mask, (row * g.ldc + jc) >> 2); }
                    if constexpr (BNB) { yv[k] = f32x4{0, 0, 0, 0}; if (ok[k]) yv[k] = *reinterpret_cast<const f32x4*>(g.bnb_y + (int64_t)i * g.ldc + jc); }
                    if constexpr (BNB && MASKED) { bbits[k] = 0u; if (ok[k]) bbits[k] = mask_bits4(g.bnb_mask, ((int64_t)i * g.ldc + jc) >> 2); }
                    if constexpr (BNB2) { y2v[k] = f32x4{0, 0, 0, 0}; if (ok[k]) y2v[k] = *reinterpret_cast<const f32x4*>(g.bnb2_y + (int64_t)i * g.ldc + jc); }
                }
#pragma unroll
                for (int k = 0; k < PIT; ++k) {
                    if constexpr (ACCUM && MASKED) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[k][e] += ((abits[k] >> e) & 1u) ? old[k][e] : 0.f;
                    } else if constexpr (ACCUM) val[k] += old[k];
                    if (BDETR_DBG(g, 1) && val[k][0] != 1234567.f) continue;
                    if (ok[k]) *reinterpret_cast<f32x4*>(dst[k]) = val[k];
                }
                if constexpr (BNB) {
#pragma unroll
                    for (int k = 0; k < PIT; ++k) {
                        if (!ok[k]) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bool on;
                            if constexpr (MASKED) on = ((bbits[k] >> e) & 1u) != 0u;
                            else on = !g.bnb_relu || (__builtin_fmaf(yv[k][e] - bnv[0][e], bnv[1][e] * bnv[2][e], bnv[3][e]) > 0.f);      // = norm.hip's bn_affine
                            const float ge = on ? val[k][e] : 0.f;
                            bnv[4][e] += ge; bnv[5][e] += ge * ((yv[k][e] - bnv[0][e]) * bnv[1][e]);
                            if constexpr (BNB2) bn2[2][e] += ge * ((y2v[k][e] - bn2[0][e]) * bn2[1][e]);
                        }
                    }
                }
            }
            if constexpr (BNB) {
                // fold the NT / V_PER_ROW threads that share a column group (fixed order), one partial row per tile_i
                static_assert(NT * 8 <= LDS_FLOATS, "reduction scratch must fit");
                __syncthreads();                                      // every thread is done reading the C tile
                *reinterpret_cast<f32x4*>(lds + tid * 8) = bnv[4];
                *reinterpret_cast<f32x4*>(lds + tid * 8 + 4) = bnv[5];
                __syncthreads();
                if (tid < V_PER_ROW && jok) {
                    f32x4 a = {0, 0, 0, 0}, b = a;
#pragma unroll
                    for (int k = 0; k < RSTEP; ++k) {
                        a += *reinterpret_cast<const f32x4*>(lds + (tid + k * V_PER_ROW) * 8);
                        b += *reinterpret_cast<const f32x4*>(lds + (tid + k * V_PER_ROW) * 8 + 4);
                    }
                    *reinterpret_cast<f32x4*>(g.bnb_sum_g + (int64_t)tile_i * g.J + jc) = a;
                    *reinterpret_cast<f32x4*>(g.bnb_sum_gx + (int64_t)tile_i * g.J + jc) = b;
                }
                if constexpr (BNB2) {                                 // the same fold for the second BatchNorm's sum
                    __syncthreads();
                    *reinterpret_cast<f32x4*>(lds + tid * 4) = bn2[2];
                    __syncthreads();
                    if (tid < V_PER_ROW && jok) {
                        f32x4 c = {0, 0, 0, 0};
#pragma unroll
                        for (int k = 0; k < RSTEP; ++k) c += *reinterpret_cast<const f32x4*>(lds + (tid + k * V_PER_ROW) * 4);
                        *reinterpret_cast<f32x4*>(g.bnb2_sum_gx + (int64_t)tile_i * g.J + jc) = c;
                    }
                }
            }
        };
        using T = std::true_type; using F = std::false_type;
        const bool accum = g.mode == ST_ACCUM;
        // host contracts: fused sums never come with a row map; a masked accumulate comes with masked sums or with none
        // (the second-BatchNorm form lives in a kernel instantiation of its own - sgemm.hip EPI_BNB2 - so that its extra float4 array
        // does not cost the other variants of the shared kernel registers: compiled in next to them it took their spills from 13 to 40)
        if constexpr (BNB2_ONLY) { vec_out(F{}, T{}, T{}, T{}, T{}, F{}); return; }
        if constexpr (COMPACT_ONLY) {               // (a kernel instantiation of its own, like BNB2_ONLY: sgemm.hip EPI_COMPACT)
            if (g.bnb_y != nullptr) vec_out(F{}, T{}, T{}, T{}, F{}, T{});
            else vec_out(F{}, T{}, F{}, T{}, F{}, T{});
            return;
        }
        if constexpr (MASKED_VARIANTS) {
            if (g.acc_mask != nullptr) {
                if (g.bnb_y != nullptr) vec_out(F{}, T{}, T{}, T{}, F{}, F{});
                else vec_out(F{}, T{}, F{}, T{}, F{}, F{});
                return;
            }
        }
        if (g.bnb_y != nullptr) vec_out(F{}, F{}, T{}, F{}, F{}, F{});
        else if (g.rowmap) { if (accum) vec_out(T{}, T{}, F{}, F{}, F{}, F{}); else vec_out(T{}, F{}, F{}, F{}, F{}, F{}); }
        else if (accum) vec_out(F{}, T{}, F{}, F{}, F{}, F{});
        else vec_out(F{}, F{}, F{}, F{}, F{}, F{});
        return;
    }
    // per-element stores / accumulates / atomics (split-K, or rows that are not 16-byte aligned): the mode is hoisted too
    auto scalar_out = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int j = j0 + wn * WTN + b * 32 + li;
            if (j >= g.J) continue;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (i < g.I) {
                        float* dst = cbase + out_row(i) * g.ldc + j;
                        const float v = acc[a][b][e];
                        if constexpr (MODE == ST_STORE) *dst = v;
                        else if constexpr (MODE == ST_ACCUM) *dst += v;
                        else atomicAdd(dst, v);
                    }
                }
            }
        }
    };
    if (g.mode == ST_STORE) scalar_out(std::integral_constant<int, ST_STORE>{});
    else if (g.mode == ST_ACCUM) scalar_out(std::integral_constant<int, ST_ACCUM>{});
    else scalar_out(std::integral_constant<int, ST_ATOMIC>{});
}

// ---- host side, defined in igemm.hip ----
int num_cus();
int gemm_mode();                    // BDETR_GEMM_* policy in force
extern bool g_prof_on;
void prof_begin(hipStream_t st, double flops, int I, int J, int R, int z, int bm, int bn, int kind);
void prof_end(hipStream_t st);
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// Deterministic split-K (the *_ws entry points): slice z of a split-K launch stores its partial C tile at ws + z * slab (mode
// ST_STORE, GemmParams::sc0 = slab) instead of adding it to C with float atomics; splitk_fold then adds the slabs to C in the
// fixed order z = 0 .. zdim - 1.  slab = splitk_slab(I * J); ws holds at least splitk * slab floats and is 16-byte aligned.
inline int64_t splitk_slab(int64_t elems) { return (elems + 3) / 4 * 4; }
int splitk_fold(const float* ws, int zdim, int64_t slab, float* dst, int64_t n, hipStream_t st);

// ---- hconv.hip: 3x3 / stride 1 / pad 1 convolutions with the input halo resident in LDS ----
int hconv_tile(int64_t rows, int W, int C, int J, bool f16);          // BM * 1000 + BN, or 0: stay on sgemm.hip's im2col kernel
int hconv_launch(int tile, bool f16, const void* x, int N, int H, int W, int C, const void* w, int J, const GemmParams& g, hipStream_t st);


// ---- hwgrad.hip: 3x3 / stride 1 / pad 1 weight gradients with both operands resident in LDS as sliding pixel windows ----
int hwgrad_slices(int N, int H, int W, int C, int K);                 // 0: not applicable (stay on sgemm.hip's im2col kernel), else the number of pixel slices
int hwgrad_launch(const void* x_bf16, const void* dy_bf16, float* dw, int N, int H, int W, int C, int K, int slices, float* slabs, long long slab, int* zdim_out, hipStream_t st);

}  // namespace bdgemm
