// gemm_common.h - pieces shared by the two MFMA GEMM translation units (gfx950):
//   igemm.hip  fp32 operands in HBM, split (or not) in registers on the way to LDS
//   sgemm.hip  operands pre-split by their producers ("P16" layout), HBM -> LDS by buffer_load ... lds
// Both compute  C[i][j] (+)= act(alpha * sum_r A(i,r) * B(j,r) + bias[j])  with 32x32 MFMA tiles held in
// f32x16 accumulators and share the problem description, the epilogue and the live-profiling hooks.
#pragma once
#include "common.h"

namespace bdgemm {

enum { ST_STORE = 0, ST_ACCUM = 1, ST_ATOMIC = 2 };
// arithmetic codes (profiling `kind` = arith * 10000 + ...): 0-2 are igemm.hip's, 3-4 sgemm.hip's
enum { AR_FP32 = 0, AR_BF16X3 = 1, AR_FP16X3 = 2, AR_P16_F16 = 3, AR_P16_BF16 = 4 };

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
};

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

// Epilogue of a BM x BN workgroup tile held as TM x TN 32x32 accumulators per wave (C/D layout of the
// 32x32 MFMAs: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)): bias + activation,
// per-column partial sums of the output (BatchNorm statistics ride the conv epilogue), then either
// LDS-transposed 16-byte row stores (store / accumulate) or per-element stores / atomics (split-K).
// `lds` must hold LDS_FLOATS >= BM * BN floats and be free to overwrite once every wave has passed the
// barrier this function starts with.
template <int BM, int BN, int WM, int WN, int NT, int LDS_FLOATS>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const GemmParams& g, float* lds,
                                              int tile_i, int i0, int j0, float* cbase) {
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    auto out_row = [&](int i) -> int64_t {
        if (!g.rowmap) return i;
        int n = i / g.rm_OHOW; int rem = i - n * g.rm_OHOW;
        int oh = rem / g.rm_OW; int ow = rem - oh * g.rm_OW;
        return ((int64_t)n * g.rm_H + (int64_t)oh * g.rm_stride) * g.rm_W + (int64_t)ow * g.rm_stride;
    };
    // bias + activation in registers, BN partial statistics from registers
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int j = j0 + wn * WTN + b * 32 + li;
        const bool jok = j < g.J;
        const float bias = (g.bias != nullptr && jok) ? g.bias[j] : 0.f;
        float csum = 0.f, csq = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = apply_act(g.alpha * acc[a][b][e] + bias, g.act);
                acc[a][b][e] = v;
                if (i < g.I && jok) { csum += v; csq += v * v; }
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

    if (g.vec_store && g.mode != ST_ATOMIC) {
        // Transpose the tile through LDS (the staging buffers are free once every wave is past the barrier) so
        // that every lane stores 16 contiguous bytes: whole rows per 16 lanes instead of 4-byte stores in 128-byte
        // segments - 4x fewer store instructions, full-line writes.
        constexpr int CLD = (BM * (BN + 4) <= LDS_FLOATS) ? BN + 4 : BN;
        static_assert(BM * CLD <= LDS_FLOATS, "epilogue tile must fit in the staging LDS");
        __syncthreads();
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    lds[(wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CLD + wn * WTN + b * 32 + li] = acc[a][b][e];
        __syncthreads();
        constexpr int V_PER_ROW = BN / 4;
        static_assert(NT % V_PER_ROW == 0, "a thread keeps one column group");
        const bool bnb = g.bnb_y != nullptr;
        const int jc = j0 + 4 * (tid % V_PER_ROW);               // this thread's 4 columns (the same in every iteration)
        f32x4 bm = {0, 0, 0, 0}, brs = bm, bgm = bm, bbt = bm, sg = bm, sgx = bm;
        if (bnb && jc < g.J) {
            bm = *reinterpret_cast<const f32x4*>(g.bnb_mean + jc); brs = *reinterpret_cast<const f32x4*>(g.bnb_rstd + jc);
            bgm = *reinterpret_cast<const f32x4*>(g.bnb_gamma + jc); bbt = *reinterpret_cast<const f32x4*>(g.bnb_beta + jc);
        }
#pragma unroll
        for (int v = tid; v < BM * V_PER_ROW; v += NT) {
            const int r = v / V_PER_ROW, c4 = v - r * V_PER_ROW;
            const int i = i0 + r, j = j0 + 4 * c4;
            if (i < g.I && j < g.J) {
                f32x4 val = *reinterpret_cast<const f32x4*>(lds + r * CLD + 4 * c4);
                f32x4* dst = reinterpret_cast<f32x4*>(cbase + out_row(i) * g.ldc + j);
                if (g.mode == ST_ACCUM) val += *dst;
                *dst = val;
                if (bnb) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(g.bnb_y + (int64_t)i * g.ldc + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool on = !g.bnb_relu || (__builtin_fmaf(yv[e] - bm[e], brs[e] * bgm[e], bbt[e]) > 0.f);      // = norm.hip's bn_affine
                        const float ge = on ? val[e] : 0.f;
                        sg[e] += ge; sgx[e] += ge * ((yv[e] - bm[e]) * brs[e]);
                    }
                }
            }
        }
        if (bnb) {
            // fold the NT / V_PER_ROW threads that share a column group (fixed order), one partial row per tile_i
            constexpr int G = NT / V_PER_ROW;
            static_assert(NT * 8 <= LDS_FLOATS, "reduction scratch must fit");
            __syncthreads();                                      // every thread is done reading the C tile
            *reinterpret_cast<f32x4*>(lds + tid * 8) = sg;
            *reinterpret_cast<f32x4*>(lds + tid * 8 + 4) = sgx;
            __syncthreads();
            if (tid < V_PER_ROW && jc < g.J) {
                f32x4 a = {0, 0, 0, 0}, b = a;
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    a += *reinterpret_cast<const f32x4*>(lds + (tid + k * V_PER_ROW) * 8);
                    b += *reinterpret_cast<const f32x4*>(lds + (tid + k * V_PER_ROW) * 8 + 4);
                }
                *reinterpret_cast<f32x4*>(g.bnb_sum_g + (int64_t)tile_i * g.J + jc) = a;
                *reinterpret_cast<f32x4*>(g.bnb_sum_gx + (int64_t)tile_i * g.J + jc) = b;
            }
        }
        return;
    }
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
                    if (g.mode == ST_STORE) *dst = v;
                    else if (g.mode == ST_ACCUM) *dst += v;
                    else atomicAdd(dst, v);
                }
            }
        }
    }
}

// ---- host side, defined in igemm.hip ----
int num_cus();
int gemm_mode();                    // BDETR_GEMM_* policy in force
extern bool g_prof_on;
void prof_begin(hipStream_t st, double flops, int I, int J, int R, int z, int bm, int bn, int kind);
void prof_end(hipStream_t st);
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace bdgemm
