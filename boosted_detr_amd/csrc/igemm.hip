// igemm.hip - MFMA implicit-GEMM family for gfx950 (MI355X): fp32 tensors, three arithmetics.
//
// One kernel template computes   C[i][j] (+)= act(alpha * sum_r A(i,r) * B(j,r) + bias[j])
// on 64-lane wavefronts with v_mfma_f32_32x32x2_f32 (exact fp32) or, splitting every operand into two
// 16-bit halves in registers, with three v_mfma_f32_32x32x16_{bf16,f16} per product (see "Split" below).
// Each operand is produced by a *loader* over its natural row-major matrix:
//   DenseLoader  - plain matrix with a leading dimension (Dense layers, 1x1 convs, attention)
//   PatchLoader  - on-the-fly im2col rows of an NHWC tensor (conv fwd / bwd-data / bwd-weight)
//   WFlipLoader  - OHWI conv weights read as [(tap',k)][c] with flipped taps (3x3 bwd-data)
// and staged into LDS either "r-contiguous" ([x][BK+4], read back with ds_read_b128) or
// "x-contiguous" ([BK][BX], read back with ds_read_b32), so that the HBM->LDS copy is always a
// straight 16-byte-per-lane coalesced copy and no operand ever needs a transposed copy in HBM.
//
// Replaces: Keras Conv2D / Dense / tf.linalg.matmul and their autodiff
// (backbone.py:37-38,76-78; transformers.py:41-48,62-65,86,97,101,174-177;
//  prediction_heads.py:40-43,106-110,175-179).
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include <vector>

using namespace bdgemm;

namespace {

constexpr int BK = 32;       // r-depth of one LDS stage
constexpr int RPAD = 4;      // row padding (floats) of r-contiguous LDS tiles: conflict-free ds_read_b128
constexpr int NTHREADS = 256;

// Split-bf16 ("bf16x3") mode: every fp32 operand value x is staged in LDS as two bf16 numbers
// hi = bf16(x) and lo = bf16(x - hi), and a product is evaluated as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-17 relative error per product (between fp32's
// 2^-24 and the TF32 2^-11 the reference's TensorFlow uses on tensor-core GPUs) at 3/16 of the fp32
// MFMA's cycles.  LDS rows are k-contiguous for both operand flavours (the MFMA wants 8 consecutive
// k per lane): [x][16 dwords hi | 16 dwords lo | 4 pad], one dword = the k-pair (2t, 2t+1); the
// 16-byte chunks of a row are XOR-swizzled with (x >> 4) & 7 so that both the ds_read_b128 fragment
// reads and the transposing ds_write_b32 of x-contiguous operands are bank-conflict free.
//
// Split-fp16 ("fp16x3", forward products): the same layout and MFMA count with fp16 halves, which carry
// 11 + 11 significant bits - fp32-grade products (~2^-23) - but only fp16's exponent range.  The lo half
// is therefore stored scaled by 2^11 (it is ~2^-12 of the value and would otherwise sink into fp16
// subnormals for |x| < 0.25) and the two cross products accumulate in a second accumulator that is folded
// in with 2^-11 in the epilogue.  Operand magnitudes must stay below 65504 (beyond that the hi half is
// inf and the output NaN - loud, not silent); gradients, whose range is unbounded, never take this path.
//
// Three-term bf16 split ("bf16x6", round 4: the fp32-grade arithmetic ON the 16-bit MFMA): x = hi + mid + lo with three bf16 terms
// (8 + 8 + 8 significant bits = fp32's 24) and a product keeps the six terms down to 2^-16 of it - hi*hi, hi*mid, mid*hi, hi*lo,
// lo*hi, mid*mid; the dropped mid*lo, lo*mid, lo*lo are 2^-24 and below - six v_mfma_f32_32x32x16_bf16 per 16-deep step: ~2^-22.5
// relative error per product at 6/16 of the fp32 MFMA's cycles, fp32's exponent range (so gradients may use it).  LDS rows grow a
// third 16-dword plane: [16 hi | 16 mid (the two-term layout's lo position) | 16 lo | 4 pad] = 52 dwords (an odd multiple of 4 like
// 36: the 16 rows of a ds_read_b128 pass still land in 16 distinct 4-bank groups).
constexpr int SLD = 36;      // dwords per LDS row in split mode
constexpr int SLD3 = 52;     // ... with three planes (bf16x6)
constexpr float LO_SCALE = 2048.f;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ARITH>
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    f32x2 v; v[0] = x0; v[1] = x1;
    if (ARITH == AR_BF16X3) {
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));        // v_cvt_pk_bf16_f32 (RNE)
        f32x2 r;
        r[0] = x0 - __builtin_bit_cast(float, hi << 16);                              // exact in fp32
        r[1] = x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    } else {
        const f16x2 h = __builtin_convertvector(v, f16x2);                            // v_cvt_pk_f16_f32 (RNE)
        hi = __builtin_bit_cast(unsigned, h);
        const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * LO_SCALE;           // residual exact, then 2^11
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
}

// three bf16 terms of two values (bf16x6): hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the residuals are exact in fp32
__device__ __forceinline__ void split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    f32x2 v; v[0] = x0; v[1] = x1;
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r;
    r[0] = x0 - __builtin_bit_cast(float, hi << 16);
    r[1] = x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    r[0] -= __builtin_bit_cast(float, mid << 16);
    r[1] -= __builtin_bit_cast(float, mid & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// ----------------------------------------------------------------------------------------
// operand descriptors + loaders
// ----------------------------------------------------------------------------------------
struct DenseOp { const float* p; int64_t ld, s0, s1; int rows, cols; };
struct PatchOp { const float* p; int N, H, W, C, OH, OW, R, S, stride, pad; int rows, cols; };
struct WFlipOp { const float* p; int K, C, R, S; int rows, cols; };

// All HBM reads go through raw buffer loads with a branch-free "invalid -> out-of-range offset"
// select: the hardware range check returns zeros for padding pixels, ragged tile edges and split-K
// tails, so the staging loads carry no control flow and stay in flight under the MFMA block (with
// branchy zero-fill code hipcc placed s_waitcnt vmcnt(0) in FRONT of the MFMAs).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0xFFFFFFF0u;              // byte offset beyond num_records -> load returns 0
constexpr unsigned NUM_RECORDS = 0xFFFFFF00u;      // operands must span < 4 GB (checked on the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)NUM_RECORDS, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

template <int VEC>
struct DenseLoader {
    using Op = DenseOp;
    struct Ctx { unsigned off; bool ok; };     // element offset of the row start
    static __device__ __forceinline__ const float* batch_base(const Op& op, int b0, int b1) {
        return op.p + (int64_t)b0 * op.s0 + (int64_t)b1 * op.s1;
    }
    static __device__ __forceinline__ void regroup(Op& op, const float* p, int x_extent, bool rc) {
        op.p = p;
        if (x_extent >= 0) { if (rc) op.rows = x_extent; else op.cols = x_extent; }
    }
    static __device__ __forceinline__ Ctx row_ctx(const Op& op, int row, int row_limit) {
        Ctx c; c.ok = row < row_limit; c.off = (unsigned)row * (unsigned)op.ld; return c;
    }
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rs, const Op& op, const Ctx& c, int col, int col_limit) {
        if (VEC == 4) {
            return buf_load4(rs, (c.ok && col < col_limit) ? (c.off + (unsigned)col) * 4u : OOB);
        } else {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = buf_load1(rs, (c.ok && col + e < col_limit) ? (c.off + (unsigned)(col + e)) * 4u : OOB);
            return v;
        }
    }
};

struct PatchLoader {
    using Op = PatchOp;
    struct Ctx { int nbase; int ih0, iw0; bool ok; };
    static __device__ __forceinline__ const float* batch_base(const Op& op, int, int) { return op.p; }
    static __device__ __forceinline__ void regroup(Op&, const float*, int, bool) {}
    static __device__ __forceinline__ Ctx row_ctx(const Op& op, int row, int row_limit) {
        Ctx c; c.ok = row < row_limit;
        int ohw = op.OH * op.OW;
        int n = row / ohw; int rem = row - n * ohw;
        int oh = rem / op.OW; int ow = rem - oh * op.OW;
        c.nbase = n * op.H * op.W;
        c.ih0 = oh * op.stride - op.pad; c.iw0 = ow * op.stride - op.pad;
        return c;
    }
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rs, const Op& op, const Ctx& c, int col, int col_limit) {
        int tap = col / op.C; int ch = col - tap * op.C;
        int r = tap / op.S; int s = tap - r * op.S;
        int ih = c.ih0 + r, iw = c.iw0 + s;
        const bool ok = c.ok && col < col_limit && (unsigned)ih < (unsigned)op.H && (unsigned)iw < (unsigned)op.W;
        const unsigned off = ((unsigned)(c.nbase + ih * op.W + iw) * (unsigned)op.C + (unsigned)ch) * 4u;
        return buf_load4(rs, ok ? off : OOB);
    }
};

struct WFlipLoader {
    using Op = WFlipOp;
    struct Ctx { unsigned off; bool ok; };
    static __device__ __forceinline__ const float* batch_base(const Op& op, int, int) { return op.p; }
    static __device__ __forceinline__ void regroup(Op&, const float*, int, bool) {}
    static __device__ __forceinline__ Ctx row_ctx(const Op& op, int row, int row_limit) {
        Ctx c; c.ok = row < row_limit;
        int tap = row / op.K; int k = row - tap * op.K;
        int r = tap / op.S; int s = tap - r * op.S;
        c.off = (unsigned)(((k * op.R + (op.R - 1 - r)) * op.S + (op.S - 1 - s)) * op.C);
        return c;
    }
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rs, const Op& op, const Ctx& c, int col, int col_limit) {
        return buf_load4(rs, (c.ok && col < col_limit) ? (c.off + (unsigned)col) * 4u : OOB);
    }
};

// ----------------------------------------------------------------------------------------
// epilogue / problem description
// ----------------------------------------------------------------------------------------
template <int BX, bool RC>
struct TileGeom {
    static constexpr int LDS_FLOATS = RC ? BX * (BK + RPAD) : BK * BX;
    static constexpr int NV = BX * BK / 4 / NTHREADS;           // float4 vectors per thread per stage
    static constexpr int VPR = RC ? BK / 4 : BX / 4;            // vectors per natural row
    static constexpr int LDS_LD = RC ? BK + RPAD : BX;
};

template <int BM, int BN, int WM, int WN, class LA, bool A_RC, class LB, bool B_RC, int ARITH>
__global__ __launch_bounds__(NTHREADS, 2)
void igemm_kernel(typename LA::Op opa, typename LB::Op opb, GemmParams g)
{
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int WTM = BM / WM, WTN = BN / WN;     // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;     // 32x32 MFMA tiles per wave
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    using GA = TileGeom<BM, A_RC>;
    using GB = TileGeom<BN, B_RC>;
    static_assert(GA::NV >= 1 && GB::NV >= 1, "tile too small for 256 threads");

    constexpr bool SPLIT = ARITH != AR_FP32;
    constexpr bool X6 = ARITH == AR_BF16X6;
    constexpr int SLDK = X6 ? SLD3 : SLD;           // dwords per LDS row
    constexpr int A_FLOATS = SPLIT ? BM * SLDK : GA::LDS_FLOATS;
    constexpr int B_FLOATS = SPLIT ? BN * SLDK : GB::LDS_FLOATS;
    constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static_assert(!SPLIT || ((A_RC || GA::NV % 2 == 0) && (B_RC || GB::NV % 2 == 0)), "split mode transposes k-pairs");
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: blocks that are dispatched to the same XCD (blockIdx % 8) get
    // consecutive logical tiles, and tile_j is the fast index, so one XCD's L2 sees the same
    // A rows (the expensive im2col gather) from neighbouring workgroups.
    // split-K launches: XCD-aware order over tiles x slices (a slice's tiles share operand rows; see sgemm.hip)
    const int nwg_ = g.tiles_i * g.tiles_j;
    const bool sk_ = g.ngroups == 0 && g.splitk > 1;
    const int lin_ = sk_ ? xcd_tile(blockIdx.z * nwg_ + blockIdx.x, nwg_ * gridDim.z) : xcd_tile(blockIdx.x, nwg_);
    const int zz_ = sk_ ? lin_ / nwg_ : blockIdx.z;
    const int wg = sk_ ? lin_ - zz_ * nwg_ : lin_;
    const int tile_i = wg / g.tiles_j, tile_j = wg - tile_i * g.tiles_j;
    const int i0 = tile_i * BM, j0 = tile_j * BN;

    int b0 = 0, b1 = 0, r_begin = 0, r_end = g.R;
    if (g.ngroups > 0) {
        // select with constant indices: a runtime index into kernel-argument arrays would push the whole
        // parameter block to scratch memory (measured: every igemm launch +25 %)
        // grouped AND split (the weight gradients of one layer's Dense kernels in one launch): blockIdx.z = group * splitk + slice
        const int z = g.splitk > 1 ? (int)blockIdx.z / g.splitk : (int)blockIdx.z;
        if (g.splitk > 1) {
            r_begin = ((int)blockIdx.z - z * g.splitk) * g.r_chunk;
            r_end = min(g.R, r_begin + g.r_chunk);
        }
#define BDETR_PICK(arr) (z == 0 ? g.arr[0] : z == 1 ? g.arr[1] : z == 2 ? g.arr[2] : g.arr[3])
        g.I = BDETR_PICK(gI); g.c = BDETR_PICK(gc); g.bias = BDETR_PICK(gbias);
        const float* pa = BDETR_PICK(ga);
        const float* pb = BDETR_PICK(gb);
#undef BDETR_PICK
        if (i0 >= g.I) return;                                  // this problem has fewer row tiles (uniform per block)
        LA::regroup(opa, pa, g.I, A_RC);
        LB::regroup(opb, pb, -1, B_RC);
    } else if (g.splitk > 1) {
        r_begin = zz_ * g.r_chunk;
        r_end = min(g.R, r_begin + g.r_chunk);
    } else {
        b0 = blockIdx.z / g.nb1; b1 = blockIdx.z - b0 * g.nb1;
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(LA::batch_base(opa, b0, b1));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(LB::batch_base(opb, b0, b1));
    // natural-matrix limits: r-contiguous operands have rows = x (I or J) and cols = r; x-contiguous the reverse
    const int a_rows = A_RC ? opa.rows : min(opa.rows, r_end), a_cols = A_RC ? min(opa.cols, r_end) : opa.cols;
    const int b_rows = B_RC ? opb.rows : min(opb.rows, r_end), b_cols = B_RC ? min(opb.cols, r_end) : opb.cols;

    // per-thread staging geometry
    typename LA::Ctx ctxA[GA::NV];
    typename LB::Ctx ctxB[GB::NV];
    int colA[GA::NV], colB[GB::NV], ldsoffA[GA::NV], ldsoffB[GB::NV], rowA[GA::NV], rowB[GB::NV];
    // split mode: ldsoff = (row base in dwords) * 32 + (swizzled dword inside the row's hi half); the lo
    // half is the same position with bit 4 flipped.  x-contiguous operands are loaded as k-row pairs
    // (p even: row 2t, p odd: row 2t+1, same 4 x columns) so that a thread can pack k-pairs itself.
    auto stage_geom = [&](int p, int vpr, int bx, bool rc, int lds_ld, int& row, int& col, int& ldsoff) {
        if (!SPLIT) {
            const int v = tid + NTHREADS * p;
            const int rt = v / vpr, c4 = v - rt * vpr;
            row = rt; col = 4 * c4; ldsoff = rt * lds_ld + 4 * c4;
        } else if (rc) {
            // 8 lanes cover one row's 16 hi dwords; a 32-lane ds_write_b64 pass therefore touches 4 rows, which
            // are chosen 4 apart (36 * 4 = 16 mod 64 banks): rows q = 16G + 4a + b -> 16G + 4b + a.
            const int v = tid + NTHREADS * p;
            const int q = v / vpr, c4 = v - q * vpr;             // vpr = 8: k = 4*c4 .. 4*c4+3
            const int rt = (q & ~15) | ((q & 3) << 2) | ((q >> 2) & 3);
            row = rt; col = 4 * c4;
            ldsoff = rt * SLDK * 32 + ((((c4 >> 1) ^ ((rt >> 4) & 7)) << 2) | ((c4 & 1) << 1));
        } else {
            const int u = tid + NTHREADS * (p >> 1);
            const int kp = u / (bx / 4), c4 = u - kp * (bx / 4);
            row = 2 * kp + (p & 1); col = 4 * c4;
            ldsoff = (4 * c4) * SLDK * 32 + ((((kp >> 2) ^ ((c4 >> 2) & 7)) << 2) | (kp & 3));
        }
    };
#pragma unroll
    for (int p = 0; p < GA::NV; ++p) {
        stage_geom(p, GA::VPR, BM, A_RC, GA::LDS_LD, rowA[p], colA[p], ldsoffA[p]);
        if (A_RC) ctxA[p] = LA::row_ctx(opa, i0 + rowA[p], a_rows);
    }
#pragma unroll
    for (int p = 0; p < GB::NV; ++p) {
        stage_geom(p, GB::VPR, BN, B_RC, GB::LDS_LD, rowB[p], colB[p], ldsoffB[p]);
        if (B_RC) ctxB[p] = LB::row_ctx(opb, j0 + rowB[p], b_rows);
    }

    // Two register staging sets: the loads of K-step t+2 are issued while K-step t computes, so an
    // HBM round trip has two full K-steps (not one) of MFMA work to hide under.
    f32x4 stA0[GA::NV], stB0[GB::NV], stA1[GA::NV], stB1[GB::NV];
    auto load_stage = [&](int r0, f32x4 (&ra)[GA::NV], f32x4 (&rb)[GB::NV]) {
#pragma unroll
        for (int p = 0; p < GA::NV; ++p) {
            if (A_RC) ra[p] = LA::load(rsA, opa, ctxA[p], r0 + colA[p], a_cols);
            else      ra[p] = LA::load(rsA, opa, LA::row_ctx(opa, r0 + rowA[p], a_rows), i0 + colA[p], a_cols);
        }
#pragma unroll
        for (int p = 0; p < GB::NV; ++p) {
            if (B_RC) rb[p] = LB::load(rsB, opb, ctxB[p], r0 + colB[p], b_cols);
            else      rb[p] = LB::load(rsB, opb, LB::row_ctx(opb, r0 + rowB[p], b_rows), j0 + colB[p], b_cols);
        }
    };
    auto write_split = [&](unsigned* base, bool rc, auto nv_c, const f32x4* r, const int* ldsoff) {
        constexpr int nv = decltype(nv_c)::value;
        // (three planes: the third sits behind the 32 interleaved hi / mid dwords at the hi position's offset inside its 16-dword half)
        if (rc) {
#pragma unroll
            for (int p = 0; p < nv; ++p) {
                unsigned* row = base + (ldsoff[p] >> 5);
                const int w = ldsoff[p] & 31;
                if constexpr (X6) {
                    unsigned h0, m0, l0, h1, m1, l1;
                    split3(r[p][0], r[p][1], h0, m0, l0);
                    split3(r[p][2], r[p][3], h1, m1, l1);
                    u32x2 hi, mid, lo;
                    hi[0] = h0; hi[1] = h1; mid[0] = m0; mid[1] = m1; lo[0] = l0; lo[1] = l1;
                    *reinterpret_cast<u32x2*>(row + w) = hi;
                    *reinterpret_cast<u32x2*>(row + (w ^ 16)) = mid;
                    *reinterpret_cast<u32x2*>(row + 32 + (w & 15)) = lo;
                } else {
                    unsigned h0, l0, h1, l1;
                    split2<ARITH>(r[p][0], r[p][1], h0, l0);
                    split2<ARITH>(r[p][2], r[p][3], h1, l1);
                    u32x2 hi, lo;
                    hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
                    *reinterpret_cast<u32x2*>(row + w) = hi;
                    *reinterpret_cast<u32x2*>(row + (w ^ 16)) = lo;
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p + 1 < nv; p += 2) {
                unsigned* row = base + (ldsoff[p] >> 5);
                const int w = ldsoff[p] & 31;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (X6) {
                        unsigned hi, mid, lo;
                        split3(r[p][e], r[p + 1][e], hi, mid, lo);
                        row[e * SLDK + w] = hi;
                        row[e * SLDK + (w ^ 16)] = mid;
                        row[e * SLDK + 32 + (w & 15)] = lo;
                    } else {
                        unsigned hi, lo;
                        split2<ARITH>(r[p][e], r[p + 1][e], hi, lo);
                        row[e * SLDK + w] = hi;
                        row[e * SLDK + (w ^ 16)] = lo;
                    }
                }
            }
        }
    };
    auto write_stage = [&](int buf, const f32x4 (&ra)[GA::NV], const f32x4 (&rb)[GB::NV]) {
        if (SPLIT) {
            unsigned* st = reinterpret_cast<unsigned*>(lds + buf * STAGE_FLOATS);
            write_split(st, A_RC, std::integral_constant<int, GA::NV>{}, ra, ldsoffA);
            write_split(st + A_FLOATS, B_RC, std::integral_constant<int, GB::NV>{}, rb, ldsoffB);
            return;
        }
#pragma unroll
        for (int p = 0; p < GA::NV; ++p) *reinterpret_cast<f32x4*>(lds + buf * STAGE_FLOATS + ldsoffA[p]) = ra[p];
#pragma unroll
        for (int p = 0; p < GB::NV; ++p) *reinterpret_cast<f32x4*>(lds + buf * STAGE_FLOATS + GA::LDS_FLOATS + ldsoffB[p]) = rb[p];
    };

    constexpr int TM2 = ARITH == AR_FP16X3 ? TM : 1, TN2 = ARITH == AR_FP16X3 ? TN : 1;
    f32x16 acc[TM][TN], acc2[TM2][TN2];       // acc2: the 2^11-scaled cross products of the fp16 split
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll
    for (int a = 0; a < TM2; ++a)
#pragma unroll
        for (int b = 0; b < TN2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][b][e] = 0.f;

    auto compute = [&](int buf) {
        if (SPLIT) {
            const unsigned* qA = reinterpret_cast<const unsigned*>(lds + buf * STAGE_FLOATS);
            const unsigned* qB = qA + A_FLOATS;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 ah[TM], al[TM], bh[TN], bl[TN];
                u32x4 a3[X6 ? TM : 1], b3[X6 ? TN : 1];      // bf16x6: al / bl hold the MID terms, a3 / b3 the third ones
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int xb = wm * WTM + a * 32 + li;
                    const int w = ((ks * 2 + lh) ^ ((xb >> 4) & 7)) << 2;
                    ah[a] = *reinterpret_cast<const u32x4*>(qA + xb * SLDK + w);
                    al[a] = *reinterpret_cast<const u32x4*>(qA + xb * SLDK + (w ^ 16));
                    if constexpr (X6) a3[a] = *reinterpret_cast<const u32x4*>(qA + xb * SLDK + 32 + (w & 15));
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int xb = wn * WTN + b * 32 + li;
                    const int w = ((ks * 2 + lh) ^ ((xb >> 4) & 7)) << 2;
                    bh[b] = *reinterpret_cast<const u32x4*>(qB + xb * SLDK + w);
                    bl[b] = *reinterpret_cast<const u32x4*>(qB + xb * SLDK + (w ^ 16));
                    if constexpr (X6) b3[b] = *reinterpret_cast<const u32x4*>(qB + xb * SLDK + 32 + (w & 15));
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        if constexpr (X6) {
#define BF8(v) __builtin_bit_cast(bf16x8, v)
                            // smallest terms first (2^-16: lo*hi, hi*lo, mid*mid; 2^-8: mid*hi, hi*mid; then hi*hi)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(a3[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(b3[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(al[a]), BF8(bl[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(al[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(bl[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
#undef BF8
                        } else if constexpr (ARITH == AR_BF16X3) {
#define BF8(v) __builtin_bit_cast(bf16x8, v)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(al[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(bl[b]), acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
#undef BF8
                        } else if constexpr (ARITH == AR_FP16X3) {
#define H8(v) __builtin_bit_cast(f16x8, v)
                            acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(al[a]), H8(bh[b]), acc2[a][b], 0, 0, 0);
                            acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah[a]), H8(bl[b]), acc2[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah[a]), H8(bh[b]), acc[a][b], 0, 0, 0);
#undef H8
                        }
                    }
            }
            return;
        }
        const float* sA = lds + buf * STAGE_FLOATS;
        const float* sB = sA + GA::LDS_FLOATS;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int xb = wm * WTM + a * 32 + li;
                if (A_RC) {
                    fa[a] = *reinterpret_cast<const f32x4*>(sA + xb * GA::LDS_LD + kb * 8 + 4 * lh);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[a][e] = sA[(kb * 8 + 4 * lh + e) * GA::LDS_LD + xb];
                }
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int xb = wn * WTN + b * 32 + li;
                if (B_RC) {
                    fb[b] = *reinterpret_cast<const f32x4*>(sB + xb * GB::LDS_LD + kb * 8 + 4 * lh);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fb[b][e] = sB[(kb * 8 + 4 * lh + e) * GB::LDS_LD + xb];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][e], fb[b][e], acc[a][b], 0, 0, 0);
        }
    };

    const int nk = (r_end - r_begin + BK - 1) / BK;
    constexpr bool PF1 = SPLIT && BM * BN >= 128 * 128;     // one staging set (register budget of the widest tile)
    if constexpr (PF1) {
        if (nk > 0) { load_stage(r_begin, stA0, stB0); write_stage(0, stA0, stB0); }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_stage(r_begin + (kt + 1) * BK, stA0, stB0);
            compute(kt & 1);
            if (kt + 1 < nk) write_stage((kt + 1) & 1, stA0, stB0);
            __syncthreads();
        }
    } else {
    if (nk > 0) {
        load_stage(r_begin, stA0, stB0);
        if (nk > 1) load_stage(r_begin + BK, stA1, stB1);
        write_stage(0, stA0, stB0);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; kt += 2) {
        // even step: tile kt is in LDS buffer 0, tile kt+1 is in flight in set 1
        if (kt + 2 < nk) load_stage(r_begin + (kt + 2) * BK, stA0, stB0);
        compute(0);
        if (kt + 1 < nk) write_stage(1, stA1, stB1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd step: tile kt+1 is in LDS buffer 1, tile kt+2 is in flight in set 0
        if (kt + 3 < nk) load_stage(r_begin + (kt + 3) * BK, stA1, stB1);
        compute(1);
        if (kt + 2 < nk) write_stage(0, stA0, stB0);
        __syncthreads();
    }
    }

    // ---------------- epilogue ----------------
    if constexpr (ARITH == AR_FP16X3) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] += acc2[a][b][e] * (1.f / LO_SCALE);
    }
    // split-K with ST_STORE (deterministic mode): slice zz_ stores its partial tile in its own slab, sc0 floats apart (fold: splitk_fold)
    gemm_epilogue<BM, BN, WM, WN, NTHREADS, 2 * STAGE_FLOATS>(acc, g, lds, tile_i, i0, j0, g.c + (int64_t)(sk_ ? zz_ : b0) * g.sc0 + (int64_t)b1 * g.sc1);
}

// ----------------------------------------------------------------------------------------
// host-side dispatch
// ----------------------------------------------------------------------------------------
struct TileChoice { int bm, bn; };

}  // namespace

namespace bdgemm {
int g_num_cus = 0;
int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cus = p.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}
}  // namespace bdgemm

namespace {

// Pick the biggest tile that still gives every CU ~2 workgroups; small-J problems get narrow tiles.
// BDETR_TILE=<bm>x<bn> (e.g. 128x64) forces a tile for tuning experiments.
// GEMM arithmetic policy (include/bdetr.h): BDETR_GEMM_PRECISION=fp32|bf16x3|mixed sets the initial value,
// bdetr_set_gemm_precision() changes it at run time.  use_split(grad) answers for one product.
}  // namespace
namespace bdgemm {
thread_local int g_gemm_mode = -1;     // per host thread (include/bdetr.h): the one mode word the library keeps
int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("BDETR_GEMM_PRECISION");
        g_gemm_mode = !e ? BDETR_GEMM_MIXED : (!strcmp(e, "fp32") || !strcmp(e, "f32")) ? BDETR_GEMM_FP32
                    : !strcmp(e, "bf16x3") ? BDETR_GEMM_BF16X3 : !strcmp(e, "split") ? BDETR_GEMM_SPLIT : !strcmp(e, "bf16x6") ? BDETR_GEMM_BF16X6 : BDETR_GEMM_MIXED;
    }
    return g_gemm_mode;
}
}  // namespace bdgemm
namespace {
int use_split(bool grad) {      // -> AR_* of one product
    switch (gemm_mode()) {
        case BDETR_GEMM_FP32:   return AR_FP32;
        case BDETR_GEMM_BF16X3: return AR_BF16X3;
        case BDETR_GEMM_SPLIT:  return grad ? AR_BF16X3 : AR_FP16X3;
        case BDETR_GEMM_BF16X6: return AR_BF16X6;
        default:                return grad ? AR_BF16X3 : AR_FP32;
    }
}

TileChoice choose_tile(int I, int J, int zdim, bool both_rc, bool split, bool patch = false) {
    static int forced_bm = -1, forced_bn = -1;
    if (forced_bm < 0) {
        forced_bm = forced_bn = 0;
        if (const char* e = getenv("BDETR_TILE")) sscanf(e, "%dx%d", &forced_bm, &forced_bn);
    }
    if (forced_bm > 0) {
        if (J <= 32) return {128, 32};
        return {forced_bm, forced_bn};
    }
    // Measured on MI355X (tools/gemm_bench.py): the 64x64 tile (4 workgroups per CU, 16 MFMAs per
    // wave between barriers) beats 128x128 / 128x64 on every shape of this model (M <= 409600,
    // N <= 2048): 97-108 vs 51-95 TF/s; 128x128 only wins once there are >= ~16 tiles per CU.
    auto tiles = [&](int bm, int bn) { return cdiv64(I, bm) * cdiv64(J, bn) * zdim; };
    if (J <= 32) return {128, 32};
    if (split) {
        // bf16x3 (tools/gemm_bench.py on MI355X): the per-element hi/lo split is VALU work that a wider tile
        // amortises over more MFMAs, so 128x128 wins 10-25 % on the r-contiguous x r-contiguous flavour
        // (conv/dense forward) once it still yields ~a workgroup per CU; the transposing flavours and the
        // small problems stay on 64x64.
        if (both_rc && J % 128 == 0 && tiles(128, 128) * 4 >= 3LL * num_cus()) return {128, 128};
        // 3x3 convolution gradients (im2col operand, R >= 1152): 128x128 is 15-25 % faster on stages 3-5
        // (bench.py per-launch dump); the weight gradient gets its parallelism from split-K, so its rule must
        // not depend on zdim (bdetr_conv2d_bwd_weight_splitk sizes z from this same choice)
        if (patch && !both_rc && I % 128 == 0 && J % 128 == 0 && (zdim > 1 || tiles(128, 128) * 4 >= 3LL * num_cus())) return {128, 128};
        return {64, 64};
    }
    if (tiles(128, 128) >= 16LL * num_cus() && J >= 1024) return {128, 128};
    return {64, 64};
}

// ---- optional live profiling of the MFMA kernel family (bench.py's roofline leg) -----------------
// When enabled, every igemm launch is bracketed by two hipEvents recorded on the launch stream; the
// host resolves the elapsed times after the timed region.  Events come from a pool that only grows.
struct ProfRec { hipEvent_t e0, e1; double flops; int I, J, R, z, bm, bn, kind; };
std::vector<ProfRec> g_prof_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
size_t g_prof_used = 0;

template <class L> struct LoaderId { static constexpr int v = 0; };
template <> struct LoaderId<PatchLoader> { static constexpr int v = 1; };
template <> struct LoaderId<WFlipLoader> { static constexpr int v = 2; };
template <> struct LoaderId<DenseLoader<1>> { static constexpr int v = 3; };

}  // namespace
namespace bdgemm {
bool g_prof_on = false;
void prof_begin(hipStream_t st, double flops, int I, int J, int R, int z, int bm, int bn, int kind) {
    if (g_prof_used == g_prof_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_prof_pool.emplace_back(a, b);
    }
    auto& ev = g_prof_pool[g_prof_used++];
    g_prof_recs.push_back({ev.first, ev.second, flops, I, J, R, z, bm, bn, kind});
    (void)hipEventRecord(ev.first, st);
}
void prof_end(hipStream_t st) { (void)hipEventRecord(g_prof_recs.back().e1, st); }
}  // namespace bdgemm
namespace {

template <int BM, int BN, int WM, int WN, class LA, bool A_RC, class LB, bool B_RC, int ARITH = AR_FP32>
int launch_cfg(const typename LA::Op& a, const typename LB::Op& b, GemmParams g, int zdim, hipStream_t st) {
    g.tiles_i = (int)cdiv64(g.I, BM);
    g.tiles_j = (int)cdiv64(g.J, BN);
    dim3 grid(g.tiles_i * g.tiles_j, 1, zdim);
    g.vec_store = (g.J % 4 == 0) && (g.ldc % 4 == 0) && (g.sc0 % 4 == 0) && (g.sc1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.c) & 15) == 0);
    const bool prof = g_prof_on;
    if (prof) prof_begin(st, 2.0 * (double)g.I * (double)g.J * (double)g.R * (g.splitk > 1 ? (g.ngroups > 0 ? (double)g.ngroups : 1.0) : (double)zdim), g.I, g.J, g.R, zdim, BM, BN,
                         ARITH * 10000 + LoaderId<LA>::v * 1000 + (A_RC ? 100 : 0) + LoaderId<LB>::v * 10 + (B_RC ? 1 : 0));
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, LA, A_RC, LB, B_RC, ARITH>), grid, dim3(NTHREADS), 0, st, a, b, g);
    if (prof) prof_end(st);
    return bdetr_launch_status("igemm");
}

template <class LA, bool A_RC, class LB, bool B_RC>
int launch_any(const typename LA::Op& a, const typename LB::Op& b, GemmParams g, int zdim, hipStream_t st, int arith, bool small_only = false) {
    if (arith == AR_FP16X3 && !(A_RC && B_RC)) arith = AR_FP32;     // the fp16 split is built for the forward flavour only
    constexpr bool PATCH = LoaderId<LA>::v == 1 || LoaderId<LB>::v == 1;
    TileChoice t = small_only ? TileChoice{64, 64} : choose_tile(g.I, g.J, g.splitk > 1 ? 2 : zdim, A_RC && B_RC, arith != AR_FP32, PATCH);
    if constexpr (LoaderId<LA>::v != 3 && LoaderId<LB>::v != 3) {
        // 16-byte loaders only; the narrow 128x32 tile (J <= 32: tiny heads) stays on the fp32 kernel
        if (arith == AR_BF16X3 && t.bn >= 64) {
            if (t.bm == 128 && t.bn == 128) return launch_cfg<128, 128, 2, 2, LA, A_RC, LB, B_RC, AR_BF16X3>(a, b, g, zdim, st);
            if (t.bm == 128 && t.bn == 64)  return launch_cfg<128, 64, 2, 2, LA, A_RC, LB, B_RC, AR_BF16X3>(a, b, g, zdim, st);
            return launch_cfg<64, 64, 2, 2, LA, A_RC, LB, B_RC, AR_BF16X3>(a, b, g, zdim, st);
        }
        if (arith == AR_BF16X6 && t.bn >= 64) {
            // (no 128x128: two stages of three planes are 106 KB of LDS - one workgroup per CU)
            if (t.bm == 128) return launch_cfg<128, 64, 2, 2, LA, A_RC, LB, B_RC, AR_BF16X6>(a, b, g, zdim, st);
            return launch_cfg<64, 64, 2, 2, LA, A_RC, LB, B_RC, AR_BF16X6>(a, b, g, zdim, st);
        }
        if constexpr (A_RC && B_RC) {
            if (arith == AR_FP16X3 && t.bn >= 64) {
                if (t.bm == 128 && t.bn == 128) return launch_cfg<128, 128, 2, 2, LA, A_RC, LB, B_RC, AR_FP16X3>(a, b, g, zdim, st);
                if (t.bm == 128 && t.bn == 64)  return launch_cfg<128, 64, 2, 2, LA, A_RC, LB, B_RC, AR_FP16X3>(a, b, g, zdim, st);
                return launch_cfg<64, 64, 2, 2, LA, A_RC, LB, B_RC, AR_FP16X3>(a, b, g, zdim, st);
            }
        }
    }
    if (t.bm == 128 && t.bn == 128) return launch_cfg<128, 128, 2, 2, LA, A_RC, LB, B_RC>(a, b, g, zdim, st);
    if (t.bm == 128 && t.bn == 64)  return launch_cfg<128, 64, 2, 2, LA, A_RC, LB, B_RC>(a, b, g, zdim, st);
    if (t.bm == 128 && t.bn == 32)  return launch_cfg<128, 32, 4, 1, LA, A_RC, LB, B_RC>(a, b, g, zdim, st);
    return launch_cfg<64, 64, 2, 2, LA, A_RC, LB, B_RC>(a, b, g, zdim, st);
}

// The loaders address an operand with 32-bit byte offsets from its base (raw buffer loads, NUM_RECORDS):
// one operand matrix / tensor must span less than 4 GB.  Larger batches shard over GPUs (or are split by the
// caller); outputs are addressed with 64-bit pointers and have no such limit.
constexpr int64_t MAX_OPERAND_ELEMS = (int64_t)(NUM_RECORDS / 4) - 64;
bool span_ok(int64_t elems) { return elems >= 0 && elems <= MAX_OPERAND_ELEMS; }
int64_t dense_span(const DenseOp& o) { return (int64_t)(o.rows - 1) * o.ld + o.cols; }

// number of partial-statistics rows the epilogue writes for an (I,J) problem: tiles_i * WM
int stat_chunks(int I, int J) {
    TileChoice t = choose_tile(I, J, 1, true, use_split(false) != AR_FP32);
    const int wm = (t.bm == 128 && t.bn == 32) ? 4 : 2;
    return (int)cdiv64(I, t.bm) * wm;
}

// dst[i] += ws[0][i] + ws[1][i] + ... in that order: the fixed-order fold of the deterministic split-K mode
__global__ __launch_bounds__(256) void splitk_fold_kernel(const float* __restrict__ ws, int zdim, int64_t slab, float* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + i);
        for (int z = 1; z < zdim; ++z) a += *reinterpret_cast<const f32x4*>(ws + (int64_t)z * slab + i);
        f32x4 d = *reinterpret_cast<f32x4*>(dst + i);
        d += a;
        *reinterpret_cast<f32x4*>(dst + i) = d;
    } else {
        for (int64_t k = i; k < n && k < i + 4; ++k) {
            float a = ws[k];
            for (int z = 1; z < zdim; ++z) a += ws[(int64_t)z * slab + k];
            dst[k] += a;
        }
    }
}

}  // namespace

namespace bdgemm {
int splitk_fold(const float* ws, int zdim, int64_t slab, float* dst, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(splitk_fold_kernel, dim3((unsigned)cdiv64(cdiv64(n, 4), 256)), dim3(256), 0, st, ws, zdim, slab, dst, n);
    return bdetr_launch_status("splitk_fold");
}
}  // namespace bdgemm

extern "C" int64_t bdetr_splitk_workspace_elems(int64_t I, int64_t J, int splitk) {
    return splitk > 1 ? (int64_t)splitk * bdgemm::splitk_slab(I * J) : 0;
}

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
extern "C" int bdetr_device_cus(void) { return num_cus(); }

extern "C" int bdetr_set_gemm_precision(int mode) {
    BDETR_CHECK_ARG(mode >= BDETR_GEMM_FP32 && mode <= BDETR_GEMM_BF16X6, "bdetr_set_gemm_precision: unknown mode %d", mode);
    bdgemm::g_gemm_mode = mode;
    return 0;
}
extern "C" int bdetr_get_gemm_precision(void) { return gemm_mode(); }

extern "C" int bdetr_prof_enable(int on) {
    g_prof_on = on != 0;
    if (on) { g_prof_recs.clear(); g_prof_used = 0; }
    return 0;
}
// Resolves all recorded launches (the caller must have synchronised the stream): total kernel time
// in ms, number of launches and the algorithmic FLOPs they performed (2*I*J*R per GEMM).
// debug aid: one CSV row per recorded launch (I,J,R,zdim,BM,BN,kind,ms,gflop); kind = LA*1000 + A_RC*100 + LB*10 + B_RC
extern "C" int bdetr_prof_dump(const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) { bdetr_set_error("bdetr_prof_dump: cannot open %s", path); return -1; }
    fprintf(f, "I,J,R,z,bm,bn,kind,ms,gflop\n");
    for (auto& r : g_prof_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) t = -1.f;
        fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.5f,%.4f\n", r.I, r.J, r.R, r.z, r.bm, r.bn, r.kind, t, r.flops * 1e-9);
    }
    fclose(f);
    return 0;
}

static int prof_sum(int arith, double* total_ms, int64_t* launches, double* flops) {
    double ms = 0, fl = 0; int64_t n = 0;
    for (auto& r : g_prof_recs) {
        if (arith >= 0 && r.kind / 10000 != arith) continue;
        float t = 0.f;
        hipError_t e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) { bdetr_set_error("bdetr_prof_read: %s", hipGetErrorString(e)); return (int)e; }
        ms += t; fl += r.flops; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    return 0;
}
extern "C" int bdetr_prof_read(double* total_ms, int64_t* launches, double* flops) { return prof_sum(-1, total_ms, launches, flops); }
extern "C" int bdetr_prof_read_arith(int arith, double* total_ms, int64_t* launches, double* flops) {
    BDETR_CHECK_ARG(arith >= AR_FP32 && arith <= AR_BF16X6, "bdetr_prof_read_arith: arith must be 0 (fp32), 1 (bf16x3), 2 (fp16x3), 3 (pre-split f16), 4 (pre-split bf16) or 5 (bf16x6)");
    return prof_sum(arith, total_ms, launches, flops);
}

// Up to 4 independent dense GEMMs with the same J, R, operand flavours and epilogue in ONE launch
// (gridDim.z = n): the Q/K/V projections of an attention block and their input gradients.
extern "C" int bdetr_gemm_grouped(const bdetr_gemm_desc* d, int n, void* stream) {
    BDETR_CHECK_ARG(d && n >= 1 && n <= 4, "bdetr_gemm_grouped: 1..4 problems");
    GemmParams g; init_params(g);
    int maxI = 0;
    bool v4 = true;
    for (int k = 0; k < n; ++k) {
        const bdetr_gemm_desc& e = d[k];
        BDETR_CHECK_ARG(e.a && e.b && e.c && e.I > 0, "bdetr_gemm_grouped: null pointer / empty problem %d", k);
        BDETR_CHECK_ARG(e.J == d[0].J && e.R == d[0].R && e.a_rcontig == d[0].a_rcontig && e.b_rcontig == d[0].b_rcontig &&
                        e.act == d[0].act && e.alpha == d[0].alpha && e.accumulate == d[0].accumulate &&
                        e.lda == d[0].lda && e.ldb == d[0].ldb && e.ldc == d[0].ldc,
                        "bdetr_gemm_grouped: problems must share J, R, leading dimensions, flavours and epilogue");
        BDETR_CHECK_ARG(e.grad == d[0].grad, "bdetr_gemm_grouped: problems must share the grad flag");
        BDETR_CHECK_ARG((e.nb0 <= 1) && (e.nb1 <= 1) && e.splitk == d[0].splitk, "bdetr_gemm_grouped: no batching inside a group; one split-K factor for all problems");
        BDETR_CHECK_ARG(e.splitk <= 1 || (!e.bias && e.act == 0 && e.I == d[0].I), "bdetr_gemm_grouped: split-K problems are plain GEMMs of one shape (C += partial products, atomics)");
        BDETR_CHECK_ARG(span_ok((int64_t)(e.a_rcontig ? e.I : e.R) * e.lda) && span_ok((int64_t)(e.b_rcontig ? e.J : e.R) * e.ldb),
                        "bdetr_gemm_grouped: operand %d spans 4 GB or more (32-bit buffer offsets)", k);
        g.ga[k] = e.a; g.gb[k] = e.b; g.gc[k] = e.c; g.gbias[k] = e.bias; g.gI[k] = e.I;
        if (e.I > maxI) maxI = e.I;
        v4 = v4 && aligned16(e.a) && aligned16(e.b) && aligned16(e.c);
    }
    const bdetr_gemm_desc& f = d[0];
    // 16-byte loads run along an operand's contiguous index: r for an r-contiguous operand, i / j otherwise (a weight gradient's
    // reduction index - the token count - is then free)
    bool dims4 = f.J % 4 == 0 && (f.a_rcontig || f.b_rcontig ? f.R % 4 == 0 : true);
    for (int k = 0; k < n; ++k) dims4 = dims4 && (f.a_rcontig || d[k].I % 4 == 0);
    BDETR_CHECK_ARG(v4 && f.lda % 4 == 0 && f.ldb % 4 == 0 && f.ldc % 4 == 0 && dims4,
                    "bdetr_gemm_grouped: operands must be 16-byte aligned with contiguous dimensions that are multiples of 4");
    g.ngroups = n;
    g.I = maxI; g.J = f.J; g.R = f.R;
    g.c = f.c; g.ldc = f.ldc; g.bias = f.bias; g.alpha = f.alpha; g.act = f.act;
    g.mode = f.accumulate ? ST_ACCUM : ST_STORE;
    int zdim = n;
    if (f.splitk > 1) {
        // every problem's r range cut into the same slices; the slices add into C with float atomics (C zeroed by the caller)
        g.r_chunk = (int)(cdiv64(cdiv64(f.R, f.splitk), BK) * BK);
        g.splitk = (int)cdiv64(f.R, g.r_chunk);
        g.mode = ST_ATOMIC;
        zdim = n * g.splitk;
    }
    DenseOp a{f.a, f.lda, 0, 0, f.a_rcontig ? maxI : f.R, f.a_rcontig ? f.R : maxI};
    DenseOp b{f.b, f.ldb, 0, 0, f.b_rcontig ? f.J : f.R, f.b_rcontig ? f.R : f.J};
    hipStream_t st = (hipStream_t)stream;
    const bool arc = f.a_rcontig != 0, brc = f.b_rcontig != 0;
    const int sp = use_split(f.grad != 0);
    if (arc && brc)   return launch_any<DenseLoader<4>, true, DenseLoader<4>, true>(a, b, g, zdim, st, sp);
    if (arc && !brc)  return launch_any<DenseLoader<4>, true, DenseLoader<4>, false>(a, b, g, zdim, st, sp);
    if (!arc && !brc) return launch_any<DenseLoader<4>, false, DenseLoader<4>, false>(a, b, g, zdim, st, sp);
    return launch_any<DenseLoader<4>, false, DenseLoader<4>, true>(a, b, g, zdim, st, sp, true);
}

extern "C" int bdetr_gemm(const bdetr_gemm_desc* d, void* stream) { return bdetr_gemm_ws(d, nullptr, 0, stream); }

template <class F>
static int with_splitk_slabs(GemmParams& g, int zdim, float* ws, int64_t ws_elems, hipStream_t st, const char* who, F&& launch) {
    // deterministic split-K: the slices store into slabs of the caller's workspace, a second launch folds them into C in order
    float* c = g.c;
    const int64_t n = (int64_t)g.I * g.J, slab = splitk_slab(n);
    BDETR_CHECK_ARG(g.ldc == g.J, "%s: the split-K workspace form needs a dense C (ldc == J)", who);
    BDETR_CHECK_ARG(aligned16(ws) && ws_elems >= (int64_t)zdim * slab, "%s: workspace too small or misaligned (%lld floats, need %lld)", who,
                    (long long)ws_elems, (long long)((int64_t)zdim * slab));
    g.c = ws; g.sc0 = slab; g.mode = ST_STORE;
    if (int e = launch()) return e;
    return splitk_fold(ws, zdim, slab, c, n, st);
}

extern "C" int bdetr_gemm_ws(const bdetr_gemm_desc* d, float* ws, int64_t ws_elems, void* stream) {
    BDETR_CHECK_ARG(d && d->a && d->b && d->c, "bdetr_gemm: null pointer");
    BDETR_CHECK_ARG(d->I > 0 && d->J > 0 && d->R >= 0, "bdetr_gemm: bad shape I=%d J=%d R=%d", d->I, d->J, d->R);
    const int nb0 = d->nb0 > 0 ? d->nb0 : 1, nb1 = d->nb1 > 0 ? d->nb1 : 1;
    const int splitk = d->splitk > 1 ? d->splitk : 1;
    BDETR_CHECK_ARG(splitk == 1 || (nb0 * nb1 == 1 && !d->bias && d->act == 0), "bdetr_gemm: splitk needs a single plain GEMM");
    hipStream_t st = (hipStream_t)stream;

    DenseOp a{d->a, d->lda, d->sa0, d->sa1, d->a_rcontig ? d->I : d->R, d->a_rcontig ? d->R : d->I};
    DenseOp b{d->b, d->ldb, d->sb0, d->sb1, d->b_rcontig ? d->J : d->R, d->b_rcontig ? d->R : d->J};
    BDETR_CHECK_ARG(span_ok(dense_span(a)) && span_ok(dense_span(b)),
                    "bdetr_gemm: an operand matrix spans 4 GB or more (32-bit buffer offsets); split the problem");
    GemmParams g; init_params(g);
    g.I = d->I; g.J = d->J; g.R = d->R; g.nb1 = nb1;
    g.c = d->c; g.ldc = d->ldc; g.sc0 = d->sc0; g.sc1 = d->sc1;
    g.bias = d->bias; g.alpha = d->alpha; g.act = d->act;
    g.mode = splitk > 1 ? ST_ATOMIC : (d->accumulate ? ST_ACCUM : ST_STORE);
    int zdim = nb0 * nb1;
    if (splitk > 1) {
        g.splitk = splitk;
        g.r_chunk = (int)(cdiv64(cdiv64(d->R, splitk), BK) * BK);
        zdim = (int)cdiv64(d->R, g.r_chunk);
        g.splitk = zdim > 1 ? zdim : 2;   // >1 keeps the split code path (r_begin/r_end)
    }
    auto vec_ok = [&](const DenseOp& o) {
        return aligned16(o.p) && o.ld % 4 == 0 && o.s0 % 4 == 0 && o.s1 % 4 == 0 && o.cols % 4 == 0;
    };
    const bool v4 = vec_ok(a) && vec_ok(b);
    const bool arc = d->a_rcontig != 0, brc = d->b_rcontig != 0;
    const int sp = use_split(d->grad != 0);
    auto launch = [&]() -> int {
        if (v4) {
            if (arc && brc)   return launch_any<DenseLoader<4>, true, DenseLoader<4>, true>(a, b, g, zdim, st, sp);
            if (arc && !brc)  return launch_any<DenseLoader<4>, true, DenseLoader<4>, false>(a, b, g, zdim, st, sp);
            if (!arc && !brc) return launch_any<DenseLoader<4>, false, DenseLoader<4>, false>(a, b, g, zdim, st, sp);
            return launch_any<DenseLoader<4>, false, DenseLoader<4>, true>(a, b, g, zdim, st, sp, true);
        }
        // unaligned fallback (scalar HBM loads): only tiny problems take it (82-wide heads, T=49 attention)
        if (arc && brc)   return launch_any<DenseLoader<1>, true, DenseLoader<1>, true>(a, b, g, zdim, st, AR_FP32, true);
        if (arc && !brc)  return launch_any<DenseLoader<1>, true, DenseLoader<1>, false>(a, b, g, zdim, st, AR_FP32, true);
        if (!arc && !brc) return launch_any<DenseLoader<1>, false, DenseLoader<1>, false>(a, b, g, zdim, st, AR_FP32, true);
        return launch_any<DenseLoader<1>, false, DenseLoader<1>, true>(a, b, g, zdim, st, AR_FP32, true);
    };
    if (ws != nullptr && splitk > 1) return with_splitk_slabs(g, zdim, ws, ws_elems, st, "bdetr_gemm_ws", launch);
    return launch();
}

static int check_conv(const bdetr_conv_desc* d, const char* who) {
    BDETR_CHECK_ARG(d != nullptr, "%s: null desc", who);
    BDETR_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->stride > 0 && d->pad >= 0,
                    "%s: bad conv geometry", who);
    BDETR_CHECK_ARG(d->C % 4 == 0, "%s: input channels must be a multiple of 4 (got %d)", who, d->C);
    BDETR_CHECK_ARG(d->OH == (d->H + 2 * d->pad - d->R) / d->stride + 1 && d->OW == (d->W + 2 * d->pad - d->S) / d->stride + 1,
                    "%s: OH/OW inconsistent with geometry", who);
    BDETR_CHECK_ARG((int64_t)d->N * d->OH * d->OW < (1LL << 31) && (int64_t)d->R * d->S * d->C < (1LL << 31), "%s: problem too large", who);
    BDETR_CHECK_ARG(span_ok((int64_t)d->N * d->H * d->W * d->C) && span_ok((int64_t)d->N * d->OH * d->OW * d->K) && span_ok((int64_t)d->K * d->R * d->S * d->C),
                    "%s: a tensor spans 4 GB or more (the loaders use 32-bit buffer offsets); use a smaller per-GPU batch", who);
    return 0;
}

extern "C" int bdetr_conv2d_fwd_stat_chunks(const bdetr_conv_desc* d) {
    if (check_conv(d, "bdetr_conv2d_fwd_stat_chunks")) return -1;
    return stat_chunks(d->N * d->OH * d->OW, d->K);
}

extern "C" int bdetr_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                const bdetr_conv_desc* d, int act, float* stat_sum, float* stat_sq, void* stream) {
    if (int e = check_conv(d, "bdetr_conv2d_fwd")) return e;
    BDETR_CHECK_ARG(x && w && y, "bdetr_conv2d_fwd: null pointer");
    BDETR_CHECK_ARG(aligned16(x) && aligned16(w), "bdetr_conv2d_fwd: x and w must be 16-byte aligned");
    BDETR_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "bdetr_conv2d_fwd: stat_sum/stat_sq must both be set or both null");
    const int M = d->N * d->OH * d->OW, Kd = d->R * d->S * d->C;
    GemmParams g; init_params(g);
    g.I = M; g.J = d->K; g.R = Kd;
    g.c = y; g.ldc = d->K; g.bias = bias; g.act = act;
    g.stat_sum = stat_sum; g.stat_sq = stat_sq;
    DenseOp wop{w, Kd, 0, 0, d->K, Kd};
    hipStream_t st = (hipStream_t)stream;
    if (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0) {
        DenseOp xop{x, d->C, 0, 0, M, d->C};
        return launch_any<DenseLoader<4>, true, DenseLoader<4>, true>(xop, wop, g, 1, st, use_split(false));
    }
    PatchOp xop{x, d->N, d->H, d->W, d->C, d->OH, d->OW, d->R, d->S, d->stride, d->pad, M, Kd};
    return launch_any<PatchLoader, true, DenseLoader<4>, true>(xop, wop, g, 1, st, use_split(false));
}

extern "C" int bdetr_conv2d_bwd_data(const float* dy, const float* w, float* dx,
                                     const bdetr_conv_desc* d, int accumulate, void* stream) {
    if (int e = check_conv(d, "bdetr_conv2d_bwd_data")) return e;
    BDETR_CHECK_ARG(dy && w && dx, "bdetr_conv2d_bwd_data: null pointer");
    BDETR_CHECK_ARG(d->K % 4 == 0, "bdetr_conv2d_bwd_data: output channels must be a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const int M = d->N * d->OH * d->OW;
    GemmParams g; init_params(g);
    g.c = dx; g.ldc = d->C; g.mode = accumulate ? ST_ACCUM : ST_STORE;
    if (d->R == 1 && d->S == 1 && d->pad == 0) {
        // dx[m][c] = sum_k dy[m][k] * w[k][c]; strided convs scatter their rows into the strided pixels
        g.I = M; g.J = d->C; g.R = d->K;
        if (d->stride > 1) {
            if (!accumulate) {
                if (int e = bdetr_zero_bytes(dx, sizeof(float) * (size_t)d->N * d->H * d->W * d->C, st)) return e;
            }
            g.rowmap = 1; g.rm_OW = d->OW; g.rm_OHOW = d->OH * d->OW; g.rm_H = d->H; g.rm_W = d->W; g.rm_stride = d->stride;
        }
        DenseOp a{dy, d->K, 0, 0, M, d->K};
        DenseOp b{w, d->C, 0, 0, d->K, d->C};
        return launch_any<DenseLoader<4>, true, DenseLoader<4>, false>(a, b, g, 1, st, use_split(true));
    }
    BDETR_CHECK_ARG(d->stride == 1, "bdetr_conv2d_bwd_data: stride>1 only for 1x1 convs");
    // dx[n,ih,iw,c] = sum_{r,s,k} dy[n, ih+pad-r, iw+pad-s, k] * w[k][r][s][c]
    //              = patch gather over dy with pad' = R-1-pad and flipped taps
    const int Mx = d->N * d->H * d->W;
    g.I = Mx; g.J = d->C; g.R = d->R * d->S * d->K;
    PatchOp a{dy, d->N, d->OH, d->OW, d->K, d->H, d->W, d->R, d->S, 1, d->R - 1 - d->pad, Mx, d->R * d->S * d->K};
    BDETR_CHECK_ARG(d->R == d->S, "bdetr_conv2d_bwd_data: square kernels only");
    WFlipOp b{w, d->K, d->C, d->R, d->S, d->R * d->S * d->K, d->C};
    return launch_any<PatchLoader, true, WFlipLoader, false>(a, b, g, 1, st, use_split(true));
}

extern "C" int bdetr_conv2d_bwd_weight_splitk(const bdetr_conv_desc* d) {
    if (check_conv(d, "bdetr_conv2d_bwd_weight_splitk")) return -1;
    const int M = d->N * d->OH * d->OW, Kd = d->R * d->S * d->C;
    TileChoice t = choose_tile(d->K, Kd, 2, false, use_split(true) != AR_FP32, !(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0));
    int64_t tiles = cdiv64(d->K, t.bm) * cdiv64(Kd, t.bn);
    int64_t want = 3LL * num_cus();
    // A handful of output tiles over a very long pixel range - the stem's 64 x 196 gradient over 1.6 M pixels, the LAST kernel of the
    // backward pass, alone on the chip with the optimizer waiting for it: 16 workgroups per CU instead of 3 (tools/stem_wgrad_probe.py,
    // round 5: split 192 -> 467 us, 1024 -> 355 us)
    const bool few_tiles = tiles <= 8 && M >= (1 << 20);      // (the pixel-count bound keeps the stage-2 layers of the igemm backward policies - bf16x6, fp32: 0.4 M pixels, 1-4 tiles - on their 3 workgroups per CU: 1,024 slices of 64-KB atomics each cost value_fp32_grade 22 %)
    if (few_tiles) want = 16LL * num_cus();
    int64_t sk = cdiv64(want, tiles);
    int64_t maxsk = cdiv64(M, 4 * BK);      // keep >= 4 stages per split
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
    if (sk > (few_tiles ? 1024 : 512)) sk = few_tiles ? 1024 : 512;
    return (int)sk;
}

extern "C" int bdetr_conv2d_bwd_weight(const float* x, const float* dy, float* dw,
                                       const bdetr_conv_desc* d, int splitk, void* stream) {
    return bdetr_conv2d_bwd_weight_ws(x, dy, dw, d, splitk, nullptr, 0, stream);
}
extern "C" int bdetr_conv2d_bwd_weight_ws(const float* x, const float* dy, float* dw,
                                          const bdetr_conv_desc* d, int splitk, float* ws, int64_t ws_elems, void* stream) {
    if (int e = check_conv(d, "bdetr_conv2d_bwd_weight")) return e;
    BDETR_CHECK_ARG(x && dy && dw, "bdetr_conv2d_bwd_weight: null pointer");
    BDETR_CHECK_ARG(d->K % 4 == 0, "bdetr_conv2d_bwd_weight: output channels must be a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const int M = d->N * d->OH * d->OW, Kd = d->R * d->S * d->C;
    if (splitk <= 0) splitk = bdetr_conv2d_bwd_weight_splitk(d);
    GemmParams g; init_params(g);
    g.I = d->K; g.J = Kd; g.R = M;
    g.c = dw; g.ldc = Kd;
    int zdim = 1;
    if (splitk > 1) {
        g.r_chunk = (int)(cdiv64(cdiv64(M, splitk), BK) * BK);
        zdim = (int)cdiv64(M, g.r_chunk);
        g.splitk = zdim > 1 ? zdim : 2;
        g.mode = ST_ATOMIC;
    }
    DenseOp a{dy, d->K, 0, 0, M, d->K};           // rows = r (pixels), cols = i (k)
    auto launch = [&]() -> int {
        if (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0) {
            DenseOp b{x, d->C, 0, 0, M, d->C};
            return launch_any<DenseLoader<4>, false, DenseLoader<4>, false>(a, b, g, zdim, st, use_split(true));
        }
        PatchOp b{x, d->N, d->H, d->W, d->C, d->OH, d->OW, d->R, d->S, d->stride, d->pad, M, Kd};
        return launch_any<DenseLoader<4>, false, PatchLoader, false>(a, b, g, zdim, st, use_split(true));
    };
    if (ws != nullptr && splitk > 1) return with_splitk_slabs(g, zdim, ws, ws_elems, st, "bdetr_conv2d_bwd_weight_ws", launch);
    return launch();
}
