// sgemm.hip - MFMA GEMM / implicit-GEMM family over PRE-SPLIT operands for gfx950 (MI355X).
//
// igemm.hip reads fp32 operands, splits every element into hi/lo 16-bit halves in registers and writes the
// halves to LDS: the global -> VGPR -> split VALU -> ds_write staging path, not the MFMA pipe, bounds it
// (profiles/README.md, round 1).  Here the PRODUCER of a tensor (BatchNorm apply / BatchNorm backward /
// the weight pack that follows the optimizer) writes the halves once, in the "P16" layout, and this kernel
// moves them HBM -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no split arithmetic, no
// ds_write) and feeds three v_mfma_f32_32x32x16_{f16,bf16} per product exactly like igemm.hip's split modes.
//
// P16 layout of a row-major [rows][C] matrix (C % 8 == 0), 4 bytes per element like fp32: every group of 8
// consecutive elements is 32 bytes = [8 x hi (16 bit)] [8 x lo (16 bit)].
//   f16 pair  (forward operands):   hi = f16(x),  lo = f16((x - hi) * 2^11)   (22 significant bits, |x| < 65504)
//   bf16 pair (gradient operands):  hi = bf16(x), lo = bf16(x - hi)           (16 significant bits, fp32 range)
// A 16-byte chunk is therefore 8 reduction elements of one half: exactly one lane's MFMA operand fragment when
// the reduction index is the contiguous one ("RR": conv forward / backward-data, ds_read_b128), and a
// 4-rows x 16-columns block for ds_read_b64_tr_b16 when it is the strided one ("XX": weight gradients, where the
// reduction runs over pixels).
//
// Replaces: Keras Conv2D and its autodiff inside tf.keras.applications ResNet-50 (backbone.py:37-38,57).
#include "gemm_common.h"
#include "p16.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

using namespace bdgemm;

namespace {

constexpr int BK = 32;                          // reduction depth of one LDS stage
constexpr unsigned OOB = 0xFFFFFFF0u;           // byte offset beyond num_records: the load writes zeros
constexpr unsigned NUM_RECORDS = 0xFFFFFF00u;   // operands must span < 4 GB (checked on the host)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned records = NUM_RECORDS) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane((int)records), 0x00020000);
}

// ----------------------------------------------------------------------------------------
// operands
// ----------------------------------------------------------------------------------------
// dense P16 matrix: RR: rows = x (I or J), cols = r.   XX: rows = r, cols = x.
struct PDense { const void* p; unsigned ld; int rows, cols; };
// im2col view of an NHWC P16 tensor: rows = output pixels (N*OH*OW), cols = (tap, channel) = R*S*C.
// The k* fields are the byte increments of the XX loader's running pixel offset (host-computed, wrapping uint32).
struct PPatch {
    const void* p; int N, H, W, C, OH, OW, R, S, stride, pad; int rows, cols;
    int d_ow, d_oh;                   // BK = d_oh * OW + d_ow: how far one K-step moves (oh, ow)
    unsigned k_step, k_wrap_ow, k_wrap_oh;
    int single_wrap = 0;              // d_oh < OH: a K-step crosses at most one image boundary (XXPatch::advance without its loop)
};

// All byte offsets are uint32 with wrapping arithmetic: operands span < 4 GB, intermediate sums of a padding pixel
// may "underflow" but are only used when the bounds test passed.
//
// ---- RR: the reduction index is the contiguous one; a lane loads 16 bytes = 8 r of one half of ONE row ----
// A lane fetches the same 16-byte chunk q of its row's 128-byte K-step segment in every load of a stage (q >> 1 = which
// 8 r, q & 1 = hi / lo; see the LDS image below), so q is folded into the row's byte offset once per tile.
struct RRDense {
    using Op = PDense;
    using Col = int;                                          // (XX-mode state: unused)
    // Rows past the end of the matrix lie beyond the buffer descriptor (num_records = rows * ld * 4): no flag
    struct Row { unsigned off; };
    struct Step {};                                           // per-K-step uniform state: none
    static __device__ __forceinline__ unsigned records(const Op& op) { return (unsigned)op.rows * op.ld * 4u; }
    static __device__ __forceinline__ Row row(const Op& op, int r, int q) { Row c; c.off = (unsigned)r * op.ld * 4u + 16u * (unsigned)q; return c; }
    static __device__ __forceinline__ Step step_init(const Op&, int) { return {}; }
    static __device__ __forceinline__ void step_next(const Op&, Step&) {}
    // klim = r_lim - 8 * (q >> 1): the chunk's first r must stay below it (the last K-step of a row may be partial)
    static __device__ __forceinline__ unsigned voff(const Op&, const Row& c, const Step&, int r0, int klim) {
        return r0 < klim ? c.off + (unsigned)r0 * 4u : OOB;
    }
};
struct RRPatch {
    using Op = PPatch;
    using Col = int;
    // base = byte offset of the row's tap-(0,0) input pixel plus the lane's chunk; mask bit (tr * S + ts) = that tap's input
    // pixel is inside the image (all clear for rows past the end).  The K-step's tap is uniform (Step), so an address
    // costs an and, a compare, an add and a select: no per-load index arithmetic.  C % 32 == 0: a K-step never straddles
    // taps or the end of the reduction.
    struct Row { unsigned base, mask; };
    struct Step { int tr, ts, c0; unsigned tapoff, tapbit; }; // tap (row, col), first channel of the K-step, their byte offset, the tap's mask bit
    static __device__ __forceinline__ unsigned records(const Op&) { return NUM_RECORDS; }
    static __device__ __forceinline__ Row row(const Op& op, int r, int q) {
        Row c;
        const int ohw = op.OH * op.OW;
        const int n = r / ohw, rem = r - n * ohw, oh = rem / op.OW, ow = rem - oh * op.OW;
        const int ih0 = oh * op.stride - op.pad, iw0 = ow * op.stride - op.pad;
        c.base = ((unsigned)(n * op.H * op.W) + (unsigned)(ih0 * op.W + iw0)) * (unsigned)op.C * 4u + 16u * (unsigned)q;
        c.mask = 0;
        if (r < op.rows) {
            unsigned colbits = 0;
            for (int ts = 0; ts < op.S; ++ts) colbits |= ((unsigned)(iw0 + ts) < (unsigned)op.W ? 1u : 0u) << ts;
            for (int tr = 0; tr < op.R; ++tr) if ((unsigned)(ih0 + tr) < (unsigned)op.H) c.mask |= colbits << (tr * op.S);
        }
        return c;
    }
    static __device__ __forceinline__ unsigned tap_bytes(const Op& op, const Step& s) { return ((unsigned)(s.tr * op.W + s.ts) * (unsigned)op.C + (unsigned)s.c0) * 4u; }
    static __device__ __forceinline__ Step step_init(const Op& op, int r0) {
        Step s; const int tap = r0 / op.C; s.c0 = r0 - tap * op.C; s.tr = tap / op.S; s.ts = tap - s.tr * op.S; s.tapoff = tap_bytes(op, s); s.tapbit = 1u << tap; return s;
    }
    static __device__ __forceinline__ void step_next(const Op& op, Step& s) {
        s.c0 += BK; s.tapoff += BK * 4u;
        if (s.c0 >= op.C) { s.c0 = 0; s.tapbit <<= 1; if (++s.ts == op.S) { s.ts = 0; ++s.tr; } s.tapoff = tap_bytes(op, s); }
    }
    static __device__ __forceinline__ unsigned voff(const Op&, const Row& c, const Step& s, int, int) {
        return (c.mask & s.tapbit) ? c.base + s.tapoff : OOB;
    }
};

// ---- XX: the reduction index is the strided one; a lane loads 16 bytes = 8 x-columns of one half of ONE r-row.
// A load's state (Col) is advanced by BK rows per K-step with adds only.
struct XXDense {
    using Op = PDense;
    using Row = int; using Step = int;                        // (RR-mode state: unused)
    static __device__ __forceinline__ unsigned records(const Op&) { return NUM_RECORDS; }
    struct Col { unsigned off; int r; bool ok; };             // running byte offset of (row r, the lane's chunk)
    // (r_uni: first row of the lane's wave-instruction - wave-uniform; lane_row: the lane's row inside it)
    static __device__ __forceinline__ Col col(const Op& op, int x0, int slot, int x_lim, int r_uni, int lane_row) {
        const int r_first = r_uni + lane_row;
        Col c; c.ok = x0 + 8 * (slot >> 1) < x_lim; c.r = r_first;
        c.off = (unsigned)r_first * op.ld * 4u + (unsigned)x0 * 4u + 16u * (unsigned)slot; return c;
    }
    template <int RPI> static __device__ __forceinline__ unsigned voff(const Op&, const Col& c, int r_lim) { return (c.ok && c.r < r_lim) ? c.off : OOB; }
    static __device__ __forceinline__ void advance(const Op& op, Col& c) { c.r += BK; c.off += (unsigned)BK * op.ld * 4u; }
};
struct XXPatch {
    using Op = PPatch;
    using Row = int; using Step = int;
    static __device__ __forceinline__ unsigned records(const Op&) { return NUM_RECORDS; }
    // (oh, ow) of the load's output pixel, (ih, iw) of the input pixel its tap reads, and the running byte offset of
    // that input pixel's channel chunk
    struct Col { int r, oh, ow, ih, iw; unsigned off; bool ok; };
    static __device__ __forceinline__ Col col(const Op& op, int x0, int slot, int x_lim, int r_uni, int lane_row) {
        const int r_first = r_uni + lane_row;
        Col c; const int x = x0 + 8 * (slot >> 1);
        c.ok = x < x_lim; c.r = r_first;
        const int tap = x / op.C, ch = x - tap * op.C;
        const int tr = tap / op.S, ts = tap - tr * op.S;
        const int ohw = op.OH * op.OW;
        const int n = r_first / ohw, rem = r_first - n * ohw;
        c.oh = rem / op.OW; c.ow = rem - c.oh * op.OW;
        c.ih = c.oh * op.stride - op.pad + tr; c.iw = c.ow * op.stride - op.pad + ts;
        c.off = ((unsigned)(n * op.H * op.W) + (unsigned)(c.ih * op.W + c.iw)) * (unsigned)op.C * 4u + (unsigned)ch * 4u + 16u * (unsigned)(slot & 1);
        return c;
    }
    template <int RPI> static __device__ __forceinline__ unsigned voff(const Op& op, const Col& c, int r_lim) {
        const bool ok = c.ok && c.r < r_lim && (unsigned)c.ih < (unsigned)op.H && (unsigned)c.iw < (unsigned)op.W;
        return ok ? c.off : OOB;
    }
    static __device__ __forceinline__ void advance(const Op& op, Col& c) {
        c.r += BK;
        c.ow += op.d_ow; c.oh += op.d_oh; c.iw += op.d_ow * op.stride; c.ih += op.d_oh * op.stride; c.off += op.k_step;
        // Selects, not branches: the per-lane `if` + `while` compiled to an exec-mask region and an inner loop per load - 100 of the
        // 146 VALU instructions and most of the 2 SALU instructions per MFMA of the 3x3 weight gradient's K-step (PMC, round 4)
        const bool w = c.ow >= op.OW;
        c.ow -= w ? op.OW : 0; c.oh += w ? 1 : 0; c.iw -= w ? op.OW * op.stride : 0; c.ih += w ? op.stride : 0; c.off += w ? op.k_wrap_ow : 0u;
        if (op.single_wrap) {                          // (uniform)
            const bool v = c.oh >= op.OH;
            c.oh -= v ? op.OH : 0; c.ih -= v ? op.OH * op.stride : 0; c.off += v ? op.k_wrap_oh : 0u;
        } else {
            while (c.oh >= op.OH) { c.oh -= op.OH; c.ih -= op.OH * op.stride; c.off += op.k_wrap_oh; }      // next image (several on maps smaller than a K-step)
        }
    }
};

// ----------------------------------------------------------------------------------------
// LDS images (one stage = A tile then B tile; 128 bytes per x-row per K-step in both modes)
//
// RR: tile[x][8 slots of 16 B]; slot s of row x holds source chunk s ^ ((x >> 1) & 7) of the row's 128-byte
//     K-step segment [hi0 lo0 hi1 lo1 hi2 lo2 hi3 lo3] - the XOR spreads the 16 rows of a ds_read_b128 lane
//     group over all 16 slots of the 256-byte bank row (conflict-free).
// XX: tile[r = 32 rows][BX/4 slots of 16 B]; slot s of row r holds source chunk s ^ swz(r),
//     swz(r) = ((r & 1) << 3) | ((r >> 1) & 1): the four rows a ds_read_b64_tr_b16 half-wave touches land in
//     four disjoint quarter bank rows (conflict-free).
// Loads write LDS linearly (wave-uniform base + lane * 16), so the swizzle is applied to the SOURCE chunk a
// lane fetches and, identically, to the slot a fragment read addresses.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int xx_swz(int r) { return ((r & 1) << 3) | ((r >> 1) & 1); }

#if defined(BDETR_SGEMM_STAMPS) && defined(BDETR_SGEMM_DIAG)
__device__ unsigned long long g_stamps[64];      // diagnostic cycle stamps (BDETR_SGEMM_DBG bit 32)
#endif

constexpr int lds_bytes_for(int bm, int bn, int ns) { return ns * (bm + bn) * 128 > bm * bn * 4 ? ns * (bm + bn) * 128 : bm * bn * 4; }

// Epilogue of the persistent RR kernels: the C tile goes from the accumulators straight to memory, one dword per lane
// (an accumulator register of a 32x32 MFMA block is two 128-byte row segments: full lines, the shape the guide measured at
// the plain-store rate).  No LDS is involved, so the staging ring is free for the NEXT tile's first stages while this
// runs.  The buffer descriptor covers exactly the tile's valid rows (rows past I are dropped by the buffer unit's range
// check), a lane's column is valid or not for the whole tile (an invalid one starts 2 GB out of range), and the row
// offset is one running VGPR: a store costs one add, and the second 32-column block rides the instruction's immediate.
enum { EPI_PLAIN = 0, EPI_ROWMAP = 1, EPI_VEC = 8, EPI_BNB2 = 16, EPI_COMPACT = 32 };      // EPI_COMPACT (with EPI_VEC): only the masked accumulate whose OLD gradient is a compact even-pixel tensor (GemmParams::acc_src)      // EPI_BNB2 (with EPI_VEC): only the masked accumulate + masked sums + second-BatchNorm sum form
template <int BM, int BN, int WM, int WN, bool STATS = true, int EPI = EPI_ROWMAP>
__device__ __forceinline__ void direct_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const GemmParams& g, int tile_i, int i0, int j0, const float* bias_pre) {
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr unsigned FAR = 0x80000000u;                    // + any in-tile offset (< 2^31) stays out of range
    if constexpr (STATS) gemm_bias_act_stats<BM, BN, WM, WN>(acc, g, tile_i, i0, j0, bias_pre);      // (false: the caller already did)
    // The thread index is laundered through an empty asm: everything derived from it here is then recomputed per tile
    // (a handful of VALU ops) instead of being hoisted out of the persistent loop and held in VGPRs across the K loop,
    // where the accumulators and fragments leave no room (the hoisted copies spilled to scratch).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

    const unsigned ldc4 = (unsigned)g.ldc * 4u;
    const int rows_here = min(BM, g.I - i0);
    const int jl = j0 + wn * WTN + li;                      // column of b = 0
    auto tile_rsrc = [&](const float* base) {                // rows i0 .. i0 + rows_here - 1 of a [I][ldc] fp32 matrix
        const unsigned long long p = reinterpret_cast<unsigned long long>(base) + (unsigned long long)i0 * ldc4;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(rows_here * (int)ldc4), 0x00020000);
    };
    // byte offset of (row of (a = 0, e = 0), column of block b) inside the tile
    unsigned vcol[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) vcol[b] = jl + b * 32 < g.J ? (unsigned)(wm * WTM + 4 * lh) * ldc4 + (unsigned)(jl + b * 32) * 4u : FAR;
    // walk the rows of the accumulator layout: (e & 3) + 8 * (e >> 2) + 32 * a
    auto for_rows = [&](auto&& f) {
        unsigned ro = 0;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) { f(a, e, ro); ro += (e & 3) == 3 ? 5u * ldc4 : ldc4; }
    };

    if constexpr ((EPI & EPI_ROWMAP) != 0) if (g.rowmap) {
        // strided scatter (1x1 stride-2 backward-data; 6 launches per step): lane l owns the row map of the wave's row l
        const __amdgpu_buffer_rsrc_t rsC = make_rsrc(g.c);
        const int i = i0 + wm * WTM + (lane % WTM);
        const int n = i / g.rm_OHOW, rem = i - n * g.rm_OHOW, oh = rem / g.rm_OW, ow = rem - oh * g.rm_OW;
        const unsigned rmap = i < g.I ? (unsigned)((n * g.rm_H + oh * g.rm_stride) * g.rm_W + ow * g.rm_stride) * ldc4 : OOB;
        const bool accum = g.mode == ST_ACCUM;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned ro = (unsigned)__shfl((int)rmap, a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, 64);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const unsigned vo = (ro != OOB && jl + b * 32 < g.J) ? ro + (unsigned)(jl + b * 32) * 4u : OOB;
                    float v = acc[a][b][e];
                    if (accum) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, vo, 0, 0));
                    if (!BDETR_DBG(g, 1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, vo, 0, 0);
                }
            }
        return;
    }

    const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(g.c);
    if (BDETR_DBG(g, 1)) return;
    for_rows([&](int a, int e, unsigned ro) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const float v = acc[a][b][e];           // (a scalar copy first: __builtin_bit_cast of a vector ELEMENT folds to element 0 on this hipcc)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, vcol[b] + ro, 0, 0);
        }
    });
}

// Epilogue of the split-K weight-gradient kernels: float atomics straight from the accumulators with the addressing of
// direct_epilogue (per-tile descriptor, running row offset, one add per atomic).  The generic per-element path spent ~85
// VALU instructions on 64-bit address arithmetic per atomic - more than the atomic unit's own issue interval.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void atomic_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const GemmParams& g, int i0, int j0) {
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr unsigned FAR = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const unsigned ldc4 = (unsigned)g.ldc * 4u;
    const int rows_here = min(BM, g.I - i0);
    const unsigned long long p = reinterpret_cast<unsigned long long>(g.c) + (unsigned long long)i0 * ldc4;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                                         __builtin_amdgcn_readfirstlane(rows_here * (int)ldc4), 0x00020000);
    unsigned vcol[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int j = j0 + wn * WTN + b * 32 + li;
        vcol[b] = j < g.J ? (unsigned)(wm * WTM + 4 * lh) * ldc4 + (unsigned)j * 4u : FAR;
    }
    unsigned ro = 0;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const float v = g.alpha * acc[a][b][e];          // (a scalar copy: see direct_epilogue)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rsC, vcol[b] + ro, 0, 0);
            }
            ro += (e & 3) == 3 ? 5u * ldc4 : ldc4;
            asm volatile("" : "+v"(ro));         // a running offset, not 32 precomputed ones (they spilled)
        }
}

// NS = LDS stages: the loads of K-step t + NS - 1 are issued while K-step t computes and a counted s_waitcnt
// vmcnt leaves the younger stages in flight across the (raw) barrier.
constexpr int default_occ(int bm, int bn, int wm, int wn, int ns) { return (wm * wn == 4 && 2 * lds_bytes_for(bm, bn, ns) <= 160 * 1024) ? 2 : (wm * wn == 8 ? 2 : 1); }

// PP ("ping-pong", 8 waves, three-stage ring): the two waves of a SIMD alternate roles in lock step - while waves 0-3 multiply
// K-step t from fragments already in registers, waves 4-7 read their fragments and issue their share of the LDS-DMA loads, then
// the halves swap (see the main loop).  A wave's MFMA segment contains nothing but MFMAs and its load segment runs in the shadow
// of its SIMD partner's MFMAs, instead of both partners stalling on the address unit at the same time.
// EPI (dense 1x1 kernels): EPI_PLAIN = persistent, register epilogue with bias / activation / statistics and plain stores only;
// EPI_ROWMAP = persistent, plus the strided scatter (with or without accumulate) of the stride-2 backward-data launches;
// EPI_VEC = NOT persistent, LDS-transposed float4 epilogue (gemm_epilogue) with its accumulate / masked accumulate / fused
// BatchNorm-backward-sum variants.  With every variant in the persistent kernel the variants' lane-derived invariants were hoisted
// over the tile loop and spilled (50 VGPRs, a scratch round trip per tile: 0.19 ms against 0.12 on the 64 -> 256 channel layer at
// 160x160), and the register epilogue's accumulate / sums - one dword per lane, up to 3 x 64 loads per lane and tile - ran at a
// third of their HBM floor (tools/dgrad_epi_probe.py).
template <int BM, int BN, int WM, int WN, int NS, class LA, class LB, bool XX, bool F16, bool PP = false, int OCC = default_occ(BM, BN, WM, WN, NS), int EPI = EPI_ROWMAP>
__global__ __launch_bounds__(WM * WN * 64, OCC)
void sgemm_kernel(typename LA::Op opa, typename LB::Op opb, GemmParams g)
{
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NIA = A_BYTES / 1024 / NW, NIB = B_BYTES / 1024 / NW;      // 1 KiB wave-instructions per wave per stage
    static_assert(NIA >= 1 && NIB >= 1 && NIA * NW * 1024 == A_BYTES && NIB * NW * 1024 == B_BYTES, "tile / wave count mismatch");
    constexpr int LDS_BYTES = lds_bytes_for(BM, BN, NS);
    static_assert(NS >= 2 && NS <= 4 && LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(BK == 32, "fragment helpers assume two 16-deep MFMA steps per stage");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware order over the WHOLE grid (tiles x split-K slices): workgroups are dealt round-robin to the 8 XCDs in
    // flattened launch order, and the tiles of one split-K slice read the same operand rows (the 9 taps of a 3x3
    // weight gradient, the row / column tiles of a 1x1 one).  Remapping the tiles of each slice alone spread a slice's
    // handful of tiles over all 8 L2s (measured: the 3x3 weight gradients fetched 13 GB per step for 1.4 GB of operands).
    //
    // The dense RR kernels (1x1 convolutions: short K, bound by the serial load -> MFMA -> store phases of a workgroup's
    // life) are PERSISTENT: the grid is min(tiles, resident workgroups) and workgroup w walks the virtual block ids
    // w, w + gridDim.x, ... (the same XCD every time: the resident count is a multiple of 8).  The patch kernels (3x3:
    // MFMA-bound, measured equal or slower when persistent, and the f16 128x128 one does not fit its accumulators plus
    // the loop-carried state in 256 VGPRs) and the XX kernels (split-K weight gradients) run one virtual block per
    // workgroup.
    constexpr bool PERSIST = !PP && !XX && std::is_same_v<LA, RRDense> && (EPI & EPI_VEC) == 0;
    static_assert(PP ? (NS == 3 && NW == 8) : NS == 2, "the plain loops are written for a two-stage ring, the ping-pong loop for three stages and 8 waves");
    const int nwg = g.tiles_i * g.tiles_j;
    const int total = PERSIST ? nwg : nwg * (int)gridDim.z;
    int vb = PERSIST ? (int)blockIdx.x : (int)(blockIdx.z * nwg + blockIdx.x);
    int tile_i, i0, j0, zz;
    auto locate = [&](int v) {
        const int lin = xcd_tile(v, total);
        zz = lin / nwg;
        const int wg = lin - zz * nwg;
        tile_i = wg / g.tiles_j;
        const int tile_j = wg - tile_i * g.tiles_j;
        i0 = tile_i * BM; j0 = tile_j * BN;
    };
    locate(vb);

    int r_begin = 0, r_end = g.R;
    if (g.splitk > 1) { r_begin = zz * g.r_chunk; r_end = min(g.R, r_begin + g.r_chunk); }

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(opa.p, LA::records(opa));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(opb.p, LB::records(opb));

    // ---------------- staging: HBM -> LDS, 16 bytes per lane, no registers ----------------
    typename LA::Row rowA[NIA]; typename LB::Row rowB[NIB]; typename LA::Col colA[NIA]; typename LB::Col colB[NIB];
    int rrA[NIA], rrB[NIB];                            // XX: row inside the stage
    // RR: the lane's source chunk.  Row x of the stage = (t * NW + wave) * 8 + (lane >> 3) is swizzled by (x >> 1) & 7 =
    // ((t * NW + wave) & 1) * 4 + (lane >> 4), the same for every load t when NW is even
    static_assert(XX || NW % 2 == 0, "one source chunk per lane needs an even wave count");
    const int qsrc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const int klimA = min(r_end, (int)opa.cols) - 8 * (qsrc >> 1), klimB = min(r_end, (int)opb.cols) - 8 * (qsrc >> 1);      // (RR)
    typename LA::Step stA; typename LB::Step stB;           // RR: per-K-step uniform state of the patch loaders (tap, first channel)
    auto setup_rows = [&]() {                               // RR: the row descriptors of tile (i0, j0)
        if constexpr (!XX) {
#pragma unroll
            for (int t = 0; t < NIA; ++t) {
                const int x = (t * NW + wave) * 8 + (lane >> 3);
                rowA[t] = LA::row(opa, i0 + x, qsrc);
            }
#pragma unroll
            for (int t = 0; t < NIB; ++t) {
                const int x = (t * NW + wave) * 8 + (lane >> 3);
                rowB[t] = LB::row(opb, j0 + x, qsrc);
            }
            stA = LA::step_init(opa, r_begin); stB = LB::step_init(opb, r_begin);
        }
    };
    setup_rows();
    if constexpr (XX) {
        constexpr int SA = BM / 4, SB = BN / 4;           // 16-byte slots per row
        static_assert(SA >= 16 && SB >= 16 && SA <= 64 && SB <= 64, "XX tiles: 64 <= BX <= 256");
#pragma unroll
        for (int t = 0; t < NIA; ++t) {
            const int ru = (t * NW + wave) * (64 / SA), r = ru + lane / SA;      // ru: wave-uniform first row of the wave-instruction
            rrA[t] = r;
            colA[t] = LA::col(opa, i0, (lane % SA) ^ xx_swz(r), opa.cols, r_begin + ru, lane / SA);
        }
#pragma unroll
        for (int t = 0; t < NIB; ++t) {
            const int ru = (t * NW + wave) * (64 / SB), r = ru + lane / SB;
            rrB[t] = r;
            colB[t] = LB::col(opb, j0, (lane % SB) ^ xx_swz(r), opb.cols, r_begin + ru, lane / SB);
        }
    }

    // one 1-KiB wave-instruction of the stage (l < NIA: A tile, else B tile)
    auto issue_one = [&](int buf, int r0, auto l_c) {
        constexpr int l = decltype(l_c)::value;
        unsigned char* sa = lds + buf * STAGE_BYTES;
        if constexpr (l < NIA) {
            constexpr int t = l;
            unsigned vo;
            if constexpr (!XX) vo = LA::voff(opa, rowA[t], stA, r0, klimA);
            else               vo = LA::template voff<256 / BM>(opa, colA[t], min(r_end, opa.rows));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(sa + (t * NW + wave) * 1024), 16, (int)vo, 0, 0, 0);
        } else {
            constexpr int t = l - NIA;
            unsigned vo;
            if constexpr (!XX) vo = LB::voff(opb, rowB[t], stB, r0, klimB);
            else               vo = LB::template voff<256 / BN>(opb, colB[t], min(r_end, opb.rows));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(sa + A_BYTES + (t * NW + wave) * 1024), 16, (int)vo, 0, 0, 0);
        }
    };
    constexpr int NLOAD = NIA + NIB;
    auto issue_all = [&](int buf, int r0) {
        if (BDETR_DBG(g, 16)) return;                    // diagnostic builds: no staging loads
        [&]<int... L>(std::integer_sequence<int, L...>) { (issue_one(buf, r0, std::integral_constant<int, L>{}), ...); }(std::make_integer_sequence<int, NLOAD>{});
        if constexpr (!XX) { LA::step_next(opa, stA); LB::step_next(opb, stB); }
        else {
#pragma unroll
            for (int t = 0; t < NIA; ++t) LA::advance(opa, colA[t]);
#pragma unroll
            for (int t = 0; t < NIB; ++t) LB::advance(opb, colB[t]);
        }
    };

    // ---------------- accumulators ----------------
    f32x16 acc[TM][TN];                       // ONE accumulator set for all three split products, f16 pairs too (p16.h)
    auto zero_acc = [&]() {
        // an opaque zero: otherwise the persistent loop keeps whole zero-filled 16-register tuples alive across the K loop
        // as the "constant" it re-initialises the accumulators from
        float z = 0.f;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = z;
    };
    zero_acc();

    // ---------------- fragment reads + MFMAs of one stage ----------------
    auto frag_rr = [&](const unsigned char* tile, int x, int ks, u32x4& hi, u32x4& lo) {
        // row x, k chunk kc = 2 * ks + lh: hi in slot (2 kc) ^ swz, lo in the neighbouring slot
        const int s = (2 * (2 * ks + lh)) ^ ((x >> 1) & 7);
        const unsigned char* p = tile + x * 128;
        hi = *reinterpret_cast<const u32x4*>(p + s * 16);
        lo = *reinterpret_cast<const u32x4*>(p + (s ^ 1) * 16);
    };
    // XX fragments: `ds_read_b64_tr_b16` as INLINE ASSEMBLY, with the s_waitcnt written by hand (frag_wait).  Round 5: through
    // __builtin_amdgcn_ds_read_tr16_b64_v4i16 the compiler's wait-count pass treated every such read as possibly aliasing the LDS-DMA
    // writes still in flight and put `s_waitcnt vmcnt(0)` in front of the first fragment read of a K-step - right behind the issue
    // of the NEXT stage's loads.  The two-stage ring was thereby a one-stage ring in every weight-gradient kernel: each K-step
    // waited out the round trip of the loads it had just issued (cycle stamps: 76 cycles waiting at the top of the step, 1,200-1,780
    // in the "MFMA" section; profiles/r05_wgrad_kstep_stamps.txt).  The plain `ds_read_b128` of the RR kernels carry a memory operand the
    // pass can reason about and never had the wait.  Ring safety is the barrier protocol's business, exactly as for the RR kernels:
    // the stage being read landed before the barrier at the top of the K-step, the stage in flight is the other buffer.
    auto frag_xx = [&](const unsigned char* tile, auto rowbytes_c, int xw, auto ks_c, u32x4& hi, u32x4& lo) {
        // ds_read_b64_tr_b16: lanes 16g .. 16g+15 read a 4-row x 16-column block, lane 4q+p supplies row q,
        // columns 4p .. 4p+3; lane i receives column i of the 4 rows.  Two reads (4 k each) per half.
        constexpr int rowbytes = decltype(rowbytes_c)::value, ks = decltype(ks_c)::value;
        const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
        const int unit = (xw + 16 * g16 + 4 * p) >> 3;                         // 8-column group
        const int r0 = 8 * lh + q;                                             // row of (ks = 0, t2 = 0); xx_swz reads bits 0-1 only: those of q
        const int sl = (2 * unit) ^ xx_swz(r0);
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)tile + (unsigned)(r0 * rowbytes + 8 * (p & 1));
        const unsigned ah = base + (unsigned)(sl * 16), al = base + (unsigned)((sl ^ 1) * 16);
        u32x2 h0, h1, l0, l1;
        constexpr int o0 = ks * 16 * rowbytes, o1 = o0 + 4 * rowbytes;
        static_assert(o1 < 65536, "ds offset field");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h0) : "v"(ah), "n"(o0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(l0) : "v"(al), "n"(o0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h1) : "v"(ah), "n"(o1));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(l1) : "v"(al), "n"(o1));
        hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
        lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    };
    // every fragment register of the K-step half passes through the wait: no use can be scheduled ahead of it
    auto frag_wait = [&](u32x4 (&ah)[TM], u32x4 (&al)[TM], u32x4 (&bh)[TN], u32x4 (&bl)[TN]) {
        if constexpr (XX) {
            static_assert(TM <= 2 && TN <= 2, "frag_wait ties at most 2 + 2 fragments");
            if constexpr (TM == 2 && TN == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
            else if constexpr (TM == 2 && TN == 1)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(bh[0]), "+v"(bl[0]));
            else if constexpr (TM == 1 && TN == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));
        }
    };
    // One K-step: the MFMAs of stage `buf` with the NEXT stage's loads issued between the MFMA groups, in program
    // order (LDS-DMA writes and ds_reads may alias as far as the compiler knows, so it keeps this order): a load's
    // issue cost (address VALU + M0 + buffer_load ... lds) then hides in the shadow of the preceding MFMAs instead of
    // running as a serial preamble in front of them.
    auto load_frags = [&](int buf, auto ks_c, u32x4 (&ah)[TM], u32x4 (&al)[TM], u32x4 (&bh)[TN], u32x4 (&bl)[TN]) {
        constexpr int ks = decltype(ks_c)::value;
        const unsigned char* tA = lds + buf * STAGE_BYTES;
        const unsigned char* tB = tA + A_BYTES;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if constexpr (!XX) frag_rr(tA, wm * WTM + a * 32 + li, ks, ah[a], al[a]);
            else               frag_xx(tA, std::integral_constant<int, BM * 4>{}, wm * WTM + a * 32, ks_c, ah[a], al[a]);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            if constexpr (!XX) frag_rr(tB, wn * WTN + b * 32 + li, ks, bh[b], bl[b]);
            else               frag_xx(tB, std::integral_constant<int, BN * 4>{}, wn * WTN + b * 32, ks_c, bh[b], bl[b]);
        }
    };
    auto mfma_ks = [&](const u32x4 (&ah)[TM], const u32x4 (&al)[TM], const u32x4 (&bh)[TN], const u32x4 (&bl)[TN]) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if constexpr (F16) {
#define H8(v) __builtin_bit_cast(f16x8, v)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(al[a]), H8(bh[b]), acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah[a]), H8(bl[b]), acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah[a]), H8(bh[b]), acc[a][b], 0, 0, 0);
#undef H8
                } else {
#define BF8(v) __builtin_bit_cast(bf16x8, v)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(al[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(bl[b]), acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[a]), BF8(bh[b]), acc[a][b], 0, 0, 0);
#undef BF8
                }
            }
    };
    // f16 pair -> bf16 pair of a fragment, element by element (the two 16-bit halves of a dword stay where they are): x = hi + lo is
    // exact in fp32 (22 bits), then the bf16 split of p16.h.  8 elements: 8 conversions up, 4 adds, 2 x (cvt_pk, 2 expands, 2 subs).
    auto f16pair_to_bf16pair = [&](u32x4& hi, u32x4& lo) {
        const unsigned hw[4] = {hi[0], hi[1], hi[2], hi[3]}, lw[4] = {lo[0], lo[1], lo[2], lo[3]};      // (scalars first: see p16_load8)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const p16_f32x2 hf = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, hw[w]), p16_f32x2);
            const p16_f32x2 lf = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, lw[w]), p16_f32x2);
            unsigned bh_, bl_;
            p16_split2_bf16(hf[0] + lf[0], hf[1] + lf[1], bh_, bl_);
            hi[w] = bh_; lo[w] = bl_;
        }
    };
    // One K-step of the plain loops: fragment reads of one 16-deep half, its MFMAs, then the other half
    auto kstep = [&](int buf) {
        if (BDETR_DBG(g, 8)) return;                     // diagnostic builds: no fragment reads / MFMAs
        auto half = [&](auto ks_c) {
            u32x4 ah[TM], al[TM], bh[TN], bl[TN];
            load_frags(buf, ks_c, ah, al, bh, bl);
            frag_wait(ah, al, bh, bl);
            if constexpr (XX && !F16) {
                if (g.b_f16) {                           // weight gradients reading the forward's f16 pair of x (see GemmParams::b_f16)
#pragma unroll
                    for (int b = 0; b < TN; ++b) f16pair_to_bf16pair(bh[b], bl[b]);
                }
            }
            mfma_ks(ah, al, bh, bl);
        };
        static_assert(BK / 16 == 2, "two 16-deep halves per stage");
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
    };

    // ---------------- main loop ----------------
    if constexpr (PP) {
        // Ping-pong over a three-stage ring.  Group X = waves 0-3, group Y = waves 4-7: wave w and wave w + 4 share a SIMD (a
        // workgroup's waves are dealt to the SIMDs cyclically), X owns the upper half of the tile's rows, Y the lower half, the B
        // tile is shared.  K-step t is two half-steps, each closed by ONE workgroup barrier:
        //   A(t): X multiplies step t (24 MFMAs, fragments already in registers) | Y reads its fragments of stage t and issues its
        //         share of stage t + 2's loads
        //   B(t): Y multiplies step t                                            | X reads its fragments of stage t + 1 and issues
        //         its share of stage t + 2's loads
        // Ring safety: stage t + 2 overwrites the buffer of stage t - 1, last read in A(t - 1) (by Y), one barrier before the first
        // issue.  Landing: X's share of stage s is issued in B(s - 2) and waited for (vmcnt(0)) at the end of A(s - 1); Y's share is
        // issued in A(s - 2) and waited for by the counted vmcnt(NLOAD) at the end of A(s - 1), which leaves Y's newest stage in
        // flight; the barrier that closes A(s - 1) precedes the first read of stage s (X in B(s - 1)).  Every wave issues every
        // stage exactly once, in order (the loaders' per-stage state advances once per issue_all).
        constexpr int NLOAD_ = NIA + NIB;
        const int nk = (r_end - r_begin + BK - 1) / BK;
        float bias_pre[TN];
        gemm_load_bias<BM, BN, WM, WN>(g, j0, bias_pre);              // (older than every stage load: retired by the first counted wait)
        const bool grpY = wave >= NW / 2;                              // wave-uniform (readfirstlane above): scalar branches
        u32x4 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
        auto read_all = [&](auto buf_c) {
            constexpr int buf = decltype(buf_c)::value;
            load_frags(buf, std::integral_constant<int, 0>{}, fah[0], fal[0], fbh[0], fbl[0]);
            load_frags(buf, std::integral_constant<int, 1>{}, fah[1], fal[1], fbh[1], fbl[1]);
            frag_wait(fah[0], fal[0], fbh[0], fbl[0]);
            frag_wait(fah[1], fal[1], fbh[1], fbl[1]);
            if constexpr (XX && !F16) {
                if (g.b_f16) {                           // (weight gradients reading the forward's f16 pair of x: converted here, in the shadow of the partner group's MFMAs)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int b = 0; b < TN; ++b) f16pair_to_bf16pair(fbh[ks][b], fbl[ks][b]);
                }
            }
        };
        auto mfma_step = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) mfma_ks(fah[ks], fal[ks], fbh[ks], fbl[ks]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (nk > 0) issue_all(0, r_begin);
        if (nk > 1) issue_all(1, r_begin + BK);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NLOAD_) : "memory");
        else        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (!grpY) read_all(std::integral_constant<int, 0>{});
        auto step = [&](int t, auto b0_c, auto b1_c, auto b2_c) {
            constexpr int b2 = decltype(b2_c)::value;
            // ---- A(t)
            if (!grpY) {
                mfma_step();
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                read_all(b0_c);
                if (t + 2 < nk) {
                    issue_all(b2, r_begin + (t + 2) * BK);
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NLOAD_) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
            // ---- B(t)
            if (grpY) {
                mfma_step();
            } else {
                if (t + 1 < nk) read_all(b1_c);
                if (t + 2 < nk) issue_all(b2, r_begin + (t + 2) * BK);
            }
            asm volatile("s_barrier" ::: "memory");
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int t = 0; t < nk; t += 3) {                 // unrolled by the ring depth: LDS addresses are lane bases + immediates
            step(t, I0{}, I1{}, I2{});
            if (t + 1 < nk) step(t + 1, I1{}, I2{}, I0{});
            if (t + 2 < nk) step(t + 2, I2{}, I0{}, I1{});
        }
        if (XX && g.mode == ST_ATOMIC && g.bias == nullptr && g.act == BDETR_ACT_NONE && g.stat_sum == nullptr && !g.rowmap && g.ldc * 4 * BM < (1ll << 31)) {
            atomic_epilogue<BM, BN, WM, WN>(acc, g, i0, j0);
            return;
        }
        gemm_epilogue<BM, BN, WM, WN, NT, LDS_BYTES / 4>(acc, g, reinterpret_cast<float*>(lds), tile_i, i0, j0, g.c + (g.splitk > 1 ? (int64_t)zz * g.sc0 : 0), bias_pre);
    } else if constexpr (PERSIST) {
        // Two-stage ring, persistent over tiles.  Per K-step: wait for everything this wave has in flight (stage kt, and
        // on a tile's first step the previous tile's C stores - on gfx9 stores count in vmcnt and return out of order with
        // loads, so a counted wait cannot tell them apart), barrier (every wave's share of stage kt has landed and every
        // wave is done reading stage kt - 1), refill the buffer stage kt - 1 used, compute.  After a tile's last K-step
        // the NEXT tile's first two stages are issued into the (idle) ring BEFORE this tile's epilogue: the C stores
        // and the next tile's first HBM round trip overlap instead of following each other (measured on the 64 -> 256
        // channel 1x1 layer at 160x160: loads + MFMAs alone 76 us, stores alone 89 us, one after the other 141 us).
        const int nk = BDETR_DBG(g, 2) ? 0 : (g.R + BK - 1) / BK;
        auto prefetch = [&]() { if (nk > 0) issue_all(0, 0); if (nk > 1) issue_all(1, BK); };
        prefetch();
        for (;;) {
            float bias_pre[TN];
            gemm_load_bias<BM, BN, WM, WN>(g, j0, bias_pre);          // lands during the K loop
            // (two K-steps per trip: the ring buffer is a compile-time constant of each half, so LDS addresses are
            // lane bases + immediates instead of per-buffer copies held - and spilled - across the loop)
            for (int kt = 0; kt < nk; kt += 2) {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (kt >= 1 && kt + 1 < nk) issue_all(1, (kt + 1) * BK);
                kstep(0);
                if (kt + 1 < nk) {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    if (kt + 2 < nk) issue_all(0, (kt + 2) * BK);
                    kstep(1);
                }
            }
            // bias / activation / statistics first (their temporaries die before the next tile's row descriptors are born), then
            // the next tile's set-up and first stages, then the stores
            gemm_bias_act_stats<BM, BN, WM, WN>(acc, g, tile_i, i0, j0, bias_pre);
            const int e_tile_i = tile_i, e_i0 = i0, e_j0 = j0;
            const int nvb = vb + (int)gridDim.x;
            const bool more = nvb < total;
            if (more) { locate(nvb); setup_rows(); }
            asm volatile("s_barrier" ::: "memory");                 // every wave is done reading the ring
            if (more) prefetch();
            direct_epilogue<BM, BN, WM, WN, false, EPI>(acc, g, e_tile_i, e_i0, e_j0, bias_pre);
            if (!more) break;
            vb = nvb;
            zero_acc();
        }
    } else {
        // One barrier per K-step: wait for my share of stage kt, barrier (every wave's share has landed and every wave
        // is done reading stage kt - 1), refill that buffer with stage kt + 1, compute.  (Loads issued early - before the
        // MFMAs, not between them - measured faster: they have the whole K-step to land.)
        const int nk = BDETR_DBG(g, 2) ? 0 : (r_end - r_begin + BK - 1) / BK;
        float bias_pre[TN];
        gemm_load_bias<BM, BN, WM, WN>(g, j0, bias_pre);              // lands during the K loop
        if (nk > 0) issue_all(0, r_begin);
#if defined(BDETR_SGEMM_STAMPS) && defined(BDETR_SGEMM_DIAG)      // diagnostic builds only (BDETR_CXXFLAGS="-DBDETR_SGEMM_DIAG -DBDETR_SGEMM_STAMPS"): the stamp array costs the kernels ~90 VGPRs
        if (BDETR_DBG(g, 32) && blockIdx.x == 1 && blockIdx.z == 0 && wave == 0) {
            // diagnostic: cycle stamps of workgroup 1's first wave over its first 12 K-steps (bdetr_sgemm_debug_stamps):
            // per step {after the load wait, after the barrier, after issuing the next stage, after issuing the MFMAs}
            unsigned long long st[48];
            auto stamp = [&](int i) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); st[i] = t; };
#pragma unroll
            for (int kt = 0; kt < 12; ++kt) {
                if (kt < nk) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(4 * kt);
                    asm volatile("s_barrier" ::: "memory"); stamp(4 * kt + 1);
                    if (kt + 1 < nk) issue_all((kt + 1) & 1, r_begin + (kt + 1) * BK);
                    stamp(4 * kt + 2);
                    kstep(kt & 1);
                    stamp(4 * kt + 3);
                }
            }
            for (int kt = 12; kt < nk; ++kt) {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (kt + 1 < nk) issue_all((kt + 1) & 1, r_begin + (kt + 1) * BK);
                kstep(kt & 1);
            }
            if (lane == 0) for (int i = 0; i < 48; ++i) g_stamps[i] = st[i];
        } else
#endif
        for (int kt = 0; kt < nk; kt += 2) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < nk) issue_all(1, r_begin + (kt + 1) * BK);
            kstep(0);
            if (kt + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (kt + 2 < nk) issue_all(0, r_begin + (kt + 2) * BK);
                kstep(1);
            }
        }
        if (XX && g.mode == ST_ATOMIC && g.bias == nullptr && g.act == BDETR_ACT_NONE && g.stat_sum == nullptr && !g.rowmap && g.ldc * 4 * BM < (1ll << 31)) {
            atomic_epilogue<BM, BN, WM, WN>(acc, g, i0, j0);
            return;
        }
        gemm_epilogue<BM, BN, WM, WN, NT, LDS_BYTES / 4, (EPI & EPI_VEC) != 0, (EPI & EPI_BNB2) != 0, (EPI & EPI_COMPACT) != 0>(acc, g, reinterpret_cast<float*>(lds), tile_i, i0, j0, g.c + (g.splitk > 1 ? (int64_t)zz * g.sc0 : 0), bias_pre);
    }
}

// ----------------------------------------------------------------------------------------
// host-side dispatch
// ----------------------------------------------------------------------------------------
// Measured and not instantiated (tools/p16_bench.py over the ResNet-50 batch-16 layers, round 2): a 3-stage ring at
// one 128x128 workgroup per CU (19.3 ms for all layers vs 15.3), 8-wave 256x128 tiles with 2 or 3 stages (17.3 / 17.4 ms;
// equal on the MFMA-bound 3x3 layers, slower on the short-K 1x1 layers), next-stage loads interleaved between the MFMA
// groups instead of issued up front (15.8 ms), 3- and 4-stage rings at two workgroups per CU on 128x64 / 64x64 tiles for
// the short-K 1x1 layers (equal or slower on every layer: those layers were bound by the serial load -> MFMA -> store
// phases of a workgroup's life, which the persistent loop overlaps, not by bytes in flight); after the epilogue
// rewrite, again a 3-stage ring with counted waits on 128x64 tiles at two workgroups per CU (13.4 ms for all layers vs
// 12.4), 128x64 two-stage tiles at three workgroups per CU (13.1 ms) and 8-wave 256x128 tiles for the patch kernels
// (13.1 ms: equal on the 40x40 3x3 layers, slower on the others).  More bytes in flight or more resident
// workgroups do not help: at 27 B/clk/CU the staging already runs near the L2 -> LDS fill rate the guide measured for
// LDS-DMA gathers (66-73 GB/s per CU), which is what a bigger FLOP-per-staged-byte ratio would have to relieve.
enum { T_128x128 = 0, T_128x64 = 1, T_64x64 = 2, T_PP256x128 = 3, T_PP256x64 = 4, T_COUNT = 5 };
const int TILE_BM[T_COUNT] = {128, 128, 64, 256, 256};
const int TILE_BN[T_COUNT] = {128, 64, 64, 128, 64};
const int TILE_WM[T_COUNT] = {2, 2, 2, 4, 4};
constexpr int DENSE128_WM = 2;                                   // wave rows of the 128x128 tile in the persistent dense (1x1) kernels
int tile_wm(int tile, bool dense) { return (dense && tile == T_128x128) ? DENSE128_WM : TILE_WM[tile]; }

int forced_tile() {
    static int forced = -2;
    if (forced == -2) {
        forced = -1;
        if (const char* e = getenv("BDETR_STILE")) {
            const char* names[T_COUNT] = {"128x128", "128x64", "64x64", "pp256x128", "pp256x64"};
            for (int t = 0; t < T_COUNT; ++t) if (!strcmp(e, names[t])) forced = t;
            if (!strcmp(e, "pp")) forced = T_PP256x128;
        }
    }
    return forced;
}

// Bigger tiles stage fewer bytes per FLOP (128x128: 32 FLOP per staged byte, 64x64: 16) but need enough
// workgroups to fill 256 CUs x 2 resident workgroups.
// patch = the A operand is an im2col view (3x3 forward / backward-data): the ping-pong tiles exist for those kernels only
// (pp_ok: bf16 launches only - the f16 flavour's two accumulator sets leave no room for a whole K-step of fragments)
int choose_tile(int64_t I, int64_t J, int64_t z, bool pp_ok = false) {
    int forced = forced_tile();
    if (forced >= T_PP256x128) {
        const bool patch = pp_ok;
        if (!patch) forced = -1;                                       // dense / weight-gradient / f16 launches keep their own rule
        else return J % 128 == 0 ? T_PP256x128 : (J % 64 == 0 ? T_PP256x64 : T_128x64);
    }
    if (forced >= 0) return (J < 64 || (TILE_BN[forced] == 128 && J < 128)) ? T_64x64 : forced;
    const int64_t cus = num_cus();
    auto tiles = [&](int bm, int bn) { return cdiv64(I, bm) * cdiv64(J, bn) * z; };
    if (J % 128 == 0 && tiles(128, 128) >= cus) return T_128x128;
    if (tiles(128, 64) >= cus) return T_128x64;
    return T_64x64;
}

template <int BM, int BN, int WM, int WN, int NS, class LA, class LB, bool XX, bool F16, bool PP = false, int OCC = default_occ(BM, BN, WM, WN, NS)>
int launch_cfg(const typename LA::Op& a, const typename LB::Op& b, GemmParams g, int zdim, hipStream_t st, int kind) {
    g.tiles_i = (int)cdiv64(g.I, BM);
    g.tiles_j = (int)cdiv64(g.J, BN);
    dim3 grid(g.tiles_i * g.tiles_j, 1, zdim);
    g.vec_store = (g.J % 4 == 0) && (g.ldc % 4 == 0) && aligned16(g.c);
#ifdef BDETR_SGEMM_DIAG
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("BDETR_SGEMM_DBG"); dbg = e ? atoi(e) : 0; }
    g.dbg = dbg;
    if (dbg & 4) g.vec_store = 0;
#endif
    const bool prof = g_prof_on;
    if (prof) prof_begin(st, 2.0 * (double)g.I * (double)g.J * (double)g.R, g.I, g.J, g.R, zdim, BM, BN * 10 + NS,
                         (F16 ? AR_P16_F16 : AR_P16_BF16) * 10000 + kind);
    if constexpr (!PP && !XX && std::is_same_v<LA, RRDense>) {
        auto go = [&](auto epi_c) {
            hipLaunchKernelGGL((sgemm_kernel<BM, BN, WM, WN, NS, LA, LB, XX, F16, PP, OCC, decltype(epi_c)::value>), grid, dim3(WM * WN * 64), 0, st, a, b, g);
            if (prof) prof_end(st);
            return bdetr_launch_status("sgemm");
        };
        BDETR_CHECK_ARG(zdim == 1 && g.splitk == 1, "sgemm: the dense 1x1 kernels take no split-K / batch dimension");
        if (!g.rowmap && (g.mode == ST_ACCUM || g.bnb_y != nullptr)) {        // accumulate / fused sums: one workgroup per tile, float4 epilogue (see EPI)
            BDETR_CHECK_ARG(g.vec_store, "sgemm: accumulate / fused BatchNorm-backward sums need 16-byte aligned C rows and J %% 4 == 0");
            BDETR_CHECK_ARG(g.acc_mask == nullptr || g.mode == ST_ACCUM, "sgemm: acc_mask without accumulate");
            BDETR_CHECK_ARG(g.bnb_mask == nullptr || (g.acc_mask != nullptr && g.bnb_y != nullptr), "sgemm: a bit-mask ReLU decision comes with a masked accumulate");
            if (g.acc_src != nullptr) {
                BDETR_CHECK_ARG(g.acc_mask != nullptr && g.bnb2_y == nullptr && g.acc_H > 0 && g.acc_W > 0 && g.acc_H % 2 == 0 && g.acc_W % 2 == 0 && g.I % (g.acc_H * g.acc_W) == 0,
                                "sgemm: a compact old gradient needs the masked accumulate, an even map that divides the rows, and no second BatchNorm");
                return go(std::integral_constant<int, EPI_VEC | EPI_COMPACT>{});
            }
            if (g.bnb2_y != nullptr) return go(std::integral_constant<int, EPI_VEC | EPI_BNB2>{});
            return go(std::integral_constant<int, EPI_VEC>{});
        }
        BDETR_CHECK_ARG(g.acc_mask == nullptr && g.bnb_y == nullptr, "sgemm: masks / fused sums are not available with a row map");
        // persistent: as many workgroups as stay resident (a multiple of 8: one XCD per workgroup for all its tiles)
        static int resident = 0;
        if (resident == 0) {
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sgemm_kernel<BM, BN, WM, WN, NS, LA, LB, XX, F16, PP, OCC, EPI_ROWMAP>, WM * WN * 64, 0) != hipSuccess || occ < 1) occ = 1;
            resident = occ * num_cus() / 8 * 8;
            if (resident < 8) resident = 8;
        }
        if ((int)grid.x > resident) grid.x = resident;
        if (!g.rowmap) return go(std::integral_constant<int, EPI_PLAIN>{});
        return go(std::integral_constant<int, EPI_ROWMAP>{});
    }
    hipLaunchKernelGGL((sgemm_kernel<BM, BN, WM, WN, NS, LA, LB, XX, F16, PP, OCC>), grid, dim3(WM * WN * 64), 0, st, a, b, g);
    if (prof) prof_end(st);
    return bdetr_launch_status("sgemm");
}

template <class LA, class LB, bool XX, bool F16>
int launch_any(const typename LA::Op& a, const typename LB::Op& b, const GemmParams& g, int zdim, hipStream_t st, int kind, int tile) {
    switch (tile) {
        case T_128x128:
            // (DENSE128_WM = 4: the persistent dense kernels on EIGHT waves - wave tile 32x64, 4 waves per SIMD at two workgroups per
            // CU, 127 VGPRs without spills in the plain-epilogue flavour.  Measured equal or slower than four waves on every 1x1
            // layer and 1.3 % slower on the step, round 3; kept as a switch.)
            if constexpr (!XX && std::is_same_v<LA, RRDense> && DENSE128_WM == 4) return launch_cfg<128, 128, 4, 2, 2, LA, LB, XX, F16, false, 4>(a, b, g, zdim, st, kind);
            else return launch_cfg<128, 128, 2, 2, 2, LA, LB, XX, F16>(a, b, g, zdim, st, kind);
        case T_128x64:     return launch_cfg<128, 64, 2, 2, 2, LA, LB, XX, F16>(a, b, g, zdim, st, kind);
        case T_PP256x128:
            // (the f16 flavour keeps two accumulator sets: with the fragments of a whole K-step resident it does not fit 256 VGPRs)
            // (round 4 also ran the weight gradients on the ping-pong loop - BDETR_WGRAD_PP: slower on every layer, removed in round 5)
            if constexpr (std::is_same_v<LA, RRPatch> && !F16) return launch_cfg<256, 128, 4, 2, 3, LA, LB, XX, F16, true>(a, b, g, zdim, st, kind);
            else return launch_cfg<128, 128, 2, 2, 2, LA, LB, XX, F16>(a, b, g, zdim, st, kind);
        case T_PP256x64:
            if constexpr (std::is_same_v<LA, RRPatch> && !F16) return launch_cfg<256, 64, 4, 2, 3, LA, LB, XX, F16, true>(a, b, g, zdim, st, kind);
            else return launch_cfg<128, 64, 2, 2, 2, LA, LB, XX, F16>(a, b, g, zdim, st, kind);
        default:           return launch_cfg<64, 64, 2, 2, 2, LA, LB, XX, F16>(a, b, g, zdim, st, kind);
    }
}

// 16 MB of slack: the row descriptors of a tile's rows past the end of a dense operand (up to 255 rows of at most 16,384
// elements) must still lie beyond the buffer descriptor without wrapping the 32-bit offset
constexpr int64_t MAX_OPERAND_ELEMS = (int64_t)(NUM_RECORDS / 4) - (4 << 20);
bool span_ok(int64_t elems) { return elems >= 0 && elems <= MAX_OPERAND_ELEMS; }

int check_conv(const bdetr_conv_desc* d, const char* who) {
    BDETR_CHECK_ARG(d != nullptr, "%s: null desc", who);
    BDETR_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->stride > 0 && d->pad >= 0,
                    "%s: bad conv geometry", who);
    BDETR_CHECK_ARG(d->OH == (d->H + 2 * d->pad - d->R) / d->stride + 1 && d->OW == (d->W + 2 * d->pad - d->S) / d->stride + 1,
                    "%s: OH/OW inconsistent with geometry", who);
    BDETR_CHECK_ARG(d->C % 8 == 0 && d->K % 8 == 0, "%s: the P16 layout needs channel counts that are multiples of 8 (C=%d K=%d)", who, d->C, d->K);
    BDETR_CHECK_ARG((d->R == 1 && d->S == 1) || d->C % BK == 0, "%s: kernels larger than 1x1 need C %% %d == 0 (C=%d)", who, BK, d->C);
    BDETR_CHECK_ARG(d->R * d->S <= 32 && (int64_t)d->R * d->S * d->C <= 16384 && d->C <= 16384 && d->K <= 16384, "%s: at most 32 taps and rows of at most 16,384 elements", who);
    BDETR_CHECK_ARG((int64_t)d->N * d->OH * d->OW < (1LL << 31) && (int64_t)d->R * d->S * d->C < (1LL << 31) && (int64_t)d->N * d->H * d->W < (1LL << 31),
                    "%s: problem too large", who);
    BDETR_CHECK_ARG(span_ok((int64_t)d->N * d->H * d->W * d->C) && span_ok((int64_t)d->N * d->OH * d->OW * d->K) && span_ok((int64_t)d->K * d->R * d->S * d->C),
                    "%s: a tensor spans 4 GB or more (the loaders use 32-bit buffer offsets); use a smaller per-GPU batch", who);
    return 0;
}

PPatch make_patch(const void* p, int N, int H, int W, int C, int OH, int OW, int R, int S, int stride, int pad, int rows, int cols) {
    PPatch o{p, N, H, W, C, OH, OW, R, S, stride, pad, rows, cols, BK % OW, BK / OW, 0u, 0u, 0u};
    const unsigned c4 = (unsigned)C * 4u;
    o.k_step = (unsigned)(o.d_ow * stride + o.d_oh * stride * W) * c4;
    o.k_wrap_ow = (unsigned)(stride * W - OW * stride) * c4;                  // ow -= OW, oh += 1   (wrapping uint32)
    o.k_wrap_oh = (unsigned)(H * W - OH * stride * W) * c4;                    // oh -= OH, next image
    o.single_wrap = o.d_oh < OH;
    return o;
}
bool is_1x1_dense(const bdetr_conv_desc* d) { return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0; }

}  // namespace

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
#if defined(BDETR_SGEMM_STAMPS) && defined(BDETR_SGEMM_DIAG)
// diagnostic builds only: the cycle stamps a launch made with BDETR_SGEMM_DBG bit 32 left behind (tools/kstep_stamps.py)
extern "C" int bdetr_sgemm_debug_stamps(uint64_t* out, int n) {
    BDETR_CHECK_ARG(out && n > 0 && n <= 64, "bdetr_sgemm_debug_stamps: 1 <= n <= 64");
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(uint64_t) * (size_t)n);
    if (e != hipSuccess) { bdetr_set_error("bdetr_sgemm_debug_stamps: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
#endif

extern "C" int bdetr_p16_supported(const bdetr_conv_desc* d) {
    if (!d || d->C % 8 || d->K % 8) return 0;
    if (!(d->R == 1 && d->S == 1) && d->C % BK) return 0;
    if (!(d->R == 1 && d->S == 1) && d->K % BK) return 0;          // backward-data gathers patches of dy: K plays C's role
    if (d->R * d->S > 32 || (int64_t)d->R * d->S * d->C > 16384 || (int64_t)d->R * d->S * d->K > 16384) return 0;   // tap mask; row pitch bound
    if (d->stride > 1 && !(d->R == 1 && d->S == 1 && d->pad == 0)) return 0;
    return 1;
}

static int fwd_tile(const bdetr_conv_desc* d) { return choose_tile((int64_t)d->N * d->OH * d->OW, d->K, 1, false); }

// 3x3 / stride 1 / pad 1: the halo-resident kernel of hconv.hip (its tile as BM * 1000 + BN), else 0
static int fwd_hconv(const bdetr_conv_desc* d) {
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return 0;
    return hconv_tile((int64_t)d->N * d->H * d->W, d->W, d->C, d->K, true);
}

extern "C" int bdetr_p16_conv2d_fwd_stat_chunks(const bdetr_conv_desc* d) {
    if (check_conv(d, "bdetr_p16_conv2d_fwd_stat_chunks")) return -1;
    if (const int ht = fwd_hconv(d)) return (int)cdiv64((int64_t)d->N * d->OH * d->OW, ht / 1000) * 4;      // tiles_i * WM (4 wave rows)
    const int t = fwd_tile(d);
    return (int)cdiv64((int64_t)d->N * d->OH * d->OW, TILE_BM[t]) * tile_wm(t, is_1x1_dense(d));          // tiles_i * WM
}

extern "C" int bdetr_p16_conv2d_fwd(const void* x_f16, const void* w_f16, const float* bias, float* y,
                                    const bdetr_conv_desc* d, int act, float* stat_sum, float* stat_sq, void* stream) {
    if (int e = check_conv(d, "bdetr_p16_conv2d_fwd")) return e;
    BDETR_CHECK_ARG(x_f16 && w_f16 && y && aligned16(x_f16) && aligned16(w_f16), "bdetr_p16_conv2d_fwd: null / unaligned pointer");
    BDETR_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "bdetr_p16_conv2d_fwd: stat_sum/stat_sq must both be set or both null");
    const int M = d->N * d->OH * d->OW, Kd = d->R * d->S * d->C;
    GemmParams g; init_params(g);
    g.I = M; g.J = d->K; g.R = Kd;
    g.c = y; g.ldc = d->K; g.bias = bias; g.act = act;
    g.alpha = 1.f / P16_W_SCALE;                          // the weights' f16 pair copy holds 2^8 w (p16.h)
    g.stat_sum = stat_sum; g.stat_sq = stat_sq;
    PDense wop{w_f16, (unsigned)Kd, d->K, Kd};
    hipStream_t st = (hipStream_t)stream;
    const int tile = fwd_tile(d);
    if (is_1x1_dense(d)) {
        PDense xop{x_f16, (unsigned)d->C, M, d->C};
        return launch_any<RRDense, RRDense, false, true>(xop, wop, g, 1, st, 0, tile);
    }
    if (const int ht = fwd_hconv(d)) return hconv_launch(ht, true, x_f16, d->N, d->H, d->W, d->C, w_f16, d->K, g, st);
    PPatch xop = make_patch(x_f16, d->N, d->H, d->W, d->C, d->OH, d->OW, d->R, d->S, d->stride, d->pad, M, Kd);
    return launch_any<RRPatch, RRDense, false, true>(xop, wop, g, 1, st, 1000, tile);
}

static int bwd_data_tile(const bdetr_conv_desc* d) {
    const bool dense = d->R == 1 && d->S == 1 && d->pad == 0;
    return choose_tile(dense ? (int64_t)d->N * d->OH * d->OW : (int64_t)d->N * d->H * d->W, d->C, 1, !dense);
}

// 1x1 stride-1 backward-data with accumulate / fused sums (EPI_VEC: one workgroup per tile, float4 epilogue with up to three tile-sized
// loads): short reductions (K <= 512) run 128x64 tiles - half the epilogue registers (no second half-pass), twice the workgroups to hide its
// round trips (masked accumulate + sums at 160x160: 0.373 -> 0.335 ms, 80x80: 0.213 -> 0.183); long ones keep the operand reuse of 128x128
static int dense_vec_tile(const bdetr_conv_desc* d) {
    const int t = bwd_data_tile(d);
    return (t == T_128x128 && d->K <= 512) ? T_128x64 : t;
}

// 3x3 / stride 1 / pad 1: the halo-resident kernel of hconv.hip (its tile as BM * 1000 + BN), else 0
static int bwd_data_hconv(const bdetr_conv_desc* d) {
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return 0;
    return hconv_tile((int64_t)d->N * d->H * d->W, d->W, d->K, d->C, false);
}

// number of partial rows bdetr_p16_conv2d_bwd_data writes per statistic when the BatchNorm-backward reduction is fused
extern "C" int bdetr_p16_conv2d_bwd_data_stat_chunks(const bdetr_conv_desc* d) {
    if (check_conv(d, "bdetr_p16_conv2d_bwd_data_stat_chunks")) return -1;
    const bool dense = d->R == 1 && d->S == 1 && d->pad == 0;
    if (const int ht = bwd_data_hconv(d)) return (int)cdiv64((int64_t)d->N * d->H * d->W, ht / 1000);      // one partial row per tile_i
    const int t = (dense && d->stride == 1) ? dense_vec_tile(d) : bwd_data_tile(d);
    // one partial row per tile_i (the LDS-transposed epilogue folds the tile's rows)
    return (int)cdiv64(dense ? (int64_t)d->N * d->OH * d->OW : (int64_t)d->N * d->H * d->W, TILE_BM[t]);
}

// dy_bf16: P16-bf16 [N,OH,OW,K]; wt_bf16: the transposed / tap-flipped P16-bf16 weight copy [C][R*S][K]
// (bdetr_p16_pack_conv_weights); dx: fp32 [N,H,W,C].  bn (may be null): dx is the gradient of the BatchNorm(+ReLU) output
// whose pre-normalisation tensor is bn->y - fuse that layer's backward reduction into this epilogue.
static int p16_bwd_data(const void* dy_bf16, const void* wt_bf16, float* dx, const bdetr_conv_desc* d, int accumulate,
                        const bdetr_bn_bwd_fuse* bn, const uint64_t* acc_mask, void* stream, const float* acc_src = nullptr) {
    if (int e = check_conv(d, "bdetr_p16_conv2d_bwd_data")) return e;
    BDETR_CHECK_ARG(dy_bf16 && wt_bf16 && dx, "bdetr_p16_conv2d_bwd_data: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int M = d->N * d->OH * d->OW;
    GemmParams g; init_params(g);
    g.c = dx; g.ldc = d->C; g.mode = accumulate ? ST_ACCUM : ST_STORE;
    g.acc_mask = reinterpret_cast<const unsigned long long*>(acc_mask);
    g.acc_src = acc_src; g.acc_H = d->H; g.acc_W = d->W;
    if (bn != nullptr) {
        const bool dense11 = d->R == 1 && d->S == 1 && d->pad == 0 && d->stride == 1;      // the persistent kernels' register epilogue
        BDETR_CHECK_ARG((!accumulate || dense11) && d->stride == 1 && d->C % 4 == 0 && aligned16(dx) && aligned16(bn->y),
                        "bdetr_p16_conv2d_bwd_data_bnstats: needs 16-byte aligned rows, stride 1, and a plain store unless the conv is 1x1");
        BDETR_CHECK_ARG(bn->relu_mask == nullptr || dense11, "bdetr_p16_conv2d_bwd_data_bnstats: a bit-mask ReLU decision needs a 1x1 stride-1 conv");
        BDETR_CHECK_ARG(bn->y && bn->mean && bn->rstd && bn->gamma && bn->beta && bn->part_g && bn->part_gx, "bdetr_p16_conv2d_bwd_data_bnstats: null pointer");
        g.bnb_y = bn->y; g.bnb_mean = bn->mean; g.bnb_rstd = bn->rstd; g.bnb_gamma = bn->gamma; g.bnb_beta = bn->beta;
        g.bnb_relu = bn->relu; g.bnb_sum_g = bn->part_g; g.bnb_sum_gx = bn->part_gx;
        g.bnb_mask = reinterpret_cast<const unsigned long long*>(bn->relu_mask);
        if (bn->y2 != nullptr) {
            BDETR_CHECK_ARG(acc_mask != nullptr && bn->relu_mask != nullptr && bn->mean2 && bn->rstd2 && bn->part_gx2 && aligned16(bn->y2),
                            "bdetr_p16_conv2d_bwd_data_masked_accum: the second BatchNorm's sums ride a masked accumulate with masked sums");
            g.bnb2_y = bn->y2; g.bnb2_mean = bn->mean2; g.bnb2_rstd = bn->rstd2; g.bnb2_sum_gx = bn->part_gx2;
        }
    }
    if (d->R == 1 && d->S == 1 && d->pad == 0) {
        g.I = M; g.J = d->C; g.R = d->K;
        if (d->stride > 1) {
            if (!accumulate) {
                if (int e = bdetr_zero_bytes(dx, sizeof(float) * (size_t)d->N * d->H * d->W * d->C, st)) return e;
            }
            g.rowmap = 1; g.rm_OW = d->OW; g.rm_OHOW = d->OH * d->OW; g.rm_H = d->H; g.rm_W = d->W; g.rm_stride = d->stride;
        }
        PDense a{dy_bf16, (unsigned)d->K, M, d->K};
        PDense b{wt_bf16, (unsigned)d->K, d->C, d->K};
        const bool vec = !g.rowmap && (g.mode == ST_ACCUM || g.bnb_y != nullptr);
        return launch_any<RRDense, RRDense, false, false>(a, b, g, 1, st, 0, vec ? dense_vec_tile(d) : bwd_data_tile(d));
    }
    BDETR_CHECK_ARG(d->stride == 1 && d->R == d->S, "bdetr_p16_conv2d_bwd_data: stride>1 only for 1x1 convs; square kernels only");
    BDETR_CHECK_ARG(d->K % BK == 0, "bdetr_p16_conv2d_bwd_data: K %% %d == 0 required for kernels larger than 1x1", BK);
    // dx[n,ih,iw,c] = sum_{r',s',k} dy[n, ih - pad' + r', iw - pad' + s', k] * wt[c][r'][s'][k],  pad' = R-1-pad, wt pre-flipped
    const int Mx = d->N * d->H * d->W, Kd = d->R * d->S * d->K;
    g.I = Mx; g.J = d->C; g.R = Kd;
    if (const int ht = bwd_data_hconv(d))                 // dy plays the input's role: same spatial size, K channels; wt rows = input channels
        return hconv_launch(ht, false, dy_bf16, d->N, d->H, d->W, d->K, wt_bf16, d->C, g, st);
    PPatch a = make_patch(dy_bf16, d->N, d->OH, d->OW, d->K, d->H, d->W, d->R, d->S, 1, d->R - 1 - d->pad, Mx, Kd);
    PDense b{wt_bf16, (unsigned)Kd, d->C, Kd};
    return launch_any<RRPatch, RRDense, false, false>(a, b, g, 1, st, 1000, bwd_data_tile(d));
}

extern "C" int bdetr_p16_conv2d_bwd_data(const void* dy_bf16, const void* wt_bf16, float* dx,
                                         const bdetr_conv_desc* d, int accumulate, void* stream) {
    return p16_bwd_data(dy_bf16, wt_bf16, dx, d, accumulate, nullptr, nullptr, stream);
}
extern "C" int bdetr_p16_conv2d_bwd_data_masked_accum(const void* dy_bf16, const void* wt_bf16, float* dx, const uint64_t* relu_mask,
                                                      const bdetr_conv_desc* d, const bdetr_bn_bwd_fuse* bn, void* stream) {
    BDETR_CHECK_ARG(relu_mask != nullptr && d != nullptr && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0,
                    "bdetr_p16_conv2d_bwd_data_masked_accum: 1x1 stride-1 convolutions only, mask required");
    BDETR_CHECK_ARG((int64_t)d->N * d->H * d->W * d->C < (1LL << 32), "bdetr_p16_conv2d_bwd_data_masked_accum: more than 2^32 elements");
    return p16_bwd_data(dy_bf16, wt_bf16, dx, d, 1, bn, relu_mask, stream);
}
extern "C" int bdetr_p16_conv2d_bwd_data_masked_accum_compact(const void* dy_bf16, const void* wt_bf16, float* dx, const float* old_even,
                                                              const uint64_t* relu_mask, const bdetr_conv_desc* d, const bdetr_bn_bwd_fuse* bn, void* stream) {
    BDETR_CHECK_ARG(relu_mask != nullptr && old_even != nullptr && d != nullptr && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->H % 2 == 0 && d->W % 2 == 0,
                    "bdetr_p16_conv2d_bwd_data_masked_accum_compact: 1x1 stride-1 convolutions on an even map only; mask and old gradient required");
    BDETR_CHECK_ARG((int64_t)d->N * d->H * d->W * d->C < (1LL << 32) && aligned16(old_even) && (bn == nullptr || bn->y2 == nullptr),
                    "bdetr_p16_conv2d_bwd_data_masked_accum_compact: more than 2^32 elements, a misaligned old gradient, or a second BatchNorm");
    return p16_bwd_data(dy_bf16, wt_bf16, dx, d, 1, bn, relu_mask, stream, old_even);
}
extern "C" int bdetr_p16_conv2d_bwd_data_bnstats(const void* dy_bf16, const void* wt_bf16, float* dx,
                                                 const bdetr_conv_desc* d, const bdetr_bn_bwd_fuse* bn, void* stream) {
    BDETR_CHECK_ARG(bn != nullptr, "bdetr_p16_conv2d_bwd_data_bnstats: null bn descriptor");
    return p16_bwd_data(dy_bf16, wt_bf16, dx, d, 0, bn, nullptr, stream);
}

static int wgrad_tile(const bdetr_conv_desc* d) {
    const int64_t Kd = (int64_t)d->R * d->S * d->C;
    // (Round 4 measured 256x128 ping-pong tiles here - BDETR_WGRAD_PP: 586 / 574 images/s for the 3x3 layers / every layer against 592;
    // one eight-wave workgroup per CU loses more to its serial prologue / atomic epilogue and to the halved slice count than the fill
    // saves.  Removed in round 5 together with its instantiations.)
    const int forced = forced_tile();
    if (forced >= 0 && d->K % TILE_BM[forced] == 0 && Kd % TILE_BN[forced] == 0) return forced;
    if (!(d->K % 128 == 0 && Kd % 128 == 0)) return T_64x64;
    // 1x1 layers: 128x64 (twice the workgroups per split-K slice, half the x-operand fragments per wave) measured 5-25 % faster than
    // 128x128 on 13 of the 16 layer shapes of ResNet-50 at batch 16 (tools/tile_sweep.py, round 3); the stride-2 projections (K = 2 C) not
    static int wide = -1;
    if (wide < 0) { const char* e = getenv("BDETR_WGRAD_1X1_TILE"); wide = (e && !strcmp(e, "128x128")) ? 1 : 0; }     // (A/B switch)
    if (d->R == 1 && d->S == 1 && !(d->stride > 1 && d->K > d->C) && !wide) return T_128x64;
    return T_128x128;
}

static bool is_3x3_same(const bdetr_conv_desc* d) { return d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1; }

extern "C" int bdetr_p16_conv2d_bwd_weight_splitk(const bdetr_conv_desc* d) {
    if (check_conv(d, "bdetr_p16_conv2d_bwd_weight_splitk")) return -1;
    if (is_3x3_same(d)) {                                 // the halo-resident kernel (hwgrad.hip) slices the pixel range itself
        const int hs = hwgrad_slices(d->N, d->H, d->W, d->C, d->K);
        if (hs > 0) return hs > 1 ? hs : 2;               // (> 1: the slices add with atomics, the caller must hand over zeros / a running sum)
    }
    const int M = d->N * d->OH * d->OW;
    const int64_t Kd = (int64_t)d->R * d->S * d->C;
    const int t = wgrad_tile(d);
    const int64_t tiles = cdiv64(d->K, TILE_BM[t]) * cdiv64(Kd, TILE_BN[t]);
    static int want_x = 0, want_3x3 = 0, min_stages = 0;
    if (want_x == 0) {
        const char* e = getenv("BDETR_WGRAD_WANT"); want_x = e ? atoi(e) : 2;      // measured: 2 workgroups per CU (5.9 ms over the ResNet-50 layers) beats 1 (6.5), 3 (6.4), 4 (6.9): atomic bytes grow with the split
        const char* e3 = getenv("BDETR_WGRAD_WANT_3X3"); want_3x3 = e3 ? atoi(e3) : want_x;      // (the same for the im2col layers alone: A/B switch)
        if (want_3x3 < 1) want_3x3 = 1;
        const char* f = getenv("BDETR_WGRAD_MINSTAGES"); min_stages = f ? atoi(f) : 8;
        if (want_x < 1) want_x = 1;
        if (min_stages < 1) min_stages = 1;
    }
    // floor, not ceil: all tiles x slices must fit the want_x * CUs resident slots at once - one workgroup more than
    // that runs alone in a second round and doubles the launch's duration
    // (3x3 layers on 64x64 tiles - the 64-channel layers of stage 2 - take three workgroups per CU: four fit, and the per-lane patch
    // addressing of their loads wants more waves to hide behind; 0.156 -> 0.130 ms at 160x160x64, round 5, profiles/r05_wgrad_tile_sweep.txt)
    static int want_64 = 0;
    if (want_64 == 0) { const char* e64 = getenv("BDETR_WGRAD_WANT_64"); want_64 = e64 ? atoi(e64) : 3; if (want_64 < 1) want_64 = 1; }
    const int want = (d->R == 1 && d->S == 1) ? want_x : (t == T_64x64 && getenv("BDETR_WGRAD_WANT_3X3") == nullptr ? want_64 : want_3x3);
    int64_t sk = ((int64_t)want * num_cus()) / tiles;
    const int64_t maxsk = cdiv64(M, (int64_t)min_stages * BK);       // keep >= min_stages K-steps per split
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
    if (sk > 512) sk = 512;
    return (int)sk;
}

// x_bf16: P16-bf16 [N,H,W,C]; dy_bf16: P16-bf16 [N,OH,OW,K]; dw: fp32 [K][R][S][C], must hold zeros (or the
// running sum) when splitk > 1 - the split-K slices add with float atomics
static int p16_bwd_weight(const void* x_bf16, int x_is_f16, const void* dy_bf16, float* dw, const bdetr_conv_desc* d, int splitk, void* stream,
                          float* ws = nullptr, int64_t ws_elems = 0) {
    if (int e = check_conv(d, "bdetr_p16_conv2d_bwd_weight")) return e;
    BDETR_CHECK_ARG(x_bf16 && dy_bf16 && dw, "bdetr_p16_conv2d_bwd_weight: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int M = d->N * d->OH * d->OW, Kd = d->R * d->S * d->C;
    if (splitk <= 0) splitk = bdetr_p16_conv2d_bwd_weight_splitk(d);
    if (!x_is_f16 && is_3x3_same(d) && hwgrad_slices(d->N, d->H, d->W, d->C, d->K) > 0) {
        // both operands as bf16 pairs, 3x3 'same': dy and x stream through LDS once for all nine taps (hwgrad.hip)
        const int64_t n = (int64_t)d->K * Kd, slab = splitk_slab(n);
        if (ws != nullptr) BDETR_CHECK_ARG(aligned16(ws) && ws_elems >= (int64_t)splitk * slab, "bdetr_p16_conv2d_bwd_weight_ws: workspace too small or misaligned");
        int zdim = 1;
        if (int e = hwgrad_launch(x_bf16, dy_bf16, dw, d->N, d->H, d->W, d->C, d->K, splitk, ws, slab, &zdim, st)) return e;
        return ws != nullptr ? splitk_fold(ws, zdim, slab, dw, n, st) : 0;
    }
    GemmParams g; init_params(g);
    g.b_f16 = x_is_f16;
    g.I = d->K; g.J = Kd; g.R = M;
    g.c = dw; g.ldc = Kd;
    int zdim = 1;
    if (splitk > 1) {
        g.r_chunk = (int)(cdiv64(cdiv64(M, splitk), BK) * BK);
        zdim = (int)cdiv64(M, g.r_chunk);
        g.splitk = zdim > 1 ? zdim : 2;
        g.mode = ST_ATOMIC;
    }
    PDense a{dy_bf16, (unsigned)d->K, M, d->K};          // rows = r (pixels), cols = i (output channels)
    const int tile = wgrad_tile(d);
    const bool slabs = ws != nullptr && splitk > 1;      // deterministic mode: partial tiles to the caller's workspace, fixed-order fold
    const int64_t n = (int64_t)d->K * Kd, slab = splitk_slab(n);
    if (slabs) {
        BDETR_CHECK_ARG(aligned16(ws) && ws_elems >= (int64_t)zdim * slab, "bdetr_p16_conv2d_bwd_weight_ws: workspace too small or misaligned (%lld floats, need %lld)",
                        (long long)ws_elems, (long long)((int64_t)zdim * slab));
        g.c = ws; g.sc0 = slab; g.mode = ST_STORE;
    }
    int e;
    if (is_1x1_dense(d)) {
        PDense b{x_bf16, (unsigned)d->C, M, d->C};
        e = launch_any<XXDense, XXDense, true, false>(a, b, g, zdim, st, 2000, tile);
    } else {
        PPatch b = make_patch(x_bf16, d->N, d->H, d->W, d->C, d->OH, d->OW, d->R, d->S, d->stride, d->pad, M, Kd);
        e = launch_any<XXDense, XXPatch, true, false>(a, b, g, zdim, st, 3000, tile);
    }
    if (e || !slabs) return e;
    return splitk_fold(ws, zdim, slab, dw, n, st);
}
extern "C" int bdetr_p16_conv2d_bwd_weight_ws(const void* x, int x_is_f16, const void* dy_bf16, float* dw,
                                              const bdetr_conv_desc* d, int splitk, float* ws, int64_t ws_elems, void* stream) {
    return p16_bwd_weight(x, x_is_f16, dy_bf16, dw, d, splitk, stream, ws, ws_elems);
}
extern "C" int bdetr_p16_conv2d_bwd_weight(const void* x_bf16, const void* dy_bf16, float* dw,
                                           const bdetr_conv_desc* d, int splitk, void* stream) {
    return p16_bwd_weight(x_bf16, 0, dy_bf16, dw, d, splitk, stream);
}
// x as the FORWARD operand of the same convolution (P16-f16): converted to bf16 pairs in registers, fragment by fragment
extern "C" int bdetr_p16_conv2d_bwd_weight_xf16(const void* x_f16, const void* dy_bf16, float* dw,
                                                const bdetr_conv_desc* d, int splitk, void* stream) {
    return p16_bwd_weight(x_f16, 1, dy_bf16, dw, d, splitk, stream);
}

// The stem's weight gradient over the space-to-depth image (p16.hip: bdetr_p16_s2d_pack_bf16): x2 P16-bf16 [N,H2,W2,16], dy P16-bf16
// [N,H2,W2,K]; dw2 fp32 [K][4][4][16], holding zeros on entry (the split-K slices add with float atomics).  The 4x4 kernel sits at
// low-side padding 2 over a size-preserving map (OH = H2: the high side has padding 1) - an asymmetric geometry bdetr_conv_desc
// cannot state, hence the entry point of its own.  Replaces the kernel gradient of keras ResNet50 conv1_conv (reference backbone.py:37-38 +
// autodiff), which igemm.hip's in-kernel-split kernel computed in 355-470 us as the LAST kernel of the backward pass.
extern "C" int bdetr_p16_stem_bwd_weight(const void* x2_bf16, const void* dy_bf16, float* dw2, int N, int H2, int W2, int K, void* stream) {
    BDETR_CHECK_ARG(x2_bf16 && dy_bf16 && dw2 && N > 0 && H2 > 0 && W2 > 0 && K > 0 && K % 64 == 0 && aligned16(x2_bf16) && aligned16(dy_bf16),
                    "bdetr_p16_stem_bwd_weight: bad arguments (K %% 64 == 0)");
    const int64_t M64 = (int64_t)N * H2 * W2;
    BDETR_CHECK_ARG(M64 < (1LL << 31) && span_ok(M64 * 16) && span_ok(M64 * K), "bdetr_p16_stem_bwd_weight: tensors of 4 GB or more");
    const int M = (int)M64, Kd = 4 * 4 * 16;
    GemmParams g; init_params(g);
    // (the transposed product - [256][K] on 128 x 64 tiles, dy staged by two row tiles instead of four column tiles - measured slower: 309 against 279 us)
    g.I = K; g.J = Kd; g.R = M; g.c = dw2; g.ldc = Kd;
    const int64_t tiles = cdiv64(K, 64) * cdiv64(Kd, 64);
    int64_t sk = (16LL * num_cus()) / tiles;                          // few output tiles over a very long pixel range: 16 workgroups per CU (igemm.hip's few-tiles rule)
    const int64_t maxsk = cdiv64(M, 8LL * BK);
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
    if (sk > 1024) sk = 1024;
    int zdim = 1;
    if (sk > 1) {
        g.r_chunk = (int)(cdiv64(cdiv64(M, sk), BK) * BK);
        zdim = (int)cdiv64(M, g.r_chunk);
        g.splitk = zdim > 1 ? zdim : 2;
        g.mode = ST_ATOMIC;
    }
    PDense a{dy_bf16, (unsigned)K, M, K};
    PPatch b = make_patch(x2_bf16, N, H2, W2, 16, H2, W2, 4, 4, 1, 2, M, Kd);
    return launch_any<XXDense, XXPatch, true, false>(a, b, g, zdim, (hipStream_t)stream, 3000, T_64x64);
}
