// hconv.hip - 3x3, stride-1, "same" convolutions on pre-split (P16) operands with the INPUT HALO resident in LDS, for gfx950.
//
// sgemm.hip treats a 3x3 convolution as an implicit GEMM over an im2col view: every K-step (one tap x 32 channels) stages a
// fresh BM x 32 patch tile AND a BN x 32 weight tile from L2 into LDS.  Measured (profiles/README.md, round 3): those kernels
// are bound by the L2 -> LDS fill path (~27 B/clk/CU), not by the MFMA pipe - a 128x128 tile needs 42 B/clk at the full MFMA
// rate, a 256x128 one still 31.  But the nine taps of one 32-channel chunk read the SAME input pixels shifted by (tr, ts):
// the patch tile of tap (tr, ts) is the patch tile of tap (0, 0) moved by tr * W + ts rows of the flattened [N*H*W][C] input.
// So this kernel stages, once per 32-channel chunk, the contiguous pixel range the tile's rows touch under all nine taps
// (BM + 2 W + 2 pixels: the "halo") and forms every tap's A fragments from it with a per-tap row offset; only the weight tile
// is staged per K-step.  Fill traffic per K-step falls from (BM + BN) x 128 B to about (BM / 6.5 + BN) x 128 B.
// Zero padding cannot come from the loader any more (the halo holds the neighbouring image row / image where the convolution
// wants zeros), so a fragment whose (row, tap) falls outside the image is zeroed in registers from a 9-bit mask per row.
//
// Schedule: the 8-wave ping-pong of sgemm.hip's PP loop (waves w and w + 4 share a SIMD; one half multiplies from registers
// while the other reads its fragments and issues its share of the loads), K order = channel chunk outer, tap inner.
//
// Replaces: Keras Conv2D (3x3, padding 'same') and its input gradient inside tf.keras.applications ResNet-50 / -101
// (reference backbone.py:37-38, 57).
#include "gemm_common.h"
#include <type_traits>
#include <utility>

using namespace bdgemm;

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;           // byte offset beyond num_records: the load writes zeros

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned records) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane((int)records), 0x00020000);
}

struct HInput { const void* p; int N, H, W, C, M; };        // P16 [N*H*W][C]; M = N*H*W = output rows
struct HWeight { const void* p; unsigned ld; int rows; };   // P16 [J][9 * C], reduction index = tap * C + channel

constexpr int LDS_TOTAL = 160 * 1024;
constexpr int halo_cap(int bn) {                            // pixels per halo buffer: two buffers + the 3-stage weight ring + a 1 KiB sink
    const int px = (LDS_TOTAL - 3 * bn * 128 - 1024) / 2 / 128 / 8 * 8;
    return px > 512 ? 512 : px;                             // 8 pieces x 8 waves x 8 pixels per chunk
}

// BM x BN output tile, 8 waves as 4 (rows) x 2 (columns); waves 0-3 = group X (upper half of the rows), 4-7 = group Y.
template <int BM, int BN, bool F16>
__global__ __launch_bounds__(512, 2)
void hconv_kernel(HInput xa, HWeight wb, GemmParams g)
{
    constexpr int NW = 8, WM = 4, WN = 2, NT = 512;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    constexpr int B_BYTES = BN * 128, NIB = B_BYTES / 1024 / NW;
    static_assert(NIB >= 1 && NIB * NW * 1024 == B_BYTES, "weight tile / wave count mismatch");
    constexpr int HALO_CAP = halo_cap(BN), HALO_BYTES = HALO_CAP * 128;
    constexpr int OFF_B = 2 * HALO_BYTES, OFF_SINK = OFF_B + 3 * B_BYTES, LDS_BYTES = OFF_SINK + 1024;
    static_assert(LDS_BYTES <= LDS_TOTAL && BM * BN * 4 <= LDS_BYTES, "LDS budget (staging, and the C tile of the epilogue)");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const bool grpY = wave >= NW / 2;

    const int nwg = g.tiles_i * g.tiles_j;
    const int lin = xcd_tile((int)blockIdx.x, nwg);
    const int tile_i = lin / g.tiles_j, tile_j = lin - tile_i * g.tiles_j;
    const int i0 = tile_i * BM, j0 = tile_j * BN;

    const int W = xa.W, C = xa.C;
    const int nchunks = C / 32, nk = 9 * nchunks;
    const unsigned c4 = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(xa.p, (unsigned)xa.M * c4);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(wb.p, (unsigned)wb.rows * wb.ld * 4u);

    // ---------------- staging descriptors ----------------
    // A lane fetches 16 bytes: chunk q of the 128-byte (32 channels x hi/lo) segment of one pixel / weight row.  Row x of an LDS
    // image is swizzled by (x >> 1) & 7; a wave-load covers 8 consecutive rows starting at a multiple of 8, so the lane's source
    // chunk is (lane & 7) ^ (((load index & 1) << 2) | (lane >> 4)), and the load index has the wave's parity in both images.
    const int qsrc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    // halo: piece ph of this wave = wave-load ph * 8 + wave = halo pixels 8 l .. 8 l + 7; halo pixel 0 = input pixel i0 - W - 1
    const int halo_px = BM + 2 * W + 2;
    // (computed per issue from one per-lane row and the wave-uniform piece index: eight precomputed offsets per lane cost
    // the 256 x 128 flavour its last registers)
    const int hrow0 = i0 - W - 1 + 8 * wave + (lane >> 3);   // input pixel of the lane in piece 0 (negative above the first image)
    const unsigned hq = 16u * (unsigned)qsrc;
    unsigned woff[NIB];                                   // weight rows j0 + x; rows past J lie beyond the buffer descriptor
#pragma unroll
    for (int t = 0; t < NIB; ++t) {
        const int x = (t * NW + wave) * 8 + (lane >> 3);
        woff[t] = (unsigned)(j0 + x) * wb.ld * 4u + 16u * (unsigned)qsrc;
    }
    auto issue_halo = [&](int chunk, int ph) {            // piece ph of `chunk` into halo buffer chunk & 1
        const int l = ph * 8 + wave, row = hrow0 + 64 * ph;
        const bool piece = 8 * l < halo_px && chunk < nchunks;                // (wave-uniform) else: a load that only keeps the counts uniform ...
        const bool ok = piece && (unsigned)row < (unsigned)xa.M;
        const unsigned vo = ok ? (unsigned)row * c4 + hq + (unsigned)chunk * 128u : OOB;
        const int dst = piece ? (chunk & 1) * HALO_BYTES + l * 1024 : OFF_SINK;    // ... and must not land in a live halo buffer
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(lds + dst), 16, (int)vo, 0, 0, 0);
    };
    auto issue_w = [&](int t, auto buf_c) {               // weight stage of K-step t into ring buffer buf
        constexpr int buf = decltype(buf_c)::value;
        const int chunk = t / 9, tap = t - 9 * chunk;
        const unsigned r0 = ((unsigned)tap * (unsigned)C + (unsigned)chunk * 32u) * 4u;
#pragma unroll
        for (int k = 0; k < NIB; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(lds + OFF_B + buf * B_BYTES + (k * NW + wave) * 1024), 16,
                                                     (int)(t < nk ? woff[k] + r0 : OOB), 0, 0, 0);
    };

    // ---------------- per-row tap masks (zero padding) ----------------
    unsigned rmask[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int i = i0 + wm * WTM + a * 32 + li;
        rmask[a] = 0;
        if (i < xa.M) {
            const int hw = xa.H * W, rem = i % hw, oh = rem / W, ow = rem - oh * W;
            unsigned colbits = 0;
#pragma unroll
            for (int ts = 0; ts < 3; ++ts) colbits |= ((unsigned)(ow - 1 + ts) < (unsigned)W ? 1u : 0u) << ts;
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) if ((unsigned)(oh - 1 + tr) < (unsigned)xa.H) rmask[a] |= colbits << (3 * tr);
        }
    }

    // ---------------- accumulators ----------------
    f32x16 acc[TM][TN];                                   // one set for all three split products, f16 pairs included (p16.h)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---------------- fragments ----------------
    u32x4 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
    // A fragments of K-step (chunk, tap): row x of the tile reads halo pixel x + tr * W + ts
    auto read_a = [&](int chunk, int tr, int ts) {
        const unsigned char* hb = lds + (chunk & 1) * HALO_BYTES;
        const int toff = tr * W + ts;
        const unsigned bit = 1u << (3 * tr + ts);
        // the lane's row is laundered through an empty asm: the addresses below are then recomputed per K-step (a handful of
        // VALU ops in the load segment) instead of being hoisted out of the chunk loop for all nine taps at once and held -
        // and spilled - across it (9 taps x TM rows x 2 halves x hi / lo)
        int row0 = wm * WTM + li;
        asm volatile("" : "+v"(row0));
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int p = row0 + a * 32 + toff;
            const bool on = (rmask[a] & bit) != 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int s = (2 * (2 * ks + lh)) ^ ((p >> 1) & 7);
                const unsigned char* q = hb + p * 128;
                u32x4 h = *reinterpret_cast<const u32x4*>(q + s * 16), l = *reinterpret_cast<const u32x4*>(q + (s ^ 1) * 16);
                if (!on) { h = u32x4{0, 0, 0, 0}; l = u32x4{0, 0, 0, 0}; }
                fah[ks][a] = h; fal[ks][a] = l;
            }
        }
    };
    auto read_b = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const unsigned char* tB = lds + OFF_B + buf * B_BYTES;
        int col0 = wn * WTN + li;
        asm volatile("" : "+v"(col0));
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int x = col0 + b * 32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int s = (2 * (2 * ks + lh)) ^ ((x >> 1) & 7);
                const unsigned char* q = tB + x * 128;
                fbh[ks][b] = *reinterpret_cast<const u32x4*>(q + s * 16);
                fbl[ks][b] = *reinterpret_cast<const u32x4*>(q + (s ^ 1) * 16);
            }
        }
    };
    auto mfma_step = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    if constexpr (F16) {
#define H8(v) __builtin_bit_cast(f16x8, v)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(fal[ks][a]), H8(fbh[ks][b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(fah[ks][a]), H8(fbl[ks][b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(fah[ks][a]), H8(fbh[ks][b]), acc[a][b], 0, 0, 0);
#undef H8
                    } else {
#define BF8(v) __builtin_bit_cast(bf16x8, v)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(fal[ks][a]), BF8(fbh[ks][b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(fah[ks][a]), BF8(fbl[ks][b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(fah[ks][a]), BF8(fbh[ks][b]), acc[a][b], 0, 0, 0);
#undef BF8
                    }
                }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---------------- main loop ----------------
    // K-step t = 9 * chunk + phi, phi = 3 * tr + ts the tap.  EVERY wave runs the same stream
    //     L(t): read my fragments of step t, issue my share of the loads   | barrier |   M(t): 24 MFMAs from registers   | barrier
    // but group Y (waves 4-7, the SIMD partners of waves 0-3) runs it ONE BARRIER LATE: while X multiplies step t, Y is in L(t);
    // while Y multiplies, X is in L(t + 1).  A wave's MFMA segment holds nothing but MFMAs, and its load segment (LDS reads,
    // LDS-DMA issue, address arithmetic) runs in the shadow of its partner's MFMAs.  Slots (barrier-delimited): X runs L(t) in
    // slot 2t and M(t) in slot 2t + 1, Y runs L(t) in slot 2t + 1 and M(t) in slot 2t + 2.
    // What L(t) issues (NLOAD = NIB + 1 loads per wave, a compile-time constant):
    //   * its share of the WEIGHT stage of step t + 2 into ring buffer (t + 2) % 3 = (ts + 2) % 3, whose previous tenant (step
    //     t - 1) was last read in Y's L(t - 1), slot 2t - 1 - before the first issue (X, slot 2t);
    //   * piece phi (< 8) of the HALO of chunk + 1 into halo buffer (chunk + 1) & 1, whose previous tenant (chunk - 1) was last
    //     read in Y's L(9 chunk - 1); phi = 8 issues a load that only keeps the count uniform (source out of range, into a sink).
    // Landing: both segments end with s_waitcnt vmcnt(NLOAD), which leaves only the wave's newest group in flight.  The weight
    // stage of step s (issued in L(s - 2)) is therefore complete in every wave by the end of slot 2s - 2 (X: end of L(s - 1);
    // Y: end of M(s - 2) ... L(s - 1)) - one barrier before its first reader, X's L(s) in slot 2s; likewise the last halo pieces
    // of chunk c + 1 (issued in L(9c + 7)) are complete by the end of each wave's L(9c + 8), before X's L(9c + 9).
    // Loads past the end of the problem are still issued (zeros into a buffer nobody reads any more).
    constexpr int NLOAD = NIB + 1;
    float bias_pre[TN];
    gemm_load_bias<BM, BN, WM, WN>(g, j0, bias_pre);
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int ph = 0; ph < 8; ++ph) issue_halo(0, ph);
    issue_w(0, I0{});
    issue_w(1, I1{});
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NIB) : "memory");
    if (grpY) asm volatile("s_barrier" ::: "memory");      // Y's one-barrier delay
    auto step = [&](int chunk, int tr, auto ts_c) {
        constexpr int TS = decltype(ts_c)::value;
        using B0 = std::integral_constant<int, TS>; using B2 = std::integral_constant<int, (TS + 2) % 3>;
        const int phi = 3 * tr + TS, t = 9 * chunk + phi;
        // ---- L(t)
        read_a(chunk, tr, TS); read_b(B0{});
        issue_w(t + 2, B2{});
        issue_halo(phi < 8 ? chunk + 1 : nchunks, phi & 7);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NLOAD) : "memory");
        // ---- M(t)
        mfma_step();
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NLOAD) : "memory");
    };
    for (int chunk = 0; chunk < nchunks; ++chunk)
        for (int tr = 0; tr < 3; ++tr) {                  // unrolled over ts: the weight ring index (t % 3 = ts) is a constant of each copy
            step(chunk, tr, std::integral_constant<int, 0>{});
            step(chunk, tr, std::integral_constant<int, 1>{});
            step(chunk, tr, std::integral_constant<int, 2>{});
        }
    if (!grpY) asm volatile("s_barrier" ::: "memory");     // X's matching extra barrier (Y is multiplying its last step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing out-of-range loads target LDS the epilogue is about to reuse
    gemm_epilogue<BM, BN, WM, WN, NT, LDS_BYTES / 4>(acc, g, reinterpret_cast<float*>(lds), tile_i, i0, j0, g.c, bias_pre);
}

template <int BM, int BN, bool F16>
int launch(const HInput& x, const HWeight& w, GemmParams g, hipStream_t st, int kind) {
    g.tiles_i = (int)cdiv64(g.I, BM);
    g.tiles_j = (int)cdiv64(g.J, BN);
    g.vec_store = (g.J % 4 == 0) && (g.ldc % 4 == 0) && aligned16(g.c);
    const bool prof = g_prof_on;
    if (prof) prof_begin(st, 2.0 * (double)g.I * (double)g.J * (double)g.R, g.I, g.J, g.R, 1, BM, BN * 10 + 3, (F16 ? AR_P16_F16 : AR_P16_BF16) * 10000 + kind);
    hipLaunchKernelGGL((hconv_kernel<BM, BN, F16>), dim3(g.tiles_i * g.tiles_j), dim3(512), 0, st, x, w, g);
    if (prof) prof_end(st);
    return bdetr_launch_status("hconv");
}

}  // namespace

namespace bdgemm {

// Tile of the halo kernel for a 3x3 / stride 1 / pad 1 convolution with `rows` output pixels of width W, C input and J output
// channels, or 0 when the launch must stay on sgemm.hip's im2col kernel.  Returned as BM * 1000 + BN.
int hconv_tile(int64_t rows, int W, int C, int J, bool f16) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("BDETR_HCONV"); enabled = e ? atoi(e) : 1; }
    if (!enabled) return 0;
    (void)f16;                                             // both flavours: one accumulator set since the f16 pair's lo half is unscaled (p16.h)
    if (C % 32 || J % 64 || rows >= (1LL << 31)) return 0;
    const int bn = J % 128 == 0 ? 128 : 64;
    const int64_t cus = num_cus();
    {   // A/B switch (tools/hconv_tile_ab.py, profiles/r04_hconv_tile_ab.json): BDETR_HCONV_TILE=256128 | 128128 | 256064 forces a tile where it fits
        static int forced = -1;
        if (forced < 0) { const char* e = getenv("BDETR_HCONV_TILE"); forced = e ? atoi(e) : 0; }
        if (forced) {
            const int fbm = forced / 1000, fbn = forced % 1000;
            if ((fbn == 128 || fbn == 64) && (fbm == 128 || fbm == 256) && !(fbm == 128 && fbn == 64) && J % fbn == 0 && fbm + 2 * W + 2 <= halo_cap(fbn)) return forced;
        }
    }
    // the bigger tile while it still gives most CUs a workgroup (one 8-wave workgroup per CU)
    int bm = cdiv64(rows, 256) * cdiv64(J, bn) * 10 >= cus * 7 ? 256 : 128;
    if (bm + 2 * W + 2 > halo_cap(bn)) bm = 128;
    if (bm + 2 * W + 2 > halo_cap(bn)) return 0;
    if (bm == 128 && bn == 64) return 0;                   // 6 MFMAs per half-step: not worth a barrier pair
    return bm * 1000 + bn;
}

// x: P16 [N,H,W,C]; w: P16 [J][9 * C]; the epilogue is GemmParams' (g.I = N*H*W, g.J = J, g.R = 9 * C set by the caller)
int hconv_launch(int tile, bool f16, const void* x, int N, int H, int W, int C, const void* w, int J, const GemmParams& g, hipStream_t st) {
    HInput xi{x, N, H, W, C, N * H * W};
    HWeight wi{w, (unsigned)(9 * C), J};
    const int kind = 4000;                                  // profiling class: halo-resident 3x3
    switch (tile) {
        case 256128: return f16 ? launch<256, 128, true>(xi, wi, g, st, kind) : launch<256, 128, false>(xi, wi, g, st, kind);
        case 128128: return f16 ? launch<128, 128, true>(xi, wi, g, st, kind) : launch<128, 128, false>(xi, wi, g, st, kind);
        case 256064: return f16 ? launch<256, 64, true>(xi, wi, g, st, kind) : launch<256, 64, false>(xi, wi, g, st, kind);
        default: bdetr_set_error("hconv_launch: no such tile %d", tile); return -1;
    }
}

}  // namespace bdgemm
