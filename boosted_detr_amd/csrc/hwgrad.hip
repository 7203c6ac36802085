// hwgrad.hip - weight gradient of a 3x3 / stride-1 / pad-1 convolution on pre-split (P16, bf16 pairs) operands with BOTH operands
// resident in LDS as sliding pixel windows, for gfx950.  The transposed analogue of hconv.hip (round 3's halo-resident forward /
// backward-data kernel).
//
//     dW[k][tr][ts][c] = sum over output pixels p of  dy[p][k] * x[p + (tr - 1) W + (ts - 1)][c]      (zero outside the image)
//
// sgemm.hip runs this as an implicit GEMM over an im2col view of x (columns = (tap, channel)): every 128-column tile of the 9 C
// columns re-stages the dy tile, and every tap re-stages the x pixels it reads - the kernel is bound by the L2 -> LDS fill path
// (~27 B/clk/CU), at 0.30 of the 3-product MFMA roof (round 3).  Here a workgroup owns a (128 output channels) x (64 input
// channels) block of dW FOR ALL NINE TAPS and streams a contiguous range of pixels through LDS once: a ring of dy stages and a
// ring of x stages that runs W + 3 pixels ahead of and behind the dy window, so that the nine taps are nine row offsets into the
// same x image.  Staged bytes per 32 pixels: 24 KB for 1,728 MFMAs (the im2col form: 32 KB for 192).
//
// Zero padding without masks: the kernel works in PADDED pixel coordinates q over an (H + 2) x (W + 2) frame per image.  The
// LDS-DMA source of a frame position that is padding is an out-of-range buffer offset (the load then writes zeros), for dy and
// for x alike - so a tap that leaves the image multiplies by a zero x row, a padding position contributes a zero dy row, and the
// reduction simply runs over q (4-21 % longer than over p, no per-pixel masks, no divisions outside the load address path).
//
// MFMA roles: D[i = k][j = c] += A[i][r] B[r][j] with r = pixel: both operands have the reduction index STRIDED in memory, so both
// fragments come from ds_read_b64_tr_b16 (sgemm.hip's "XX" LDS image: rows = pixels, 16-byte slots XOR-ed with xx_swz(row)).
// 8 waves = 4 (k) x 2 (c) tiles of 32 x 32, nine accumulators (one per tap) per wave.
//
// Replaces: the kernel gradient of Keras Conv2D (3x3, padding 'same') inside tf.keras.applications ResNet-50 / -101 (reference
// backbone.py:37-38, 57 + autodiff).
#include "gemm_common.h"
#include "p16.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

using namespace bdgemm;

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t hw_rsrc(const void* p, unsigned records) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane((int)records), 0x00020000);
}
__device__ __forceinline__ int hw_swz(int r) { return ((r & 1) << 3) | ((r >> 1) & 1); }       // = sgemm.hip xx_swz

constexpr int BKO = 128, BCI = 64;          // output-channel x input-channel block of one workgroup
constexpr int SP = 32;                      // pixels per stage
constexpr int PFD = 2;                      // stages in flight
constexpr int NAS = PFD + 1;                // dy ring stages
constexpr int A_STAGE = SP * BKO * 4, X_STAGE = SP * BCI * 4;
constexpr int MAX_XL = 96;                  // round_up_32(W + 3): W <= 93
constexpr int MAX_NXS = 2 * MAX_XL / SP + PFD + 1;
constexpr int LDS_BYTES = NAS * A_STAGE + MAX_NXS * X_STAGE;      // 48 KB + 72 KB

struct HwArgs {
    const void* dy; const void* x;          // P16-bf16 [M][K], [M][C]
    int N, H, W, C, K;
    int PW, PH, PHW;                        // padded frame: W + 2, H + 2, PH * PW
    int XL, NXS;                            // x runs XL = round_up_32(W + 3) padded pixels ahead of / behind dy; x ring stages = 2 XL / 32 + PFD + 1
    int stages_total, stages_per_slice;     // 32-pixel stages of the padded pixel range, and per blockIdx.z
    int tiles_c;
    float* dw; int ldw;                     // fp32 [K][9 C]
    int store_slabs; long long slab;        // deterministic mode: slice z stores into dw + z * slab instead of adding
};

// (n, row, col) of a padded pixel index, advanced by 32 per stage with adds only
struct PadPos { int n, row, col; };
__device__ __forceinline__ PadPos pad_pos(int q, int PW, int PHW) {
    PadPos s;
    int n = q / PHW, rem = q - n * PHW;
    if (rem < 0) { rem += PHW; n -= 1; }
    s.n = n; s.row = rem / PW; s.col = rem - s.row * PW;
    return s;
}
__device__ __forceinline__ void pad_advance(PadPos& s, int PW, int PH) {      // branch-free: PW >= 16 (at most two column wraps per 32 pixels), PH >= 3
    s.col += SP;
    bool w = s.col >= PW; s.col -= w ? PW : 0; s.row += w ? 1 : 0;
    w = s.col >= PW;      s.col -= w ? PW : 0; s.row += w ? 1 : 0;
    w = s.row >= PH;      s.row -= w ? PH : 0; s.n += w ? 1 : 0;
}
// byte offset of the pixel's channel row in a [M][ch] P16 tensor, or OOB where the frame position is padding / outside the batch
__device__ __forceinline__ unsigned pad_src(const PadPos& s, const HwArgs& a, unsigned ch4, unsigned tail) {
    const bool ok = ((unsigned)s.n < (unsigned)a.N) & ((unsigned)(s.row - 1) < (unsigned)a.H) & ((unsigned)(s.col - 1) < (unsigned)a.W);     // (&: no short-circuit branches)
    const unsigned p = (unsigned)((s.n * a.H + s.row - 1) * a.W + s.col - 1);
    return ok ? p * ch4 + tail : OOB;
}

__global__ __launch_bounds__(512, 1) void hwgrad_kernel(HwArgs a) {
    __shared__ __attribute__((aligned(64))) unsigned char lds[LDS_BYTES];
    unsigned char* const aring = lds;
    unsigned char* const xring = lds + NAS * A_STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wc = wave & 1, lh = lane >> 5;
    const int tile_k = (int)blockIdx.x / a.tiles_c, tile_c = (int)blockIdx.x - tile_k * a.tiles_c;
    const int k0 = tile_k * BKO, c0 = tile_c * BCI;
    const int s_begin = (int)blockIdx.z * a.stages_per_slice, s_end = min(a.stages_total, s_begin + a.stages_per_slice);
    const int nst = s_end - s_begin;
    if (nst <= 0) return;
    const int qa = s_begin * SP, xa = qa - a.XL;                     // first padded pixel of the dy / x windows
    const int XS = 2 * a.XL / SP, NXS = a.NXS, RX = NXS * SP;        // x stage u + XS must have landed before dy stage u is multiplied
    const unsigned K4 = (unsigned)a.K * 4u, C4 = (unsigned)a.C * 4u;
    const __amdgpu_buffer_rsrc_t rsA = hw_rsrc(a.dy, (unsigned)(a.N * a.H * a.W) * K4);
    const __amdgpu_buffer_rsrc_t rsX = hw_rsrc(a.x, (unsigned)(a.N * a.H * a.W) * C4);

    // ---------------- staging state ----------------
    // dy stage: 32 rows x 512 B = 16 wave-loads, wave w issues loads w and w + 8 (rows 2 l, 2 l + 1; 32 lanes per row)
    // x  stage: 32 rows x 256 B =  8 wave-loads, wave w issues load w          (rows 4 w .. 4 w + 3; 16 lanes per row)
    PadPos pa[2], px;
    unsigned tailA[2], tailX;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = 2 * (wave + 8 * t) + (lane >> 5);
        pa[t] = pad_pos(qa + r, a.PW, a.PHW);
        tailA[t] = (unsigned)k0 * 4u + 16u * (unsigned)((lane & 31) ^ hw_swz(r));
    }
    {
        const int r = 4 * wave + (lane >> 4);
        px = pad_pos(xa + r, a.PW, a.PHW);
        tailX = (unsigned)c0 * 4u + 16u * (unsigned)((lane & 15) ^ hw_swz(r));
    }
    int a_issued = 0, x_issued = 0;                                  // stages issued so far (ring position = count mod ring size)
    int a_slot = 0, x_slot = 0;
    auto issue_a = [&]() {                                           // dy stage a_issued (zeros beyond the slice: nobody multiplies them)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned vo = a_issued < nst ? pad_src(pa[t], a, K4, tailA[t]) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(aring + a_slot * A_STAGE + (wave + 8 * t) * 1024), 16, (int)vo, 0, 0, 0);
            pad_advance(pa[t], a.PW, a.PH);
        }
        a_issued += 1; a_slot = a_slot + 1 == NAS ? 0 : a_slot + 1;
    };
    auto issue_x = [&]() {
        const unsigned vo = x_issued < nst + XS ? pad_src(px, a, C4, tailX) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(xring + x_slot * X_STAGE + wave * 1024), 16, (int)vo, 0, 0, 0);
        pad_advance(px, a.PW, a.PH);
        x_issued += 1; x_slot = x_slot + 1 == NXS ? 0 : x_slot + 1;
    };

    // ---------------- fragment addressing ----------------
    // ds_read_b64_tr_b16 (sgemm.hip frag_xx): lanes 16 g .. 16 g + 15 read a 4-row x 16-column block; lane 4 ql + p supplies row ql,
    // columns 4 p .. 4 p + 3 and receives column (lane & 15) of the four rows.  Two reads per half give 8 consecutive pixels.
    const int g16 = (lane >> 4) & 1, ql = (lane >> 2) & 3, pp = lane & 3;
    const int laneoff0 = 8 * lh + ql, laneoff1 = laneoff0 + 4;
    const int unitA = (32 * wk + 16 * g16 + 4 * pp) >> 3, unitX = (32 * wc + 16 * g16 + 4 * pp) >> 3;
    // dy: rows never wrap inside a stage; row parity bits = (16 ks + laneoff) & 3 = ql
    const int colA = (((2 * unitA) ^ hw_swz(ql)) << 4) + 8 * (pp & 1);
    // x: the swizzle of ring row (rel + laneoff) depends on (rel + ql) & 3, and rel & 3 = off(tap) & 3 (everything else is a multiple of 4)
    int colX[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int off = (tap / 3 - 1) * a.PW + (tap % 3 - 1);
        colX[tap] = (((2 * unitX) ^ hw_swz((off + ql) & 3)) << 4) + 8 * (pp & 1);
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // LDS addresses as integer byte offsets from the array's LDS-space base: a pointer that went through integer arithmetic comes back
    // as a GENERIC pointer, and every read then pays the flat -> LDS address-space cast (null / aperture checks, branches)
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_u8* const lbase = (lds_u8*)lds;
    auto tr_read = [&](int off) -> u32x2 {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lbase + off)));
    };
    constexpr int XOFF = NAS * A_STAGE;

    // ---------------- prologue ----------------
    // x stages 0 .. XS + PFD - 1 and dy stages 0 .. PFD - 1; per stage every wave issues 3 loads (2 dy + 1 x), the x-only ones 1
    for (int u = 0; u < XS; ++u) issue_x();
#pragma unroll
    for (int u = 0; u < PFD; ++u) { issue_a(); issue_x(); }

    // ---------------- main loop ----------------
    int a_read = 0;                                                  // ring slot of dy stage s
    int relbase = a.XL;                                              // (xa-relative) padded pixel of dy stage s's first row, modulo the x ring
    for (int s = 0; s < nst; ++s) {
        // stage s of dy and stage s + XS of x have landed in every wave: all but the newest PFD - 1 groups of 3 loads are complete
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(3 * (PFD - 1)) : "memory");
        // the slots these loads overwrite were last read in stage s - 1, before the barrier above
        issue_a(); issue_x();
        const int tA = a_read * A_STAGE;
        // Fragments.  The compiler issues an LDS read right in front of its first use (TR TR s_waitcnt lgkmcnt(0) MFMA: the whole read
        // latency exposed once per tap - measured 0.22 of the MFMA roof), so the order is pinned by hand: the reads of step i + 1 are
        // issued BEFORE the three MFMAs of step i (scheduling barriers), into the other half of a two-deep fragment buffer; the
        // waitcnt the compiler then needs in front of step i's MFMAs leaves step i + 1's four reads in flight.
        u32x4 ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r0 = tA + (16 * ks + laneoff0) * (BKO * 4) + colA;
            const int r1 = tA + (16 * ks + laneoff1) * (BKO * 4) + colA;
            const u32x2 h0 = tr_read(r0), l0 = tr_read(r0 ^ 16);
            const u32x2 h1 = tr_read(r1), l1 = tr_read(r1 ^ 16);
            ah[ks][0] = h0[0]; ah[ks][1] = h0[1]; ah[ks][2] = h1[0]; ah[ks][3] = h1[1];
            al[ks][0] = l0[0]; al[ks][1] = l0[1]; al[ks][2] = l1[0]; al[ks][3] = l1[1];
        }
        constexpr int BD = 3;                                        // fragment buffers: reads run BD - 1 steps ahead of the MFMAs
        u32x4 bh[BD], bl[BD];
        auto read_b = [&](int ks, int tap, u32x4& h, u32x4& l) {
            const int off = (tap / 3 - 1) * a.PW + (tap % 3 - 1);
            int rel = relbase + 16 * ks + off;                       // wave-uniform; -RX < rel < 2 RX
            rel = rel < 0 ? rel + RX : (rel >= RX ? rel - RX : rel);
            int row0 = rel + laneoff0, row1 = rel + laneoff1;
            row0 = min((unsigned)row0, (unsigned)(row0 - RX));       // wrap: row - RX underflows to a huge value when row < RX
            row1 = min((unsigned)row1, (unsigned)(row1 - RX));
            const int r0 = XOFF + row0 * (BCI * 4) + colX[tap];
            const int r1 = XOFF + row1 * (BCI * 4) + colX[tap];
            const u32x2 h0 = tr_read(r0), l0 = tr_read(r0 ^ 16);
            const u32x2 h1 = tr_read(r1), l1 = tr_read(r1 ^ 16);
            h[0] = h0[0]; h[1] = h0[1]; h[2] = h1[0]; h[3] = h1[1];
            l[0] = l0[0]; l[1] = l0[1]; l[2] = l1[0]; l[3] = l1[1];
        };
#pragma unroll
        for (int i = 0; i < BD - 1; ++i) read_b(i / 9, i % 9, bh[i], bl[i]);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int ks = i / 9, tap = i % 9;
            if (i + BD - 1 < 18) read_b((i + BD - 1) / 9, (i + BD - 1) % 9, bh[(i + BD - 1) % BD], bl[(i + BD - 1) % BD]);
            __builtin_amdgcn_sched_barrier(0);
#define BF8(v) __builtin_bit_cast(bf16x8, v)
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(al[ks]), BF8(bh[i % BD]), acc[tap], 0, 0, 0);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[ks]), BF8(bl[i % BD]), acc[tap], 0, 0, 0);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(ah[ks]), BF8(bh[i % BD]), acc[tap], 0, 0, 0);
#undef BF8
            __builtin_amdgcn_sched_barrier(0);
        }
        a_read = a_read + 1 == NAS ? 0 : a_read + 1;
        relbase += SP; if (relbase >= RX) relbase -= RX;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the trailing out-of-range loads still target this workgroup's LDS

    // ---------------- epilogue: dW[k0 + 32 wk + row][tap][c0 + 32 wc + col] ----------------
    float* base = a.dw + (a.store_slabs ? (long long)blockIdx.z * a.slab : 0ll);
    const unsigned ldw4 = (unsigned)a.ldw * 4u;
    const __amdgpu_buffer_rsrc_t rsC = hw_rsrc(base, (unsigned)a.K * ldw4);
    const unsigned rowoff = (unsigned)(k0 + 32 * wk + 4 * lh) * ldw4 + (unsigned)(c0 + 32 * wc + (lane & 31)) * 4u;
    auto out = [&](auto slab_c) {
        constexpr bool SLAB = decltype(slab_c)::value;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const unsigned tb = rowoff + (unsigned)tap * C4;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[tap][e];                         // (a scalar copy first: see sgemm.hip direct_epilogue)
                const unsigned off = tb + (unsigned)((e & 3) + 8 * (e >> 2)) * ldw4;
                if constexpr (SLAB) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, off, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rsC, off, 0, 0);
            }
        }
    };
    if (a.store_slabs) out(std::true_type{}); else out(std::false_type{});
}

}  // namespace

namespace bdgemm {

// Is the halo-resident weight-gradient kernel applicable, and with how many pixel slices?  0: stay on sgemm.hip's im2col kernel.
int hwgrad_slices(int N, int H, int W, int C, int K) {
    // OFF by default (BDETR_HWGRAD=1 switches it on).  Measured at batch 16 (tools/p16_bench.py, profiles/r04_hwgrad_ab.json), ms per launch:
    //                  im2col + atomics   this kernel + atomics   im2col + slabs + fold   this kernel + slabs + fold
    //   80x80x128            0.108               0.149                    0.122                     0.129
    //   40x40x256            0.103               0.144                    0.107                     0.108
    //   20x20x512            0.109               0.136                    0.113                     0.113
    // Staging traffic is 6x lower and results are bit-for-bit within the bf16-pair tolerance, but (a) a (128 x 64 x 9-tap) block per
    // workgroup and one workgroup per CU mean 32 pixel slices: 256 x 295 KB = 75 MB of float atomics per launch (58 us at the chip's
    // 1.3 TB/s atomic rate; the im2col kernel's 14 slices send 33 MB), and (b) its main loop runs at about twice the MFMA bound (40
    // transposing LDS reads with per-lane ring addresses per 27 MFMAs).  With plain slab stores + a fold launch it only draws level.
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("BDETR_HWGRAD"); enabled = e ? atoi(e) : 0; }
    if (!enabled || K % BKO || C % BCI) return 0;
    const int XL = (W + 3 + SP - 1) / SP * SP;
    if (XL > MAX_XL) return 0;
    const long long Mp = (long long)N * (H + 2) * (W + 2);
    if (Mp >= (1ll << 30) || (long long)N * H * W * (K > C ? K : C) * 4 >= (1ll << 32) - 256) return 0;
    const int stages = (int)cdiv64(Mp, SP);
    const int tiles = (K / BKO) * (C / BCI);
    int slices = num_cus() / tiles;                               // one 8-wave workgroup per CU
    const int min_stages = 2 * (2 * XL / SP) + 4;                 // a slice's x prologue (2 XL pixels) must stay a small part of its work
    if (slices > stages / min_stages) slices = stages / min_stages;
    return slices < 1 ? 1 : slices;
}

int hwgrad_launch(const void* x_bf16, const void* dy_bf16, float* dw, int N, int H, int W, int C, int K, int slices, float* slabs, long long slab, int* zdim_out, hipStream_t st) {
    HwArgs a;
    a.dy = dy_bf16; a.x = x_bf16; a.N = N; a.H = H; a.W = W; a.C = C; a.K = K;
    a.PW = W + 2; a.PH = H + 2; a.PHW = a.PW * a.PH;
    a.XL = (W + 3 + SP - 1) / SP * SP; a.NXS = 2 * a.XL / SP + PFD + 1;
    a.stages_total = (int)cdiv64((long long)N * a.PHW, SP);
    a.stages_per_slice = (int)cdiv64(a.stages_total, slices);
    const int zdim = (int)cdiv64(a.stages_total, a.stages_per_slice);
    a.tiles_c = C / BCI;
    a.dw = slabs ? slabs : dw; a.ldw = 9 * C; a.store_slabs = slabs != nullptr; a.slab = slab;
    const bool prof = g_prof_on;
    if (prof) prof_begin(st, 2.0 * (double)K * 9.0 * (double)C * (double)N * H * W, K, 9 * C, N * H * W, zdim, BKO, BCI * 10 + 5, AR_P16_BF16 * 10000 + 5000);
    hipLaunchKernelGGL(hwgrad_kernel, dim3((K / BKO) * (C / BCI), 1, zdim), dim3(512), 0, st, a);
    if (prof) prof_end(st);
    if (zdim_out) *zdim_out = zdim;
    return bdetr_launch_status("hwgrad");
}

}  // namespace bdgemm
