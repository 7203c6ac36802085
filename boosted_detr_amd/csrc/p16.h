// p16.h - the pre-split "P16" operand layout of sgemm.hip (see its header): device helpers that split eight
// consecutive fp32 values into the 32-byte group [8 x hi][8 x lo] and back.
#pragma once
#include "common.h"

typedef unsigned int p16_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p16_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 p16_f16x2 __attribute__((ext_vector_type(2)));
typedef float p16_f32x2 __attribute__((ext_vector_type(2)));

// f16 pair: hi = f16(x), lo = f16(x - hi), both UNSCALED, so that the three split products hi*hi + hi*lo + lo*hi share one
// accumulator (round 2 kept lo scaled by 2^11 in a second accumulator set: 64 more VGPRs per 128x128 tile, no room for wider
// wave tiles).  For |x| < 2^-3 the lo half is an f16 subnormal: its absolute error is then at most 2^-25 - fp32-grade relative to
// an operand tensor of RMS ~1, which BatchNorm outputs are.  v_mfma_f32_32x32x16_f16 keeps subnormal operands (measured on
// gfx950: tools/probes/mfma_f16_denorm.hip).  Weights are small (|w| ~ 1e-2): their forward copy holds 2^8 w, which puts its lo
// half back into the normal range; the convolution's epilogue multiplies by 2^-8.
constexpr float P16_W_SCALE = 256.f;
constexpr float P16_F16_LIMIT = 65504.f;        // |x| at or beyond this cannot be represented by the f16 pair

__device__ __forceinline__ void p16_split2_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    p16_f32x2 v; v[0] = x0; v[1] = x1;
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, p16_bf16x2));            // v_cvt_pk_bf16_f32 (RNE)
    p16_f32x2 r;
    r[0] = x0 - __builtin_bit_cast(float, hi << 16);                                      // exact in fp32
    r[1] = x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, p16_bf16x2));
}
__device__ __forceinline__ void p16_split2_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
    p16_f32x2 v; v[0] = x0; v[1] = x1;
    const p16_f16x2 h = __builtin_convertvector(v, p16_f16x2);                            // v_cvt_pk_f16_f32 (RNE)
    hi = __builtin_bit_cast(unsigned, h);
    const p16_f32x2 r = v - __builtin_convertvector(h, p16_f32x2);                        // residual: exact in fp32
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, p16_f16x2));
}

// eight consecutive elements (element index a multiple of 8) -> the group at `group` (32-byte aligned)
template <bool F16>
__device__ __forceinline__ void p16_store8(void* group, const float (&v)[8]) {
    p16_u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned h, l;
        if (F16) p16_split2_f16(v[2 * e], v[2 * e + 1], h, l);
        else     p16_split2_bf16(v[2 * e], v[2 * e + 1], h, l);
        hi[e] = h; lo[e] = l;
    }
    reinterpret_cast<p16_u32x4*>(group)[0] = hi;
    reinterpret_cast<p16_u32x4*>(group)[1] = lo;
}
template <bool F16>
__device__ __forceinline__ void p16_load8(const void* group, float (&v)[8]) {
    const p16_u32x4 hi4 = reinterpret_cast<const p16_u32x4*>(group)[0], lo4 = reinterpret_cast<const p16_u32x4*>(group)[1];
    // scalars first: __builtin_bit_cast applied directly to a vector element (hi4[e]) was folded to element 0 by
    // hipcc (ROCm 7.2) - every pair came out as pair 0
    const unsigned hw[4] = {hi4[0], hi4[1], hi4[2], hi4[3]}, lw[4] = {lo4[0], lo4[1], lo4[2], lo4[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned h = hw[e], l = lw[e];
        if (F16) {
            const p16_f32x2 hf = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, h), p16_f32x2);
            const p16_f32x2 lf = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, l), p16_f32x2);
            v[2 * e] = hf[0] + lf[0]; v[2 * e + 1] = hf[1] + lf[1];
        } else {
            v[2 * e] = __builtin_bit_cast(float, h << 16) + __builtin_bit_cast(float, l << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, h & 0xFFFF0000u) + __builtin_bit_cast(float, l & 0xFFFF0000u);
        }
    }
}
// true when any of the eight values is outside the f16 pair's range (or not finite)
__device__ __forceinline__ bool p16_f16_overflow(const float (&v)[8]) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));       // fmaxf drops NaNs: test them separately
    bool bad = !(m < P16_F16_LIMIT);
#pragma unroll
    for (int e = 0; e < 8; ++e) bad = bad || (v[e] != v[e]);
    return bad;
}

// ---- four consecutive elements per lane (element index a multiple of 4): lanes 2j / 2j+1 hold the two halves of
// one 8-element group.  Both lanes of a pair must be active.
__device__ __forceinline__ unsigned p16_swap_pair(unsigned v) {      // value of the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false);
}
// Writes the pair's 32-byte group with two fully coalesced 16-byte stores: the even lane stores the hi chunk, the odd
// lane the lo chunk, each at base + f4_index * 16 (the address a plain float4 store would use).
template <bool F16>
__device__ __forceinline__ void p16_store4(void* base, int64_t f4_index, float v0, float v1, float v2, float v3) {
    unsigned h01, l01, h23, l23;
    if (F16) { p16_split2_f16(v0, v1, h01, l01); p16_split2_f16(v2, v3, h23, l23); }
    else     { p16_split2_bf16(v0, v1, h01, l01); p16_split2_bf16(v2, v3, h23, l23); }
    const bool odd = (f4_index & 1) != 0;
    const unsigned ra = p16_swap_pair(odd ? h01 : l01), rb = p16_swap_pair(odd ? h23 : l23);   // even lane receives the partner's hi, odd its lo
    p16_u32x4 w;
    if (odd) { w[0] = ra; w[1] = rb; w[2] = l01; w[3] = l23; }
    else     { w[0] = h01; w[1] = h23; w[2] = ra; w[3] = rb; }
    reinterpret_cast<p16_u32x4*>(base)[f4_index] = w;
}
// the lane's four values back from an f16 pair tensor
__device__ __forceinline__ void p16_load4_f16(const void* base, int64_t f4_index, float (&v)[4]) {
    const char* g = reinterpret_cast<const char*>(base) + (f4_index >> 1) * 32 + (f4_index & 1) * 8;
    const unsigned h0 = reinterpret_cast<const unsigned*>(g)[0], h1 = reinterpret_cast<const unsigned*>(g)[1];
    const unsigned l0 = reinterpret_cast<const unsigned*>(g + 16)[0], l1 = reinterpret_cast<const unsigned*>(g + 16)[1];
    const p16_f32x2 a = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, h0), p16_f32x2), b = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, h1), p16_f32x2);
    const p16_f32x2 c = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, l0), p16_f32x2), d = __builtin_convertvector(__builtin_bit_cast(p16_f16x2, l1), p16_f32x2);
    v[0] = a[0] + c[0]; v[1] = a[1] + c[1];
    v[2] = b[0] + d[0]; v[3] = b[1] + d[1];
}
// bit e set when element e of the lane's four is > 0 in a bf16 pair tensor (the hi half decides: bf16 keeps fp32's
// exponent range, so hi > 0 <=> x > 0 for every normal x)
__device__ __forceinline__ unsigned p16_positive4_bf16(const void* base, int64_t f4_index) {
    const char* g = reinterpret_cast<const char*>(base) + (f4_index >> 1) * 32 + (f4_index & 1) * 8;
    const unsigned h0 = reinterpret_cast<const unsigned*>(g)[0], h1 = reinterpret_cast<const unsigned*>(g)[1];
    auto pos = [](unsigned h16) { return (h16 & 0x8000u) == 0u && (h16 & 0x7FFFu) != 0u; };
    return (pos(h0 & 0xFFFFu) ? 1u : 0u) | (pos(h0 >> 16) ? 2u : 0u) | (pos(h1 & 0xFFFFu) ? 4u : 0u) | (pos(h1 >> 16) ? 8u : 0u);
}
