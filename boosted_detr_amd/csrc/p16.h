// p16.h - the pre-split "P16" operand layout of sgemm.hip (see its header): device helpers that split eight
// consecutive fp32 values into the 32-byte group [8 x hi][8 x lo] and back.
#pragma once
#include "common.h"

typedef unsigned int p16_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p16_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 p16_f16x2 __attribute__((ext_vector_type(2)));
typedef float p16_f32x2 __attribute__((ext_vector_type(2)));

constexpr float P16_LO_SCALE = 2048.f;          // the f16 lo half is stored scaled by 2^11 (keeps it out of f16 subnormals)
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
    const p16_f32x2 r = (v - __builtin_convertvector(h, p16_f32x2)) * P16_LO_SCALE;       // residual exact, then 2^11
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
            v[2 * e] = hf[0] + lf[0] * (1.f / P16_LO_SCALE); v[2 * e + 1] = hf[1] + lf[1] * (1.f / P16_LO_SCALE);
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
