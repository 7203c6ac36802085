// Shared host/device helpers for libbdetr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/bdetr.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void bdetr_set_error(const char* fmt, ...);

#define BDETR_CHECK_ARG(cond, ...)                     \
    do {                                               \
        if (!(cond)) {                                 \
            bdetr_set_error(__VA_ARGS__);              \
            return -1;                                 \
        }                                              \
    } while (0)

static inline int bdetr_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        bdetr_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zero `bytes` bytes at p on `st` with a kernel of this library (elementwise.hip).  Not hipMemsetAsync: a memset NODE inside a
// hipGraph that is relaunched went wrong on ROCm 7.2 with the runtime's pre-built packet path (the first tensor whose
// fingerprint differed between an eager step and its replay was the one tile_batch zero-fills) - a kernel node is what every other
// operation of the step already is.  BDETR_ZERO_MEMSET=1 switches back (A/B).
int bdetr_zero_bytes(void* p, size_t bytes, hipStream_t st);

// grid size for grid-stride elementwise kernels: enough blocks to fill 256 CUs x 8
static inline int ew_grid(int64_t n, int block = 256, int per_thread = 4) {
    int64_t g = cdiv64(n, (int64_t)block * per_thread);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
