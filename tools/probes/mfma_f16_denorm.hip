// Does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL operands (or flush them to zero)?  Decides whether the lo half of an f16 pair
// can be stored unscaled (one accumulator for all three split products).  Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const _Float16* a, const _Float16* b, float* out) {
    f16x8 va, vb;
    for (int j = 0; j < 8; ++j) { va[j] = a[0]; vb[j] = b[0]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
    // VALU reference: the same product through v_fma_mix / cvt (default denorm mode of a HIP kernel)
    if (threadIdx.x == 0) out[1] = 16.f * (float)a[0] * (float)b[0];
}

int main() {
    _Float16 *da, *db; float* dout;
    hipMalloc(&da, 2); hipMalloc(&db, 2); hipMalloc(&dout, 8);
    const float as[] = {5.9604645e-8f /* 2^-24: smallest subnormal */, 9.5367432e-7f /* 2^-20 */, 3.0517578e-5f /* 2^-15: largest-ish subnormal */, 6.1035156e-5f /* 2^-14: smallest normal */};
    int flushed = 0;
    for (float av : as) {
        _Float16 ha = (_Float16)av, hb = (_Float16)1024.f;
        hipMemcpy(da, &ha, 2, hipMemcpyHostToDevice); hipMemcpy(db, &hb, 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
        float o[2]; hipMemcpy(o, dout, 8, hipMemcpyDeviceToHost);
        const float want = 16.f * av * 1024.f;
        printf("a = %.9g  (f16 %s)  mfma %.9g  valu %.9g  want %.9g  %s\n", av, av < 6.1035156e-5f ? "subnormal" : "normal", o[0], o[1], want, o[0] == want ? "kept" : "FLUSHED/WRONG");
        if (o[0] != want) flushed = 1;
    }
    printf(flushed ? "MFMA_F16_DENORM: FLUSHED\n" : "MFMA_F16_DENORM: KEPT\n");
    return 0;
}
