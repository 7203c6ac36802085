// optim.hip - fused multi-tensor optimizer step for gfx950 (HBM-bound).
//
// Replaces the Keras optimizer the notebooks compile the model with
// (DETR_COCO.ipynb cell 26: SGD(momentum=.9, nesterov=True, clipnorm=.1)), SURVEY S15:
//   g  <- g * grad_scale ; g <- g * min(1, clipnorm/||g||_2)        (per tensor)
//   v  <- m*v - lr*g ;  w <- w + m*v - lr*g
#include "common.h"

namespace {

struct TensorTriple { float* w; float* g; float* v; };

// one workgroup per (tensor, slab): deterministic per-slab partial sums of g^2
constexpr int SLAB = 16384;

__global__ __launch_bounds__(256) void sqnorm_kernel(const uint64_t* __restrict__ ptrs, const int64_t* __restrict__ sizes,
                                                     const int64_t* __restrict__ slab_tensor, const int64_t* __restrict__ slab_first,
                                                     float* __restrict__ partial) {
    __shared__ float sh[4];
    const int64_t t = slab_tensor[blockIdx.x];
    const int64_t off = (blockIdx.x - slab_first[t]) * (int64_t)SLAB;
    const float* g = reinterpret_cast<const float*>(ptrs[3 * t + 1]);
    const int64_t n = sizes[t];
    float s = 0.f;
    // 16-byte loads over the slab's whole quads (every tensor's slot in the flat buffers is 16-byte aligned and a slab starts at a
    // multiple of 16,384 elements), the last 1-3 elements one by one: the 4-byte form ran at 2.1 TB/s (round 5)
    const int64_t end = min(n, off + SLAB), nq = (reinterpret_cast<uintptr_t>(g) & 15) == 0 ? (end - off) >> 2 : 0;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + off);
    for (int64_t q = threadIdx.x; q < nq; q += 256) { const f32x4 x = g4[q]; s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]; }
    for (int64_t i = off + 4 * nq + threadIdx.x; i < end; i += 256) { float x = g[i]; s += x * x; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// guard (optional): raised when a tensor's gradient norm is not finite - a NaN / Inf born in the backward pass, which the loss
// check of the range guard cannot see; sgd_apply_kernel (the next launch) then applies nothing at all.
__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, const int64_t* __restrict__ slab_first, int ntensors,
                                                         float* __restrict__ norms, int* __restrict__ guard) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntensors) return;
    double s = 0;
    for (int64_t k = slab_first[t]; k < slab_first[t + 1]; ++k) s += partial[k];
    const float nrm = (float)sqrt(s);
    norms[t] = nrm;
    if (guard != nullptr && !(fabsf(nrm) <= 3.0e38f)) *guard = 1;
}

__global__ __launch_bounds__(256) void sgd_apply_kernel(const uint64_t* __restrict__ ptrs, const int64_t* __restrict__ sizes,
                                                        const int64_t* __restrict__ slab_tensor, const int64_t* __restrict__ slab_first,
                                                        const float* __restrict__ norms, const float* __restrict__ lr_p,
                                                        float momentum, float clipnorm, float grad_scale, const int* __restrict__ skip_flag) {
    if (skip_flag != nullptr && *skip_flag != 0) return;      // the step overflowed the f16 pair range / went non-finite: apply nothing
    const int64_t t = slab_tensor[blockIdx.x];
    const int64_t off = (blockIdx.x - slab_first[t]) * (int64_t)SLAB;
    float* w = reinterpret_cast<float*>(ptrs[3 * t + 0]);
    const float* g = reinterpret_cast<const float*>(ptrs[3 * t + 1]);
    float* v = reinterpret_cast<float*>(ptrs[3 * t + 2]);
    const int64_t n = sizes[t];
    const float lr = *lr_p;
    float scale = grad_scale;
    if (clipnorm > 0.f) {
        const float nrm = norms[t] * fabsf(grad_scale);
        if (nrm > clipnorm) scale *= clipnorm / nrm;
    }
    const bool aligned = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    const int64_t end = min(n, off + SLAB), nq = aligned ? (end - off) >> 2 : 0;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + off);
    f32x4* v4 = reinterpret_cast<f32x4*>(v + off);
    f32x4* w4 = reinterpret_cast<f32x4*>(w + off);
    for (int64_t q = threadIdx.x; q < nq; q += 256) {             // (the same arithmetic per element as the scalar tail below)
        const f32x4 gq = g4[q], vq = v4[q], wq = w4[q];
        f32x4 vn, wn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = gq[e] * scale;
            vn[e] = momentum * vq[e] - lr * gi;
            wn[e] = wq[e] + momentum * vn[e] - lr * gi;
        }
        v4[q] = vn; w4[q] = wn;
    }
    for (int64_t i = off + 4 * nq + threadIdx.x; i < end; i += 256) {
        const float gi = g[i] * scale;
        const float vn = momentum * v[i] - lr * gi;
        v[i] = vn;
        w[i] = w[i] + momentum * vn - lr * gi;
    }
}

}  // namespace

// The slab table (slab_tensor[nslabs], slab_first[ntensors+1]) is built once by the host and
// lives on the device next to ptrs/sizes; partial: nslabs floats; norms: ntensors floats.
extern "C" int bdetr_sgd_slab_elems(void) { return SLAB; }

extern "C" int bdetr_sgd_nesterov_clipnorm(const uint64_t* ptrs, const int64_t* sizes, int ntensors,
                                           const int64_t* slab_tensor, const int64_t* slab_first, int nslabs,
                                           float* partial, float* norms, const float* lr, float momentum,
                                           float clipnorm, float grad_scale, int* skip_flag, void* stream) {
    BDETR_CHECK_ARG(ptrs && sizes && slab_tensor && slab_first && partial && norms && lr && ntensors > 0 && nslabs > 0,
                    "bdetr_sgd_nesterov_clipnorm: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nslabs), dim3(256), 0, st, ptrs, sizes, slab_tensor, slab_first, partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3((ntensors + 255) / 256), dim3(256), 0, st, partial, slab_first, ntensors, norms, skip_flag);
    hipLaunchKernelGGL(sgd_apply_kernel, dim3(nslabs), dim3(256), 0, st, ptrs, sizes, slab_tensor, slab_first, norms, lr, momentum, clipnorm, grad_scale, skip_flag);
    return bdetr_launch_status("sgd_nesterov_clipnorm");
}
