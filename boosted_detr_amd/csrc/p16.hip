// p16.hip - stand-alone producers of the P16 operand layout (sgemm.hip): fp32 -> f16 / bf16 pairs, the
// per-step weight packs (forward copy + transposed, tap-flipped backward-data copy) and the inverse map.
// The hot producers are fused: BatchNorm apply / BatchNorm backward write P16 directly (norm.hip).
#include "p16.h"

namespace {

// f16_scale: the f16 pair holds f16_scale * x (1 for activations, P16_W_SCALE for conv weights: p16.h)
__global__ __launch_bounds__(256) void p16_pack_kernel(const float* __restrict__ x, int64_t n8, void* __restrict__ f16_out, void* __restrict__ bf16_out,
                                                       int* __restrict__ overflow_flag, float f16_scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (f16_out) {
            const float vs[8] = {v[0] * f16_scale, v[1] * f16_scale, v[2] * f16_scale, v[3] * f16_scale, v[4] * f16_scale, v[5] * f16_scale, v[6] * f16_scale, v[7] * f16_scale};
            p16_store8<true>(reinterpret_cast<char*>(f16_out) + i * 32, vs);
            if (overflow_flag && p16_f16_overflow(vs)) *overflow_flag = 1;
        }
        if (bf16_out) p16_store8<false>(reinterpret_cast<char*>(bf16_out) + i * 32, v);
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void p16_unpack_kernel(const void* __restrict__ p, int64_t n8, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        p16_load8<F16>(reinterpret_cast<const char*>(p) + i * 32, v);
        reinterpret_cast<f32x4*>(out)[2 * i] = f32x4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4*>(out)[2 * i + 1] = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// w [K][R][S][C] fp32 -> wt [C][R][S][K] P16-bf16 with the taps flipped: wt[c][r'][s'][k] = w[k][R-1-r'][S-1-s'][c].
// One thread = one group of 8 k for one (c, tap); adjacent threads take adjacent c (coalesced reads).
__global__ __launch_bounds__(256) void p16_pack_wt_kernel(const float* __restrict__ w, int K, int R, int S, int C, void* __restrict__ wt) {
    const int64_t total = (int64_t)(K / 8) * R * S * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); int64_t t = i / C;
        const int tap = (int)(t % (R * S)); const int kg = (int)(t / (R * S));
        const int r = tap / S, s = tap - r * S;
        const int src_tap = (R - 1 - r) * S + (S - 1 - s);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[((int64_t)(kg * 8 + e) * R * S + src_tap) * C + c];
        p16_store8<false>(reinterpret_cast<char*>(wt) + (((int64_t)c * R * S + tap) * K + kg * 8) * 4, v);
    }
}

// every registered conv weight in ONE launch: row t of the table = {w, w_f16, wt_bf16, K, R, S, C} (int64 each);
// blockIdx.y = tensor, the blocks of a row grid-stride over its 8-element groups (both copies)
__global__ __launch_bounds__(256) void p16_pack_weights_multi_kernel(const int64_t* __restrict__ table, int* __restrict__ overflow_flag) {
    const int64_t* row = table + (int64_t)blockIdx.y * 7;
    const float* w = reinterpret_cast<const float*>(row[0]);
    char* wf = reinterpret_cast<char*>(row[1]);
    char* wt = reinterpret_cast<char*>(row[2]);
    const int K = (int)row[3], R = (int)row[4], S = (int)row[5], C = (int)row[6];
    const int64_t n8 = (int64_t)K * R * S * C / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        if (wf != nullptr) {
            const f32x4 a = reinterpret_cast<const f32x4*>(w)[2 * i], b = reinterpret_cast<const f32x4*>(w)[2 * i + 1];
            const float v[8] = {a[0] * P16_W_SCALE, a[1] * P16_W_SCALE, a[2] * P16_W_SCALE, a[3] * P16_W_SCALE,
                                b[0] * P16_W_SCALE, b[1] * P16_W_SCALE, b[2] * P16_W_SCALE, b[3] * P16_W_SCALE};      // the forward copy holds 2^8 w (p16.h)
            p16_store8<true>(wf + i * 32, v);
            if (overflow_flag && p16_f16_overflow(v)) *overflow_flag = 1;
        }
        if (wt != nullptr) {
            const int c = (int)(i % C); int64_t t = i / C;
            const int tap = (int)(t % (R * S)); const int kg = (int)(t / (R * S));
            const int r = tap / S, s2 = tap - r * S;
            const int src_tap = (R - 1 - r) * S + (S - 1 - s2);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w[((int64_t)(kg * 8 + e) * R * S + src_tap) * C + c];
            p16_store8<false>(wt + (((int64_t)c * R * S + tap) * K + kg * 8) * 4, v);
        }
    }
}

}  // namespace

extern "C" int bdetr_p16_pack_conv_weights_multi(const int64_t* table, int ntensors, int* overflow_flag, void* stream) {
    BDETR_CHECK_ARG(table && ntensors > 0, "bdetr_p16_pack_conv_weights_multi: bad arguments");
    hipLaunchKernelGGL(p16_pack_weights_multi_kernel, dim3(48, ntensors), dim3(256), 0, (hipStream_t)stream, table, overflow_flag);
    return bdetr_launch_status("p16_pack_conv_weights_multi");
}

extern "C" int bdetr_p16_pack(const float* x, int64_t n, void* f16_out, void* bf16_out, int* overflow_flag, void* stream) {
    BDETR_CHECK_ARG(x && n > 0 && n % 8 == 0 && (f16_out || bf16_out), "bdetr_p16_pack: bad arguments (n %% 8 == 0 required)");
    hipLaunchKernelGGL(p16_pack_kernel, dim3(ew_grid(n / 8, 256, 2)), dim3(256), 0, (hipStream_t)stream, x, n / 8, f16_out, bf16_out, overflow_flag, 1.f);
    return bdetr_launch_status("p16_pack");
}

extern "C" int bdetr_p16_unpack(const void* p, int is_f16, int64_t n, float* out, void* stream) {
    BDETR_CHECK_ARG(p && out && n > 0 && n % 8 == 0, "bdetr_p16_unpack: bad arguments (n %% 8 == 0 required)");
    if (is_f16) hipLaunchKernelGGL((p16_unpack_kernel<true>), dim3(ew_grid(n / 8, 256, 2)), dim3(256), 0, (hipStream_t)stream, p, n / 8, out);
    else        hipLaunchKernelGGL((p16_unpack_kernel<false>), dim3(ew_grid(n / 8, 256, 2)), dim3(256), 0, (hipStream_t)stream, p, n / 8, out);
    return bdetr_launch_status("p16_unpack");
}

// Both operand copies of one conv / dense weight tensor w [K][R][S][C] (OHWI):
//   w_f16   P16-f16  [K][R*S*C]        forward B operand; holds 2^8 w (p16.h), the forward convolution rescales
//   wt_bf16 P16-bf16 [C][R*S][K]       backward-data B operand (transposed, taps flipped)
// Either output may be null.
extern "C" int bdetr_p16_pack_conv_weights(const float* w, int K, int R, int S, int C, void* w_f16, void* wt_bf16, int* overflow_flag, void* stream) {
    BDETR_CHECK_ARG(w && K > 0 && R > 0 && S > 0 && C > 0 && K % 8 == 0 && C % 8 == 0, "bdetr_p16_pack_conv_weights: K and C must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)K * R * S * C;
    if (w_f16) hipLaunchKernelGGL(p16_pack_kernel, dim3(ew_grid(n / 8, 256, 2)), dim3(256), 0, st, w, n / 8, w_f16, (void*)nullptr, overflow_flag, P16_W_SCALE);
    if (wt_bf16) hipLaunchKernelGGL(p16_pack_wt_kernel, dim3(ew_grid(n / 8, 256, 1)), dim3(256), 0, st, w, K, R, S, C, wt_bf16);
    return bdetr_launch_status("p16_pack_conv_weights");
}

// ---- the stem's weight gradient on the pre-split path (round 5) -------------------------------------------------------------------
// keras ResNet50 conv1_conv (7x7 / stride 2 on the ZeroPadding2D(3) image; reference backbone.py:37-38) reads a 4-channel image - no
// P16 layout for that (groups of 8 channels).  Space-to-depth turns it into a size-preserving stride-1 convolution: x2[n][i][j][(a, b, c)] =
// x[n][2 i + a][2 j + b][c] (16 channels on the H/2 x W/2 grid), tap (r, s) of the 7x7 kernel = tap (r', s') of a 4x4 kernel over x2 with
// r = 2 r' + a - 1 (r = -1: no such tap), low-side padding 2.  The weight gradient then is sgemm.hip's XX kernel over x2's bf16 pairs
// ([K][4][4][16] fp32) and bdetr_p16_s2d_unpack_dw folds it back into the [K][7][7][4] layout.
namespace {
__global__ __launch_bounds__(256) void s2d_pack_bf16_kernel(const float* __restrict__ x, int N, int H, int W, void* __restrict__ out) {
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)N * H2 * W2 * 2;                  // one 8-element group (a; b = 0, 1; c = 0..3) per thread
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int a = (int)(t & 1); int64_t p = t >> 1;
        const int j = (int)(p % W2); p /= W2; const int i = (int)(p % H2); const int b_ = (int)(p / H2);
        const f32x4* row = reinterpret_cast<const f32x4*>(x) + ((int64_t)b_ * H + 2 * i + a) * W + 2 * j;
        const f32x4 p0 = row[0], p1 = row[1];
        const float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
        p16_store8<false>(reinterpret_cast<char*>(out) + t * 32, v);
    }
}
__global__ __launch_bounds__(256) void s2d_unpack_dw_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int K) {
    const int n = K * 7 * 7 * 4;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int c = t & 3; int q = t >> 2;
        const int s = q % 7; q /= 7; const int r = q % 7; const int k = q / 7;
        const int rp = (r + 1) >> 1, a = (r + 1) & 1, sp = (s + 1) >> 1, b = (s + 1) & 1;
        dw[t] = dw2[k * 256 + (rp * 4 + sp) * 16 + (a * 2 + b) * 4 + c];
    }
}
}  // namespace

extern "C" int bdetr_p16_s2d_pack_bf16(const float* x, int N, int H, int W, void* out_bf16, void* stream) {
    BDETR_CHECK_ARG(x && out_bf16 && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "bdetr_p16_s2d_pack_bf16: a [N,H,W,4] image with even H and W");
    const int64_t n = (int64_t)N * (H / 2) * (W / 2) * 2;
    hipLaunchKernelGGL(s2d_pack_bf16_kernel, dim3(ew_grid(n, 256, 2)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, out_bf16);
    return bdetr_launch_status("p16_s2d_pack_bf16");
}
extern "C" int bdetr_p16_s2d_unpack_dw(const float* dw2, float* dw, int K, void* stream) {
    BDETR_CHECK_ARG(dw2 && dw && K > 0, "bdetr_p16_s2d_unpack_dw: bad arguments");
    hipLaunchKernelGGL(s2d_unpack_dw_kernel, dim3((K * 196 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw2, dw, K);
    return bdetr_launch_status("p16_s2d_unpack_dw");
}
