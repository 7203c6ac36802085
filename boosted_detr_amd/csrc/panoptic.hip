// panoptic.hip - the HBM-bound pieces of the panoptic head for gfx950: bilinear NHWC
// resize, channel LayerNormalization + leaky ReLU on channel counts that are not multiples of 4, column-block
// copies (channel concatenation / padding) and the final NHWC -> NCHW transpose.  The convolutions of the head
// (Conv2D k=2, Conv2DTranspose k=2 as a padded conv with flipped taps, Conv2D k=3 s=4) run on igemm.hip with the
// channel dimension zero-padded to a multiple of 4.
//
// Replaces: panoptic_neck.py:20-21 (Reshape + Resizing), 118-121 / 164-167 (LayerNormalization + ReLU(negative_slope)),
// 33-45 (Concatenate), 46-47 (transpose + reshape); transformers.py:519 (LayerNormalization of PanopticAttention).
#include "common.h"

namespace {

// tf.keras.layers.Resizing(bilinear): half-pixel centres, no antialias, edge clamp; C % 4 == 0 (padded)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int B, int h, int w, int C4,
                                                              float* __restrict__ out, int H, int W) {
    const int64_t n = (int64_t)B * H * W * C4;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4); int64_t t = i / C4;
        const int ox = (int)(t % W); t /= W; const int oy = (int)(t % H); const int b = (int)(t / H);
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float fy0 = floorf(fy), fx0 = floorf(fx);
        const int y0 = max((int)fy0, 0), y1 = min((int)fy0 + 1, h - 1);
        const int x0 = max((int)fx0, 0), x1 = min((int)fx0 + 1, w - 1);
        const float ly = fy - fy0, lx = fx - fx0;
        const f32x4* base = reinterpret_cast<const f32x4*>(in) + (int64_t)b * h * w * C4;
        const f32x4 p00 = base[((int64_t)y0 * w + x0) * C4 + c4], p01 = base[((int64_t)y0 * w + x1) * C4 + c4];
        const f32x4 p10 = base[((int64_t)y1 * w + x0) * C4 + c4], p11 = base[((int64_t)y1 * w + x1) * C4 + c4];
        const f32x4 top = p00 + (p01 - p00) * lx, bot = p10 + (p11 - p10) * lx;
        reinterpret_cast<f32x4*>(out)[i] = top + (bot - top) * ly;
    }
}

// one wave per row: LayerNormalization over the first C of ld columns (biased variance), y = leaky(gamma * xhat + beta),
// columns C .. ldo-1 of the output row are written as zeros (channel padding for the next convolution)
__global__ __launch_bounds__(256) void layernorm_act_kernel(const float* __restrict__ x, int64_t rows, int C, int ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float slope,
                                                            float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float t = xr[c] - mean; q += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float* orow = out + row * ldo;
    for (int c = lane; c < ldo; c += 64) {
        float v = 0.f;
        if (c < C) { v = (xr[c] - mean) * rstd * gamma[c] + beta[c]; v = v >= 0.f ? v : v * slope; }
        orow[c] = v;
    }
}

// dst[r][col0 + c] = src[r][c] for c < C
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int64_t rows, int C, int lds_, float* __restrict__ dst, int ldd, int col0) {
    const int64_t n = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const int64_t r = i / C;
        dst[r * ldd + col0 + c] = src[r * lds_ + c];
    }
}

// out[b][c][p] = in[b][p][c] (c < C of ldin columns)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int B, int P, int C, int ldin, float* __restrict__ out) {
    const int64_t n = (int64_t)B * C * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % P); int64_t t = i / P; const int c = (int)(t % C); const int b = (int)(t / C);
        out[i] = in[((int64_t)b * P + p) * ldin + c];
    }
}

}  // namespace

extern "C" int bdetr_resize_bilinear_nhwc(const float* in, int B, int h, int w, int C, float* out, int H, int W, void* stream) {
    BDETR_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bdetr_resize_bilinear_nhwc: bad arguments (C %% 4 == 0)");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ew_grid((int64_t)B * H * W * (C / 4), 256, 1)), dim3(256), 0, (hipStream_t)stream, in, B, h, w, C / 4, out, H, W);
    return bdetr_launch_status("resize_bilinear_nhwc");
}

extern "C" int bdetr_layernorm_act_fwd(const float* x, int64_t rows, int C, int ldx, const float* gamma, const float* beta, float eps, float slope,
                                       float* out, int ldo, void* stream) {
    BDETR_CHECK_ARG(x && gamma && beta && out && rows > 0 && C > 0 && ldx >= C && ldo >= C, "bdetr_layernorm_act_fwd: bad arguments");
    hipLaunchKernelGGL(layernorm_act_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, ldx, gamma, beta, eps, slope, out, ldo);
    return bdetr_launch_status("layernorm_act_fwd");
}

extern "C" int bdetr_copy_cols(const float* src, int64_t rows, int C, int ld_src, float* dst, int ld_dst, int dst_col0, void* stream) {
    BDETR_CHECK_ARG(src && dst && rows > 0 && C > 0 && ld_src >= C && dst_col0 >= 0 && ld_dst >= dst_col0 + C, "bdetr_copy_cols: bad arguments");
    hipLaunchKernelGGL(copy_cols_kernel, dim3(ew_grid(rows * C, 256, 4)), dim3(256), 0, (hipStream_t)stream, src, rows, C, ld_src, dst, ld_dst, dst_col0);
    return bdetr_launch_status("copy_cols");
}

extern "C" int bdetr_nhwc_to_nchw(const float* in, int B, int P, int C, int ld_in, float* out, void* stream) {
    BDETR_CHECK_ARG(in && out && B > 0 && P > 0 && C > 0 && ld_in >= C, "bdetr_nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((int64_t)B * P * C, 256, 4)), dim3(256), 0, (hipStream_t)stream, in, B, P, C, ld_in, out);
    return bdetr_launch_status("nhwc_to_nchw");
}
