// augment.hip - on-GPU training augmentations of the input step, HBM-bound.
//
// Replaces the tf.image calls of pipeline.py:274-341 (Augmentations): random_downsizer_with_pad
// (bilinear down-size, shift, zero pad back to the original size), random_contrast, random_brightness,
// random_saturation.  The random draws are made on the host (TF's RNG stream is not reproducible) and
// handed over per image; the arithmetic of each op follows tf.image (resize: bilinear, half-pixel
// centres, no antialias; adjust_contrast: (x-mean_c)*f+mean_c; adjust_brightness: x+delta;
// adjust_saturation: TF's rgb->hsv, s*=f clamped to [0,1], hsv->rgb).  random_jpeg_quality
// (pipeline.py:319-325) needs a JPEG codec round trip and is not built.
#include "common.h"

namespace {

constexpr int AUG_BLOCKS_PER_IMAGE = 128;

// iparams per image: new_h, new_w, off_h, off_w ; fparams per image: contrast, brightness, saturation
__global__ __launch_bounds__(256) void augment_geometry_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               const int32_t* __restrict__ ip, int H, int W,
                                                               float* __restrict__ part /* [B][blocks][3] */) {
    __shared__ float sh[4][3];
    const int b = blockIdx.y;
    const int nh = ip[4 * b + 0], nw = ip[4 * b + 1], oh = ip[4 * b + 2], ow = ip[4 * b + 3];
    const float sy = (float)H / (float)nh, sx = (float)W / (float)nw;
    const float* src = in + (int64_t)b * H * W * 3;
    float* dst = out + (int64_t)b * H * W * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        float rgb[3] = {0.f, 0.f, 0.f};                                  // pad_to_bounding_box pads with zeros
        const int ry = y - oh, rx = x - ow;
        if (ry >= 0 && ry < nh && rx >= 0 && rx < nw) {
            if (nh == H && nw == W) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb[c] = src[(int64_t)i * 3 + c];
            } else {
                const float fy = ((float)ry + 0.5f) * sy - 0.5f, fx = ((float)rx + 0.5f) * sx - 0.5f;
                const float fy0 = floorf(fy), fx0 = floorf(fx);
                const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), H - 1);
                const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), W - 1);
                const float ly = fy - fy0, lx = fx - fx0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float p00 = src[((int64_t)y0 * W + x0) * 3 + c], p01 = src[((int64_t)y0 * W + x1) * 3 + c];
                    const float p10 = src[((int64_t)y1 * W + x0) * 3 + c], p11 = src[((int64_t)y1 * W + x1) * 3 + c];
                    const float top = p00 + (p01 - p00) * lx, bot = p10 + (p11 - p10) * lx;
                    rgb[c] = top + (bot - top) * ly;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(int64_t)i * 3 + c] = rgb[c];
        s0 += rgb[0]; s1 += rgb[1]; s2 += rgb[2];
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = s0; sh[threadIdx.x >> 6][1] = s1; sh[threadIdx.x >> 6][2] = s2; }
    __syncthreads();
    if (threadIdx.x < 3)
        part[((int64_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__device__ __forceinline__ void adjust_saturation(float& r, float& g, float& b, float scale) {
    // tensorflow/core/kernels/image/adjust_saturation_op.cc (rgb_to_hsv / hsv_to_rgb), restated
    const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b)), range = vmax - vmin;
    float s = vmax > 0.f ? range / vmax : 0.f;
    const float norm = 1.0f / (6.0f * range);
    float h;
    if (r == vmax) h = norm * (g - b);
    else if (g == vmax) h = norm * (b - r) + 2.0f / 6.0f;
    else h = norm * (r - g) + 4.0f / 6.0f;
    if (range <= 0.f) h = 0.f;
    if (h < 0.f) h += 1.0f;
    const float v = vmax;
    s = fminf(1.0f, fmaxf(0.f, s * scale));
    const float c = s * v, m = v - c, dh = h * 6.0f;
    float fm = dh;
    while (fm <= 0.f) fm += 2.0f;
    while (fm >= 2.0f) fm -= 2.0f;
    const float x = c * (1.0f - fabsf(fm - 1.0f));
    float rr = 0.f, gg = 0.f, bb = 0.f;
    switch ((int)dh) {
        case 0: rr = c; gg = x; break;
        case 1: rr = x; gg = c; break;
        case 2: gg = c; bb = x; break;
        case 3: gg = x; bb = c; break;
        case 4: rr = x; bb = c; break;
        case 5: rr = c; bb = x; break;
        default: break;
    }
    r = rr + m; g = gg + m; b = bb + m;
}

__global__ __launch_bounds__(256) void augment_color_kernel(float* __restrict__ img, const float* __restrict__ fp, const float* __restrict__ part,
                                                            int nparts, int H, int W) {
    __shared__ float mean[3];
    const int b = blockIdx.y;
    if (threadIdx.x < 3) {
        double s = 0;
        for (int p = 0; p < nparts; ++p) s += part[((int64_t)b * nparts + p) * 3 + threadIdx.x];
        mean[threadIdx.x] = (float)(s / ((double)H * W));
    }
    __syncthreads();
    const float cf = fp[3 * b + 0], delta = fp[3 * b + 1], sat = fp[3 * b + 2];
    float* p = img + (int64_t)b * H * W * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        float r = p[(int64_t)i * 3 + 0], g = p[(int64_t)i * 3 + 1], bl = p[(int64_t)i * 3 + 2];
        r = (r - mean[0]) * cf + mean[0]; g = (g - mean[1]) * cf + mean[1]; bl = (bl - mean[2]) * cf + mean[2];   // adjust_contrast
        r += delta; g += delta; bl += delta;                                                                        // adjust_brightness
        adjust_saturation(r, g, bl, sat);
        p[(int64_t)i * 3 + 0] = r; p[(int64_t)i * 3 + 1] = g; p[(int64_t)i * 3 + 2] = bl;
    }
}

}  // namespace

extern "C" int bdetr_augment_ws_floats(int B) { return B * AUG_BLOCKS_PER_IMAGE * 3; }

extern "C" int bdetr_augment(const float* in, float* out, const int32_t* iparams, const float* fparams,
                             int B, int H, int W, float* ws, void* stream) {
    BDETR_CHECK_ARG(in && out && iparams && fparams && ws && B > 0 && H > 0 && W > 0 && in != out, "bdetr_augment: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(AUG_BLOCKS_PER_IMAGE, B);
    hipLaunchKernelGGL(augment_geometry_kernel, grid, dim3(256), 0, st, in, out, iparams, H, W, ws);
    hipLaunchKernelGGL(augment_color_kernel, grid, dim3(256), 0, st, out, fparams, ws, AUG_BLOCKS_PER_IMAGE, H, W);
    return bdetr_launch_status("augment");
}
