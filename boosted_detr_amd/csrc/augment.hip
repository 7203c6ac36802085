// augment.hip - on-GPU training augmentations of the input step, HBM-bound.
//
// Replaces the tf.image calls of pipeline.py:274-341 (Augmentations): random_downsizer_with_pad
// (bilinear down-size, shift, zero pad back to the original size), random_contrast, random_brightness,
// random_saturation.  The random draws are made on the host (TF's RNG stream is not reproducible) and
// handed over per image; the arithmetic of each op follows tf.image (resize: bilinear, half-pixel
// centres, no antialias; adjust_contrast: (x-mean_c)*f+mean_c; adjust_brightness: x+delta;
// adjust_saturation: TF's rgb->hsv, s*=f clamped to [0,1], hsv->rgb).  random_jpeg_quality
// (pipeline.py:319-325) is the lossy part of a baseline JPEG round trip (entropy coding is lossless and is skipped): see the
// jpeg_* kernels below, bit-exact against libjpeg-turbo through oracle/jpeg_oracle.py.
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

// stages: bit 0 = contrast + brightness, bit 1 = saturation (the JPEG round trip sits between the two, pipeline.py:364-383)
__global__ __launch_bounds__(256) void augment_color_kernel(float* __restrict__ img, const float* __restrict__ fp, const float* __restrict__ part,
                                                            int nparts, int H, int W, int stages) {
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
        if (stages & 1) {
            r = (r - mean[0]) * cf + mean[0]; g = (g - mean[1]) * cf + mean[1]; bl = (bl - mean[2]) * cf + mean[2];   // adjust_contrast
            r += delta; g += delta; bl += delta;                                                                        // adjust_brightness
        }
        if (stages & 2) adjust_saturation(r, g, bl, sat);
        p[(int64_t)i * 3 + 0] = r; p[(int64_t)i * 3 + 1] = g; p[(int64_t)i * 3 + 2] = bl;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// tf.image.adjust_jpeg_quality (random_jpeg_quality, pipeline.py:319-325): float [0,1] -> uint8 (x * 255.5 truncated, saturating) ->
// baseline JPEG at the image's quality with 4:2:0 chroma -> decode (slow integer IDCT, fancy up-sampling) -> / 255.  The third-party
// codec is libjpeg-turbo (bundled with TensorFlow); its lossy stages are restated from the published algorithm - jccolor.c rgb_ycc_convert,
// jcsample.c h2v2_downsample + edge expansion, jfdctint.c, jcdctmgr.c quantize, jidctint.c, jdsample.c h2v2_fancy_upsample, jdcolor.c -
// and are bit-exact against it (tests/test_jpeg_quality.py).  One workgroup per 16x16 MCU; planes of the decoded Y / Cb / Cr go through
// HBM because the up-sampling filter reaches into the neighbouring MCUs.
__device__ const unsigned char JPEG_STD_LUMA[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                                    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__device__ const unsigned char JPEG_STD_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
constexpr int J_CONST_BITS = 13, J_PASS1_BITS = 2;
constexpr int J_0_298 = 2446, J_0_390 = 3196, J_0_541 = 4433, J_0_765 = 6270, J_0_899 = 7373, J_1_175 = 9633;
constexpr int J_1_501 = 12299, J_1_847 = 15137, J_1_961 = 16069, J_2_053 = 16819, J_2_562 = 20995, J_3_072 = 25172;
__device__ __forceinline__ int j_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass of jpeg_fdct_islow over p[0], p[stride], ...
__device__ __forceinline__ void jpeg_fdct_pass(int* p, int stride, bool first) {
    int d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = p[i * stride];
    const int t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6];
    const int t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    const int sh = first ? J_CONST_BITS - J_PASS1_BITS : J_CONST_BITS + J_PASS1_BITS;
    int o[8];
    if (first) { o[0] = (t10 + t11) << J_PASS1_BITS; o[4] = (t10 - t11) << J_PASS1_BITS; }
    else { o[0] = j_descale(t10 + t11, J_PASS1_BITS); o[4] = j_descale(t10 - t11, J_PASS1_BITS); }
    int z1 = (t12 + t13) * J_0_541;
    o[2] = j_descale(z1 + t13 * J_0_765, sh);
    o[6] = j_descale(z1 - t12 * J_1_847, sh);
    z1 = t4 + t7; int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * J_1_175;
    const int a4 = t4 * J_0_298, a5 = t5 * J_2_053, a6 = t6 * J_3_072, a7 = t7 * J_1_501;
    z1 = -z1 * J_0_899; z2 = -z2 * J_2_562; z3 = -z3 * J_1_961 + z5; z4 = -z4 * J_0_390 + z5;
    o[7] = j_descale(a4 + z1 + z3, sh); o[5] = j_descale(a5 + z2 + z4, sh); o[3] = j_descale(a6 + z2 + z3, sh); o[1] = j_descale(a7 + z1 + z4, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i * stride] = o[i];
}
// one 8-point pass of jpeg_idct_islow (the second pass adds the level shift and clamps)
__device__ __forceinline__ void jpeg_idct_pass(int* p, int stride, bool first) {
    int c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = p[i * stride];
    int z2 = c[2], z3 = c[6];
    int z1 = (z2 + z3) * J_0_541;
    const int e2 = z1 - z3 * J_1_847, e3 = z1 + z2 * J_0_765;
    const int e0 = (c[0] + c[4]) << J_CONST_BITS, e1 = (c[0] - c[4]) << J_CONST_BITS;
    const int t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
    int t0 = c[7], t1 = c[5], t2 = c[3], t3 = c[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; int z4 = t1 + t3;
    const int z5 = (z3 + z4) * J_1_175;
    t0 *= J_0_298; t1 *= J_2_053; t2 *= J_3_072; t3 *= J_1_501;
    z1 = -z1 * J_0_899; z2 = -z2 * J_2_562; z3 = -z3 * J_1_961 + z5; z4 = -z4 * J_0_390 + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int sh = first ? J_CONST_BITS - J_PASS1_BITS : J_CONST_BITS + J_PASS1_BITS + 3;
    int o[8] = {j_descale(t10 + t3, sh), j_descale(t11 + t2, sh), j_descale(t12 + t1, sh), j_descale(t13 + t0, sh),
                j_descale(t13 - t0, sh), j_descale(t12 - t1, sh), j_descale(t11 - t2, sh), j_descale(t10 - t3, sh)};
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i * stride] = first ? o[i] : min(max(o[i] + 128, 0), 255);
}
__device__ __forceinline__ void jpeg_load_ycc(const float* __restrict__ img, int W, int y, int x, int& Y, int& Cb, int& Cr) {
    const float* px = img + ((int64_t)y * W + x) * 3;
    const int r = (int)fminf(fmaxf(floorf(px[0] * 255.5f), 0.f), 255.f), g = (int)fminf(fmaxf(floorf(px[1] * 255.5f), 0.f), 255.f),
              b = (int)fminf(fmaxf(floorf(px[2] * 255.5f), 0.f), 255.f);
    Y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    Cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
    Cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

// grid (MCUs per row, MCU rows, B) x 256 threads; planes: Y [B][H16][W16], then Cb, Cr [B][H16/2][W16/2] (uint8)
__global__ __launch_bounds__(256) void jpeg_mcu_kernel(const float* __restrict__ in, const int32_t* __restrict__ quality, int H, int W,
                                                       unsigned char* __restrict__ planeY, unsigned char* __restrict__ planeCb, unsigned char* __restrict__ planeCr) {
    __shared__ int blk[6][64];                       // Y00 Y01 Y10 Y11 Cb Cr, each [row][col]
    __shared__ int qt[2][64];
    const int b = blockIdx.z, mx = blockIdx.x, my = blockIdx.y, t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int W16 = gridDim.x * 16, H16 = gridDim.y * 16;
    const float* img = in + (int64_t)b * H * W * 3;
    if (t < 128) {                                   // jpeg_quality_scaling + jpeg_add_quant_table(force_baseline)
        const int q = min(max(quality[b], 1), 100), scale = q < 50 ? 5000 / q : 200 - 2 * q;
        const int base = t < 64 ? JPEG_STD_LUMA[t] : JPEG_STD_CHROMA[t - 64];
        qt[t >> 6][t & 63] = min(max((base * scale + 50) / 100, 1), 255);
    }
    {   // luma: the thread's own pixel, edges replicated (expand_right_edge / expand_bottom_edge)
        int Y, Cb, Cr;
        jpeg_load_ycc(img, W, min(my * 16 + ty, H - 1), min(mx * 16 + tx, W - 1), Y, Cb, Cr);
        blk[(ty >> 3) * 2 + (tx >> 3)][(ty & 7) * 8 + (tx & 7)] = Y - 128;
    }
    if (t < 64) {
        // chroma: 2x2 box with the rounding bias alternating 1, 2 along a row (h2v2_downsample).  Columns are replicated before the
        // down-sampling, rows only up to the next even row: below that the last DOWN-SAMPLED row is replicated (pre_process_data).
        const int cx = t & 7, cy = t >> 3;
        const int ch = (H + 1) >> 1, rc = min(my * 8 + cy, ch - 1);
        const int y0 = min(2 * rc, H - 1), y1 = min(2 * rc + 1, H - 1);
        const int x0 = min(mx * 16 + 2 * cx, W - 1), x1 = min(mx * 16 + 2 * cx + 1, W - 1);
        int sb = 0, sr = 0, Y, Cb, Cr;
        jpeg_load_ycc(img, W, y0, x0, Y, Cb, Cr); sb += Cb; sr += Cr;
        jpeg_load_ycc(img, W, y0, x1, Y, Cb, Cr); sb += Cb; sr += Cr;
        jpeg_load_ycc(img, W, y1, x0, Y, Cb, Cr); sb += Cb; sr += Cr;
        jpeg_load_ycc(img, W, y1, x1, Y, Cb, Cr); sb += Cb; sr += Cr;
        const int bias = 1 + ((mx * 8 + cx) & 1);
        blk[4][cy * 8 + cx] = ((sb + bias) >> 2) - 128;
        blk[5][cy * 8 + cx] = ((sr + bias) >> 2) - 128;
    }
    __syncthreads();
    if (t < 48) jpeg_fdct_pass(&blk[t >> 3][(t & 7) * 8], 1, true);          // rows
    __syncthreads();
    if (t < 48) jpeg_fdct_pass(&blk[t >> 3][t & 7], 8, false);               // columns
    __syncthreads();
    for (int i = t; i < 384; i += 256) {             // quantize (divisor 8 * q, round half away from zero) and de-quantise
        const int k = i >> 6, e = i & 63, q = qt[k >= 4][e], div = q * 8, c = blk[k][e];
        const int m = (abs(c) + (div >> 1)) / div;
        blk[k][e] = (c < 0 ? -m : m) * q;
    }
    __syncthreads();
    if (t < 48) jpeg_idct_pass(&blk[t >> 3][t & 7], 8, true);                // columns
    __syncthreads();
    if (t < 48) jpeg_idct_pass(&blk[t >> 3][(t & 7) * 8], 1, false);         // rows (+128, clamped)
    __syncthreads();
    planeY[((int64_t)b * H16 + my * 16 + ty) * W16 + mx * 16 + tx] = (unsigned char)blk[(ty >> 3) * 2 + (tx >> 3)][(ty & 7) * 8 + (tx & 7)];
    if (t < 128) {
        const int k = 4 + (t >> 6), e = t & 63;
        unsigned char* pl = k == 4 ? planeCb : planeCr;
        pl[((int64_t)b * (H16 / 2) + my * 8 + (e >> 3)) * (W16 / 2) + mx * 8 + (e & 7)] = (unsigned char)blk[k][e];
    }
}

// h2v2_fancy_upsample of one chroma plane at output pixel (y, x): 3/4 of the nearer and 1/4 of the further sample in each direction;
// beyond the REAL chroma extent (ch x cw) the nearer sample stands alone; cw <= 2: plain replication (jinit_upsampler)
__device__ __forceinline__ int jpeg_fancy(const unsigned char* __restrict__ c, int pitch, int ch, int cw, int y, int x) {
    const int cy = y >> 1, cx = x >> 1;
    if (cw <= 2) return c[cy * pitch + cx];
    const int fy = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
    const int cur = 3 * c[cy * pitch + cx] + c[fy * pitch + cx];
    if (x & 1) {
        if (cx == cw - 1) return (cur * 4 + 7) >> 4;
        return (3 * cur + 3 * c[cy * pitch + cx + 1] + c[fy * pitch + cx + 1] + 7) >> 4;
    }
    if (cx == 0) return (cur * 4 + 8) >> 4;
    return (3 * cur + 3 * c[cy * pitch + cx - 1] + c[fy * pitch + cx - 1] + 8) >> 4;
}
__global__ __launch_bounds__(256) void jpeg_finish_kernel(float* __restrict__ out, int H, int W, int H16, int W16, const unsigned char* __restrict__ planeY,
                                                          const unsigned char* __restrict__ planeCb, const unsigned char* __restrict__ planeCr) {
    const int b = blockIdx.y;
    const int ch = (H + 1) >> 1, cw = (W + 1) >> 1, cp = W16 / 2;
    const unsigned char* pb = planeCb + (int64_t)b * (H16 / 2) * cp;
    const unsigned char* pr = planeCr + (int64_t)b * (H16 / 2) * cp;
    float* dst = out + (int64_t)b * H * W * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        const int Y = planeY[((int64_t)b * H16 + y) * W16 + x];
        const int cb = jpeg_fancy(pb, cp, ch, cw, y, x) - 128, cr = jpeg_fancy(pr, cp, ch, cw, y, x) - 128;
        const int r = Y + ((91881 * cr + 32768) >> 16), bl = Y + ((116130 * cb + 32768) >> 16), g = Y + ((-22554 * cb - 46802 * cr + 32768) >> 16);   // ycc_rgb_convert
        dst[(int64_t)i * 3 + 0] = (float)min(max(r, 0), 255) / 255.0f;
        dst[(int64_t)i * 3 + 1] = (float)min(max(g, 0), 255) / 255.0f;
        dst[(int64_t)i * 3 + 2] = (float)min(max(bl, 0), 255) / 255.0f;
    }
}

}  // namespace

extern "C" int bdetr_augment_ws_floats(int B) { return B * AUG_BLOCKS_PER_IMAGE * 3; }

extern "C" int64_t bdetr_jpeg_quality_ws_bytes(int B, int H, int W) {
    const int64_t H16 = (H + 15) / 16 * 16, W16 = (W + 15) / 16 * 16;
    return (int64_t)B * (H16 * W16 + 2 * (H16 / 2) * (W16 / 2));
}
// in / out: float [B,H,W,3] in [0,1] (in == out allowed); quality: int32 [B] on the device; ws: bdetr_jpeg_quality_ws_bytes bytes
extern "C" int bdetr_jpeg_quality(const float* in, float* out, const int32_t* quality, int B, int H, int W, void* ws, void* stream) {
    BDETR_CHECK_ARG(in && out && quality && ws && B > 0 && H > 0 && W > 0, "bdetr_jpeg_quality: bad arguments");
    BDETR_CHECK_ARG((int64_t)H * W < (1LL << 31) && B <= 65535 && (H + 15) / 16 <= 65535, "bdetr_jpeg_quality: image too large");
    hipStream_t st = (hipStream_t)stream;
    const int H16 = (H + 15) / 16 * 16, W16 = (W + 15) / 16 * 16;
    unsigned char* pY = reinterpret_cast<unsigned char*>(ws);
    unsigned char* pCb = pY + (int64_t)B * H16 * W16;
    unsigned char* pCr = pCb + (int64_t)B * (H16 / 2) * (W16 / 2);
    hipLaunchKernelGGL(jpeg_mcu_kernel, dim3(W16 / 16, H16 / 16, B), dim3(256), 0, st, in, quality, H, W, pY, pCb, pCr);
    hipLaunchKernelGGL(jpeg_finish_kernel, dim3(AUG_BLOCKS_PER_IMAGE, B), dim3(256), 0, st, out, H, W, H16, W16, pY, pCb, pCr);
    return bdetr_launch_status("jpeg_quality");
}

// quality (optional, int32 [B]) + jpeg_ws: the JPEG round trip between brightness and saturation, as the reference orders it
extern "C" int bdetr_augment_jpeg(const float* in, float* out, const int32_t* iparams, const float* fparams, const int32_t* quality,
                                  int B, int H, int W, float* ws, void* jpeg_ws, void* stream) {
    BDETR_CHECK_ARG(in && out && iparams && fparams && ws && B > 0 && H > 0 && W > 0 && in != out, "bdetr_augment: bad arguments");
    BDETR_CHECK_ARG(quality == nullptr || jpeg_ws != nullptr, "bdetr_augment_jpeg: quality without workspace");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(AUG_BLOCKS_PER_IMAGE, B);
    hipLaunchKernelGGL(augment_geometry_kernel, grid, dim3(256), 0, st, in, out, iparams, H, W, ws);
    if (quality == nullptr) {
        hipLaunchKernelGGL(augment_color_kernel, grid, dim3(256), 0, st, out, fparams, ws, AUG_BLOCKS_PER_IMAGE, H, W, 3);
        return bdetr_launch_status("augment");
    }
    hipLaunchKernelGGL(augment_color_kernel, grid, dim3(256), 0, st, out, fparams, ws, AUG_BLOCKS_PER_IMAGE, H, W, 1);
    if (int e = bdetr_jpeg_quality(out, out, quality, B, H, W, jpeg_ws, stream)) return e;
    hipLaunchKernelGGL(augment_color_kernel, grid, dim3(256), 0, st, out, fparams, ws, AUG_BLOCKS_PER_IMAGE, H, W, 2);
    return bdetr_launch_status("augment_jpeg");
}
extern "C" int bdetr_augment(const float* in, float* out, const int32_t* iparams, const float* fparams,
                             int B, int H, int W, float* ws, void* stream) {
    return bdetr_augment_jpeg(in, out, iparams, fparams, nullptr, B, H, W, ws, nullptr, stream);
}
