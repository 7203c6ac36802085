// elementwise.hip - HBM-bound pointwise / pooling kernels for gfx950 (fp32, grid-stride,
// 16-byte accesses where the layout allows).
//
// Replaces: backbone.py:49-56 (image preparation), ResNet-50 pool1 (max-pool 3x3/2),
// prediction_heads.py:44,180 (sigmoid variants), transformers.py Add layers, Keras
// autodiff of tanh/relu.
#include "common.h"

namespace {

__device__ __forceinline__ float prep_px(float v) {        // clip happened before the resize
    float q = floorf(v * 255.5f);                           // tf.image.convert_image_dtype(float->uint8): saturate_cast(x*255.5)
    return fminf(fmaxf(q, 0.f), 255.f);
}

__global__ __launch_bounds__(256) void image_prep_kernel(const float* __restrict__ in, int B, int h, int w,
                                                         float* __restrict__ out, int H, int W) {
    const int64_t npix = (int64_t)B * H * W;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W); const int64_t t = i / W; const int oy = (int)(t % H); const int b = (int)(t / H);
        float rgb[3];
        if (h == H && w == W) {
            const float* p = in + i * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = fminf(fmaxf(p[c], 0.f), 1.f);
        } else {
            // tf.keras.layers.Resizing: bilinear, half-pixel centres, no antialias
            const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
            const float fy0 = floorf(fy), fx0 = floorf(fx);
            const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), h - 1);
            const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), w - 1);
            const float ly = fy - fy0, lx = fx - fx0;
            const float* base = in + (int64_t)b * h * w * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                auto px = [&](int yy, int xx) { return fminf(fmaxf(base[((int64_t)yy * w + xx) * 3 + c], 0.f), 1.f); };
                const float top = px(y0, x0) + (px(y0, x1) - px(y0, x0)) * lx;
                const float bot = px(y1, x0) + (px(y1, x1) - px(y1, x0)) * lx;
                rgb[c] = top + (bot - top) * ly;
            }
        }
        // caffe-mode preprocess_input: RGB->BGR, subtract ImageNet mean
        f32x4 o;
        o[0] = prep_px(rgb[2]) - 103.939f;
        o[1] = prep_px(rgb[1]) - 116.779f;
        o[2] = prep_px(rgb[0]) - 123.68f;
        o[3] = 0.f;
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int N, int H, int W, int C, int OH, int OW) {
    const int c4n = C / 4;
    const int64_t n4 = (int64_t)N * OH * OW * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n); int64_t t = i / c4n;
        const int ow = (int)(t % OW); t /= OW; const int oh = (int)(t % OH); const int n = (int)(t / OH);
        // zero padding (ZeroPadding2D) == -inf padding here because the input is post-ReLU; keep the literal zero-pad semantics
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ih = oh * 2 - 1 + r, iw = ow * 2 - 1 + s;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = reinterpret_cast<const f32x4*>(x)[(((int64_t)n * H + ih) * W + iw) * c4n + c4];
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        reinterpret_cast<f32x4*>(y)[i] = m;
    }
}

// gather form (no atomics): an input pixel receives dy of every window whose max it equals.
// Ties only occur at post-ReLU zeros, whose gradient is masked by the producer's ReLU anyway.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int OH, int OW) {
    const int c4n = C / 4;
    const int64_t n4 = (int64_t)N * H * W * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n); int64_t t = i / c4n;
        const int iw = (int)(t % W); t /= W; const int ih = (int)(t % H); const int n = (int)(t / H);
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows oh with oh*2-1 <= ih <= oh*2+1
        const int oh_lo = max((ih) / 2, 0), oh_hi = min((ih + 1) / 2, OH - 1);
        const int ow_lo = max((iw) / 2, 0), ow_hi = min((iw + 1) / 2, OW - 1);
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * c4n + c4;
                const f32x4 yv = reinterpret_cast<const f32x4*>(y)[o];
                const f32x4 dv = reinterpret_cast<const f32x4*>(dy)[o];
#pragma unroll
                for (int e = 0; e < 4; ++e) if (xv[e] == yv[e]) g[e] += dv[e];
            }
        reinterpret_cast<f32x4*>(dx)[i] = g;
    }
}

enum { EW_ZERO, EW_ADD, EW_SIGMOID, EW_SIGMOID_BWD, EW_BOXSIG, EW_BOXSIG_BWD, EW_TANH_BWD, EW_RELU_BWD, EW_AXPY };

template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float r;
        if (OP == EW_ZERO) r = 0.f;
        else if (OP == EW_ADD) r = a[i] + b[i];
        else if (OP == EW_SIGMOID) r = 1.0f / (1.0f + expf(-a[i]));
        else if (OP == EW_SIGMOID_BWD) { float y = a[i]; r = b[i] * y * (1.0f - y); }
        else if (OP == EW_BOXSIG) r = 3.0f * (1.0f / (1.0f + expf(-(a[i] / 100.0f)))) - 1.0f;
        else if (OP == EW_BOXSIG_BWD) { float s = (a[i] + 1.0f) / 3.0f; r = b[i] * 3.0f * s * (1.0f - s) / 100.0f; }
        else if (OP == EW_TANH_BWD) { float y = a[i]; r = b[i] * (1.0f - y * y); }
        else if (OP == EW_RELU_BWD) r = a[i] > 0.f ? b[i] : 0.f;
        else r = o[i] + alpha * a[i];
        o[i] = r;
    }
}

__global__ __launch_bounds__(256) void add_bcast_rows_kernel(const float* __restrict__ a, const float* __restrict__ row, float* __restrict__ out,
                                                             int64_t n, int64_t rowlen) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + row[i % rowlen];
}

__global__ __launch_bounds__(256) void sum_over_batch_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t batch, int64_t n, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = accumulate ? out[i] : 0.f;
        for (int64_t b = 0; b < batch; ++b) s += x[b * n + i];
        out[i] = s;
    }
}

// column sums of [rows][cols] (any cols), two deterministic levels: per-chunk partials, then a
// fixed-order fp64 sum of the partials.  block = 64 columns x 4 row phases.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t rows_per_chunk,
                                                             float* __restrict__ part, float* __restrict__ acc_out) {
    __shared__ float s[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float a = 0.f;
    if (c < cols) for (int64_t r = r0 + ry; r < r1; r += 4) a += x[r * cols + c];
    s[threadIdx.x] = a;
    __syncthreads();
    if (ry == 0 && c < cols) {
        const float v = s[cx] + s[cx + 64] + s[cx + 128] + s[cx + 192];
        if (acc_out != nullptr) atomicAdd(acc_out + c, v);
        else part[(int64_t)blockIdx.y * cols + c] = v;
    }
}
// up to four independent column sums of the same width in one launch (blockIdx.z = member): out[m][c] += sum over rows of x[m][.][c]
struct ColsumGroup { const float* x[4]; float* out[4]; int64_t rows[4]; int64_t rpc[4]; };
__global__ __launch_bounds__(256) void colsum_group_kernel(ColsumGroup g, int cols) {
    __shared__ float s[256];
    const int z = blockIdx.z;
    // (selects with constant indices: a run-time index into the by-value argument would push it to scratch memory)
    const float* x = z == 0 ? g.x[0] : z == 1 ? g.x[1] : z == 2 ? g.x[2] : g.x[3];
    float* out = z == 0 ? g.out[0] : z == 1 ? g.out[1] : z == 2 ? g.out[2] : g.out[3];
    const int64_t rows = z == 0 ? g.rows[0] : z == 1 ? g.rows[1] : z == 2 ? g.rows[2] : g.rows[3];
    const int64_t rpc = z == 0 ? g.rpc[0] : z == 1 ? g.rpc[1] : z == 2 ? g.rpc[2] : g.rpc[3];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
    if (r0 >= rows) return;                                    // (uniform: a member with fewer row chunks than the grid)
    float a = 0.f;
    if (c < cols) for (int64_t r = r0 + ry; r < r1; r += 4) a += x[r * cols + c];
    s[threadIdx.x] = a;
    __syncthreads();
    if (ry == 0 && c < cols) atomicAdd(out + c, s[cx] + s[cx + 64] + s[cx + 128] + s[cx + 192]);
}
// 32 columns x 8 part-phases per block, fp64, fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nparts, int cols, float* __restrict__ out) {
    __shared__ double s[256];
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double a = 0;
    if (c < cols) for (int p = py; p < nparts; p += 8) a += part[(int64_t)p * cols + c];
    s[threadIdx.x] = a;
    __syncthreads();
    if (py == 0 && c < cols) {
        for (int k = 1; k < 8; ++k) a += s[cx + 32 * k];
        out[c] = (float)a;
    }
}

template <int OP>
int ew_launch(const char* who, const float* a, const float* b, float* o, int64_t n, float alpha, void* stream) {
    BDETR_CHECK_ARG(o != nullptr && n >= 0, "%s: bad arguments", who);
    if (n == 0) return 0;
    hipLaunchKernelGGL((ew_kernel<OP>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, o, n, alpha);
    return bdetr_launch_status(who);
}

// Targets that are already integer ids and resident in HBM (tokenizers.py:40-82 after the StringLookup):
// category ids are range-checked, attribute id slots are scattered into the multi-hot matrix
// (one_hot + reduce_max over slots; PAD slots set bit 0).  An id outside its vocabulary maps to <OOV> = 1,
// which is what StringLookup does with an unknown string.
__global__ __launch_bounds__(256) void tokens_prepare_kernel(const int* __restrict__ cat_in, const int* __restrict__ att_in, int64_t rows, int slots,
                                                             int C, int A, int* __restrict__ cat_out, float* __restrict__ hot) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int c = cat_in[r];
        cat_out[r] = (unsigned)c < (unsigned)C ? c : 1;
        float* h = hot + r * A;
        for (int a = 0; a < A; ++a) h[a] = 0.f;
        for (int s = 0; s < slots; ++s) {
            const int a = att_in[r * slots + s];
            h[(unsigned)a < (unsigned)A ? a : 1] = 1.f;
        }
    }
}

__global__ __launch_bounds__(256) void flag_nonfinite_kernel(const float* __restrict__ x, int64_t n, int* __restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (!(fabsf(x[i]) <= 3.0e38f)) *flag = 1;
}

}  // namespace

extern "C" int bdetr_tokens_prepare(const int* cat_ids, const int* att_ids, int64_t rows, int slots, int C, int A,
                                    int* cat_out, float* att_hot, void* stream) {
    BDETR_CHECK_ARG(cat_ids && att_ids && cat_out && att_hot && rows > 0 && slots >= 0 && C > 1 && A > 1, "bdetr_tokens_prepare: bad arguments");
    hipLaunchKernelGGL(tokens_prepare_kernel, dim3(ew_grid(rows, 256, 1)), dim3(256), 0, (hipStream_t)stream, cat_ids, att_ids, rows, slots, C, A, cat_out, att_hot);
    return bdetr_launch_status("tokens_prepare");
}

// Last kernel of a guarded training step (eager or inside the captured optimizer segment): one lane advances the device-resident
// step ordinal and logs the guard word against it in host-visible (pinned, device-mapped) memory - ring[1 + ordinal % len] = *flag,
// then ring[0] = ordinal, system-scope stores with a system fence between.  A host that finds ring[0] >= k may read the entry of
// step k without synchronising the stream.  Part of the step, not an operation enqueued between steps: a device operation that reads
// the guard word BETWEEN the hipGraph launches of a segmented step (a D2H memcpy or a kernel alike) corrupted the replays that
// followed on ROCm 7.2 (round 3, tools/graph_debug.py), one inside the captured segment does not.
__global__ void flag_snapshot_kernel(const int* __restrict__ flag, int* __restrict__ ordinal, int* ring, int ring_len) {
    if (threadIdx.x == 0) {
        const int k = *ordinal + 1;
        *ordinal = k;
        __hip_atomic_store(&ring[1 + k % ring_len], *flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&ring[0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" int bdetr_flag_snapshot(const int* flag, int* ordinal, int* host_ring, int ring_len, void* stream) {
    BDETR_CHECK_ARG(flag && ordinal && host_ring && ring_len > 0, "bdetr_flag_snapshot: bad arguments");
    hipLaunchKernelGGL(flag_snapshot_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, ordinal, host_ring, ring_len);
    return bdetr_launch_status("flag_snapshot");
}

// Diagnostic (tools/graph_segment_checksums.py): an ORDER-INDEPENDENT fingerprint of a buffer - the wrapping sum of its bit patterns
// and the count of its non-finite elements - appended to a device-resident log behind a device-resident cursor, so that the launch
// can be captured into a hipGraph segment and every replay appends its own entry.  Two launches: the grid adds into scratch[0..1]
// with integer atomics, one lane moves scratch to log[cursor++] and clears it.
namespace {
__global__ __launch_bounds__(256) void debug_checksum_kernel(const float* __restrict__ x, int64_t n, unsigned long long* __restrict__ scratch) {
    unsigned long long s = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned u = __float_as_uint(x[i]);
        s += u;
        bad += ((u & 0x7F800000u) == 0x7F800000u) ? 1ull : 0ull;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); bad += __shfl_xor(bad, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&scratch[0], s); atomicAdd(&scratch[1], bad); }
}
__global__ void debug_checksum_commit_kernel(unsigned long long* __restrict__ scratch, unsigned long long* __restrict__ log, int* __restrict__ cursor, int cap, unsigned long long tag) {
    if (threadIdx.x == 0) {
        const int c = atomicAdd(cursor, 1);                    // (entries of two streams may interleave: compare per tag)
        if (c < cap) { log[3 * c] = scratch[0]; log[3 * c + 1] = scratch[1]; log[3 * c + 2] = tag; }
        scratch[0] = 0; scratch[1] = 0;
    }
}
}  // namespace
extern "C" int bdetr_debug_checksum(const float* x, int64_t n, uint64_t* scratch, uint64_t* log, int* cursor, int cap, uint64_t tag, void* stream) {
    BDETR_CHECK_ARG(x && scratch && log && cursor && n > 0 && cap > 0, "bdetr_debug_checksum: bad arguments");
    const int grid = (int)(n / 256 / 8 > 2048 ? 2048 : (n / 256 / 8 < 1 ? 1 : n / 256 / 8));
    hipLaunchKernelGGL(debug_checksum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, reinterpret_cast<unsigned long long*>(scratch));
    hipLaunchKernelGGL(debug_checksum_commit_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(scratch),
                       reinterpret_cast<unsigned long long*>(log), cursor, cap, (unsigned long long)tag);
    return bdetr_launch_status("debug_checksum");
}

extern "C" int bdetr_flag_nonfinite(const float* x, int64_t n, int* flag, void* stream) {
    BDETR_CHECK_ARG(x && flag && n > 0, "bdetr_flag_nonfinite: bad arguments");
    hipLaunchKernelGGL(flag_nonfinite_kernel, dim3(ew_grid(n, 256, 4)), dim3(256), 0, (hipStream_t)stream, x, n, flag);
    return bdetr_launch_status("flag_nonfinite");
}

extern "C" int bdetr_image_prep(const float* in, int B, int h, int w, float* out, int H, int W, void* stream) {
    BDETR_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bdetr_image_prep: bad arguments");
    int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(image_prep_kernel, dim3(ew_grid(n, 256, 1)), dim3(256), 0, (hipStream_t)stream, in, B, h, w, out, H, W);
    return bdetr_launch_status("image_prep");
}

extern "C" int bdetr_maxpool3x3s2_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream) {
    BDETR_CHECK_ARG(x && y && C % 4 == 0 && OH == (H + 2 - 3) / 2 + 1 && OW == (W + 2 - 3) / 2 + 1, "bdetr_maxpool3x3s2_fwd: bad arguments");
    int64_t n4 = (int64_t)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(n4, 256, 1)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, OH, OW);
    return bdetr_launch_status("maxpool_fwd");
}
extern "C" int bdetr_maxpool3x3s2_bwd(const float* x, const float* y, const float* dy, float* dx,
                                      int N, int H, int W, int C, int OH, int OW, void* stream) {
    BDETR_CHECK_ARG(x && y && dy && dx && C % 4 == 0 && OH == (H + 2 - 3) / 2 + 1 && OW == (W + 2 - 3) / 2 + 1, "bdetr_maxpool3x3s2_bwd: bad arguments");
    int64_t n4 = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(n4, 256, 1)), dim3(256), 0, (hipStream_t)stream, x, y, dy, dx, N, H, W, C, OH, OW);
    return bdetr_launch_status("maxpool_bwd");
}

// Zero fill by a kernel of this library, not by hipMemsetAsync: see bdetr_zero_bytes in common.h for why.
namespace {
__global__ __launch_bounds__(256) void zero_fill_kernel(f32x4* __restrict__ p4, int64_t n4, unsigned char* __restrict__ tail, int ntail) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) p4[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
}  // namespace
int bdetr_zero_bytes(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    static int use_memset = -1;
    if (use_memset < 0) { const char* e = getenv("BDETR_ZERO_MEMSET"); use_memset = (e && e[0] == '1') ? 1 : 0; }
    if (use_memset) {                                       // (A/B switch: the runtime's memset node, as in rounds 1-3)
        hipError_t e = hipMemsetAsync(p, 0, bytes, st);
        if (e != hipSuccess) { bdetr_set_error("bdetr_zero_bytes: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    unsigned char* b = reinterpret_cast<unsigned char*>(p);
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15;
    if (head >= bytes) {                                    // tiny and unaligned: bytes only
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (f32x4*)nullptr, (int64_t)0, b, (int)bytes);
        return bdetr_launch_status("zero_fill");
    }
    if (head) hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (f32x4*)nullptr, (int64_t)0, b, (int)head);
    const int64_t n4 = (int64_t)((bytes - head) / 16);
    const int ntail = (int)((bytes - head) % 16);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(ew_grid(n4 > 0 ? n4 : 1, 256, 4)), dim3(256), 0, st, reinterpret_cast<f32x4*>(b + head), n4, b + head + (size_t)n4 * 16, ntail);
    return bdetr_launch_status("zero_fill");
}

extern "C" int bdetr_zero(float* p, int64_t n, void* stream) {
    BDETR_CHECK_ARG(p != nullptr && n >= 0, "bdetr_zero: bad arguments");
    if (n == 0) return 0;
    if (int e = bdetr_zero_bytes(p, sizeof(float) * (size_t)n, (hipStream_t)stream)) return e;
    return 0;
}
extern "C" int bdetr_add(const float* a, const float* b, float* out, int64_t n, void* stream) { return ew_launch<EW_ADD>("bdetr_add", a, b, out, n, 0.f, stream); }
extern "C" int bdetr_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream) { return ew_launch<EW_SIGMOID>("bdetr_sigmoid_fwd", x, nullptr, y, n, 0.f, stream); }
extern "C" int bdetr_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) { return ew_launch<EW_SIGMOID_BWD>("bdetr_sigmoid_bwd", y, dy, dx, n, 0.f, stream); }
extern "C" int bdetr_boxsigmoid_fwd(const float* x, float* y, int64_t n, void* stream) { return ew_launch<EW_BOXSIG>("bdetr_boxsigmoid_fwd", x, nullptr, y, n, 0.f, stream); }
extern "C" int bdetr_boxsigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) { return ew_launch<EW_BOXSIG_BWD>("bdetr_boxsigmoid_bwd", y, dy, dx, n, 0.f, stream); }
extern "C" int bdetr_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) { return ew_launch<EW_TANH_BWD>("bdetr_tanh_bwd", y, dy, dx, n, 0.f, stream); }
extern "C" int bdetr_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) { return ew_launch<EW_RELU_BWD>("bdetr_relu_bwd", y, dy, dx, n, 0.f, stream); }
extern "C" int bdetr_axpy(float alpha, const float* x, float* y, int64_t n, void* stream) { return ew_launch<EW_AXPY>("bdetr_axpy", x, nullptr, y, n, alpha, stream); }

extern "C" int bdetr_add_bcast_rows(const float* a, const float* row, float* out, int64_t rows, int64_t rowlen, void* stream) {
    BDETR_CHECK_ARG(a && row && out && rows > 0 && rowlen > 0, "bdetr_add_bcast_rows: bad arguments");
    int64_t n = rows * rowlen;
    hipLaunchKernelGGL(add_bcast_rows_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, row, out, n, rowlen);
    return bdetr_launch_status("add_bcast_rows");
}
extern "C" int bdetr_sum_over_batch(const float* x, float* out, int64_t batch, int64_t n, int accumulate, void* stream) {
    BDETR_CHECK_ARG(x && out && batch > 0 && n > 0, "bdetr_sum_over_batch: bad arguments");
    hipLaunchKernelGGL(sum_over_batch_kernel, dim3(ew_grid(n, 256, 1)), dim3(256), 0, (hipStream_t)stream, x, out, batch, n, accumulate);
    return bdetr_launch_status("sum_over_batch");
}
static int64_t colsum_rows_per_chunk(int64_t rows, int cols) {
    // enough chunks to fill the chip with (cols/64) x nch blocks, at least 32 rows per chunk
    int64_t colblocks = (cols + 63) / 64;
    int64_t want = cdiv64(1024, colblocks);
    int64_t rpc = cdiv64(rows, want);
    if (rpc < 32) rpc = 32;
    return rpc;
}
// out[c] += sum over rows of x[.][c], one launch: every block reduces its row chunk in LDS and adds the 64 column sums with
// float atomics (out must hold zeros or the running sum: the flat gradient buffer is zero-filled once per step)
extern "C" int bdetr_colsum_accumulate(const float* x, int64_t rows, int cols, float* out, void* stream) {
    BDETR_CHECK_ARG(x && out && rows > 0 && cols > 0, "bdetr_colsum_accumulate: bad arguments");
    int64_t rpc = colsum_rows_per_chunk(rows, cols);
    int nch = (int)cdiv64(rows, rpc);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, nch), dim3(256), 0, (hipStream_t)stream, x, rows, cols, rpc, (float*)nullptr, out);
    return bdetr_launch_status("colsum_accumulate");
}

// n <= 4 column sums of one width in ONE launch (the bias gradients of an attention block's Q / K / V projections, transformers.py:68-70
// + autodiff): outs[m][c] += sum over rows[m] of xs[m][.][c], float atomics like bdetr_colsum_accumulate (outs must hold zeros or the
// running sum).  xs / rows / outs are HOST arrays of n entries.
extern "C" int bdetr_colsum_accumulate_group(const float* const* xs, const int64_t* rows, int cols, float* const* outs, int n, void* stream) {
    BDETR_CHECK_ARG(xs && rows && outs && cols > 0 && n >= 1 && n <= 4, "bdetr_colsum_accumulate_group: bad arguments (1 <= n <= 4)");
    ColsumGroup g{};
    int nch = 1;
    for (int m = 0; m < n; ++m) {
        BDETR_CHECK_ARG(xs[m] && outs[m] && rows[m] > 0, "bdetr_colsum_accumulate_group: null / empty member %d", m);
        g.x[m] = xs[m]; g.out[m] = outs[m]; g.rows[m] = rows[m]; g.rpc[m] = colsum_rows_per_chunk(rows[m], cols);
        nch = max(nch, (int)cdiv64(rows[m], g.rpc[m]));
    }
    for (int m = n; m < 4; ++m) { g.x[m] = xs[0]; g.out[m] = outs[0]; g.rows[m] = 0; g.rpc[m] = 1; }
    hipLaunchKernelGGL(colsum_group_kernel, dim3((cols + 63) / 64, nch, n), dim3(256), 0, (hipStream_t)stream, g, cols);
    return bdetr_launch_status("colsum_accumulate_group");
}

extern "C" int bdetr_colsum_chunks(int64_t rows) { return (int)cdiv64(rows, 32); }   // upper bound for any cols
extern "C" int bdetr_colsum(const float* x, int64_t rows, int cols, float* out, float* ws, void* stream) {
    BDETR_CHECK_ARG(x && out && ws && rows > 0 && cols > 0, "bdetr_colsum: bad arguments");
    int64_t rpc = colsum_rows_per_chunk(rows, cols);
    int nch = (int)cdiv64(rows, rpc);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, nch), dim3(256), 0, (hipStream_t)stream, x, rows, cols, rpc, ws, (float*)nullptr);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, ws, nch, cols, out);
    return bdetr_launch_status("colsum");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Placement of the side stream (round 5).  HIP hands a new stream one of a few hardware queues in creation order, and which queue a
// stream sits on relative to the caller's stream decides how well the two overlap: measured on MI355X / ROCm 7.2 with the training step
// (tools/dp_gc_probe.py, BDETR_SIDE_QUEUE_SKIP = 0..3) the SAME step takes 25.1 / 25.3 / 25.2 / 44.3 ms on the four queues a
// low-priority stream can land on - on the worst one every small kernel of the critical path waits for the dispatch of the side stream's
// large grids (3-8 x their duration) - and which of the four a process gets depends on what else created streams before (creating
// the RCCL process group before or after the model moved the step from 25.7 to 45 ms).  So the stream is CHOSEN by measurement: `ncand`
// low-priority streams are created, each is loaded with `loads` launches of a streaming kernel over a 64-MB buffer while `ticks`
// one-workgroup kernels run back to back on the caller's stream, and the time those take under each candidate comes back in scores_ms
// (the second pass of two; 0.76 against 4.5 ms on the bad queue; unloaded_ms = the same ticks with nothing beside them).  The candidates
// are returned; the host keeps the first good one and destroys the others (bdetr_stream_destroy: three more idle low-priority queues
// measured 0.5 % on the step), a data-parallel model asks for more and settles between the good ones by timing steps (training.Model).
// Synchronises the streams involved: call it outside any capture.
namespace {
__global__ __launch_bounds__(256) void sidecal_load_kernel(f32x4* __restrict__ x, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { f32x4 v = x[i]; v = v * 1.0001f + 1.f; x[i] = v; }
}
__global__ void sidecal_tick_kernel(int* p) { if (threadIdx.x == 0) p[0] += 1; }
}  // namespace

extern "C" int bdetr_stream_destroy(void* stream) {
    BDETR_CHECK_ARG(stream != nullptr, "bdetr_stream_destroy: null stream");
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { bdetr_set_error("bdetr_stream_destroy: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int bdetr_side_stream_candidates(void* main_stream, int ncand, int loads, int ticks, void** streams_out, float* scores_ms, float* unloaded_ms) {
    BDETR_CHECK_ARG(streams_out != nullptr && scores_ms != nullptr && ncand >= 1 && ncand <= 8 && loads >= 1 && ticks >= 1,
                    "bdetr_side_stream_candidates: bad arguments (1 <= ncand <= 8)");
    hipStream_t cand[8] = {};
    for (int c = 0; c < ncand; ++c) {
        const int rc = bdetr_low_priority_stream_create((void**)&cand[c]);
        if (rc != 0) { for (int d = 0; d < c; ++d) (void)hipStreamDestroy(cand[d]); return rc; }
    }
    const int64_t n4 = (int64_t)4 << 20;                       // 64 MB
    f32x4* buf = nullptr; int* cnt = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t ms = (hipStream_t)main_stream;
    hipError_t e = hipMalloc((void**)&buf, n4 * sizeof(f32x4));
    if (e == hipSuccess) e = hipMalloc((void**)&cnt, sizeof(int));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, sizeof(int), ms);
    if (e == hipSuccess) e = hipMemsetAsync(buf, 0, n4 * sizeof(f32x4), ms);
    if (e == hipSuccess) e = hipStreamSynchronize(ms);
    if (unloaded_ms != nullptr) {                               // the reference: the same ticks with nothing beside them
        float ms_0 = 0.f;
        for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
            e = hipEventRecord(e0, ms);
            for (int t = 0; t < ticks; ++t) hipLaunchKernelGGL(sidecal_tick_kernel, dim3(1), dim3(64), 0, ms, cnt);
            if (e == hipSuccess) e = hipEventRecord(e1, ms);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms_0, e0, e1);
        }
        *unloaded_ms = ms_0;
    }
    for (int c = 0; c < ncand && e == hipSuccess; ++c) {
        float ms_c = 0.f;
        for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
            for (int l = 0; l < loads; ++l) hipLaunchKernelGGL(sidecal_load_kernel, dim3((unsigned)(n4 / 256)), dim3(256), 0, cand[c], buf, n4);
            e = hipEventRecord(e0, ms);
            for (int t = 0; t < ticks; ++t) hipLaunchKernelGGL(sidecal_tick_kernel, dim3(1), dim3(64), 0, ms, cnt);
            if (e == hipSuccess) e = hipEventRecord(e1, ms);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipStreamSynchronize(cand[c]);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms_c, e0, e1);
        }
        scores_ms[c] = ms_c;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (buf) (void)hipFree(buf);
    if (cnt) (void)hipFree(cnt);
    if (e != hipSuccess) {
        for (int c = 0; c < ncand; ++c) (void)hipStreamDestroy(cand[c]);
        bdetr_set_error("bdetr_side_stream_candidates: %s", hipGetErrorString(e));
        return (int)e;
    }
    for (int c = 0; c < ncand; ++c) streams_out[c] = (void*)cand[c];
    return bdetr_launch_status("side_stream_candidates");
}
