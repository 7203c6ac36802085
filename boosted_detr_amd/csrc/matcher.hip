// matcher.hip - on-GPU set criterion for gfx950: pairwise cost matrix, exact rectangular
// linear-sum-assignment (one image per workgroup, wavefront-parallel shortest augmenting
// path with all solver state in LDS, fp64, scipy's scan order and tie rule), and the masked
// loss reduction with its sparse backward.
//
// Replaces the reference's only device->host hop (losses_and_metrics.py:234-251:
// tf.numpy_function -> Python loop -> scipy.optimize.linear_sum_assignment) and the TF/TFA
// loss graph around it (losses_and_metrics.py:14-72, 111-161, 176-192).
//
// Compiled with -ffp-contract=off: the solver must evaluate ((minVal + c) - u[i]) - v[j]
// exactly like scipy's C++ (no fused multiply-add anywhere in this file).
#include "common.h"

namespace {

constexpr float KEPS = 1e-7f;     // tf.keras.backend.epsilon()
constexpr float CLIP_LO = 0.001f, CLIP_HI = 0.999f;   // safe_clip (losses_and_metrics.py:26-27)
constexpr float FOCAL_ALPHA = 0.25f;                  // TFA SigmoidFocalCrossEntropy defaults, gamma = 2

__device__ __forceinline__ float safe_clip(float p) { return fminf(fmaxf(p, CLIP_LO), CLIP_HI); }
__device__ __forceinline__ bool in_clip(float p) { return p >= CLIP_LO && p <= CLIP_HI; }

// Keras binary_crossentropy element (S9) for y in {0,1}
__device__ __forceinline__ float bce_elem(float y, float p) {
    float o = fminf(fmaxf(p, KEPS), 1.0f - KEPS);
    return -(y * logf(o + KEPS) + (1.0f - y) * logf(1.0f - o + KEPS));
}

// TFA sigmoid_focal_crossentropy element on q = safe_clip(p) (S10)
__device__ __forceinline__ float focal_elem(float y, float q) {
    float ce = bce_elem(y, q);
    float p_t = y * q + (1.0f - y) * (1.0f - q);
    float a_t = y * FOCAL_ALPHA + (1.0f - y) * (1.0f - FOCAL_ALPHA);
    float m = 1.0f - p_t;
    return a_t * (m * m) * ce;
}
// d focal / d q
__device__ __forceinline__ float focal_grad(float y, float q) {
    float o = fminf(fmaxf(q, KEPS), 1.0f - KEPS);   // identity on the clipped range
    float ce = -(y * logf(o + KEPS) + (1.0f - y) * logf(1.0f - o + KEPS));
    float dce = -(y / (o + KEPS) - (1.0f - y) / (1.0f - o + KEPS));
    float p_t = y * q + (1.0f - y) * (1.0f - q);
    float dp_t = 2.0f * y - 1.0f;
    float a_t = y * FOCAL_ALPHA + (1.0f - y) * (1.0f - FOCAL_ALPHA);
    float m = 1.0f - p_t;
    return a_t * (m * m * dce - 2.0f * m * dp_t * ce);
}

struct Box4 { float ymin, xmin, ymax, xmax; };
__device__ __forceinline__ Box4 coco_to_tf(const float* b) {   // losses_and_metrics.py:59-66
    Box4 r; r.ymin = b[1]; r.xmin = b[0]; r.ymax = b[1] + b[3]; r.xmax = b[0] + b[2]; return r;
}
__device__ __forceinline__ float div_no_nan(float a, float b) { return b == 0.f ? 0.f : a / b; }

// TFA _calculate_giou (S11). mode 0 = iou, 1 = giou
__device__ __forceinline__ float giou_fwd(const Box4& a, const Box4& b, int mode) {
    float aw = fmaxf(0.f, a.xmax - a.xmin), ah = fmaxf(0.f, a.ymax - a.ymin);
    float bw = fmaxf(0.f, b.xmax - b.xmin), bh = fmaxf(0.f, b.ymax - b.ymin);
    float aa = aw * ah, ba = bw * bh;
    float iw = fmaxf(0.f, fminf(a.xmax, b.xmax) - fmaxf(a.xmin, b.xmin));
    float ih = fmaxf(0.f, fminf(a.ymax, b.ymax) - fmaxf(a.ymin, b.ymin));
    float inter = iw * ih;
    float uni = aa + ba - inter;
    float iou = div_no_nan(inter, uni);
    if (mode == 0) return iou;
    float ew = fmaxf(0.f, fmaxf(a.xmax, b.xmax) - fminf(a.xmin, b.xmin));
    float eh = fmaxf(0.f, fmaxf(a.ymax, b.ymax) - fminf(a.ymin, b.ymin));
    float enc = ew * eh;
    return iou - div_no_nan(enc - uni, enc);
}

// reverse-mode gradient of giou(a, b) w.r.t. b (the prediction); g = upstream gradient
__device__ __forceinline__ void giou_bwd(const Box4& a, const Box4& b, float g, float db[4] /* ymin,xmin,ymax,xmax */) {
    float aw = fmaxf(0.f, a.xmax - a.xmin), ah = fmaxf(0.f, a.ymax - a.ymin);
    float bwr = b.xmax - b.xmin, bhr = b.ymax - b.ymin;
    float bw = fmaxf(0.f, bwr), bh = fmaxf(0.f, bhr);
    float aa = aw * ah, ba = bw * bh;
    float ixmax = fminf(a.xmax, b.xmax), ixmin = fmaxf(a.xmin, b.xmin);
    float iymax = fminf(a.ymax, b.ymax), iymin = fmaxf(a.ymin, b.ymin);
    float iwr = ixmax - ixmin, ihr = iymax - iymin;
    float iw = fmaxf(0.f, iwr), ih = fmaxf(0.f, ihr);
    float inter = iw * ih;
    float uni = aa + ba - inter;
    float exmax = fmaxf(a.xmax, b.xmax), exmin = fminf(a.xmin, b.xmin);
    float eymax = fmaxf(a.ymax, b.ymax), eymin = fminf(a.ymin, b.ymin);
    float ewr = exmax - exmin, ehr = eymax - eymin;
    float ew = fmaxf(0.f, ewr), eh = fmaxf(0.f, ehr);
    float enc = ew * eh;
    // giou = inter/uni - (enc - uni)/enc
    float g_inter = 0.f, g_uni = 0.f, g_enc = 0.f;
    if (uni != 0.f) { g_inter += g / uni; g_uni += -g * inter / (uni * uni); }
    if (enc != 0.f) { g_enc += -g * uni / (enc * enc); g_uni += g / enc; }   // d/denc[-(enc-uni)/enc] = -uni/enc^2 ; d/duni = 1/enc
    float g_ba = g_uni; g_inter += -g_uni;
    float g_iw = g_inter * ih, g_ih = g_inter * iw;
    float g_ew = g_enc * eh, g_eh = g_enc * ew;
    float g_bw = g_ba * bh, g_bh = g_ba * bw;
    float g_xmin = 0.f, g_xmax = 0.f, g_ymin = 0.f, g_ymax = 0.f;
    // tf.maximum(zero, d): gradient reaches d only where d > 0
    if (bwr > 0.f) { g_xmax += g_bw; g_xmin -= g_bw; }
    if (bhr > 0.f) { g_ymax += g_bh; g_ymin -= g_bh; }
    if (iwr > 0.f) {   // ixmax = min(a.xmax, b.xmax) ; ixmin = max(a.xmin, b.xmin)
        if (b.xmax <= a.xmax) g_xmax += (b.xmax == a.xmax ? 0.5f : 1.f) * g_iw;
        if (b.xmin >= a.xmin) g_xmin -= (b.xmin == a.xmin ? 0.5f : 1.f) * g_iw;
    }
    if (ihr > 0.f) {
        if (b.ymax <= a.ymax) g_ymax += (b.ymax == a.ymax ? 0.5f : 1.f) * g_ih;
        if (b.ymin >= a.ymin) g_ymin -= (b.ymin == a.ymin ? 0.5f : 1.f) * g_ih;
    }
    if (ewr > 0.f) {   // exmax = max, exmin = min
        if (b.xmax >= a.xmax) g_xmax += (b.xmax == a.xmax ? 0.5f : 1.f) * g_ew;
        if (b.xmin <= a.xmin) g_xmin -= (b.xmin == a.xmin ? 0.5f : 1.f) * g_ew;
    }
    if (ehr > 0.f) {
        if (b.ymax >= a.ymax) g_ymax += (b.ymax == a.ymax ? 0.5f : 1.f) * g_eh;
        if (b.ymin <= a.ymin) g_ymin -= (b.ymin == a.ymin ? 0.5f : 1.f) * g_eh;
    }
    db[0] = g_ymin; db[1] = g_xmin; db[2] = g_ymax; db[3] = g_xmax;
}

// BoxLoss (losses_and_metrics.py:68-72): 2*(1-giou) + 5*mean4((10*yt - 10*yp)^2)
__device__ __forceinline__ float box_loss_fwd(const float* tb, const float* pb) {
    Box4 t = coco_to_tf(tb), p = coco_to_tf(pb);
    float giou_loss = 1.0f - giou_fwd(t, p, 1);
    float d0 = 10.0f * t.ymin - 10.0f * p.ymin, d1 = 10.0f * t.xmin - 10.0f * p.xmin;
    float d2 = 10.0f * t.ymax - 10.0f * p.ymax, d3 = 10.0f * t.xmax - 10.0f * p.xmax;
    float l2 = (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) / 4.0f;
    return 2.0f * giou_loss + 5.0f * l2;
}
// gradient w.r.t. the predicted COCO box [xmin,ymin,w,h]
__device__ __forceinline__ void box_loss_bwd(const float* tb, const float* pb, float g, float dpb[4]) {
    Box4 t = coco_to_tf(tb), p = coco_to_tf(pb);
    float db[4];
    giou_bwd(t, p, -2.0f * g, db);
    const float tv[4] = {t.ymin, t.xmin, t.ymax, t.xmax}, pv[4] = {p.ymin, p.xmin, p.ymax, p.xmax};
#pragma unroll
    for (int k = 0; k < 4; ++k) db[k] += g * 5.0f * 0.25f * 2.0f * (10.0f * tv[k] - 10.0f * pv[k]) * (-10.0f);
    // tf box = [y, x, y+h, x+w]
    dpb[0] = db[1] + db[3];   // d/dx
    dpb[1] = db[0] + db[2];   // d/dy
    dpb[2] = db[3];           // d/dw
    dpb[3] = db[2];           // d/dh
}

__device__ __forceinline__ float cat_cost_elem(float p, int C) {     // CategoryLoss mean over C of the single live term
    return -logf(safe_clip(p) + KEPS) / (float)C;
}

__device__ float att_cost_elem(const float* hot_m, const float* pred_n, int A) {   // AttributeLoss: mean over A of focal
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += focal_elem(hot_m[a], safe_clip(pred_n[a]));
    return s / (float)A;
}

// ------------------------------------------------------------------------------------
// K10 cost matrix: grid (M, B), 64 threads; rows m >= num_objects[b] are written as zero
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cost_matrix_kernel(bdetr_loss_desc d, const float* __restrict__ cat_pred, const float* __restrict__ att_pred,
                                                         const float* __restrict__ box_pred, const int32_t* __restrict__ cat_ids,
                                                         const float* __restrict__ att_hot, const float* __restrict__ bbox,
                                                         const int32_t* __restrict__ num_objects,
                                                         float* __restrict__ cost, float* __restrict__ ccat, float* __restrict__ catt, float* __restrict__ cbox) {
    const int m = blockIdx.x, b = blockIdx.y;
    const bool live = num_objects == nullptr || m < num_objects[b];
    const int64_t row = ((int64_t)b * d.M + m) * d.N;
    const int cid = cat_ids[(int64_t)b * d.M + m];
    const float* tb = bbox + ((int64_t)b * d.M + m) * 4;
    const bool use_att = d.attribute_weight != 0.f && att_hot != nullptr && d.A > 0;
    for (int n = threadIdx.x; n < d.N; n += 64) {
        float c = 0.f, a = 0.f, bx = 0.f;
        if (live) {
            c = d.category_weight * cat_cost_elem(cat_pred[((int64_t)b * d.N + n) * d.C + cid], d.C);
            bx = d.box_weight * box_loss_fwd(tb, box_pred + ((int64_t)b * d.N + n) * 4);
            if (use_att) a = d.attribute_weight * att_cost_elem(att_hot + ((int64_t)b * d.M + m) * d.A, att_pred + ((int64_t)b * d.N + n) * d.A, d.A);
        }
        cost[row + n] = (c + bx) + a;      // category_cost + box_cost + attribute_cost (line 130)
        if (ccat) ccat[row + n] = c;
        if (catt) catt[row + n] = a;
        if (cbox) cbox[row + n] = bx;
    }
}

// ------------------------------------------------------------------------------------
// K11 exact LSA: one wavefront (64 lanes) per image.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// LDS layout (dynamic): double u[nr], v[nc], spc[nc]; int path[nc], col4row[nr], row4col[nc],
// remaining[nc]; uchar SR[nr], SC[nc]; then (optionally) the fp32 cost block.
template <bool COST_IN_LDS>
__global__ __launch_bounds__(64) void lsa_kernel(const float* __restrict__ cost, const int32_t* __restrict__ num_objects,
                                                 int M, int N, int32_t* __restrict__ match, int maxdim) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    int n_obj = num_objects[b];
    n_obj = max(0, min(n_obj, M));
    int32_t* mrow = match + (int64_t)b * M;
    for (int m = lane; m < M; m += 64) mrow[m] = -1;
    if (n_obj == 0 || N == 0) return;
    const float* cmat = cost + (int64_t)b * M * N;
    // scipy transposes a tall matrix; rows of the solver are then predictions
    const bool tr = N < n_obj;
    const int nr = tr ? N : n_obj, nc = tr ? n_obj : N;

    double* u = reinterpret_cast<double*>(smem);
    double* v = u + maxdim;
    double* spc = v + maxdim;
    int* path = reinterpret_cast<int*>(spc + maxdim);
    int* col4row = path + maxdim;
    int* row4col = col4row + maxdim;
    int* remaining = row4col + maxdim;
    unsigned char* SR = reinterpret_cast<unsigned char*>(remaining + maxdim);
    unsigned char* SC = SR + maxdim;
    float* lcost = reinterpret_cast<float*>(SC + maxdim + ((16 - (2 * maxdim) % 16) % 16));
    __shared__ int s_bad;

    if (lane == 0) s_bad = 0;
    for (int k = lane; k < nr; k += 64) { u[k] = 0.0; col4row[k] = -1; }
    for (int k = lane; k < nc; k += 64) { v[k] = 0.0; row4col[k] = -1; path[k] = -1; }
    __syncthreads();
    {   // validity test (scipy raises on NaN / -inf) + optional LDS copy, in solver orientation
        int bad = 0;
        for (int k = lane; k < n_obj * N; k += 64) {
            float x = cmat[k];
            if (x != x || x == -INFINITY) bad = 1;
            if (COST_IN_LDS) {
                int i = k / N, j = k - i * N;
                if (tr) lcost[j * nc + i] = x; else lcost[k] = x;
            }
        }
        if (bad) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) return;      // match stays -1 (the Python host reports ValueError like scipy)

    auto C = [&](int i, int j) -> double {
        if (COST_IN_LDS) return (double)lcost[i * nc + j];
        return (double)(tr ? cmat[(int64_t)j * N + i] : cmat[(int64_t)i * N + j]);
    };

    bool infeasible = false;           // wave-uniform
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; spc[it] = INFINITY; SC[it] = 0; }
        for (int k = lane; k < nr; k += 64) SR[k] = 0;
        __syncthreads();
        int sink = -1, i = cur;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            // each lane scans its strided positions in scan order; it keeps (local lowest, first
            // position with that value, last position with that value whose column is unassigned)
            double lo = INFINITY; int first_it = 0x7fffffff, last_un = -1;
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = minVal + C(i, j) - ui - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const bool un = row4col[j] == -1;
                if (s < lo) { lo = s; first_it = it; last_un = un ? it : -1; }
                else if (s == lo) { if (first_it == 0x7fffffff) first_it = it; if (un) last_un = it; }
            }
            const double lowest = wave_min_f64(lo);
            // sequential semantics: `index` = last unassigned position among the entries equal to
            // the minimum, else the first such position.  With lowest == +inf no entry ever
            // satisfies `<`, only the (== && unassigned) arm - reproduced by the same rule.
            const bool mine = (lo == lowest) && (num_remaining > lane);
            const int g_last_un = wave_max_i32(mine ? last_un : -1);
            const int g_first = wave_min_i32((mine && first_it != 0x7fffffff) ? first_it : 0x7fffffff);
            minVal = lowest;
            if (lowest == INFINITY) { infeasible = true; break; }   // infeasible cost matrix (uniform)
            const int index = g_last_un >= 0 ? g_last_un : g_first;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            __syncthreads();                   // everyone has read remaining[index] / row4col[j]
            if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num_remaining - 1]; }
            --num_remaining;
            __syncthreads();
        }
        if (infeasible) break;
        // dual updates
        for (int k = lane; k < nr; k += 64) {
            if (k == cur) u[k] += minVal;
            else if (SR[k]) u[k] += minVal - spc[col4row[k]];
        }
        for (int k = lane; k < nc; k += 64) if (SC[k]) v[k] -= minVal - spc[k];
        __syncthreads();
        if (lane == 0) {                       // augment along the alternating path
            int j = sink;
            for (;;) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r]; col4row[r] = j; j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (infeasible) return;            // match stays -1
    if (!tr) { for (int k = lane; k < nr; k += 64) mrow[k] = col4row[k]; }
    else     { for (int k = lane; k < nr; k += 64) mrow[col4row[k]] = k; }   // solver row k = prediction, col = object
}

__global__ void match_to_mask_kernel(const int32_t* __restrict__ match, float* __restrict__ mask, int B, int M, int N) {
    int64_t n = (int64_t)B * M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int col = (int)(i % N); int64_t bm = i / N;
        mask[i] = match[bm] == col ? 1.0f : 0.0f;
    }
}

// ------------------------------------------------------------------------------------
// K12 loss + sparse backward: one workgroup (256 threads) per image
// ------------------------------------------------------------------------------------
__device__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void set_loss_kernel(bdetr_loss_desc d, const float* __restrict__ cat_pred, const float* __restrict__ att_pred,
                                                       const float* __restrict__ box_pred, const int32_t* __restrict__ cat_ids,
                                                       const float* __restrict__ att_hot, const float* __restrict__ bbox,
                                                       const int32_t* __restrict__ num_objects, const int32_t* __restrict__ match,
                                                       float* __restrict__ losses, float* __restrict__ d_cat, float* __restrict__ d_att,
                                                       float* __restrict__ d_box, float loss_scale) {
    __shared__ float sh[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    int* assigned = reinterpret_cast<int*>(smem2);            // [N] -> matched object or -1
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool use_att = d.attribute_weight != 0.f && att_hot != nullptr && d.A > 0 && att_pred != nullptr;
    // tot = 1 + sum over the (replica's) batch of num_objects (lines 144-145)
    float cnt = 0.f;
    for (int k = tid; k < d.B; k += 256) cnt += (float)num_objects[k];
    const float tot = 1.0f + block_sum(cnt, sh);
    const float np1 = 1.0f + (float)d.N;
    const int n_obj = max(0, min(num_objects[b], d.M));

    for (int n = tid; n < d.N; n += 256) assigned[n] = -1;
    // zero this image's gradient slices
    if (d_cat) for (int k = tid; k < d.N * d.C; k += 256) d_cat[(int64_t)b * d.N * d.C + k] = 0.f;
    if (d_att && d.A > 0) for (int k = tid; k < d.N * d.A; k += 256) d_att[(int64_t)b * d.N * d.A + k] = 0.f;
    if (d_box) for (int k = tid; k < d.N * 4; k += 256) d_box[(int64_t)b * d.N * 4 + k] = 0.f;
    __syncthreads();
    for (int m = tid; m < n_obj; m += 256) { int n = match[(int64_t)b * d.M + m]; if (n >= 0 && n < d.N) assigned[n] = m; }
    __syncthreads();

    float s_cat = 0.f, s_att = 0.f, s_box = 0.f, s_iou = 0.f, s_exist = 0.f;
    // matched pairs: one thread per prediction n that owns an object
    for (int n = tid; n < d.N; n += 256) {
        const int m = assigned[n];
        const float* pc = cat_pred + ((int64_t)b * d.N + n) * d.C;
        // existence loss on class 0 (lines 139-140): target = 1 - assigned
        {
            const float y = m >= 0 ? 0.f : 1.f;
            const float p0 = pc[0], q = safe_clip(p0);
            s_exist += d.exist_weight * bce_elem(y, q);
            if (d_cat && in_clip(p0)) {
                const float o = fminf(fmaxf(q, KEPS), 1.0f - KEPS);
                const float dq = -(y / (o + KEPS) - (1.0f - y) / (1.0f - o + KEPS));
                d_cat[((int64_t)b * d.N + n) * d.C + 0] += loss_scale * d.exist_weight * dq / ((float)d.N * np1);
            }
        }
        if (m < 0) continue;
        const int cid = cat_ids[(int64_t)b * d.M + m];
        const float* tb = bbox + ((int64_t)b * d.M + m) * 4;
        const float* pb = box_pred + ((int64_t)b * d.N + n) * 4;
        const float pcid = pc[cid];
        s_cat += d.category_weight * cat_cost_elem(pcid, d.C);
        if (d_cat && in_clip(pcid))
            d_cat[((int64_t)b * d.N + n) * d.C + cid] += loss_scale * d.category_weight * (-1.0f / (safe_clip(pcid) + KEPS)) / ((float)d.C * tot);
        s_box += d.box_weight * box_loss_fwd(tb, pb);
        if (d_box && d.box_weight != 0.f) {
            float g4[4];
            box_loss_bwd(tb, pb, loss_scale * d.box_weight / tot, g4);
#pragma unroll
            for (int k = 0; k < 4; ++k) d_box[((int64_t)b * d.N + n) * 4 + k] = g4[k];
        }
        // IoU metric on the RAW COCO-format boxes (losses_and_metrics.py:188) - quirk reproduced
        {
            Box4 t{tb[0], tb[1], tb[2], tb[3]}, p{pb[0], pb[1], pb[2], pb[3]};
            s_iou += giou_fwd(t, p, 0);
        }
        if (use_att) {
            const float* hot = att_hot + ((int64_t)b * d.M + m) * d.A;
            const float* pa = att_pred + ((int64_t)b * d.N + n) * d.A;
            s_att += d.attribute_weight * att_cost_elem(hot, pa, d.A);
            if (d_att) {
                const float gs = loss_scale * d.attribute_weight / ((float)d.A * tot);
                for (int a = 0; a < d.A; ++a) {
                    const float p = pa[a];
                    if (in_clip(p)) d_att[((int64_t)b * d.N + n) * d.A + a] = gs * focal_grad(hot[a], safe_clip(p));
                }
            }
        }
    }
    const float t_cat = block_sum(s_cat, sh) / tot;
    const float t_att = block_sum(s_att, sh) / tot;
    const float t_box = block_sum(s_box, sh) / tot;
    const float t_iou = block_sum(s_iou, sh) / tot;
    const float t_exist = (block_sum(s_exist, sh) / (float)d.N) / np1;
    if (tid == 0) {
        losses[0 * d.B + b] = ((t_cat + t_att) + t_box) + t_exist;   // line 153 order
        losses[1 * d.B + b] = t_cat;
        losses[2 * d.B + b] = t_att;
        losses[3 * d.B + b] = t_box;
        losses[4 * d.B + b] = t_exist;
        losses[5 * d.B + b] = t_iou;
    }
}

int check_loss_desc(const bdetr_loss_desc* d, const char* who) {
    BDETR_CHECK_ARG(d != nullptr, "%s: null desc", who);
    BDETR_CHECK_ARG(d->B > 0 && d->M > 0 && d->N > 0 && d->C > 0 && d->A >= 0, "%s: bad sizes B=%d M=%d N=%d C=%d A=%d", who, d->B, d->M, d->N, d->C, d->A);
    return 0;
}

}  // namespace

extern "C" int bdetr_cost_matrix(const bdetr_loss_desc* d, const float* cat_pred, const float* att_pred,
                                 const float* box_pred, const int32_t* cat_ids, const float* att_hot,
                                 const float* bbox, const int32_t* num_objects, float* cost, float* cost_cat, float* cost_att,
                                 float* cost_box, void* stream) {
    if (int e = check_loss_desc(d, "bdetr_cost_matrix")) return e;
    BDETR_CHECK_ARG(cat_pred && box_pred && cat_ids && bbox && cost, "bdetr_cost_matrix: null pointer");
    BDETR_CHECK_ARG(d->attribute_weight == 0.f || d->A == 0 || (att_pred && att_hot), "bdetr_cost_matrix: attribute tensors required when attribute_weight != 0");
    hipLaunchKernelGGL(cost_matrix_kernel, dim3(d->M, d->B), dim3(64), 0, (hipStream_t)stream, *d, cat_pred, att_pred, box_pred, cat_ids, att_hot, bbox,
                       num_objects, cost, cost_cat, cost_att, cost_box);
    return bdetr_launch_status("cost_matrix");
}

extern "C" int bdetr_lsa(const float* cost, const int32_t* num_objects, int B, int M, int N, int32_t* match, void* stream) {
    BDETR_CHECK_ARG(cost && num_objects && match && B > 0 && M > 0 && N > 0, "bdetr_lsa: bad arguments");
    const int maxdim = M > N ? M : N;
    size_t state = (size_t)maxdim * (3 * sizeof(double) + 4 * sizeof(int) + 2) + 16;
    size_t with_cost = state + (size_t)M * N * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (with_cost <= 150 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsa_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)with_cost);
        if (e != hipSuccess) { bdetr_set_error("bdetr_lsa: cannot reserve %zu bytes of LDS: %s", with_cost, hipGetErrorString(e)); return (int)e; }
        hipLaunchKernelGGL((lsa_kernel<true>), dim3(B), dim3(64), with_cost, st, cost, num_objects, M, N, match, maxdim);
    } else {
        BDETR_CHECK_ARG(state <= 150 * 1024, "bdetr_lsa: problem too large (max(M,N)=%d)", maxdim);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsa_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)state);
        if (e != hipSuccess) { bdetr_set_error("bdetr_lsa: cannot reserve %zu bytes of LDS: %s", state, hipGetErrorString(e)); return (int)e; }
        hipLaunchKernelGGL((lsa_kernel<false>), dim3(B), dim3(64), state, st, cost, num_objects, M, N, match, maxdim);
    }
    return bdetr_launch_status("lsa");
}

extern "C" int bdetr_match_to_mask(const int32_t* match, float* mask, int B, int M, int N, void* stream) {
    BDETR_CHECK_ARG(match && mask && B > 0 && M > 0 && N > 0, "bdetr_match_to_mask: bad arguments");
    hipLaunchKernelGGL(match_to_mask_kernel, dim3(ew_grid((int64_t)B * M * N)), dim3(256), 0, (hipStream_t)stream, match, mask, B, M, N);
    return bdetr_launch_status("match_to_mask");
}

extern "C" int bdetr_set_loss(const bdetr_loss_desc* d, const float* cat_pred, const float* att_pred,
                              const float* box_pred, const int32_t* cat_ids, const float* att_hot,
                              const float* bbox, const int32_t* num_objects, const int32_t* match,
                              float* losses, float* d_cat, float* d_att, float* d_box, float loss_scale,
                              void* stream) {
    if (int e = check_loss_desc(d, "bdetr_set_loss")) return e;
    BDETR_CHECK_ARG(cat_pred && box_pred && cat_ids && bbox && num_objects && match && losses, "bdetr_set_loss: null pointer");
    hipLaunchKernelGGL(set_loss_kernel, dim3(d->B), dim3(256), (size_t)d->N * sizeof(int), (hipStream_t)stream, *d, cat_pred, att_pred, box_pred,
                       cat_ids, att_hot, bbox, num_objects, match, losses, d_cat, d_att, d_box, loss_scale);
    return bdetr_launch_status("set_loss");
}
