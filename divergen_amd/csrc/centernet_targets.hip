// CenterNet dense target assignment for gfx950: one lane per (image, level, y, x) location, the
// image's GT list staged in LDS, no M x N temporaries.  Restates
// CN/modeling/dense_heads/centernet.py:338-436 (+ :505-530, :551-562, :576-592) with the same fp32
// operation order (compiled with -ffp-contract=off): reg_targets are bit-exact, the heat-map
// differs from the CPU only by expf's last ulp.
#include "dgx_common.h"

#define CT_MAX_LEVELS 8
struct CtLevels {
    int h[CT_MAX_LEVELS], w[CT_MAX_LEVELS], stride[CT_MAX_LEVELS];
    int loc_base[CT_MAX_LEVELS];   // offset of the level inside one image's M locations
    int64_t out_base[CT_MAX_LEVELS];  // offset of the level block in the (level, image, y, x) layout
    float lo[CT_MAX_LEVELS], hi[CT_MAX_LEVELS];
    int L, M, B;
};

__global__ __launch_bounds__(256) void centernet_targets_kernel(const float* __restrict__ gt, const int32_t* __restrict__ offs,
                                                                CtLevels P, float coef, float min_r2,
                                                                float* __restrict__ reg, float* __restrict__ hm) {
    extern __shared__ float sb[];  // per GT: x1,y1,x2,y2,cx,cy,rad2
    const int img = blockIdx.y;
    const int g0 = offs[img], n = offs[img + 1] - g0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float x1 = gt[4 * (g0 + i)], y1 = gt[4 * (g0 + i) + 1], x2 = gt[4 * (g0 + i) + 2], y2 = gt[4 * (g0 + i) + 3];
        const float area = (x2 - x1) * (y2 - y1);
        sb[7 * i] = x1; sb[7 * i + 1] = y1; sb[7 * i + 2] = x2; sb[7 * i + 3] = y2;
        sb[7 * i + 4] = (x1 + x2) / 2;
        sb[7 * i + 5] = (y1 + y2) / 2;
        sb[7 * i + 6] = fmaxf(coef * area, min_r2);
    }
    __syncthreads();
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= P.M) return;
    int l = 0;
    while (l + 1 < P.L && m >= P.loc_base[l + 1]) ++l;
    const int loc = m - P.loc_base[l];
    const int yy = loc / P.w[l], xx = loc - yy * P.w[l];
    const int s = P.stride[l];
    const float st = (float)s;
    const float gx = (float)(xx * s) + (float)(s / 2), gy = (float)(yy * s) + (float)(s / 2);
    const int64_t o = P.out_base[l] + (int64_t)img * P.h[l] * P.w[l] + loc;
    const float INF = 100000000.0f;
    float best = INF, bestw = INFINITY;
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int i = 0; i < n; ++i) {
        const float x1 = sb[7 * i], y1 = sb[7 * i + 1], x2 = sb[7 * i + 2], y2 = sb[7 * i + 3];
        const float cx = sb[7 * i + 4], cy = sb[7 * i + 5];
        const float le = gx - x1, to = gy - y1, ri = x2 - gx, bo = y2 - gy;
        const float dx_ = (float)((int)(cx / st)) * st + st / 2, dy_ = (float)((int)(cy / st)) * st + st / 2;
        const bool peak = (gx == dx_) && (gy == dy_);
        const bool inbox = fminf(fminf(le, to), fminf(ri, bo)) > 0.0f;
        const bool c33 = (fabsf(gx - dx_) <= st) && (fabsf(gy - dy_) <= st) && inbox;
        const float sx = le + ri, sy = to + bo;
        const float crit = sqrtf(sx * sx + sy * sy) / 2;
        const bool cared = crit >= P.lo[l] && crit <= P.hi[l];
        const float ddx = gx - cx, ddy = gy - cy;
        float d2 = ddx * ddx + ddy * ddy;
        if (peak) d2 = 0.0f;
        const float wd = d2 / sb[7 * i + 6];
        bestw = fminf(bestw, wd);
        const float dist = (c33 && cared) ? wd : INF;
        if (dist < best) { best = dist; r0 = le; r1 = to; r2 = ri; r3 = bo; }
    }
    if (n == 0 || best == INF) { r0 = r1 = r2 = r3 = -INF; }
    reg[4 * o + 0] = r0 / st;
    reg[4 * o + 1] = r1 / st;
    reg[4 * o + 2] = r2 / st;
    reg[4 * o + 3] = r3 / st;
    float hv = 0.0f;
    if (n > 0) {
        hv = expf(-bestw);
        if (hv < 1e-4f) hv = 0.0f;
    }
    hm[o] = hv;
}

extern "C" int dgx_centernet_targets(const float* gt_boxes, const int32_t* gt_offsets, int B, const int32_t* level_hw,
                                     const int32_t* strides, const float* soi, int L, float delta, float min_radius,
                                     float* reg_targets, float* heatmap, void* stream) {
    // level_hw / strides / soi are small HOST arrays (they are configuration, not data)
    if (B <= 0) return DGX_OK;
    if (!gt_offsets || !level_hw || !strides || !soi || !reg_targets || !heatmap || L < 1 || L > CT_MAX_LEVELS)
        return DGX_ERR_BAD_ARG;
    CtLevels P = {};
    P.L = L; P.B = B;
    int m = 0;
    int64_t ob = 0;
    for (int l = 0; l < L; ++l) {
        P.h[l] = level_hw[2 * l]; P.w[l] = level_hw[2 * l + 1]; P.stride[l] = strides[l];
        P.lo[l] = soi[2 * l]; P.hi[l] = soi[2 * l + 1];
        P.loc_base[l] = m; P.out_base[l] = ob;
        m += P.h[l] * P.w[l];
        ob += (int64_t)B * P.h[l] * P.w[l];
    }
    P.M = m;
    if (m == 0) return DGX_OK;
    const float coef = (float)((double)delta * (double)delta * 2.0);
    const float min_r2 = (float)((double)min_radius * (double)min_radius);
    // GT list per image must fit LDS: 7 floats per box -> up to 1170 boxes per image
    const size_t sm = 32 * 1024;
    hipLaunchKernelGGL(centernet_targets_kernel, dim3((m + 255) / 256, B), dim3(256), sm, (hipStream_t)stream, gt_boxes,
                       gt_offsets, P, coef, min_r2, reg_targets, heatmap);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}


// Positive-location indices of the CenterNet losses (CN/modeling/dense_heads/centernet.py:439-483 `_get_label_inds`): for
// every ground-truth box and every FPN level the flat index (level-major layout: level, image, y, x) of the location that
// holds the box centre, and whether the box's size falls into the level's size-of-interest range.  One thread per
// (box, level); the float sequence is the reference's (centre = (x0 + x1) / 2, / stride, truncation; half diagonal).
__global__ __launch_bounds__(256) void centernet_label_inds_kernel(const float* __restrict__ gt, const int32_t* __restrict__ offs, CtLevels P,
                                                                   int total, int64_t* __restrict__ ind, uint8_t* __restrict__ cared) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total * P.L) return;
    const int n = t / P.L, l = t - n * P.L;
    int img = 0;
    while (img + 1 < P.B && n >= offs[img + 1]) ++img;
    const float x0 = gt[4 * n], y0 = gt[4 * n + 1], x1 = gt[4 * n + 2], y1 = gt[4 * n + 3];
    const float st = (float)P.stride[l];
    const int64_t cx = (int64_t)(((x0 + x1) / 2.0f) / st), cy = (int64_t)(((y0 + y1) / 2.0f) / st);
    ind[t] = P.out_base[l] + (int64_t)img * P.h[l] * P.w[l] + cy * P.w[l] + cx;
    const float dx = x1 - x0, dy = y1 - y0;
    const float crit = sqrtf(dx * dx + dy * dy) / 2.0f;
    cared[t] = (crit >= P.lo[l] && crit <= P.hi[l]) ? 1 : 0;
}
extern "C" int dgx_centernet_label_inds(const float* gt_boxes, const int32_t* gt_offsets, int B, int total, const int32_t* level_hw,
                                        const int32_t* strides, const float* soi, int L, int64_t* ind, uint8_t* cared, void* stream) {
    if (B <= 0 || total <= 0) return DGX_OK;
    if (!gt_boxes || !gt_offsets || !level_hw || !strides || !soi || !ind || !cared || L < 1 || L > CT_MAX_LEVELS) return DGX_ERR_BAD_ARG;
    CtLevels P = {};
    P.L = L; P.B = B;
    int64_t ob = 0;
    for (int l = 0; l < L; ++l) {
        P.h[l] = level_hw[2 * l]; P.w[l] = level_hw[2 * l + 1]; P.stride[l] = strides[l];
        P.lo[l] = soi[2 * l]; P.hi[l] = soi[2 * l + 1];
        P.out_base[l] = ob;
        ob += (int64_t)B * P.h[l] * P.w[l];
    }
    hipLaunchKernelGGL(centernet_label_inds_kernel, dim3((total * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, gt_boxes, gt_offsets, P,
                       total, ind, cared);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
