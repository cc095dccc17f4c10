// Detic box-head losses in one pass (DG/divergen/modeling/roi_heads/detic_fast_rcnn.py:160-235):
//   loss_cls = sum_{r, c<C} w[c] * BCEWithLogits(x[r,c], [gt[r]==c]) / R      (sigmoid CE with the federated-loss /
//              zero-category class weights folded into w;  :271-304)
//   loss_box = sum_{selected rows} |deltas - get_deltas(prop, gt_box)|_1 / max(4*rows, 1)   (L1, class-agnostic; :160-235)
//   + the three classification statistics of D2 fast_rcnn.py:88-114.
// The eager form is ~60 tiny launches per cascade stage (one-hot target, masks, nonzero, gathers, reductions) and as
// many again in backward; here: one workgroup per RoI row computes the row's loss terms AND its gradient, a second
// kernel folds the rows in a fixed order (deterministic).  HBM-bound on the (R, C+1) logits: read once, gradient
// written once.
#include "dgx_common.h"

namespace {
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }
constexpr int NPART = 8;   // per-row partials: cls loss, box loss, box row selected, correct, fg correct, false neg, fg, unused
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void detic_rows_kernel(const T* __restrict__ logits, const T* __restrict__ deltas,
                                                         const int64_t* __restrict__ gt, const float* __restrict__ class_w,
                                                         const float* __restrict__ prop, const float* __restrict__ gtb,
                                                         const int64_t* __restrict__ src, int R, int C, float wx, float wy,
                                                         float ww, float wh, T* __restrict__ dlogits, float* __restrict__ dsign,
                                                         float* __restrict__ part) {
    __shared__ float red_l[4];
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const int r = blockIdx.x;
    const int64_t g = gt[r];
    const T* x = logits + (int64_t)r * (C + 1);
    T* dx = dlogits + (int64_t)r * (C + 1);
    const bool row_on = g >= 0;      // gt < 0: "ignore" row (dropped by the reference before the loss): no loss, no gradient
    float loss = 0.f, best = -INFINITY;
    int besti = 0x7fffffff;
    for (int c = threadIdx.x; c <= C; c += 256) {
        const float v = ld1<T>(x + c);
        if (v > best || (v == best && c < besti)) { best = v; besti = c; }
        if (c < C) {
            const float w = class_w ? class_w[c] : 1.0f;
            const float t = (g == c) ? 1.0f : 0.0f;
            // torch's binary_cross_entropy_with_logits: (1 - t) x + m + log(exp(-m) + exp(-x - m)),  m = max(-x, 0)
            const float m = fmaxf(-v, 0.0f);
            if (row_on) loss += w * ((1.0f - t) * v + m + logf(expf(-m) + expf(-v - m)));
            st1<T>(dx + c, row_on ? w * (1.0f / (1.0f + expf(-v)) - t) : 0.0f);      // scaled by 1/rows in backward
        } else {
            st1<T>(dx + c, 0.0f);      // the background column carries no loss
        }
    }
    // block reductions: loss sum and argmax (smallest index among equal maxima)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        loss += __shfl_xor(loss, o);
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red_l[w] = loss; red_v[w] = best; red_i[w] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float L = red_l[0] + red_l[1] + red_l[2] + red_l[3];
        float bv = red_v[0];
        int bi = red_i[0];
        for (int k = 1; k < 4; ++k)
            if (red_v[k] > bv || (red_v[k] == bv && red_i[k] < bi)) { bv = red_v[k]; bi = red_i[k]; }
        const bool fg = g >= 0 && g < C;
        // box regression (class-agnostic L1 on the Box2Box deltas), rows selected by foreground [& instance_source == 0]
        float lb = 0.f, sel = 0.f;
        float sg[4] = {0.f, 0.f, 0.f, 0.f};
        if (fg && (!src || src[r] == 0)) {
            const float* p = prop + 4 * (int64_t)r;
            const float* q = gtb + 4 * (int64_t)r;
            const float sw = p[2] - p[0], sh = p[3] - p[1];
            const float sx = p[0] + 0.5f * sw, sy = p[1] + 0.5f * sh;
            const float tw = q[2] - q[0], th = q[3] - q[1];
            const float tx = q[0] + 0.5f * tw, ty = q[1] + 0.5f * th;
            const float tg[4] = {wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * logf(tw / sw), wh * logf(th / sh)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = ld1<T>(deltas + 4 * (int64_t)r + k) - tg[k];
                lb += fabsf(d);
                sg[k] = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
            }
            sel = 1.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dsign[4 * (int64_t)r + k] = sg[k];
        float* o = part + (int64_t)r * NPART;
        o[0] = L;
        o[1] = lb;
        o[2] = sel;
        o[3] = (row_on && bi == g) ? 1.0f : 0.0f;
        o[4] = (bi == g && fg) ? 1.0f : 0.0f;
        o[5] = (bi == C && fg) ? 1.0f : 0.0f;
        o[6] = fg ? 1.0f : 0.0f;
        o[7] = row_on ? 1.0f : 0.0f;
    }
}

// out[0..7] = column sums of part over rows (fixed order), then the final normalisations
__global__ __launch_bounds__(256) void detic_fold_kernel(const float* __restrict__ part, int R, float* __restrict__ out) {
    __shared__ float red[4][NPART];
    float a[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) a[k] = 0.f;
    for (int r = threadIdx.x; r < R; r += 256)
#pragma unroll
        for (int k = 0; k < NPART; ++k) a[k] += part[(int64_t)r * NPART + k];
#pragma unroll
    for (int k = 0; k < NPART; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NPART; ++k) red[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    if (threadIdx.x < NPART) {
        const int k = threadIdx.x;
        out[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float rows = out[2];
        const float nrow = fmaxf(out[7], 1.0f);            // rows that take part (all of them unless some are "ignore")
        out[14] = 1.0f / nrow;                             // backward scale of the classification term
        out[8] = out[0] / nrow;                            // loss_cls
        out[9] = out[1] / fmaxf(4.0f * rows, 1.0f);        // loss_box_reg
        out[10] = 1.0f / fmaxf(4.0f * rows, 1.0f);         // backward scale of the box term
        const float nfg = fmaxf(out[6], 1.0f);
        out[11] = out[3] / nrow;                           // cls_accuracy
        out[12] = out[4] / nfg;                            // fg_cls_accuracy
        out[13] = out[5] / nfg;                            // false_negative
    }
}

extern "C" int dgx_detic_losses(const void* logits, const void* deltas, const int64_t* gt_classes, const float* class_w,
                                const float* prop, const float* gtb, const int64_t* src, int R, int C, float wx, float wy,
                                float ww, float wh, void* dlogits, float* dsign, float* out16, float* part, int dtype,
                                void* stream) {
    if (!out16) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (R <= 0) {
        (void)hipMemsetAsync(out16, 0, 16 * sizeof(float), st);
        return DGX_OK;
    }
    if (!logits || !deltas || !gt_classes || !prop || !gtb || !dlogits || !dsign || !part || C <= 0) return DGX_ERR_BAD_ARG;
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(detic_rows_kernel<uint16_t>, dim3(R), dim3(256), 0, st, (const uint16_t*)logits, (const uint16_t*)deltas,
                           gt_classes, class_w, prop, gtb, src, R, C, wx, wy, ww, wh, (uint16_t*)dlogits, dsign, part);
    else
        hipLaunchKernelGGL(detic_rows_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)logits, (const float*)deltas, gt_classes,
                           class_w, prop, gtb, src, R, C, wx, wy, ww, wh, (float*)dlogits, dsign, part);
    hipLaunchKernelGGL(detic_fold_kernel, dim3(1), dim3(256), 0, st, part, R, out16);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
