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
                                                         float* __restrict__ part, int64_t ldx, int64_t ldd, int64_t ldg,
                                                         int gcols) {
    __shared__ float red_l[4];
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const int r = blockIdx.x;
    const int64_t g = gt[r];
    const T* x = logits + (int64_t)r * ldx;
    T* dx = dlogits + (int64_t)r * ldg;
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
                const float d = ld1<T>(deltas + ldd * (int64_t)r + k) - tg[k];
                lb += fabsf(d);
                sg[k] = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
            }
            sel = 1.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dsign[4 * (int64_t)r + k] = sg[k];
        // joint layout (gcols > C + 1): the gradient row also carries the box-delta signs behind the logits and zero columns
        // up to gcols -- the K-padded operand of the predictor's input- and weight-gradient GEMMs (scaled later in place)
        for (int k = C + 1; k < gcols; ++k) st1<T>(dx + k, k < C + 5 ? sg[k - C - 1] : 0.0f);
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

extern "C" int dgx_detic_losses_strided(const void* logits, int64_t ld_logits, const void* deltas, int64_t ld_deltas,
                                        const int64_t* gt_classes, const float* class_w, const float* prop, const float* gtb,
                                        const int64_t* src, int R, int C, float wx, float wy, float ww, float wh, void* dlogits,
                                        int64_t ld_dlogits, int grad_cols, float* dsign, float* out16, float* part, int dtype,
                                        void* stream) {
    if (!out16) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (R <= 0) {
        (void)hipMemsetAsync(out16, 0, 16 * sizeof(float), st);
        return DGX_OK;
    }
    if (!logits || !deltas || !gt_classes || !prop || !gtb || !dlogits || !dsign || !part || C <= 0 || ld_logits < C + 1 ||
        ld_deltas < 4 || ld_dlogits < C + 1 || (grad_cols > C + 1 && (grad_cols < C + 5 || grad_cols > ld_dlogits)))
        return DGX_ERR_BAD_ARG;
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(detic_rows_kernel<uint16_t>, dim3(R), dim3(256), 0, st, (const uint16_t*)logits, (const uint16_t*)deltas,
                           gt_classes, class_w, prop, gtb, src, R, C, wx, wy, ww, wh, (uint16_t*)dlogits, dsign, part, ld_logits,
                           ld_deltas, ld_dlogits, grad_cols);
    else
        hipLaunchKernelGGL(detic_rows_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)logits, (const float*)deltas, gt_classes,
                           class_w, prop, gtb, src, R, C, wx, wy, ww, wh, (float*)dlogits, dsign, part, ld_logits, ld_deltas,
                           ld_dlogits, grad_cols);
    hipLaunchKernelGGL(detic_fold_kernel, dim3(1), dim3(256), 0, st, part, R, out16);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_detic_losses(const void* logits, const void* deltas, const int64_t* gt_classes, const float* class_w,
                                const float* prop, const float* gtb, const int64_t* src, int R, int C, float wx, float wy,
                                float ww, float wh, void* dlogits, float* dsign, float* out16, float* part, int dtype,
                                void* stream) {
    return dgx_detic_losses_strided(logits, C + 1, deltas, 4, gt_classes, class_w, prop, gtb, src, R, C, wx, wy, ww, wh, dlogits,
                                    C + 1, 0, dsign, out16, part, dtype, stream);
}

// In-place scaling of a joint gradient buffer (rows, ld) written by dgx_detic_losses_strided: columns [0, C + 1) by
// g_cls * out16[14], columns [C + 1, C + 5) by g_box * out16[10] -- the two loss terms' backward scales, all on the device.
template <typename T>
__global__ __launch_bounds__(256) void detic_grad_scale_kernel(T* __restrict__ dy, int64_t ld, int rows, int C1, const float* __restrict__ out16,
                                                               const float* __restrict__ g_cls, const float* __restrict__ g_box) {
    const float sc = g_cls[0] * out16[14], sb = g_box[0] * out16[10];
    const int cols = C1 + 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        T* p = dy + (int64_t)r * ld + c;
        st1<T>(p, ld1<T>(p) * (c < C1 ? sc : sb));
    }
}
extern "C" int dgx_detic_grad_scale(void* dy, int64_t ld, int rows, int C, const float* out16, const float* g_cls, const float* g_box,
                                    int dtype, void* stream) {
    if (rows <= 0) return DGX_OK;
    if (!dy || !out16 || !g_cls || !g_box || C <= 0 || ld < C + 5) return DGX_ERR_BAD_ARG;
    const int64_t n = (int64_t)rows * (C + 5);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(detic_grad_scale_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint16_t*)dy, ld, rows, C + 1,
                           out16, g_cls, g_box);
    else
        hipLaunchKernelGGL(detic_grad_scale_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)dy, ld, rows, C + 1, out16,
                           g_cls, g_box);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// Federated-loss class set (DG/divergen/modeling/utils.py:16-28 get_fed_loss_inds) as a 0/1 mask over C + 1 classes, one
// workgroup: the classes that appear among gt_classes, plus -- when fewer than K appear -- the (K - #appeared) classes with
// the largest  prob[c] / expo[c]  among the others (prob > 0).  torch.multinomial(prob, n, replacement=False) IS the top-n of
// prob / Exponential(1); the caller draws `expo` with torch's generator (one exponential_ call of C + 1 values, the reference's
// position in the random stream), so the class SET is the reference's for the same generator state.
__global__ __launch_bounds__(1024) void fed_class_mask_kernel(const int64_t* __restrict__ gt, int R, const float* __restrict__ prob,
                                                              const float* __restrict__ expo, int C, int K, uint8_t* __restrict__ mask) {
    extern __shared__ float q[];                    // C + 1 keys, then C + 1 appearance flags (as ints)
    int* app = reinterpret_cast<int*>(q + C + 1);
    __shared__ int n_app;
    for (int c = threadIdx.x; c <= C; c += blockDim.x) app[c] = 0;
    if (threadIdx.x == 0) n_app = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int64_t g = gt[r];
        if (g >= 0 && g <= C) app[(int)g] = 1;      // benign race: every writer stores 1
    }
    __syncthreads();
    int cnt = 0;
    for (int c = threadIdx.x; c <= C; c += blockDim.x) {
        cnt += app[c];
        q[c] = (app[c] || c == C) ? 0.0f : prob[c] / expo[c];
    }
    atomicAdd(&n_app, cnt);
    __syncthreads();
    const int need = K - n_app;
    for (int c = threadIdx.x; c <= C; c += blockDim.x) {
        int take = 0;
        const float v = q[c];
        if (need > 0 && v > 0.0f) {
            int rank = 0;                           // keys that sort in front of this one (value descending, index ascending)
            for (int o = 0; o <= C; ++o) rank += (q[o] > v || (q[o] == v && o < c)) ? 1 : 0;
            take = rank < need;
        }
        mask[c] = (uint8_t)(app[c] | take);
    }
}
extern "C" int dgx_fed_class_mask(const int64_t* gt_classes, int R, const float* prob, const float* expo, int C, int K, uint8_t* mask,
                                  void* stream) {
    if (!prob || !expo || !mask || C <= 0 || R < 0 || (R > 0 && !gt_classes)) return DGX_ERR_BAD_ARG;
    const size_t sm = (size_t)(C + 1) * 8;
    if (sm > 64 * 1024) return DGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fed_class_mask_kernel, dim3(1), dim3(1024), sm, (hipStream_t)stream, gt_classes, R, prob, expo, C, K, mask);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
