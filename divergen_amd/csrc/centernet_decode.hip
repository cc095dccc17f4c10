// CenterNet proposal decoding, the element-wise parts around torch's top-k / sort and the NMS kernel
// (CN/modeling/dense_heads/centernet.py:627-737 `predict_instances` / `predict_single_level` / `nms_and_topK`):
//   dgx_centernet_scores    all FPN levels' agnostic heat-map logits -> one (B, M) score row per image: sigmoid, and -1 where the
//                           score does not pass INFERENCE_TH (centernet.py:657-660); also clears the per-image counters;
//   dgx_centernet_decode    candidates (B, Kc) location indices -> boxes from the grid centre and the regression maps
//                           (:672-681: reg * stride, x0/y0/x1/y1 with the 0.01 minimum extent), scores sqrt'ed (:704), the
//                           number of candidates above the threshold per image;
//   dgx_centernet_finalize  the survivors of the NMS (indices into the score-sorted candidates) -> fixed-length box / score /
//                           validity rows.
// The composed form is ~75 torch launches per step (per-level casts, sigmoids, permuting copies, concatenations, gathers, the
// decode arithmetic, masks); the float sequences here are the same expressions in the same order.
#include "dgx_common.h"

namespace {
constexpr int CD_MAXL = 8;
struct CdLevels {
    const void* hm[CD_MAXL];       // logits of level l: (B, h, w, hm_ps) channels-last, channel hm_co
    const void* reg[CD_MAXL];      // regression maps: (B, h, w, reg_ps), channels reg_co .. reg_co + 3
    int h[CD_MAXL], w[CD_MAXL], stride[CD_MAXL], off[CD_MAXL + 1];
    int hm_ps, hm_co, reg_ps, reg_co;
    int L, B, M;
};
template <typename T> __device__ __forceinline__ float cd_ld(const void* p, int64_t i);
template <> __device__ __forceinline__ float cd_ld<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float cd_ld<uint16_t>(const void* p, int64_t i) { return bf2f(((const uint16_t*)p)[i]); }
__device__ __forceinline__ int cd_level(const CdLevels& P, int m) {
    int l = 0;
    while (l + 1 < P.L && m >= P.off[l + 1]) ++l;
    return l;
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void cn_scores_kernel(CdLevels P, float thr, float* __restrict__ scores, int32_t* __restrict__ n_valid) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < P.B) n_valid[i] = 0;
    if (i >= (int64_t)P.B * P.M) return;
    const int b = (int)(i / P.M), m = (int)(i - (int64_t)b * P.M);
    const int l = cd_level(P, m), p = m - P.off[l];
    const float x = cd_ld<T>(P.hm[l], ((int64_t)b * P.h[l] * P.w[l] + p) * P.hm_ps + P.hm_co);
    const float s = 1.0f / (1.0f + expf(-x));
    scores[i] = s > thr ? s : -1.0f;
}

template <typename T>
__global__ __launch_bounds__(256) void cn_decode_kernel(CdLevels P, const int64_t* __restrict__ idx, int Kc, const float* __restrict__ scores,
                                                        float thr, float* __restrict__ boxes, float* __restrict__ out_sc,
                                                        int32_t* __restrict__ n_valid) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)P.B * Kc) return;
    const int b = (int)(i / Kc);
    const int m = (int)idx[i];
    const int l = cd_level(P, m), p = m - P.off[l];
    const int y = p / P.w[l], x = p - y * P.w[l];
    const float st = (float)P.stride[l];
    const float gx = (float)(x * P.stride[l]) + (float)(P.stride[l] / 2), gy = (float)(y * P.stride[l]) + (float)(P.stride[l] / 2);
    const int64_t ro = ((int64_t)b * P.h[l] * P.w[l] + p) * P.reg_ps + P.reg_co;
    const float r0 = cd_ld<T>(P.reg[l], ro) * st, r1 = cd_ld<T>(P.reg[l], ro + 1) * st;
    const float r2 = cd_ld<T>(P.reg[l], ro + 2) * st, r3 = cd_ld<T>(P.reg[l], ro + 3) * st;
    const float x0 = gx - r0, y0 = gy - r1;
    boxes[4 * i] = x0;
    boxes[4 * i + 1] = y0;
    boxes[4 * i + 2] = fmaxf(gx + r2, x0 + 0.01f);
    boxes[4 * i + 3] = fmaxf(gy + r3, y0 + 0.01f);
    const float v = scores[(int64_t)b * P.M + m];
    const bool ok = v > thr;
    out_sc[i] = ok ? sqrtf(fmaxf(v, 0.0f)) : -1.0f;
    if (ok) atomicAdd(&n_valid[b], 1);
}

__global__ __launch_bounds__(256) void cn_finalize_kernel(const float* __restrict__ boxes, const float* __restrict__ sc, const int32_t* __restrict__ keep_idx,
                                                          const int32_t* __restrict__ num_keep, int B, int K, int cap, float* __restrict__ o_box,
                                                          float* __restrict__ o_sc, uint8_t* __restrict__ o_valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * cap) return;
    const int b = i / cap, j = i - b * cap;
    const bool v = j < num_keep[b];
    int k = keep_idx[i];
    k = k < 0 ? 0 : k;
    const int64_t s = (int64_t)b * K + k;
#pragma unroll
    for (int c = 0; c < 4; ++c) o_box[4 * (int64_t)i + c] = v ? boxes[4 * s + c] : 0.0f;
    o_sc[i] = v ? sc[s] : 0.0f;
    o_valid[i] = v ? 1 : 0;
}

static int cd_fill(CdLevels& P, const void* const* hm, const void* const* reg, const int32_t* level_hw, const int32_t* strides, int L, int B,
                   int hm_ps, int hm_co, int reg_ps, int reg_co) {
    if (L < 1 || L > CD_MAXL || B <= 0 || !level_hw || !strides) return DGX_ERR_BAD_ARG;
    P.L = L; P.B = B;
    int m = 0;
    for (int l = 0; l < L; ++l) {
        P.hm[l] = hm ? hm[l] : nullptr;
        P.reg[l] = reg ? reg[l] : nullptr;
        P.h[l] = level_hw[2 * l]; P.w[l] = level_hw[2 * l + 1]; P.stride[l] = strides[l];
        P.off[l] = m;
        m += P.h[l] * P.w[l];
    }
    P.off[L] = m;
    P.M = m;
    P.hm_ps = hm_ps; P.hm_co = hm_co; P.reg_ps = reg_ps; P.reg_co = reg_co;
    return DGX_OK;
}

extern "C" int dgx_centernet_scores(const void* const* hm_levels, int hm_pixel_stride, int hm_channel, const int32_t* level_hw,
                                    const int32_t* strides, int L, int B, float thr, float* scores, int32_t* n_valid, int dtype,
                                    void* stream) {
    CdLevels P = {};
    if (!hm_levels || !scores || !n_valid) return DGX_ERR_BAD_ARG;
    const int rc = cd_fill(P, hm_levels, nullptr, level_hw, strides, L, B, hm_pixel_stride, hm_channel, 0, 0);
    if (rc != DGX_OK) return rc;
    const int64_t n = (int64_t)B * P.M;
    if (n <= 0) return DGX_OK;
    const int grid = (int)((n + 255) / 256);
    if (dtype == DGX_BF16) hipLaunchKernelGGL(cn_scores_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, thr, scores, n_valid);
    else hipLaunchKernelGGL(cn_scores_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, thr, scores, n_valid);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_centernet_decode(const void* const* reg_levels, int reg_pixel_stride, int reg_channel, const int32_t* level_hw,
                                    const int32_t* strides, int L, int B, const int64_t* cand_idx, int Kc, const float* scores, float thr,
                                    float* boxes, float* out_scores, int32_t* n_valid, int dtype, void* stream) {
    CdLevels P = {};
    if (!reg_levels || !cand_idx || !scores || !boxes || !out_scores || !n_valid || Kc <= 0) return DGX_ERR_BAD_ARG;
    const int rc = cd_fill(P, nullptr, reg_levels, level_hw, strides, L, B, 0, 0, reg_pixel_stride, reg_channel);
    if (rc != DGX_OK) return rc;
    const int64_t n = (int64_t)B * Kc;
    const int grid = (int)((n + 255) / 256);
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(cn_decode_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, cand_idx, Kc, scores, thr, boxes, out_scores, n_valid);
    else
        hipLaunchKernelGGL(cn_decode_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, cand_idx, Kc, scores, thr, boxes, out_scores, n_valid);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_centernet_finalize(const float* sorted_boxes, const float* sorted_scores, const int32_t* keep_idx, const int32_t* num_keep,
                                      int B, int K, int cap, float* out_boxes, float* out_scores, uint8_t* out_valid, void* stream) {
    if (B <= 0 || cap <= 0) return DGX_OK;
    if (!sorted_boxes || !sorted_scores || !keep_idx || !num_keep || !out_boxes || !out_scores || !out_valid || K <= 0) return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(cn_finalize_kernel, dim3((B * cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, sorted_boxes, sorted_scores, keep_idx,
                       num_keep, B, K, cap, out_boxes, out_scores, out_valid);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
