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
    const bool live = i < (int64_t)P.B * Kc;
    const int b = live ? (int)(i / Kc) : -1;
    bool ok = false;
    if (live) {
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
    ok = v > thr;
    out_sc[i] = ok ? sqrtf(fmaxf(v, 0.0f)) : -1.0f;
    }
    // one atomic per wave and image instead of one per candidate: with most candidates above the threshold (early training) the
    // 2 x 4000 same-address atomics of the per-lane form took 0.2 ms -- the whole kernel
    const int b0 = __shfl(b, 0);                   // a wave spans at most two images when Kc >= 64; handle any number anyway
    int bb = b0;
    for (;;) {
        const unsigned long long mine = __ballot(ok && b == bb);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&n_valid[bb], (int)__popcll(mine));
        const unsigned long long rest = __ballot(live && b > bb);
        if (!rest) break;
        bb = __shfl(b, (int)__builtin_ctzll(rest));
    }
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

// rows of 16 bytes (a box) picked by an index list per image: dst[b][k] = src[b][order[b][k]] -- the candidates' boxes brought into
// score order behind torch.sort (torch.gather over the expanded index ran 87 us for 2 x 9 344 boxes: one lane per element)
__global__ __launch_bounds__(256) void cn_gather_boxes_kernel(const float4* __restrict__ src, const int64_t* __restrict__ order, int B, int K,
                                                              float4* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * K) return;
    const int b = (int)(i / K);
    dst[i] = src[(int64_t)b * K + order[i]];
}

extern "C" int dgx_gather_boxes(const float* boxes, const int64_t* order, int B, int K, float* out, void* stream) {
    if (B <= 0 || K <= 0) return DGX_OK;
    if (!boxes || !order || !out) return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(cn_gather_boxes_kernel, dim3((int)(((int64_t)B * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)boxes,
                       order, B, K, (float4*)out);
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

// ------------------------------------------------------------------------------------------------------------------
// dgx_centernet_head_outputs / _bwd: the tail of CenterNetHead.forward + the flattening of CenterNet.forward
// (CN/modeling/dense_heads/centernet_head.py:113-131 `agn_hm`, `F.relu(self.scales[l](bbox_pred(...)))`; centernet.py:179-235
// `_flatten_outputs`) over ALL levels in one launch each way.  Input per level: the grouped predictor output, channels-last
// (rows, C) bf16 with the heat-map logit in channel 0 and the four regression channels behind it; outputs: the loss operands
// reg (M, 4) f32 = relu(float(x[1..4]) * scale_l) and hm (M,) f32 = float(x[0]), levels stacked.  Backward: dx rows (all C channels:
// the unused ones zero) bf16 = [g_hm, g_reg * (reg > 0) * scale_l, 0...] and d scale_l = sum g_reg * (reg > 0) * float(x) -- block
// partials in a fixed order, folded by a second launch (bit-reproducible).  The composed form is ~60 tiny launches per step
// inside the tower graph (per-level slices, scale, ReLU, permuting views, concatenations, casts and their backward).
namespace {
constexpr int HO_MAXL = 8;
struct HoLevels {
    const uint16_t* x[HO_MAXL];
    uint16_t* dx[HO_MAXL];
    const float* scale[HO_MAXL];
    int rows[HO_MAXL], off[HO_MAXL + 1];
    int L, C;
};
}  // namespace

__global__ __launch_bounds__(256) void cn_head_outputs_kernel(HoLevels P, float* __restrict__ reg, float* __restrict__ hm) {
    const int l = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.rows[l]) return;
    const u32x4 v = *reinterpret_cast<const u32x4*>(P.x[l] + (int64_t)r * P.C);          // channels 0 .. 7 of the row
    const float s = P.scale[l][0];
    const float c0 = __uint_as_float(v[0] << 16), c1 = __uint_as_float(v[0] & 0xffff0000u), c2 = __uint_as_float(v[1] << 16),
                c3 = __uint_as_float(v[1] & 0xffff0000u), c4 = __uint_as_float(v[2] << 16);
    const int64_t m = P.off[l] + r;
    hm[m] = c0;
    *reinterpret_cast<float4*>(reg + 4 * m) = make_float4(fmaxf(c1 * s, 0.f), fmaxf(c2 * s, 0.f), fmaxf(c3 * s, 0.f), fmaxf(c4 * s, 0.f));
}

// 8 lanes per row (16 bytes each): lane 0 of a row holds the live channels, the others write zeros
__global__ __launch_bounds__(256) void cn_head_outputs_bwd_kernel(HoLevels P, const float* __restrict__ g_reg, const float* __restrict__ g_hm,
                                                                  float* __restrict__ part, int max_blocks) {
    __shared__ float red[4];
    const int l = blockIdx.y;
    const int cpr = P.C >> 3;                       // 16-byte chunks per row
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = i / cpr;
    const int ch = (int)(i - r * cpr);
    float ds = 0.f;
    if (r < P.rows[l]) {
        u32x4 o = {0u, 0u, 0u, 0u};
        if (ch == 0) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(P.x[l] + r * P.C);
            const float s = P.scale[l][0];
            const float c[4] = {__uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u),
                                __uint_as_float(v[2] << 16)};
            const int64_t m = P.off[l] + r;
            const float4 g4 = *reinterpret_cast<const float4*>(g_reg + 4 * m);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = c[k] * s > 0.f ? g[k] : 0.f;      // ReLU': the forward value relu(c * s) is positive
                d[k] = gk * s;
                ds += gk * c[k];
            }
            o = u32x4{pack_bf2(g_hm[m], d[0]), pack_bf2(d[1], d[2]), pack_bf2(d[3], 0.f), 0u};
        }
        *reinterpret_cast<u32x4*>(P.dx[l] + r * P.C + 8 * ch) = o;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ds;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)l * max_blocks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void cn_head_scale_grad_kernel(HoLevels P, const float* __restrict__ part, int max_blocks, float* __restrict__ d_scale) {
    const int l = blockIdx.x;
    const int nb = (int)(((int64_t)P.rows[l] * (P.C >> 3) + 255) / 256);
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) s += part[(int64_t)l * max_blocks + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) d_scale[l] = s;
}

static int ho_fill(HoLevels& P, const dgx_head_level* levels, int n, int C, bool bwd) {
    if (!levels || n <= 0 || n > HO_MAXL || C < 8 || (C & 7)) return DGX_ERR_BAD_ARG;
    P.L = n; P.C = C;
    int off = 0, maxr = 0;
    for (int l = 0; l < n; ++l) {
        if (!levels[l].x || !levels[l].scale || levels[l].rows <= 0 || (bwd && !levels[l].dx)) return DGX_ERR_BAD_ARG;
        if (((uintptr_t)levels[l].x & 15) || ((uintptr_t)levels[l].dx & 15)) return DGX_ERR_UNSUPPORTED;
        P.x[l] = (const uint16_t*)levels[l].x; P.dx[l] = (uint16_t*)levels[l].dx; P.scale[l] = levels[l].scale;
        P.rows[l] = levels[l].rows; P.off[l] = off;
        off += levels[l].rows;
        maxr = levels[l].rows > maxr ? levels[l].rows : maxr;
    }
    P.off[n] = off;
    return maxr;
}

extern "C" int dgx_centernet_head_outputs(const dgx_head_level* levels, int n, int C, float* reg, float* hm, void* stream) {
    HoLevels P;
    const int maxr = ho_fill(P, levels, n, C, false);
    if (maxr < 0) return maxr;
    if (!reg || !hm || ((uintptr_t)reg & 15)) return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(cn_head_outputs_kernel, dim3((maxr + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, P, reg, hm);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int64_t dgx_centernet_head_outputs_bwd_workspace_floats(const dgx_head_level* levels, int n, int C) {
    if (!levels || n <= 0 || C < 8) return 0;
    int maxr = 0;
    for (int l = 0; l < n; ++l) maxr = levels[l].rows > maxr ? levels[l].rows : maxr;
    return (int64_t)n * (((int64_t)maxr * (C >> 3) + 255) / 256);
}

extern "C" int dgx_centernet_head_outputs_bwd(const dgx_head_level* levels, int n, int C, const float* g_reg, const float* g_hm,
                                              float* d_scale, float* workspace, void* stream) {
    HoLevels P;
    const int maxr = ho_fill(P, levels, n, C, true);
    if (maxr < 0) return maxr;
    if (!g_reg || !g_hm || !d_scale || !workspace || ((uintptr_t)g_reg & 15)) return DGX_ERR_BAD_ARG;
    const int max_blocks = (int)(((int64_t)maxr * (C >> 3) + 255) / 256);
    hipLaunchKernelGGL(cn_head_outputs_bwd_kernel, dim3(max_blocks, n), dim3(256), 0, (hipStream_t)stream, P, g_reg, g_hm, workspace, max_blocks);
    hipLaunchKernelGGL(cn_head_scale_grad_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, P, workspace, max_blocks, d_scale);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
