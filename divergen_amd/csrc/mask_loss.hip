// Mask loss: binary cross-entropy with logits (mean) over the (R, S, S) foreground mask logits, its gradient and the three
// training statistics of mask_rcnn_loss (D2/modeling/roi_heads/mask_head.py:35-110) in one pass over the logits (gfx950).
// HBM-bound and tiny (R*S*S ~ 2e5 elements); the point is one launch pair instead of ~12 elementwise / reduction launches and a
// deterministic two-stage sum.
#include "dgx_common.h"

namespace {
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }

constexpr int NV = 5;   // loss sum, incorrect, false positive, false negative, positives

// logits may be a strided (R, S*S) view: element i lives at (i / inner) * row_stride + i % inner
template <typename T>
__global__ __launch_bounds__(256) void mask_bce_partial_kernel(const T* __restrict__ logits, int64_t row_stride, int64_t inner,
                                                               const uint8_t* __restrict__ gt, int64_t n, float inv_n,
                                                               T* __restrict__ grad, float* __restrict__ part) {
    float acc[NV] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / inner;
        const float x = ld(logits + r * row_stride + (i - r * inner));
        const bool t = gt[i] != 0;
        const float tf = t ? 1.0f : 0.0f;
        // max(x, 0) - x t + log(1 + exp(-|x|))
        acc[0] += fmaxf(x, 0.0f) - x * tf + log1pf(expf(-fabsf(x)));
        const bool wrong = (x > 0.0f) != t;
        acc[1] += wrong ? 1.0f : 0.0f;
        acc[2] += (wrong && !t) ? 1.0f : 0.0f;
        acc[3] += (wrong && t) ? 1.0f : 0.0f;
        acc[4] += tf;
        if (grad) st(grad + i, (1.0f / (1.0f + expf(-x)) - tf) * inv_n);
    }
    __shared__ float red[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NV) part[(int64_t)blockIdx.x * NV + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void mask_bce_final_kernel(const float* __restrict__ part, int blocks, float inv_n, float* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= NV) return;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += part[(int64_t)b * NV + k];   // fixed order: reproducible
    out[k] = k == 0 ? s * inv_n : s;
}
}  // namespace

extern "C" int64_t dgx_mask_bce_workspace_floats(int64_t n) {
    const int64_t blocks = (n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024;
    return (blocks < 1 ? 1 : blocks) * NV;
}

extern "C" int dgx_mask_bce(const void* logits, int64_t row_stride, int64_t inner, const uint8_t* gt, int64_t n, void* grad,
                            float* out, float* workspace, int dtype, void* stream) {
    if (!out) return DGX_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) { (void)hipMemsetAsync(out, 0, NV * sizeof(float), s); return DGX_OK; }
    if (!logits || !gt || !workspace || inner <= 0 || row_stride < inner) return DGX_ERR_BAD_ARG;
    if (dtype != DGX_F32 && dtype != DGX_BF16) return DGX_ERR_BAD_ARG;
    const int blocks = (int)(dgx_mask_bce_workspace_floats(n) / NV);
    const float inv_n = 1.0f / (float)n;
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(mask_bce_partial_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)logits, row_stride, inner, gt, n,
                           inv_n, (uint16_t*)grad, workspace);
    else
        hipLaunchKernelGGL(mask_bce_partial_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)logits, row_stride, inner, gt, n, inv_n,
                           (float*)grad, workspace);
    hipLaunchKernelGGL(mask_bce_final_kernel, dim3(1), dim3(64), 0, s, workspace, blocks, inv_n, out);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
