// Fused per-element gradient value clip + AdamW + EMA over a flat fp32 arena (one HBM sweep:
// read g, read/write p, m, v, ema = 36 B/param, optional +2 B bf16 shadow).  Replaces ~400 per-tensor
// param groups + a python loop of ~400 EMA lerps (custom_solver.py:19-77, D2/solver/build.py:24-75,
// ema.py:49-58).  EMA uses the PRE-step weights, as the reference calls model_ema.update(model)
// before optimizer.step() (train_net.py:262-284).
#include "dgx_common.h"

__global__ __launch_bounds__(256) void adamw_ema_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                        float4* __restrict__ m, float4* __restrict__ v,
                                                        float4* __restrict__ ema, uint2* __restrict__ pbf, int64_t n4,
                                                        float lr, float b1, float b2, float eps, float wd, float clip,
                                                        float gscale, float bc1, float bc2s, float decay,
                                                        const float* __restrict__ lr_scale,
                                                        const int64_t* __restrict__ seg_end, int n_seg,
                                                        const int32_t* __restrict__ found_inf) {
    if (found_inf && *found_inf) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float lre = lr;
        if (lr_scale) {  // binary search of the segment of element 4*i
            int lo = 0, hi = n_seg - 1;
            const int64_t e = 4 * i;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > e) hi = mid; else lo = mid + 1; }
            lre = lr * lr_scale[lo];
        }
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        float pe[4] = {P.x, P.y, P.z, P.w}, ge[4] = {G.x, G.y, G.z, G.w}, me[4] = {M.x, M.y, M.z, M.w},
              ve[4] = {V.x, V.y, V.z, V.w};
        if (ema) {
            float4 E = ema[i];
            E.x = E.x * decay + (1.0f - decay) * pe[0];
            E.y = E.y * decay + (1.0f - decay) * pe[1];
            E.z = E.z * decay + (1.0f - decay) * pe[2];
            E.w = E.w * decay + (1.0f - decay) * pe[3];
            ema[i] = E;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gr = ge[k] * gscale;
            gr = fminf(fmaxf(gr, -clip), clip);
            pe[k] *= 1.0f - lre * wd;
            me[k] = me[k] + (gr - me[k]) * (1.0f - b1);
            ve[k] = ve[k] * b2 + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(ve[k]) / bc2s + eps;
            pe[k] -= (lre / bc1) * (me[k] / denom);
        }
        p[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        m[i] = make_float4(me[0], me[1], me[2], me[3]);
        v[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
        if (pbf) pbf[i] = make_uint2(pack_bf2(pe[0], pe[1]), pack_bf2(pe[2], pe[3]));
    }
}

extern "C" int dgx_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, float clip_value,
                                  float grad_scale, int step, float ema_decay, const float* lr_scale,
                                  const int64_t* seg_end, int n_seg, const int32_t* found_inf, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!p || !g || !m || !v || (n & 3) || step < 1 || ((lr_scale != nullptr) != (seg_end != nullptr))) return DGX_ERR_BAD_ARG;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, (float4*)ema, (uint2*)p_bf16, n4, lr, beta1, beta2, eps, weight_decay,
                       clip_value > 0 ? clip_value : INFINITY, grad_scale, bc1, bc2s, ema_decay, lr_scale, seg_end,
                       n_seg, found_inf);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
