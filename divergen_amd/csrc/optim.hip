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
                                                        float gscale, const float* __restrict__ scale_dev, float bc1, float bc2s,
                                                        float decay, const float* __restrict__ lr_scale,
                                                        const int64_t* __restrict__ seg_end, int n_seg,
                                                        const int32_t* __restrict__ found_inf) {
    if (found_inf && *found_inf) return;
    if (scale_dev) gscale *= *scale_dev;       // full-model norm-clip coefficient (dgx_clip_coef_f32), kept on the device
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    constexpr int AU = 2;      // quads per trip: all their streams are requested before the first is consumed (10 x 16 B in flight per lane; 4 quads: no further gain)
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += AU * stride) {
      float4 Pq[AU], Gq[AU], Mq[AU], Vq[AU], Eq[AU];
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const int64_t i = i0 + u * stride < n4 ? i0 + u * stride : i0;
        Pq[u] = p[i]; Gq[u] = g[i]; Mq[u] = m[i]; Vq[u] = v[i];
        if (ema) Eq[u] = ema[i];
      }
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const int64_t i = i0 + u * stride;
        if (i >= n4) break;
        float lre = lr;
        if (lr_scale) {  // binary search of the segment of element 4*i
            int lo = 0, hi = n_seg - 1;
            const int64_t e = 4 * i;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > e) hi = mid; else lo = mid + 1; }
            lre = lr * lr_scale[lo];
        }
        float4 P = Pq[u], G = Gq[u], M = Mq[u], V = Vq[u];
        float pe[4] = {P.x, P.y, P.z, P.w}, ge[4] = {G.x, G.y, G.z, G.w}, me[4] = {M.x, M.y, M.z, M.w},
              ve[4] = {V.x, V.y, V.z, V.w};
        if (ema) {
            float4 E = Eq[u];
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
}

extern "C" int dgx_adamw_ema_step_scaled(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16, int64_t n,
                                         float lr, float beta1, float beta2, float eps, float weight_decay, float clip_value,
                                         float grad_scale, const float* grad_scale_dev, int step, float ema_decay,
                                         const float* lr_scale, const int64_t* seg_end, int n_seg, const int32_t* found_inf,
                                         void* stream) {
    if (n <= 0) return DGX_OK;
    if (!p || !g || !m || !v || (n & 3) || step < 1 || ((lr_scale != nullptr) != (seg_end != nullptr))) return DGX_ERR_BAD_ARG;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, (float4*)ema, (uint2*)p_bf16, n4, lr, beta1, beta2, eps, weight_decay,
                       clip_value > 0 ? clip_value : INFINITY, grad_scale, grad_scale_dev, bc1, bc2s, ema_decay, lr_scale, seg_end,
                       n_seg, found_inf);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, float clip_value,
                                  float grad_scale, int step, float ema_decay, const float* lr_scale,
                                  const int64_t* seg_end, int n_seg, const int32_t* found_inf, void* stream) {
    return dgx_adamw_ema_step_scaled(p, g, m, v, ema, p_bf16, n, lr, beta1, beta2, eps, weight_decay, clip_value, grad_scale, nullptr,
                                     step, ema_decay, lr_scale, seg_end, n_seg, found_inf, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// SGD with momentum (the 'SGD' branch of build_custom_optimizer, custom_solver.py:64-68 = torch.optim.SGD): per element
//   d = clip(g * grad_scale [* *scale_dev]) + wd * p;  buf = first ? d : momentum * buf + d;
//   d = nesterov ? d + momentum * buf : buf;  p -= lr * d
// with the EMA lerp of the PRE-step weights and the bf16 shadow as in the AdamW kernel.  scale_dev (optional, device scalar) carries
// the full-model norm-clip coefficient of dgx_clip_coef_f32 without a host read.
__global__ __launch_bounds__(256) void sgd_ema_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ buf,
                                                      float4* __restrict__ ema, uint2* __restrict__ pbf, int64_t n4, float lr,
                                                      float mom, int nesterov, float wd, float clip, float gscale,
                                                      const float* __restrict__ scale_dev, int first, float decay,
                                                      const float* __restrict__ lr_scale, const int64_t* __restrict__ seg_end,
                                                      int n_seg, const int32_t* __restrict__ found_inf) {
    if (found_inf && *found_inf) return;
    if (scale_dev) gscale *= *scale_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float lre = lr;
        if (lr_scale) {
            int lo = 0, hi = n_seg - 1;
            const int64_t e = 4 * i;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > e) hi = mid; else lo = mid + 1; }
            lre = lr * lr_scale[lo];
        }
        const float4 P = p[i], G = g[i];
        float pe[4] = {P.x, P.y, P.z, P.w}, ge[4] = {G.x, G.y, G.z, G.w}, be[4] = {0.f, 0.f, 0.f, 0.f};
        if (mom != 0.f && !first) { const float4 B = buf[i]; be[0] = B.x; be[1] = B.y; be[2] = B.z; be[3] = B.w; }
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
            float d = ge[k] * gscale;
            d = fminf(fmaxf(d, -clip), clip);
            d = d + wd * pe[k];
            if (mom != 0.f) {
                be[k] = first ? d : mom * be[k] + d;
                d = nesterov ? d + mom * be[k] : be[k];
            }
            pe[k] = pe[k] - lre * d;
        }
        p[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        if (mom != 0.f) buf[i] = make_float4(be[0], be[1], be[2], be[3]);
        if (pbf) pbf[i] = make_uint2(pack_bf2(pe[0], pe[1]), pack_bf2(pe[2], pe[3]));
    }
}

extern "C" int dgx_sgd_ema_step(float* p, const float* g, float* buf, float* ema, void* p_bf16, int64_t n, float lr, float momentum,
                                int nesterov, float weight_decay, float clip_value, float grad_scale, const float* grad_scale_dev,
                                int step, float ema_decay, const float* lr_scale, const int64_t* seg_end, int n_seg,
                                const int32_t* found_inf, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!p || !g || (momentum != 0.f && !buf) || (n & 3) || step < 1 || ((lr_scale != nullptr) != (seg_end != nullptr)) ||
        (nesterov && momentum <= 0.f))
        return DGX_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(sgd_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g, (float4*)buf,
                       (float4*)ema, (uint2*)p_bf16, n4, lr, momentum, nesterov, weight_decay, clip_value > 0 ? clip_value : INFINITY,
                       grad_scale, grad_scale_dev, step == 1 ? 1 : 0, ema_decay, lr_scale, seg_end, n_seg, found_inf);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// Full-model gradient-norm clipping (FullModelGradientClippingOptimizer, custom_solver.py:46-60 = torch.nn.utils.clip_grad_norm_ over
// every parameter): coef = min(1, max_norm / (||g * grad_scale||_2 + 1e-6)) as a DEVICE scalar (no host read): 1024 block partial sums
// in a fixed order, folded by the last stage in double precision -- the same result on every call.
constexpr int CLIP_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float4* __restrict__ g, int64_t n4, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = g[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void clip_coef_kernel(const float* __restrict__ part, int nb, float gscale, float max_norm,
                                                       float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += (double)part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s) * fabsf(gscale);
        out[0] = fminf(1.0f, max_norm / (norm + 1e-6f));
        out[1] = norm;
    }
}

extern "C" int64_t dgx_clip_coef_workspace_floats(void) { return CLIP_BLOCKS; }
extern "C" int dgx_clip_coef_f32(const float* g, int64_t n, float grad_scale, float max_norm, float* workspace, float* out2,
                                 void* stream) {
    if (!g || !workspace || !out2 || n <= 0 || (n & 3) || !(max_norm > 0.f)) return DGX_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < CLIP_BLOCKS ? (n4 + 255) / 256 : CLIP_BLOCKS);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)g, n4, workspace);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, grid, grad_scale, max_norm, out2);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// Zero a list of ranges of a flat f32 arena in one launch: ranges i64 (n, 2) DEVICE = (first element, count), each a multiple of 4
// and at most 65 536 elements (one workgroup per range).  The gradient arena minus the segments that their first writer overwrites
// (solver.FlatArena.zero_grad(lazy=True)).
__global__ __launch_bounds__(256) void zero_ranges_kernel(float4* __restrict__ g, const int64_t* __restrict__ ranges) {
    const int64_t p4 = ranges[2 * (int64_t)blockIdx.x] >> 2, n4 = ranges[2 * (int64_t)blockIdx.x + 1] >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 256) g[p4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
extern "C" int dgx_zero_ranges_f32(float* g, const int64_t* ranges, int64_t n, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!g || !ranges || ((uintptr_t)g & 15) || n >= (1ll << 31)) return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (float4*)g, ranges);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
