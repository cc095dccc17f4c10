// Residual + stochastic-depth epilogues of a Swin block (gfx950), HBM-bound, 16 B per lane.
//   forward : out[t] = x[t] + scale[b] * y[pos(t)]      y bf16 in token order (ws == 0) or in WINDOW
//             order (ws > 0: window_reverse + roll(+shift) + crop folded into the read address)
//   backward: dy[pos(t)] = bf16(scale[b] * g[t])         (padding rows of the window layout = 0)
// scale[b] is the per-sample DropPath factor floor(keep + U)/keep (timm drop_path; ones when off).
// Replaces window_reverse/roll/crop + DropPath (rand, floor, div, mul) + add at
// swintransformer.py:239-255: 1 launch instead of ~7 each way.
#include "dgx_common.h"

struct RMap { int B, H, W, ws, shift, nWh, nWw; };

__device__ __forceinline__ int64_t r_win_pos(const RMap& m, int b, int hh0, int ww0) {
    const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
    int hs = hh0 - m.shift, wsx = ww0 - m.shift;
    if (hs < 0) hs += Hp;
    if (wsx < 0) wsx += Wp;
    const int wr = hs / m.ws, wc = wsx / m.ws;
    const int n = (hs - wr * m.ws) * m.ws + (wsx - wc * m.ws);
    return (((int64_t)b * m.nWh + wr) * m.nWw + wc) * (m.ws * m.ws) + n;
}

template <typename XT> struct Vec;   // 8 elements of the residual stream per lane
template <> struct Vec<float> {
    static __device__ void ld(const float* p, float* v) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ void st(float* p, const float* v) {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec<uint16_t> {
    static __device__ void ld(const uint16_t* p, float* v) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
    static __device__ void st(uint16_t* p, const float* v) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
};

template <typename XT>
__global__ __launch_bounds__(256) void residual_fwd_kernel(const XT* __restrict__ x, const uint16_t* __restrict__ y,
                                                           const float* __restrict__ scale, XT* __restrict__ out, int vecC,
                                                           int64_t total, RMap m) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t t = i / vecC;
        const int64_t tok = t;
        const int ww0 = (int)(t % m.W);
        t /= m.W;
        const int hh0 = (int)(t % m.H);
        const int b = (int)(t / m.H);
        const int64_t src = m.ws ? r_win_pos(m, b, hh0, ww0) : tok;
        float xv[8], yv[8];
        Vec<XT>::ld(x + (tok * vecC + v) * 8, xv);
        Vec<uint16_t>::ld(y + (src * vecC + v) * 8, yv);
        const float s = scale ? scale[b] : 1.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] += s * yv[k];
        Vec<XT>::st(out + (tok * vecC + v) * 8, xv);
    }
}

// one lane per 8 channels of an OUTPUT row (window order incl. padding when ws > 0)
template <typename XT>
__global__ __launch_bounds__(256) void residual_bwd_kernel(const XT* __restrict__ g, const float* __restrict__ scale,
                                                           uint16_t* __restrict__ dy, int vecC, int64_t total_out, RMap m) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_out; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t orow = i / vecC;
        int64_t tok;
        int b;
        if (m.ws) {
            const int N = m.ws * m.ws;
            const int n = (int)(orow % N);
            int64_t t = orow / N;
            const int wc = (int)(t % m.nWw);
            t /= m.nWw;
            const int wr = (int)(t % m.nWh);
            b = (int)(t / m.nWh);
            int hh = wr * m.ws + n / m.ws + m.shift, ww = wc * m.ws + n % m.ws + m.shift;
            const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
            if (hh >= Hp) hh -= Hp;
            if (ww >= Wp) ww -= Wp;
            tok = (hh < m.H && ww < m.W) ? ((int64_t)b * m.H + hh) * m.W + ww : -1;
        } else {
            tok = orow;
            b = (int)(orow / ((int64_t)m.H * m.W));
        }
        float gv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tok >= 0) {
            Vec<XT>::ld(g + (tok * vecC + v) * 8, gv);
            const float s = scale ? scale[b] : 1.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) gv[k] *= s;
        }
        Vec<uint16_t>::st(dy + i * 8, gv);
    }
}

static RMap rmap(int B, int H, int W, int ws, int shift) {
    RMap m = {B, H, W, ws, shift, 0, 0};
    if (ws > 0) { m.nWh = (H + ws - 1) / ws; m.nWw = (W + ws - 1) / ws; }
    return m;
}

extern "C" int dgx_residual_fwd(const void* x, const void* y_bf16, const float* scale, void* out, int B, int H, int W, int C,
                                int ws, int shift, int x_dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!x || !y_bf16 || !out || (C & 7) || shift < 0 || (ws > 0 && shift >= ws)) return DGX_ERR_BAD_ARG;
    const RMap m = rmap(B, H, W, ws, shift);
    const int vecC = C / 8;
    const int64_t total = (int64_t)B * H * W * vecC;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == DGX_BF16)
        hipLaunchKernelGGL(residual_fwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)y_bf16,
                           scale, (uint16_t*)out, vecC, total, m);
    else
        hipLaunchKernelGGL(residual_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (const uint16_t*)y_bf16, scale,
                           (float*)out, vecC, total, m);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_residual_bwd(const void* g, const float* scale, void* dy_bf16, int B, int H, int W, int C, int ws, int shift,
                                int g_dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!g || !dy_bf16 || (C & 7) || shift < 0 || (ws > 0 && shift >= ws)) return DGX_ERR_BAD_ARG;
    const RMap m = rmap(B, H, W, ws, shift);
    const int vecC = C / 8;
    const int64_t rows = ws > 0 ? (int64_t)B * m.nWh * m.nWw * ws * ws : (int64_t)B * H * W;
    const int64_t total = rows * vecC;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t st = (hipStream_t)stream;
    if (g_dtype == DGX_BF16)
        hipLaunchKernelGGL(residual_bwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)g, scale, (uint16_t*)dy_bf16,
                           vecC, total, m);
    else
        hipLaunchKernelGGL(residual_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)g, scale, (uint16_t*)dy_bf16, vecC,
                           total, m);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// FPN top-down step (D2/modeling/backbone/fpn.py:139-145: `F.interpolate(prev, scale_factor=2, mode="nearest")`, then
// `lateral + top_down`): NHWC maps, 8 channels per lane.
//   forward : out[n][y][x][c] = lat[n][y][x][c] + top[n][y >> 1][x >> 1][c]           (one rounding of the fp32 sum)
//   backward: d lat = g (the caller aliases it);  d top[n][y][x][c] = g[n][2y][2x][c] + g[n][2y][2x+1][c] + g[n][2y+1][2x][c] + g[n][2y+1][2x+1][c]
//             (fp32 sum of four values, one rounding -- what autograd's upsample_nearest2d_backward computes)
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_add_fwd_kernel(const T* __restrict__ lat, const T* __restrict__ top, T* __restrict__ out,
                                                                 int N, int H, int W, int C) {
    const int CV = C >> 3;
    const int64_t total = (int64_t)N * H * W * CV;
    const int Ht = H >> 1, Wt = W >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int n = (int)(r / H);
        float a[8], b[8];
        Vec<T>::ld(lat + i * 8, a);
        Vec<T>::ld(top + ((((int64_t)n * Ht + (y >> 1)) * Wt + (x >> 1)) * CV + cv) * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        Vec<T>::st(out + i * 8, a);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_add_bwd_kernel(const T* __restrict__ g, T* __restrict__ gtop, int N, int H, int W, int C) {
    const int CV = C >> 3;
    const int Ht = H >> 1, Wt = W >> 1;
    const int64_t total = (int64_t)N * Ht * Wt * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int x = (int)(r % Wt); r /= Wt;
        const int y = (int)(r % Ht);
        const int n = (int)(r / Ht);
        const T* p = g + ((((int64_t)n * H + 2 * y) * W + 2 * x) * CV + cv) * 8;
        float a[8], b[8];
        Vec<T>::ld(p, a);
        Vec<T>::ld(p + (int64_t)CV * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        Vec<T>::ld(p + (int64_t)W * CV * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        Vec<T>::ld(p + ((int64_t)W + 1) * CV * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        Vec<T>::st(gtop + i * 8, a);
    }
}

extern "C" int dgx_upsample2x_add_fwd(const void* lat, const void* top, void* out, int N, int H, int W, int C, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!lat || !top || !out) return DGX_ERR_BAD_ARG;
    if ((C & 7) || (H & 1) || (W & 1) || (((uintptr_t)lat | (uintptr_t)top | (uintptr_t)out) & 15)) return DGX_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)N * H * W * (C >> 3);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(upsample2x_add_fwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)lat, (const uint16_t*)top, (uint16_t*)out, N, H, W, C);
    else if (dtype == DGX_F32)
        hipLaunchKernelGGL(upsample2x_add_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)lat, (const float*)top, (float*)out, N, H, W, C);
    else return DGX_ERR_BAD_ARG;
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_upsample2x_add_bwd(const void* g, void* gtop, int N, int H, int W, int C, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!g || !gtop) return DGX_ERR_BAD_ARG;
    if ((C & 7) || (H & 1) || (W & 1) || (((uintptr_t)g | (uintptr_t)gtop) & 15)) return DGX_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)N * (H >> 1) * (W >> 1) * (C >> 3);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(upsample2x_add_bwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)g, (uint16_t*)gtop, N, H, W, C);
    else if (dtype == DGX_F32)
        hipLaunchKernelGGL(upsample2x_add_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)g, (float*)gtop, N, H, W, C);
    else return DGX_ERR_BAD_ARG;
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
