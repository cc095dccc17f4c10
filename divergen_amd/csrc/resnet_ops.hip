// ResNet-50 glue of the R50 configurations (BASELINE configs[0], DG/configs/Base-C2_L_R5021k_640b64_4x.yaml,
// DG/divergen/modeling/backbone/timm.py:27-151 over timm 0.4.9's ResNet / Bottleneck): what sits between the GEMMs.
//   * stem 7x7 / stride 2 / pad 3 convolution: the image unfolded into GEMM rows (K = 3*49 = 147, zero-padded to 152 so that
//     rows are 16-byte multiples), contraction on dgx_gemm_bf16_nt;
//   * FrozenBatchNorm2d (+ residual add) (+ ReLU) after every convolution as ONE pass over the channels-last activation
//     (per-channel scale / shift precomputed on the host side from weight, bias, running_mean, running_var), and its backward;
//   * max-pool 3x3 / stride 2 / pad 1 with the arg-max tap kept as a byte, backward as a gather (one writer per input pixel).
// All HBM-bound, 16 bytes per lane along the channel dimension.
#include "dgx_common.h"

namespace {
__device__ __forceinline__ void rn_unpack8(const uint4 r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 rn_pack8(const float (&v)[8]) {
    return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
}  // namespace

// rows[(n, oy, ox)][c*49 + ky*7 + kx] = x[n][c][2 oy - 3 + ky][2 ox - 3 + kx] (0 outside), columns 147 .. 151 zero
__global__ __launch_bounds__(256) void stem_im2col7x7_kernel(const float* __restrict__ x, uint4* __restrict__ rows, int N, int H, int W,
                                                             int Ho, int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * 19;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 19);
        int64_t t = i / 19;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = 8 * ch + k;
            float val = 0.f;
            if (col < 147) {
                const int c = col / 49, r = col - c * 49, ky = r / 7, kx = r - ky * 7;
                const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = x[(((int64_t)n * 3 + c) * H + iy) * W + ix];
            }
            v[k] = val;
        }
        rows[i] = rn_pack8(v);
    }
}

extern "C" int dgx_stem_im2col7x7(const float* x_nchw, void* rows_bf16, int N, int H, int W, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!x_nchw || !rows_bf16) return DGX_ERR_BAD_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * 19;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(stem_im2col7x7_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw, (uint4*)rows_bf16, N, H, W, Ho, Wo);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// y = act(x * scale[c] + shift[c] (+ res)), rows x C bf16, C % 8 == 0
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const uint4* __restrict__ res,
                                                             uint4* __restrict__ y, int64_t total, int vecC, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % vecC) * 8;
        float v[8], r[8];
        rn_unpack8(x[i], v);
        if (res) rn_unpack8(res[i], r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float o = v[k] * scale[c0 + k] + shift[c0 + k];
            if (res) o += r[k];
            v[k] = (relu && !(o > 0.f)) ? 0.f : o;
        }
        y[i] = rn_pack8(v);
    }
}
// g = dy * [y > 0] (ReLU) ; dx = g * scale[c] ; dres = g (when asked for; may be the same buffer as dy)
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y,
                                                             const float* __restrict__ scale, uint4* __restrict__ dx, uint4* dres,
                                                             int64_t total, int vecC, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % vecC) * 8;
        float g[8], o[8], d[8];
        rn_unpack8(dy[i], g);
        if (relu) {
            rn_unpack8(y[i], o);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = o[k] > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = g[k] * scale[c0 + k];
        dx[i] = rn_pack8(d);
        if (dres) dres[i] = rn_pack8(g);
    }
}

extern "C" int dgx_affine_act_fwd(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows,
                                  int C, int relu, void* stream) {
    if (rows <= 0) return DGX_OK;
    if (!x || !scale || !shift || !y || C <= 0 || (C & 7)) return DGX_ERR_BAD_ARG;
    const int64_t total = rows * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine_act_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, scale, shift,
                       (const uint4*)residual, (uint4*)y, total, C / 8, relu);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
extern "C" int dgx_affine_act_bwd(const void* dy, const void* y, const float* scale, void* dx, void* dres, int64_t rows, int C, int relu,
                                  void* stream) {
    if (rows <= 0) return DGX_OK;
    if (!dy || !scale || !dx || (relu && !y) || C <= 0 || (C & 7)) return DGX_ERR_BAD_ARG;
    const int64_t total = rows * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine_act_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)dy, (const uint4*)y, scale,
                       (uint4*)dx, (uint4*)dres, total, C / 8, relu);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// max-pool 3x3 / 2 / pad 1 over (N, H, W, C) bf16; idx[(n, oy, ox, c)] = winning tap 0..8 (first maximum in scan order, the rule of
// ATen's max_pool2d: a later tap wins only if strictly greater)
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, uint2* __restrict__ idx,
                                                               int N, int H, int W, int vecC, int Ho, int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t t = i / vecC;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[8];
        uint32_t arg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; arg[k] = 0; }
        bool any = false;
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            float c[8];
            rn_unpack8(x[(((int64_t)n * H + iy) * W + ix) * vecC + v], c);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (!any || c[k] > best[k] || c[k] != c[k]) { best[k] = c[k]; arg[k] = (uint32_t)tap; }
            any = true;
        }
        y[i] = rn_pack8(best);
        idx[i] = make_uint2(arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24), arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24));
    }
}
// dx[(n, iy, ix, c)] = sum of dy over the (<= 4) windows that contain the pixel and whose arg-max is this pixel
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const uint4* __restrict__ dy, const uint2* __restrict__ idx,
                                                               uint4* __restrict__ dx, int N, int H, int W, int vecC, int Ho, int Wo) {
    const int64_t total = (int64_t)N * H * W * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t t = i / vecC;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int oy = (iy >= 1 ? (iy - 1 + 1) / 2 : 0); oy <= (iy + 1) / 2 && oy < Ho; ++oy)
            for (int ox = (ix >= 1 ? (ix - 1 + 1) / 2 : 0); ox <= (ix + 1) / 2 && ox < Wo; ++ox) {
                const int ky = iy - (2 * oy - 1), kx = ix - (2 * ox - 1);
                if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
                const uint32_t tap = (uint32_t)(ky * 3 + kx);
                const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * vecC + v;
                const uint2 a = idx[o];
                float g[8];
                rn_unpack8(dy[o], g);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w = k < 4 ? (a.x >> (8 * k)) & 0xffu : (a.y >> (8 * (k - 4))) & 0xffu;
                    if (w == tap) acc[k] += g[k];
                }
            }
        dx[i] = rn_pack8(acc);
    }
}

extern "C" int dgx_maxpool3x3s2_fwd(const void* x, void* y, void* idx_u8, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!x || !y || !idx_u8 || C <= 0 || (C & 7)) return DGX_ERR_BAD_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, (uint2*)idx_u8,
                       N, H, W, C / 8, Ho, Wo);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
extern "C" int dgx_maxpool3x3s2_bwd(const void* dy, const void* idx_u8, void* dx, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!dy || !idx_u8 || !dx || C <= 0 || (C & 7)) return DGX_ERR_BAD_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * H * W * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)dy, (const uint2*)idx_u8,
                       (uint4*)dx, N, H, W, C / 8, Ho, Wo);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
