// Exact (erf) GELU of the Swin MLP (DG/divergen/modeling/backbone/swintransformer.py:40-46, nn.GELU()) on bf16
// activations, forward and backward; the backward also folds the fc1 bias gradient (column sums of its output)
// into the same pass.  HBM-bound streams, 16 bytes per lane.
//   fwd:  a = 0.5 x (1 + erf(x / sqrt 2))
//   bwd:  dx = dy * (0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi));  part[slab][n] = sum over the slab's rows of dx
#include "dgx_common.h"

namespace {
__device__ __forceinline__ void unpack8g(const u32x4 r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8g(const float (&v)[8]) {
    return u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;
}  // namespace

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8], o[8];
        unpack8g(x[i], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = 0.5f * v[k] * (1.0f + erff(v[k] * kInvSqrt2));
        y[i] = pack8g(o);
    }
}

// grid (ceil(N/512), slabs): a wave owns whole rows of a 512-column panel (like colsum_partial_kernel)
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                              uint16_t* __restrict__ dx, float* __restrict__ part, int M, int N,
                                                              int rows_per_slab) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + 8 * lane;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {
        for (int r = r0 + w; r < r1; r += 4) {
            const int64_t o = (int64_t)r * N + c0;
            float g[8], v[8], d[8];
            unpack8g(*reinterpret_cast<const u32x4*>(dy + o), g);
            unpack8g(*reinterpret_cast<const u32x4*>(x + o), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float cdf = 0.5f * (1.0f + erff(v[k] * kInvSqrt2));
                const float pdf = __expf(-0.5f * v[k] * v[k]) * kInvSqrt2Pi;
                d[k] = g[k] * (cdf + v[k] * pdf);
            }
            const u32x4 pk = pack8g(d);
            *reinterpret_cast<u32x4*>(dx + o) = pk;
            // the bias gradient sums the bf16-ROUNDED dx (what a separate pass over dx would read)
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += __uint_as_float(pk[k] << 16); acc[2 * k + 1] += __uint_as_float(pk[k] & 0xffff0000u); }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[w][8 * lane + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int col = blockIdx.x * 512 + c;
        if (col < N) part[(int64_t)blockIdx.y * N + col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

__global__ __launch_bounds__(1024) void gelu_colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int slabs,
                                                                 float beta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (n < N)
        for (int b = rg; b < slabs; b += 16) s += part[(int64_t)b * N + n];
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float a = red[0][lane];
#pragma unroll
        for (int r = 1; r < 16; ++r) a += red[r][lane];
        out[n] = beta != 0.f ? beta * out[n] + a : a;
    }
}

static int gelu_slabs(int M, int N) {
    const int panels = (N + 511) / 512;
    int slabs = (1024 + panels - 1) / panels;
    if (slabs > 256) slabs = 256;
    const int mx = (M + 15) / 16;
    if (slabs > mx) slabs = mx;
    return slabs < 1 ? 1 : slabs;
}

extern "C" int dgx_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!x || !y || (n & 7)) return DGX_ERR_BAD_ARG;
    const int64_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)y, n8);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int64_t dgx_gelu_bwd_workspace_bytes(int M, int N) { return (M <= 0 || N <= 0) ? 0 : (int64_t)gelu_slabs(M, N) * N * 4; }

extern "C" int dgx_gelu_bwd_colsum(const void* dy, const void* x, void* dx, float* bias_grad, int M, int N, float beta,
                                   void* workspace, void* stream) {
    if (M <= 0 || N <= 0) return DGX_OK;
    if (!dy || !x || !dx || !workspace || (N & 7)) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int slabs = gelu_slabs(M, N);
    const int rows = (M + slabs - 1) / slabs;
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((N + 511) / 512, slabs), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x,
                       (uint16_t*)dx, (float*)workspace, M, N, rows);
    if (bias_grad)
        hipLaunchKernelGGL(gelu_colsum_final_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)workspace, bias_grad, N, slabs,
                           beta);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
