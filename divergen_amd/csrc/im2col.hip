// 3x3 (pad 1, stride 1|2) im2col / col2im on channels-last tensors for gfx950.  HBM-bound index
// copies, 16 bytes per lane, consecutive lanes walk the channel dimension so every tap is a
// contiguous C*esize-byte run.  The contraction itself runs as a plain library GEMM on the column
// matrix; these two kernels replace MIOpen, whose bf16 NHWC path fell back to naive direct
// convolution on this image (profiles/r01_smoke_swinT_miopen_kernel_stats.csv).
//   col[(n,oy,ox)][(ky*3+kx)*C + c] = x[n][oy*s-1+ky][ox*s-1+kx][c]   (0 outside)
//   dx[n][y][x][c] = sum over the <=9 (ky,kx,oy,ox) that read (y,x) of dcol[...]   (gather form)
#include "dgx_common.h"

__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint4* __restrict__ x, uint4* __restrict__ col, int N, int H,
                                                        int W, int vecC, int Ho, int Wo, int stride) {
    const int64_t total = (int64_t)N * Ho * Wo * 9 * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t t = i / vecC;
        const int tap = (int)(t % 9);
        t /= 9;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
        uint4 val = {0u, 0u, 0u, 0u};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = x[(((int64_t)n * H + iy) * W + ix) * vecC + v];
        col[i] = val;
    }
}

template <typename T> struct V4;  // 16-byte vector of T with fp32 accumulate
template <> struct V4<float> {
    static constexpr int N = 4;
    static __device__ void add(float* acc, const uint4& v) {
        acc[0] += __uint_as_float(v.x); acc[1] += __uint_as_float(v.y);
        acc[2] += __uint_as_float(v.z); acc[3] += __uint_as_float(v.w);
    }
    static __device__ uint4 pack(const float* a) {
        return make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
    }
};
template <> struct V4<uint16_t> {
    static constexpr int N = 8;
    static __device__ void add(float* acc, const uint4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += __uint_as_float(w[k] << 16);
            acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    static __device__ uint4 pack(const float* a) {
        return make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(a[4], a[5]), pack_bf2(a[6], a[7]));
    }
};

template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const uint4* __restrict__ dcol, uint4* __restrict__ dx, int N, int H,
                                                        int W, int vecC, int Ho, int Wo, int stride) {
    const int64_t total = (int64_t)N * H * W * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t t = i / vecC;
        const int xx = (int)(t % W);
        t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        float acc[V4<T>::N];
#pragma unroll
        for (int k = 0; k < V4<T>::N; ++k) acc[k] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = yy + 1 - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = xx + 1 - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                V4<T>::add(acc, dcol[((((int64_t)n * Ho + oy) * Wo + ox) * 9 + ky * 3 + kx) * vecC + v]);
            }
        }
        dx[i] = V4<T>::pack(acc);
    }
}

static int conv_dims(int H, int W, int stride, int& Ho, int& Wo) {
    if (stride != 1 && stride != 2) return DGX_ERR_UNSUPPORTED;
    Ho = (H + 2 - 3) / stride + 1;
    Wo = (W + 2 - 3) / stride + 1;
    return DGX_OK;
}

extern "C" int dgx_im2col3x3(const void* x, void* col, int N, int H, int W, int C, int stride, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    const int es = dtype == DGX_BF16 ? 2 : 4;
    int Ho, Wo;
    if (conv_dims(H, W, stride, Ho, Wo)) return DGX_ERR_UNSUPPORTED;
    if (!x || !col || (C * es) % 16) return DGX_ERR_BAD_ARG;
    const int vecC = C * es / 16;
    const int64_t total = (int64_t)N * Ho * Wo * 9 * vecC;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)col, N, H,
                       W, vecC, Ho, Wo, stride);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_col2im3x3(const void* dcol, void* dx, int N, int H, int W, int C, int stride, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    const int es = dtype == DGX_BF16 ? 2 : 4;
    int Ho, Wo;
    if (conv_dims(H, W, stride, Ho, Wo)) return DGX_ERR_UNSUPPORTED;
    if (!dcol || !dx || (C * es) % 16) return DGX_ERR_BAD_ARG;
    const int vecC = C * es / 16;
    const int64_t total = (int64_t)N * H * W * vecC;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(col2im3x3_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)dcol,
                           (uint4*)dx, N, H, W, vecC, Ho, Wo, stride);
    else
        hipLaunchKernelGGL(col2im3x3_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)dcol,
                           (uint4*)dx, N, H, W, vecC, Ho, Wo, stride);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}


// Zero-bordered copy of an NHWC bf16 image for the implicit 3x3 convolution (dgx_conv3x3_gemm): rows = (W + 3) zero slack rows,
// then the (N, H + 2, W + 2) grid with the image in its interior, then (W + 3) zero slack rows; every tap of every grid position
// is then a constant row shift inside the allocation.  2 x the image in HBM traffic instead of the 9 x column matrix of im2col.
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int N, int H, int W,
                                                       int vecC, int64_t total_rows) {
    const int wp = W + 2, hp = H + 2, slack = W + 3;
    const int64_t total = total_rows * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        const int64_t row = i / vecC - slack;
        uint4 val = {0u, 0u, 0u, 0u};
        if (row >= 0 && row < (int64_t)N * hp * wp) {
            const int n = (int)(row / (hp * wp)), r = (int)(row - (int64_t)n * hp * wp);
            const int yp = r / wp, xp = r - yp * wp;
            if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) val = x[(((int64_t)n * H + yp - 1) * W + xp - 1) * vecC + v];
        }
        out[i] = val;
    }
}

// The same copy with ReLU' folded in: element e of the copy = g[e] where the saved activation y[e] (bf16, the convolution's own
// ReLU-ed output) is positive, else 0 -- the output gradient of a convolution with a fused ReLU, masked on its way into the padded
// image that its input- and weight-gradient kernels read (round 2: a compare and a multiply launch in front of the copy).
__global__ __launch_bounds__(256) void pad_nhwc_relu_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y, uint4* __restrict__ out,
                                                            int N, int H, int W, int vecC, int64_t total_rows) {
    const int wp = W + 2, hp = H + 2, slack = W + 3;
    const int64_t total = total_rows * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        const int64_t row = i / vecC - slack;
        uint4 val = {0u, 0u, 0u, 0u};
        if (row >= 0 && row < (int64_t)N * hp * wp) {
            const int n = (int)(row / (hp * wp)), r = (int)(row - (int64_t)n * hp * wp);
            const int yp = r / wp, xp = r - yp * wp;
            if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) {
                const int64_t src = (((int64_t)n * H + yp - 1) * W + xp - 1) * vecC + v;
                const uint4 gv = g[src], yv = y[src];
                const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {      // bf16 > 0: sign bit clear and not (+)zero
                    const uint32_t a = yw[k];
                    const uint32_t lo = ((a & 0x8000u) || !(a & 0x7fffu)) ? 0u : 0xffffu;
                    const uint32_t hi = ((a & 0x80000000u) || !(a & 0x7fff0000u)) ? 0u : 0xffff0000u;
                    o[k] = gw[k] & (lo | hi);
                }
                val = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        out[i] = val;
    }
}

extern "C" int dgx_conv3x3_pad_relu_grad(const void* g, const void* y, void* gpad, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!g || !y || !gpad || C <= 0 || (C & 7)) return DGX_ERR_BAD_ARG;
    const int64_t rows = (int64_t)N * (H + 2) * (W + 2) + 2 * (int64_t)(W + 3);
    const int64_t total = rows * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(pad_nhwc_relu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, (const uint4*)y, (uint4*)gpad, N, H,
                       W, C / 8, rows);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// Several images (the FPN levels of one tower layer) in one launch: blockIdx.y = the image set
constexpr int PAD_MAXI = 8;
struct PadMulti { const uint4* x[PAD_MAXI]; uint4* out[PAD_MAXI]; int N[PAD_MAXI], H[PAD_MAXI], W[PAD_MAXI]; int64_t rows[PAD_MAXI]; };
__global__ __launch_bounds__(256) void pad_nhwc_multi_kernel(PadMulti M, int vecC) {
    const int k = blockIdx.y;
    const uint4* __restrict__ x = M.x[k];
    uint4* __restrict__ out = M.out[k];
    const int N = M.N[k], H = M.H[k], W = M.W[k];
    const int wp = W + 2, hp = H + 2, slack = W + 3;
    const int64_t total = M.rows[k] * vecC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        const int64_t row = i / vecC - slack;
        uint4 val = {0u, 0u, 0u, 0u};
        if (row >= 0 && row < (int64_t)N * hp * wp) {
            const int n = (int)(row / (hp * wp)), r = (int)(row - (int64_t)n * hp * wp);
            const int yp = r / wp, xp = r - yp * wp;
            if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) val = x[(((int64_t)n * H + yp - 1) * W + xp - 1) * vecC + v];
        }
        out[i] = val;
    }
}

extern "C" int dgx_conv3x3_pad_multi(const dgx_pad_item* items, int n, int C, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!items || n > PAD_MAXI || C <= 0 || (C & 7)) return n > PAD_MAXI ? DGX_ERR_UNSUPPORTED : DGX_ERR_BAD_ARG;
    PadMulti M;
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) {
        const dgx_pad_item& a = items[i];
        if (!a.x || !a.xpad || a.N <= 0 || a.H <= 0 || a.W <= 0) return DGX_ERR_BAD_ARG;
        M.x[i] = (const uint4*)a.x; M.out[i] = (uint4*)a.xpad; M.N[i] = a.N; M.H[i] = a.H; M.W[i] = a.W;
        M.rows[i] = (int64_t)a.N * (a.H + 2) * (a.W + 2) + 2 * (int64_t)(a.W + 3);
        const int64_t t = M.rows[i] * (C / 8);
        mx = t > mx ? t : mx;
    }
    for (int i = n; i < PAD_MAXI; ++i) { M.x[i] = M.x[0]; M.out[i] = M.out[0]; M.N[i] = M.N[0]; M.H[i] = M.H[0]; M.W[i] = M.W[0]; M.rows[i] = 0; }
    const int grid = (int)((mx + 255) / 256 < 4096 ? (mx + 255) / 256 : 4096);
    hipLaunchKernelGGL(pad_nhwc_multi_kernel, dim3(grid, n), dim3(256), 0, (hipStream_t)stream, M, C / 8);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_conv3x3_pad(const void* x, void* xpad, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!x || !xpad || C <= 0 || (C & 7)) return DGX_ERR_BAD_ARG;
    const int64_t rows = (int64_t)N * (H + 2) * (W + 2) + 2 * (int64_t)(W + 3);
    const int64_t total = rows * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)xpad, N, H, W, C / 8, rows);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(kernel 2, stride 2) = ONE GEMM over the weight as stored, (Cin, Cout*2*2), + a pixel shuffle
// (mask_head.py:209-284 `deconv`; D2 wrappers.ConvTranspose2d): the GEMM's row m = (n, h, w) holds columns (co, dy, dx);
//   dgx_deconv2x2_shuffle             out[n][2h+dy][2w+dx][co] = y2[m][4 co + 2 dy + dx]                     (forward)
//   dgx_deconv2x2_unshuffle_relu_grad g2[m][4 co + 2 dy + dx]  = yout[p][co] > 0 ? gy[p][co] : 0, p = that pixel (backward: the
//                                     un-shuffle of the output gradient with the ReLU' of the layer's fused ReLU folded in;
//                                     yout NULL = no ReLU)
// One lane = 8 channels of one input pixel: 64 contiguous bytes on the GEMM side, four 16-byte chunks on the image side.  The
// composed form was a permuting copy each way + compare + multiply launches (0.15 ms per step on the mask head).
__device__ __forceinline__ uint32_t dc_pick(const uint32_t w[16], int e) {     // bf16 element e (0 .. 31) of 16 packed words
    const uint32_t v = w[e >> 1];
    return (e & 1) ? (v >> 16) : (v & 0xffffu);
}
__global__ __launch_bounds__(256) void deconv2x2_shuffle_kernel(const uint4* __restrict__ y2, uint4* __restrict__ out, int64_t M, int H, int W,
                                                                int C8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C8) return;
    const int64_t m = i / C8;
    const int c8 = (int)(i - m * C8);
    const int w_ = (int)(m % W), h = (int)((m / W) % H);
    const int64_t n = m / ((int64_t)W * H);
    uint32_t in[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = y2[(m * C8 + c8) * 4 + q];
        in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                  // k = 2 dy + dx: element j of the chunk = input element 4 j + k
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = dc_pick(in, 8 * j + k) | (dc_pick(in, 8 * j + 4 + k) << 16);
        const int64_t p = ((n * 2 * H + 2 * h + (k >> 1)) * 2 * W + 2 * w_ + (k & 1));
        out[p * C8 + c8] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
__global__ __launch_bounds__(256) void deconv2x2_unshuffle_kernel(const uint4* __restrict__ gy, const uint4* __restrict__ yout,
                                                                  uint4* __restrict__ g2, int64_t M, int H, int W, int C8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C8) return;
    const int64_t m = i / C8;
    const int c8 = (int)(i - m * C8);
    const int w_ = (int)(m % W), h = (int)((m / W) % H);
    const int64_t n = m / ((int64_t)W * H);
    uint32_t ch[4][4];                             // ch[k][j]: channels 2j, 2j+1 of sub-pixel k
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t p = ((n * 2 * H + 2 * h + (k >> 1)) * 2 * W + 2 * w_ + (k & 1));
        const uint4 g = gy[p * C8 + c8];
        uint32_t gw[4] = {g.x, g.y, g.z, g.w};
        if (yout) {
            const uint4 yv = yout[p * C8 + c8];
            const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // bf16 > 0: sign bit clear and not (+)zero
                const uint32_t a = yw[j];
                const uint32_t lo = ((a & 0x8000u) || !(a & 0x7fffu)) ? 0u : 0xffffu;
                const uint32_t hi = ((a & 0x80000000u) || !(a & 0x7fff0000u)) ? 0u : 0xffff0000u;
                gw[j] &= lo | hi;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ch[k][j] = gw[j];
    }
    // output element e = 4 c + k (c = 0 .. 7): word e / 2 holds (c, k = 0 | 2) low and (c, k = 1 | 3) high
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 8 * q + 2 * t;           // even element: channel c = e / 4, sub-pixel k = e % 4 (0 or 2), and k + 1 behind it
            const int c = e >> 2, k = e & 3;
            const uint32_t lo = (c & 1) ? (ch[k][c >> 1] >> 16) : (ch[k][c >> 1] & 0xffffu);
            const uint32_t hi = (c & 1) ? (ch[k + 1][c >> 1] >> 16) : (ch[k + 1][c >> 1] & 0xffffu);
            o[t] = lo | (hi << 16);
        }
        g2[(m * C8 + c8) * 4 + q] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int dgx_deconv2x2_shuffle(const void* y2, void* out, int N, int H, int W, int Cout, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!y2 || !out || Cout <= 0 || (Cout & 7) || ((uintptr_t)y2 & 15) || ((uintptr_t)out & 15)) return DGX_ERR_BAD_ARG;
    const int64_t M = (int64_t)N * H * W, total = M * (Cout / 8);
    if ((total + 255) / 256 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(deconv2x2_shuffle_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)y2,
                       (uint4*)out, M, H, W, Cout / 8);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
extern "C" int dgx_deconv2x2_unshuffle_relu_grad(const void* gy, const void* yout, void* g2, int N, int H, int W, int Cout, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!gy || !g2 || Cout <= 0 || (Cout & 7) || ((uintptr_t)gy & 15) || ((uintptr_t)g2 & 15) || ((uintptr_t)yout & 15)) return DGX_ERR_BAD_ARG;
    const int64_t M = (int64_t)N * H * W, total = M * (Cout / 8);
    if ((total + 255) / 256 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(deconv2x2_unshuffle_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)gy,
                       (const uint4*)yout, (uint4*)g2, M, H, W, Cout / 8);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
