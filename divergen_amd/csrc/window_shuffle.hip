// Window gather / scatter: pad + cyclic shift + window_partition folded into one index map
// (swintransformer.py:216-233, :239-251).  Pure HBM-bound copy: 16 bytes per lane, one token row
// per group of C*esize/16 lanes, no intermediate padded / rolled tensors.
#include "dgx_common.h"

template <bool GATHER>
__global__ __launch_bounds__(256) void window_shuffle_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                             int B, int H, int W, int vecC, int ws, int shift,
                                                             int nWh, int nWw) {
    const int N = ws * ws;
    const int64_t total = (int64_t)B * nWh * nWw * N * vecC;
    const int Hp = nWh * ws, Wp = nWw * ws;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecC);
        int64_t t = i / vecC;                 // window-major token index
        const int n = (int)(t % N);
        t /= N;
        const int wc = (int)(t % nWw);
        t /= nWw;
        const int wr = (int)(t % nWh);
        const int b = (int)(t / nWh);
        // position in the shifted, padded frame -> position in the un-shifted padded frame
        int hs = wr * ws + n / ws, wsx = wc * ws + n % ws;
        int hh = hs + shift, ww = wsx + shift;   // roll(-shift): shifted[h] = x[(h + shift) % Hp]
        if (hh >= Hp) hh -= Hp;
        if (ww >= Wp) ww -= Wp;
        const bool inside = hh < H && ww < W;
        const int64_t xi = (((int64_t)b * H + hh) * W + ww) * vecC + v;
        if (GATHER) {
            uint4 z = {0u, 0u, 0u, 0u};
            dst[i] = inside ? src[xi] : z;
        } else if (inside) {
            dst[xi] = src[i];
        }
    }
}

static int window_shuffle(bool gather, const void* src, void* dst, int B, int H, int W, int C, int ws, int shift,
                          int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return DGX_OK;
    const int es = dtype == DGX_BF16 ? 2 : 4;
    if (!src || !dst || ws <= 0 || shift < 0 || shift >= ws || (C * es) % 16) return DGX_ERR_BAD_ARG;
    const int vecC = C * es / 16;
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
    const int64_t total = (int64_t)B * nWh * nWw * ws * ws * vecC;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (gather)
        hipLaunchKernelGGL(window_shuffle_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)src, (uint4*)dst, B, H, W, vecC, ws, shift, nWh, nWw);
    else
        hipLaunchKernelGGL(window_shuffle_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)src, (uint4*)dst, B, H, W, vecC, ws, shift, nWh, nWw);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_window_gather(const void* x, void* xw, int B, int H, int W, int C, int ws, int shift, int dtype,
                                 void* stream) {
    return window_shuffle(true, x, xw, B, H, W, C, ws, shift, dtype, stream);
}
extern "C" int dgx_window_scatter(const void* xw, void* x, int B, int H, int W, int C, int ws, int shift, int dtype,
                                  void* stream) {
    return window_shuffle(false, xw, x, B, H, W, C, ws, shift, dtype, stream);
}
