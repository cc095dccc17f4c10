// Transposed bf16 weight images for the input-gradient GEMMs (gfx950).
// dx = dy W of a Linear (autograd's mm(dy, W), reached from every F.linear on the path: swintransformer.py:40-46,133,155,296)
// needs W with the OUTPUT index contiguous; dgx_gemm_bf16_nt wants both operands contraction-contiguous.  Instead of a
// second GEMM variant with transposing LDS reads, the bf16 shadow arena gets a twin that holds every matrix parameter
// transposed, rebuilt once per optimizer step by ONE grouped launch: 2 x 2 B per parameter of HBM traffic (~1 GB per step
// for Swin-L CenterNet2, ~0.2 ms) against ~4.5 TFLOP of input-gradient GEMMs that then run on the forward kernel.
#include "dgx_common.h"

namespace {
// matrix (rows, cols) at element offset `off` in both arenas; cin > 0: 3x3 convolution weight, tap-flipped twin
struct TJob { int64_t off; int rows, cols; int64_t tile0; int64_t cin; };

__global__ __launch_bounds__(256) void transpose_grouped_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                                const TJob* __restrict__ jobs, int njobs, int64_t total_tiles) {
    __shared__ uint16_t tile[64][72];   // 144-byte rows: 16-byte aligned, 36 dwords (conflict-light column walks)
    for (int64_t T = blockIdx.x; T < total_tiles; T += gridDim.x) {
        int lo = 0, hi = njobs - 1;     // last job with tile0 <= T
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].tile0 <= T) lo = mid; else hi = mid - 1;
        }
        const TJob j = jobs[lo];
        const int tc = (j.cols + 63) >> 6;
        const int64_t local = T - j.tile0;
        const int r0 = (int)(local / tc) * 64, c0 = (int)(local % tc) * 64;
        const uint16_t* s = src + j.off;
        uint16_t* d = dst + j.off;
        const int t = threadIdx.x;
        const bool vec = ((j.rows | j.cols) & 7) == 0 && (j.off & 7) == 0;
        // destination of source element (r, c): plain transpose d[c * rows + r]; conv twin (cols = 9 cin, c = tap * cin + ci):
        // d[ci * 9 rows + (8 - tap) * rows + r] -- a 64-column tile lies inside one tap (cin % 64 == 0)
        int64_t dbase = (int64_t)c0 * j.rows, dstride = j.rows;
        if (j.cin > 0) {
            const int tap = c0 / (int)j.cin, ci0 = c0 - tap * (int)j.cin;
            dbase = (int64_t)ci0 * 9 * j.rows + (int64_t)(8 - tap) * j.rows;
            dstride = 9 * (int64_t)j.rows;
        }
        __syncthreads();
        if (vec) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int r = (t >> 3) + 32 * p, c = (t & 7) * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (r0 + r < j.rows && c0 + c < j.cols) v = *reinterpret_cast<const u32x4*>(s + (int64_t)(r0 + r) * j.cols + c0 + c);
                *reinterpret_cast<u32x4*>(&tile[r][c]) = v;
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int oc = (t >> 3) + 32 * p, orr = (t & 7) * 8;      // output row = source column
                if (c0 + oc < j.cols && r0 + orr < j.rows) {
                    uint32_t w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[orr + 2 * k][oc] | ((uint32_t)tile[orr + 2 * k + 1][oc] << 16);
                    *reinterpret_cast<u32x4*>(d + dbase + (int64_t)oc * dstride + r0 + orr) = u32x4{w[0], w[1], w[2], w[3]};
                }
            }
        } else {
            for (int i = t; i < 64 * 64; i += 256) {
                const int r = i >> 6, c = i & 63;
                tile[r][c] = (r0 + r < j.rows && c0 + c < j.cols) ? s[(int64_t)(r0 + r) * j.cols + c0 + c] : (uint16_t)0;
            }
            __syncthreads();
            for (int i = t; i < 64 * 64; i += 256) {
                const int oc = i >> 6, orr = i & 63;
                if (c0 + oc < j.cols && r0 + orr < j.rows) d[dbase + (int64_t)oc * dstride + r0 + orr] = tile[orr][oc];
            }
        }
    }
}
}  // namespace

extern "C" int dgx_transpose_bf16_grouped(const void* src, void* dst, const void* jobs, int njobs, int64_t total_tiles, void* stream) {
    if (njobs <= 0 || total_tiles <= 0) return DGX_OK;
    if (!src || !dst || !jobs) return DGX_ERR_BAD_ARG;
    const int grid = (int)(total_tiles < 16384 ? total_tiles : 16384);
    hipLaunchKernelGGL(transpose_grouped_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst,
                       (const TJob*)jobs, njobs, total_tiles);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
