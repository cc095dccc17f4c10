// Bias gradient of a Linear layer: out[n] = beta*out[n] + sum_m dY[m][n], dY bf16 row-major (M x N),
// out fp32 (a slice of the gradient arena).  HBM-bound: M*N*2 bytes read once.
//   stage 1: grid (column panels of 512, row slabs); a wave owns whole rows of its panel (1 KiB
//            contiguous per row, 16 B per lane), 8 fp32 accumulators per lane; the 4 waves of a block
//            fold through LDS into part[slab][N].
//   stage 2: out[n] = beta*out[n] + sum_slab part[slab][n]   (deterministic order).
#include "dgx_common.h"

namespace {
constexpr int PANEL = 512;   // columns per block: 64 lanes x 8 bf16
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const uint16_t* __restrict__ dy, float* __restrict__ part,
                                                             int M, int N, int rows_per_slab) {
    __shared__ float red[4][PANEL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * PANEL + 8 * lane;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {
        const uint16_t* p = dy + (int64_t)r0 * N + c0;
#pragma unroll 4
        for (int r = r0 + w; r < r1; r += 4) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(p + (int64_t)(r - r0) * N);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += __uint_as_float(v[i] << 16);
                acc[2 * i + 1] += __uint_as_float(v[i] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[w][8 * lane + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < PANEL; c += 256) {
        const int col = blockIdx.x * PANEL + c;
        if (col < N) part[(int64_t)blockIdx.y * N + col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

// block = 64 columns x 16 slab groups (1024 threads): each thread folds every 16th slab, LDS folds the groups
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N,
                                                            int slabs, float beta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int b = rg; b < slabs; b += 16) s += part[(int64_t)b * N + n];
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float a = red[0][lane];
#pragma unroll
        for (int r = 1; r < 16; ++r) a += red[r][lane];
        out[n] = beta != 0.f ? beta * out[n] + a : a;
    }
}

static int colsum_slabs(int M, int N) {
    const int panels = (N + PANEL - 1) / PANEL;
    int slabs = (768 + panels - 1) / panels;            // ~3 blocks per CU
    if (slabs > 256) slabs = 256;
    const int max_slabs = (M + 15) / 16;                // at least 16 rows per block
    if (slabs > max_slabs) slabs = max_slabs;
    return slabs < 1 ? 1 : slabs;
}

extern "C" int64_t dgx_colsum_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (int64_t)colsum_slabs(M, N) * N * 4;
}

extern "C" int dgx_colsum_bf16(const void* dy, float* out, int M, int N, float beta, void* workspace, void* stream) {
    if (N <= 0) return DGX_OK;
    if (!out || (M > 0 && (!dy || !workspace)) || (N & 7)) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int slabs = M > 0 ? colsum_slabs(M, N) : 0;
    if (M > 0) {
        const int rows = (M + slabs - 1) / slabs;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + PANEL - 1) / PANEL, slabs), dim3(256), 0, st,
                           (const uint16_t*)dy, (float*)workspace, M, N, rows);
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)workspace, out, N, slabs,
                       beta);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
