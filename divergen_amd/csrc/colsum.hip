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


// ---- grouped form: the bias gradients of several Linear layers (the four of a Swin block) in two launches ----
namespace {
constexpr int MAXC = 8;
struct ColProb { const uint16_t* dy; float* out; float* part; int M, N, panels, slabs, rows, blk0, fin0; };
struct ColParams { ColProb p[MAXC]; int n; float beta; };
}  // namespace

__global__ __launch_bounds__(256) void colsum_grouped_partial_kernel(ColParams P) {
    __shared__ float red[4][PANEL];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXC; ++i)
        if (i < P.n && (int)blockIdx.x >= P.p[i].blk0) pi = i;
    const ColProb q = P.p[pi];
    const int local = blockIdx.x - q.blk0;
    const int panel = local % q.panels, slab = local / q.panels;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = panel * PANEL + 8 * lane;
    const int r0 = slab * q.rows, r1 = min(q.M, r0 + q.rows);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < q.N) {
        const uint16_t* p = q.dy + (int64_t)r0 * q.N + c0;
#pragma unroll 4
        for (int r = r0 + w; r < r1; r += 4) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(p + (int64_t)(r - r0) * q.N);
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
        const int col = panel * PANEL + c;
        if (col < q.N) q.part[(int64_t)slab * q.N + col] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

__global__ __launch_bounds__(1024) void colsum_grouped_final_kernel(ColParams P) {
    __shared__ float red[16][64];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXC; ++i)
        if (i < P.n && (int)blockIdx.x >= P.p[i].fin0) pi = i;
    const ColProb& q = P.p[pi];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = (blockIdx.x - q.fin0) * 64 + lane;
    float s = 0.f;
    if (n < q.N) {
#pragma unroll 4
        for (int b = rg; b < q.slabs; b += 16) s += q.part[(int64_t)b * q.N + n];
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && n < q.N) {
        float a = red[0][lane];
#pragma unroll
        for (int r = 1; r < 16; ++r) a += red[r][lane];
        q.out[n] = P.beta != 0.f ? P.beta * q.out[n] + a : a;
    }
}

static int grouped_slabs(int M, int N, int nprob) {
    const int panels = (N + PANEL - 1) / PANEL;
    int slabs = (768 / (nprob > 0 ? nprob : 1) + panels - 1) / panels;
    if (slabs > 256) slabs = 256;
    const int max_slabs = (M + 15) / 16;
    if (slabs > max_slabs) slabs = max_slabs;
    return slabs < 1 ? 1 : slabs;
}

extern "C" int64_t dgx_colsum_grouped_workspace_bytes(const dgx_colsum_problem* pr, int n) {
    if (!pr || n <= 0 || n > MAXC) return 0;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) tot += (int64_t)grouped_slabs(pr[i].M, pr[i].N, n) * pr[i].N;
    return tot * 4;
}

extern "C" int dgx_colsum_grouped(const dgx_colsum_problem* pr, int n, float beta, void* workspace, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!pr || n > MAXC || !workspace) return DGX_ERR_BAD_ARG;
    ColParams P;
    P.n = n;
    P.beta = beta;
    int blk = 0, fin = 0;
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (!pr[i].dy || !pr[i].out || pr[i].M <= 0 || pr[i].N <= 0 || (pr[i].N & 7)) return DGX_ERR_BAD_ARG;
        ColProb& q = P.p[i];
        q.dy = (const uint16_t*)pr[i].dy;
        q.out = pr[i].out;
        q.M = pr[i].M;
        q.N = pr[i].N;
        q.panels = (q.N + PANEL - 1) / PANEL;
        q.slabs = grouped_slabs(q.M, q.N, n);
        q.rows = (q.M + q.slabs - 1) / q.slabs;
        q.part = (float*)workspace + off;
        q.blk0 = blk;
        q.fin0 = fin;
        off += (int64_t)q.slabs * q.N;
        blk += q.panels * q.slabs;
        fin += (q.N + 63) / 64;
    }
    for (int i = n; i < MAXC; ++i) P.p[i] = P.p[0];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_grouped_partial_kernel, dim3(blk), dim3(256), 0, st, P);
    hipLaunchKernelGGL(colsum_grouped_final_kernel, dim3(fin), dim3(1024), 0, st, P);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
