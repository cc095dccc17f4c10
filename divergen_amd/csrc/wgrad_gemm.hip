// Weight-gradient GEMM for gfx950:  C[Nn][Kk] (fp32, += ) = sum_m A[m][Nn] * B[m][Kk]
// with A = dY and B = X both row-major bf16 and M (tokens) up to 131072 while Nn, Kk are a few
// hundred to a few thousand.  The library GEMM runs this "TN, very long K" shape at 65-500 TF/s
// (tools/wgrad_probe.py) because the small output gives it too few workgroups; here the M axis is
// split across workgroups (split-K):
//   tile 128(n) x 128(k) per workgroup, 4 waves x (4x4 MFMA 16x16x32 bf16 tiles), M-depth 64 per
//   stage.  Both operands are stored in LDS exactly as they sit in memory ([m][n], 16-byte stores)
//   and the k-contiguous MFMA fragments come out of ds_read_b64_tr_b16 transpose reads, so no
//   transposing pass exists anywhere.  Double-buffered LDS, next stage prefetched into registers
//   under the MFMAs; fp32 partial tiles go to a workspace and a second kernel folds the S slabs into
//   the gradient arena (beta = 1).
#include "dgx_common.h"

namespace {
constexpr int BT = 128;            // tile edge (both n and k)
constexpr int BM = 64;             // m-depth per stage
constexpr int RSB = BT * 2 + 16;   // LDS row stride in bytes (+16: consecutive rows shift 4 banks)

struct Stage { bf16x8 a[4], b[4]; };

// thread t: rows r = (t >> 4) + 16*i (i = 0..3), 16-byte column chunk c = t & 15
__device__ __forceinline__ void load_stage(Stage& st, const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                           int64_t lda, int64_t ldb, int m0, int M, int n0, int Nn, int k0, int Kk, int tid) {
    const int r = tid >> 4, c = tid & 15;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r + 16 * i;
        const bool mok = m < M;
        st.a[i] = (mok && n0 + 8 * c < Nn) ? *reinterpret_cast<const bf16x8*>(A + (int64_t)m * lda + n0 + 8 * c) : z;
        st.b[i] = (mok && k0 + 8 * c < Kk) ? *reinterpret_cast<const bf16x8*>(B + (int64_t)m * ldb + k0 + 8 * c) : z;
    }
}

__device__ __forceinline__ void store_stage(const Stage& st, unsigned char* Ai, unsigned char* Bi, int tid) {
    const int r = tid >> 4, c = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<bf16x8*>(Ai + (r + 16 * i) * RSB + 16 * c) = st.a[i];
        *reinterpret_cast<bf16x8*>(Bi + (r + 16 * i) * RSB + 16 * c) = st.b[i];
    }
}
}  // namespace

__global__ __launch_bounds__(256) void wgrad_partial_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                            float* __restrict__ ws, int M, int Nn, int Kk, int64_t lda,
                                                            int64_t ldb, int tiles_k, int slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [buf][A|B][BM*RSB]
    const int tile = blockIdx.x, s = blockIdx.y;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    const int m_begin = s * slab, m_end = min(M, m_begin + slab);
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    const int wn = (w >> 1) * 64, wk = (w & 1) * 64;    // this wave's 64x64 quadrant

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stage st;
    int buf = 0;
    if (m_begin < m_end) load_stage(st, A, B, lda, ldb, m_begin, m_end, n0, Nn, k0, Kk, tid);
    for (int m0 = m_begin; m0 < m_end; m0 += BM) {
        unsigned char* Ai = lds_raw + (size_t)(2 * buf) * BM * RSB;
        unsigned char* Bi = Ai + (size_t)BM * RSB;
        store_stage(st, Ai, Bi, tid);
        __syncthreads();
        if (m0 + BM < m_end) load_stage(st, A, B, lda, ldb, m0 + BM, m_end, n0, Nn, k0, Kk, tid);
        // this lane's chunk: row 8g + (c16 >> 2) of the 32-row k-step, 8-byte column chunk c16 & 3
        DGX_LDS const uint16_t* a_l = lds_opaque(reinterpret_cast<const uint16_t*>(Ai + (8 * g + (c16 >> 2)) * RSB) + wn + 4 * (c16 & 3));
        DGX_LDS const uint16_t* b_l = lds_opaque(reinterpret_cast<const uint16_t*>(Bi + (8 * g + (c16 >> 2)) * RSB) + wk + 4 * (c16 & 3));
#pragma unroll
        for (int ks = 0; ks < BM / 32; ++ks) {
            constexpr int RSE = RSB / 2;   // row stride in elements
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = tr_frag(a_l, 32 * ks * RSE + 16 * i, (32 * ks + 4) * RSE + 16 * i);
                bfr[i] = tr_frag(b_l, 32 * ks * RSE + 16 * i, (32 * ks + 4) * RSE + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
        }
        buf ^= 1;   // the other buffer was last read one iteration ago, behind this iteration's barrier
    }
    float* out = ws + (int64_t)s * Nn * Kk;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn + 16 * i + 4 * g + r;
            if (n >= Nn) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + wk + 16 * j + c16;
                if (k < Kk) out[(int64_t)n * Kk + k] = acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4* __restrict__ ws, float4* __restrict__ C, int64_t n4,
                                                           int S, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            const float4 v = ws[(int64_t)s * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (beta != 0.f) {
            const float4 c = C[i];
            a.x += beta * c.x; a.y += beta * c.y; a.z += beta * c.z; a.w += beta * c.w;
        }
        C[i] = a;
    }
}

static int wgrad_splits(int M, int Nn, int Kk) {
    const int tiles = ((Nn + BT - 1) / BT) * ((Kk + BT - 1) / BT);
    int S = (1024 + tiles - 1) / tiles;
    const int maxS = (M + 4 * BM - 1) / (4 * BM);
    if (S > maxS) S = maxS;
    return S < 1 ? 1 : S;
}

extern "C" int64_t dgx_wgrad_workspace_bytes(int M, int Nn, int Kk) {
    if (M <= 0 || Nn <= 0 || Kk <= 0) return 0;
    return (int64_t)wgrad_splits(M, Nn, Kk) * Nn * Kk * 4;
}

extern "C" int dgx_linear_wgrad(const void* dy, const void* x, float* gw, int M, int Nn, int Kk, float beta,
                                void* workspace, void* stream) {
    if (M <= 0 || Nn <= 0 || Kk <= 0) return DGX_OK;
    if (!dy || !x || !gw || !workspace || (Nn & 7) || (Kk & 7) || ((int64_t)Nn * Kk & 3)) return DGX_ERR_BAD_ARG;
    const int tiles_n = (Nn + BT - 1) / BT, tiles_k = (Kk + BT - 1) / BT;
    const int S = wgrad_splits(M, Nn, Kk);
    int slab = (M + S - 1) / S;
    slab = (slab + BM - 1) / BM * BM;
    hipStream_t st = (hipStream_t)stream;
    DgxProfScope prof(DGX_PROF_WGRAD, stream, 2.0 * M * Nn * Kk, 2.0 * M * ((double)Nn + Kk) + 8.0 * Nn * Kk);
    const size_t sm = (size_t)4 * BM * RSB;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)wgrad_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        once = true;
    }
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(tiles_n * tiles_k, S), dim3(256), sm, st, (const uint16_t*)dy,
                       (const uint16_t*)x, (float*)workspace, M, Nn, Kk, (int64_t)Nn, (int64_t)Kk, tiles_k, slab);
    const int64_t n4 = (int64_t)Nn * Kk / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float4*)workspace, (float4*)gw, n4, S, beta);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
