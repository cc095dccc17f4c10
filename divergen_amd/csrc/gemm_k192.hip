// The K = 192 GEMMs of Swin stage 0 (fc1 forward / fc2 input gradient with their GELU tails, qkv, proj: 131 072 - 139 392 rows, 0.8 ms of a
// step) as a kernel for what bounds them -- HBM, not the matrix pipe: 50 MB in and up to 400 MB out per launch for 38 GFLOP.
//
// gemm_nt / gemm_lw move a tile through serial phases (operands in, K-steps, read-out), two workgroups per CU at most: 3.0-3.5 TB/s where a
// fill writes 6.9.  Here the whole contraction is resident: a workgroup keeps its 192-column panel of W (192 x 192 bf16, 72 KiB)
// in LDS for its whole life, and each of its six waves walks 32-row tiles of X ON ITS OWN -- its private 12 KiB of LDS takes the tile's
// three K-tiles by LDS-direct loads, 144 MFMAs contract them against the shared panel, and the 32 x 192 results leave straight from the
// accumulators through the fused tails of gemm_common.h (bias | bias + GELU with both tensors | x GELU'(f1) | window-reverse + DropPath +
// residual), eight columns per lane after one exchange with the neighbouring lane group.  No barrier after the panel has landed: while one
// wave waits for its next tile (requested BEFORE its read-out) five others compute or store, which is what keeps enough bytes in flight.
//   LDS image of a K-tile: rows of 128 B, 16-byte chunk p of row r holds logical chunk p ^ ((r >> 1) & 7) (gemm_nt's layout and fragment reads)
//   accumulator (i, j)[r] = out[m0 + 16 i + c][n0 + 16 j + 4 g + r]   (lane = (g, c); B fragment first: D[n][m])
#include "gemm_common.h"

namespace {
constexpr int KW_BROWS = 192, KW_KT = 3;
constexpr int KW_BKT = KW_BROWS * 128;                                     // bytes of one K-tile image of the panel
constexpr int KW_B_BYTES = KW_KT * KW_BKT;                                 // 73 728 B; the waves' tiles take as much again: 147 456 B of LDS
// AR rows per wave tile: 32 (six waves; the plain tails: fewer panel reads per row) or 16 (twelve waves; the GELU tails: their ~12 VALU
// operations per element run beside other waves' loads and MFMAs only if there are enough other waves)
template <int AR> struct KwCfg { static constexpr int WAVES = 192 / AR, AKT = AR * 128, A_BYTES = KW_KT * AKT, RI = AR / 16, RQ = AR / 8; };
struct KwParams {
    GemmP g;
    int ntn, row_tiles, wg_per_xcd;
};

template <int MC, int AR>      // the tail's mode compiled in (gemm_nt.hip: MC); 6 = mode 3 with an fp32 residual stream
__global__ __launch_bounds__(KwCfg<AR>::WAVES * 64, 1) void gemm_k192_kernel(KwParams K) {
    constexpr int KW_WAVES = KwCfg<AR>::WAVES, KW_AKT = KwCfg<AR>::AKT, KW_A_BYTES = KwCfg<AR>::A_BYTES, KW_AROWS = AR, RI = KwCfg<AR>::RI,
                  RQ = KwCfg<AR>::RQ;
    GemmP& P = K.g;
    if constexpr (MC == 6) { P.mode = 3; P.res_dtype = DGX_F32; }
    else if constexpr (MC == 3) { P.mode = 3; P.res_dtype = DGX_BF16; }
    else P.mode = MC;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = l & 15, g = l >> 4;
    // workgroups of one XCD (blockIdx % 8: one L2) share row tiles across the column panels: slot s of the XCD takes panel s % ntn and is
    // row group s / ntn of it there; panel k owns groups_k row groups over the chip and strides its row tiles over their waves
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int panel = slot % K.ntn;
    const int full = K.wg_per_xcd / K.ntn, extra = K.wg_per_xcd - full * K.ntn;
    const int groups = 8 * (full + (panel < extra ? 1 : 0));
    const int wave_rank = ((slot / K.ntn) * 8 + xcd) * KW_WAVES + w, wave_count = groups * KW_WAVES;
    const int n0 = panel * KW_BROWS;

    const int rsub = l >> 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;
    // ---- the panel: 3 K-tiles x 24 groups of 8 rows, one LDS-direct load instruction each, dealt over the six waves
    {
        const u32x4 rB = g_rsrc(P.B, (uint32_t)((int64_t)P.N * P.ldb * 2));
        for (int idx = w; idx < KW_KT * 24; idx += KW_WAVES) {
            const int kt = idx / 24, q = idx - kt * 24;
            const int lc = (l & 7) ^ (((q & 1) << 2) | (rsub >> 1));
            const int n = n0 + 8 * q + rsub;
            const uint32_t voff = (uint32_t)(((int64_t)n * P.ldb + lc * 8) * 2);
            g_load_lds16(voff, rB, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(kt * KW_BKT + q * 1024)), (uint32_t)kt * 128u);
        }
    }
    const uint32_t abase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(KW_B_BYTES + w * KW_A_BYTES));
    const u32x4 rA = g_rsrc(P.A, (uint32_t)((int64_t)P.M * P.lda * 2));
    uint32_t lcq[RQ];                              // logical chunk of this lane in row group q of a tile
#pragma unroll
    for (int q = 0; q < RQ; ++q) lcq[q] = (uint32_t)(((l & 7) ^ (((q & 1) << 2) | (rsub >> 1))) * 16);
    auto issue_a = [&](int tile) {
        const int m0 = tile * KW_AROWS;
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int m = m0 + 8 * q + rsub;
            const uint32_t voff = m < P.M ? (uint32_t)((int64_t)m * P.lda * 2) + lcq[q] : G_OOB;
#pragma unroll
            for (int kt = 0; kt < KW_KT; ++kt) g_load_lds16(voff, rA, abase + (uint32_t)(kt * KW_AKT + q * 1024), (uint32_t)kt * 128u);
        }
    };
    // this lane's bias entries: columns n0 + 16 j + 4 g .. + 3
    u32x2 braw[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) braw[j] = P.bias ? *reinterpret_cast<const u32x2*>(P.bias + n0 + 16 * j + 4 * g) : u32x2{0u, 0u};
    int tile = wave_rank;
    if (tile < K.row_tiles) issue_a(tile);
    g_vmcnt<0>();
    __syncthreads();                               // the panel (and every wave's first tile) is in LDS

    const int swz = (c >> 1) & 7;
    const uint32_t fa = (uint32_t)(c * 128 + ((g ^ swz) << 4));
    DGX_LDS const unsigned char* pa = lds_opaque((const unsigned char*)lds_raw + KW_B_BYTES + w * KW_A_BYTES + fa);
    DGX_LDS const unsigned char* pa1 = lds_opaque((const unsigned char*)lds_raw + KW_B_BYTES + w * KW_A_BYTES + (fa ^ 64u));
    DGX_LDS const unsigned char* pb = lds_opaque((const unsigned char*)lds_raw + fa);
    DGX_LDS const unsigned char* pb1 = lds_opaque((const unsigned char*)lds_raw + (fa ^ 64u));
    const bool even = (g & 1) == 0;

    // this lane's chunk columns: even lane groups finish the chunk of column tile 2p (their piece + the next group's), odd ones that of 2p + 1
    int gn[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) gn[p] = n0 + 16 * (2 * p + (even ? 0 : 1)) + 4 * (even ? g : g - 1);
    for (; tile < K.row_tiles; tile += wave_count) {
        // GELU': the saved pre-activation chunks of the tile are requested in front of the MFMAs (their latency under the contraction)
        u32x4 xpre[MC == 4 ? RI : 1][6];
        if constexpr (MC == 4) {
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                const int gm = tile * KW_AROWS + 16 * i + c;
#pragma unroll
                for (int p = 0; p < 6; ++p) xpre[i][p] = *reinterpret_cast<const u32x4*>(P.aux + (int64_t)(gm < P.M ? gm : 0) * P.ldaux + gn[p]);
            }
        }
        f32x4 acc[RI][12];
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < 12; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KW_KT; ++kt)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                DGX_LDS const unsigned char* a_ = (kh ? pa1 : pa) + kt * KW_AKT;
                DGX_LDS const unsigned char* b_ = (kh ? pb1 : pb) + kt * KW_BKT;
                bf16x8 af[RI];
#pragma unroll
                for (int i = 0; i < RI; ++i) af[i] = *reinterpret_cast<DGX_LDS const bf16x8*>(a_ + 2048 * i);
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const bf16x8 b = *reinterpret_cast<DGX_LDS const bf16x8*>(b_ + 2048 * j);
#pragma unroll
                    for (int i = 0; i < RI; ++i) acc[i][j] = mfma16(b, af[i], acc[i][j]);
                }
            }
        // the tile's images have been read: the next tile's operands fly under this tile's read-out
        const int m0 = tile * KW_AROWS;
        const int next = tile + wave_count;
        g_lgkm0();
        if (next < K.row_tiles) issue_a(next);
        // ---- read-out, one 16-row half at a time: bias, bf16, 8-column chunks (lane groups g and g ^ 1 exchange one 4-column piece per
        // column-tile pair), the tail's operands of the half requested together, then the tails
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const int gm = m0 + 16 * i + c;
            int64_t tok = 0;
            float sc = 1.0f;
            bool ok = gm < P.M;
            if (P.mode == 3 && ok) {
                int b = 0;
                tok = g_row_token(P.map, gm, b);
                ok = tok >= 0;
                if (ok && P.scale) sc = P.scale[b];
            }
            u32x4 y[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                u32x2 pk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * p + h;
                    const f32x4 a = acc[i][j];
                    const float b0 = __uint_as_float(braw[j][0] << 16), b1 = __uint_as_float(braw[j][0] & 0xffff0000u);
                    const float b2 = __uint_as_float(braw[j][1] << 16), b3 = __uint_as_float(braw[j][1] & 0xffff0000u);
                    pk[h] = u32x2{pack_bf2(a[0] + b0, a[1] + b1), pack_bf2(a[2] + b2, a[3] + b3)};
                }
                const u32x2 give = even ? pk[1] : pk[0];
                const u32x2 got = u32x2{(uint32_t)__shfl_xor((int)give[0], 16), (uint32_t)__shfl_xor((int)give[1], 16)};
                y[p] = even ? u32x4{pk[0][0], pk[0][1], got[0], got[1]} : u32x4{got[0], got[1], pk[1][0], pk[1][1]};
            }
            u32x4 xa[6], xb[6];
            if constexpr (MC == 4) {
#pragma unroll
                for (int p = 0; p < 6; ++p) { xa[p] = xpre[i][p]; xb[p] = xa[p]; }
            } else if (P.mode >= 3) {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    xa[p] = u32x4{0u, 0u, 0u, 0u}; xb[p] = xa[p];
                    g_epi_prefetch(P, ok ? gm : 0, gn[p], ok ? tok : 0, xa[p], xb[p]);
                }
            }
            if (ok) {
#pragma unroll
                for (int p = 0; p < 6; ++p) g_epi_finish(P, gm, gn[p], y[p], tok, sc, xa[p], xb[p]);
            }
        }
        // the next tile's operands must have landed; this tile's stores, issued behind them (the counter retires in order), need not:
        // a full tile issues a known number of them
        constexpr int NST = (MC == 1 || MC == 4) ? RI * 6 : MC == 2 ? RI * 12 : -1;
        if (NST >= 0 && m0 + KW_AROWS <= P.M) g_vmcnt<(NST >= 0 ? NST : 0)>();
        else g_vmcnt<0>();
    }
}

template <int MC, int AR>
int launch_k192(KwParams& K, hipStream_t st, int grid) {
    static bool once = false;
    const size_t sm = (size_t)KW_B_BYTES + (size_t)KwCfg<AR>::WAVES * KwCfg<AR>::A_BYTES;
    K.row_tiles = (K.g.M + AR - 1) / AR;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)gemm_k192_kernel<MC, AR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        once = true;
    }
    hipLaunchKernelGGL((gemm_k192_kernel<MC, AR>), dim3(grid), dim3(KwCfg<AR>::WAVES * 64), sm, st, K);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
}  // namespace

// gemm_nt.hip's dispatch hands the K = 192 problems it takes here (bool: the shape and the tail are this kernel's)
bool dgx_gemm_k192_takes(const GemmP& P) {
    return P.K == 192 && P.lda >= 192 && P.N >= 192 && P.N % 192 == 0 && P.M >= 32768 && !P.conv_kc && P.ngrp == 0 && !P.relu && P.mode >= 0 &&
           P.mode <= 4 && !P.dbg && ((uintptr_t)P.A & 15) == 0 && ((uintptr_t)P.B & 15) == 0 && (P.lda & 7) == 0 && (P.ldb & 7) == 0;
}
int dgx_gemm_k192_launch(const GemmP& P0, hipStream_t st) {
    extern int dgx_get_reserved_cus(void);
    KwParams K;
    K.g = P0;
    K.ntn = P0.N / KW_BROWS;
    K.row_tiles = 0;                               // (set per instantiation: rows per wave tile)
    K.wg_per_xcd = 32 - dgx_get_reserved_cus() / 8;
    if (K.wg_per_xcd < K.ntn) return DGX_ERR_UNSUPPORTED;
    const int grid = 8 * K.wg_per_xcd;
    switch (P0.mode) {
        case 0: case 1: return launch_k192<1, 32>(K, st, grid);
        case 2: return launch_k192<2, 16>(K, st, grid);
        case 3: return P0.res_dtype == DGX_BF16 ? launch_k192<3, 16>(K, st, grid) : launch_k192<6, 16>(K, st, grid);
        case 4: return launch_k192<4, 16>(K, st, grid);
        default: return DGX_ERR_UNSUPPORTED;
    }
}
