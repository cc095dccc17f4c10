// bf16 GEMM with fused epilogues for gfx950, loader-wave form:  C[M][N] = epi( sum_k A[m][k] * B[n][k] )  -- the contract, the
// LDS image and the fused tails of gemm_nt.hip (every Linear / 1x1 / 3x3 forward and input gradient of the reference:
// swintransformer.py:133,155,40-46,296, fpn.py:126-154, box_head.py:26-98), with the operand traffic moved off the MFMA waves.
//
// Why: gemm_nt's 8 waves issue their own LDS-direct loads.  One `buffer_load_dwordx4 ... lds` (1 KiB) holds its wave for ~60-70
// cycles (the CU's address path moves 64 B per clock), and a wave issues in order, so the 5-7 loads per wave and K-tile sit on the
// critical path of the two-group schedule: measured 2 189 / 1 740 / 1 291 cycles per K-tile for 256x192 / 192x192 / 128x192 tiles
// against 1 536 / 1 152 / 768 of MFMA issue, and exactly  T = (MFMA_group + loads) + max(MFMA_group, reads + loads) + 2 barriers.
// Here a workgroup has 12 waves (3 per SIMD):
//   * waves 8-11 are LOADERS: each owns a quarter of the row groups (8 rows x 128 B per instruction) of both operand tiles, keeps
//     the ring of K-tiles full and does nothing else in the main loop; waits are counted per loader (`vmcnt`) and published by the
//     workgroup barrier every wave passes, exactly as before;
//   * waves 0-7 are MFMA waves (2 (M) x 4 (N), two groups one phase apart): read phase = all B fragments + the first A fragments of
//     the K-tile, MFMA phase = 2*WMF*WNF MFMAs with the remaining A fragments read from LDS just in time (three 16-row fragment
//     pairs in flight), so a wave needs <= 168 registers and three waves fit a SIMD; they never execute a vector-memory instruction
//     in the main loop;
//   * the A and B rings have their own depths (256x192: 3 x 32 KB + 2 x 24 KB = 144 KB): the activations (cold, every column tile
//     of an XCD misses on them together) get 1.5 K-tile periods of flight, the weights (L2 resident) one;
//   * read-out as gemm_nt.hip: the tile is staged as bf16 in LDS and streamed out as whole rows by all 768 threads with the fused
//     tail (bias | bias + exact GELU with both tensors | x GELU'(f1) | x ReLU'(act) | window-reverse + roll + crop + DropPath +
//     residual add).
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {
template <int BM, int BN, int NSA_, int NSB_> struct LwCfg {
    static constexpr int NSA = NSA_, NSB = NSB_;  // ring depths (K-tiles resident or in flight) of the A / B operand
    static_assert(NSA >= NSB && NSA - NSB <= 1 && NSB >= 2, "ring depths: NSA == NSB or NSB + 1");
    static constexpr int WMF = BM / 32;           // 16-row MFMA fragments per wave along M (2 groups)
    static constexpr int WNF = BN / 64;           // 16-column fragments per wave along N (4 waves)
    static constexpr int NLA = BM / 32;           // LDS-direct loads per LOADER wave per K-tile, A rows (BM / 8 row groups over 4 loaders)
    static constexpr int NLB = BN / 32;
    static constexpr int SA = BM * 128;           // bytes per A slot
    static constexpr int SBb = BN * 128;          //           B slot
    static constexpr int B0 = NSA * SA;           // B ring behind the A ring
    static constexpr int RING = NSA * SA + NSB * SBb;
    static constexpr int SROW = BN * 2 + 16;      // epilogue staging row stride (bytes)
    static constexpr int EPI = BM * SROW + BM * 8;
    static constexpr int LDS = (RING > EPI ? RING : EPI);
    static_assert(LDS <= 160 * 1024, "LDS per workgroup");
};
constexpr int LW_THREADS = 768;

// lgkmcnt(0) as the BUILTIN: the compiler's wait-count pass sees it, so the fragment reads issued behind the barrier are waited for
// with their own counts (behind an inline-asm wait it re-waits lgkmcnt(0) in front of the first MFMA of the phase)
__device__ __forceinline__ void lw_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }
__device__ __forceinline__ uint32_t lw_sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// s_waitcnt vmcnt(n') with the largest n' <= n out of {PRO, STEADY, STEP, 0}
template <int PRO, int STEADY, int STEP>
__device__ __forceinline__ void lw_vmcnt_le(int n) {
    if (PRO > STEADY && n >= PRO) g_vmcnt<PRO>();
    else if (STEADY > 0 && n >= STEADY) g_vmcnt<STEADY>();
    else if (STEP > 0 && STEP < STEADY && n >= STEP) g_vmcnt<STEP>();
    else g_vmcnt<0>();
}
}  // namespace

template <int BM, int BN, int NSA, int NSB>
__global__ __launch_bounds__(LW_THREADS) void gemm_lw_kernel(GemmP P) {
    using Cfg = LwCfg<BM, BN, NSA, NSB>;
    constexpr int WMF = Cfg::WMF, WNF = Cfg::WNF, NLA = Cfg::NLA, NLB = Cfg::NLB, SA = Cfg::SA, SBb = Cfg::SBb, B0 = Cfg::B0;
    constexpr int SROW = Cfg::SROW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int L0 = (blockIdx.x & 7) * P.per_xcd + (blockIdx.x >> 3);
    if (L0 >= P.total * P.splits) return;
    int L = L0 / P.splits;
    const int split = L0 - L * P.splits;
    if (P.ngrp > 0) {                              // grouped convolution: this tile's image (gemm_nt.hip)
        L = __builtin_amdgcn_readfirstlane(L);
        dgxgemm::GemmP::Grp q = P.grp[0];
#pragma unroll
        for (int k = 1; k < dgxgemm::GEMM_MAXG; ++k)
            if (k < P.ngrp && L >= P.grp[k].tile0) q = P.grp[k];
        P.A = q.A; P.C = q.C; P.M = q.M;
        P.cmap_n = q.cn; P.cmap_h = q.ch; P.cmap_w = q.cw; P.conv_wp = q.wp;
        L -= q.tile0;
    }
#define LCLK(i) do { if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    LCLK(0);
    const int tm = L / P.tiles_n, tn = L - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = (w >> 2) & 1, wc = w & 3;      // MFMA waves: group = M half of the tile
    const int c = l & 15, g = l >> 4;
    const int NTK = (P.K + GBK - 1) / GBK;
    const int ktail = P.K - (NTK - 1) * GBK;
    const int kt0 = split * P.kt_per_split;
    const int NT = min(P.kt_per_split, NTK - kt0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;
    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Barrier numbering (every wave executes #0 .. #(2 NT + 1)); I_k = the interval that barrier #k closes.  MFMA group 0 reads
    // tile t (B fragments, first A fragments) in I_{2t+1} and multiplies (+ streams the other A fragments) in I_{2t+2}; group 1 one
    // interval later.  So the B slot of tile t is free behind #(2t+2) and its A slot behind #(2t+3); tile t must be visible at
    // #(2t).  Loaders: I_{2u+3} issue B(u + NSB); I_{2u+4} issue A(u + NSA), wait for tile u + 2, barrier.
    if (w >= 8) {
        // ---------------------------------------------------------------- loader waves
        const int lw = w - 8;
        const int rsub = l >> 3;
        const int lc = (l & 7) ^ (((lw & 1) << 2) | (rsub >> 1));     // row group q = lw + 4 s: (q & 1) == (lw & 1)
        uint32_t voffA[NLA], voffB[NLB];
#pragma unroll
        for (int s = 0; s < NLA; ++s) {
            const int m = m0 + 8 * (lw + 4 * s) + rsub;
            int64_t arow = m;
            if (P.conv_kc) {                       // output pixel (n, y, x) -> its position in the zero-bordered image
                const int hw = P.cmap_h * P.cmap_w;
                const int n = m / hw, r = m - n * hw;
                const int y = r / P.cmap_w, x = r - y * P.cmap_w;
                arow = ((int64_t)n * (P.cmap_h + 2) + y + 1) * P.conv_wp + x + 1;
            }
            voffA[s] = m < P.M ? (uint32_t)((arow * P.lda + lc * 8) * 2) : G_OOB;
        }
#pragma unroll
        for (int s = 0; s < NLB; ++s) {
            const int n = n0 + 8 * (lw + 4 * s) + rsub;
            voffB[s] = n < P.N ? (uint32_t)(((int64_t)n * P.ldb + lc * 8) * 2) : G_OOB;
        }
        const bool kt_ok = lc * 8 < ktail;
        const u32x4 rA = g_rsrc(P.A, (uint32_t)((P.conv_kc ? (int64_t)P.cmap_n * (P.cmap_h + 2) * P.conv_wp + 2 * P.conv_wp + 2 : (int64_t)P.M) * P.lda * 2));
        const u32x4 rB = g_rsrc(P.B, (uint32_t)((int64_t)P.N * P.ldb * 2));
        const uint32_t ldsq = lw_sgpr(lds0 + 1024u * lw);
        auto issue_a = [&](int t) {
            if (t >= NT) return;
            const int kta = kt0 + t;
            uint32_t soffA = (uint32_t)kta * (GBK * 2);
            if (P.conv_kc) {
                const int tap = kta / P.conv_kc, kc = kta - tap * P.conv_kc;
                soffA = (uint32_t)((tap / 3) * P.conv_wp + tap % 3) * (uint32_t)(P.lda * 2) + (uint32_t)kc * (GBK * 2);
            }
            const bool tail = (kta == NTK - 1) && (ktail != GBK) && !kt_ok;
            const uint32_t dst = ldsq + (uint32_t)(t % NSA) * SA;
#pragma unroll
            for (int s = 0; s < NLA; ++s) g_load_lds16(tail ? G_OOB : voffA[s], rA, dst + 4096u * s, lw_sgpr(soffA));
        };
        auto issue_b = [&](int t) {
            if (t >= NT) return;
            const int kta = kt0 + t;
            const uint32_t soff = (uint32_t)kta * (GBK * 2);
            const bool tail = (kta == NTK - 1) && (ktail != GBK) && !kt_ok;
            const uint32_t dst = ldsq + B0 + (uint32_t)(t % NSB) * SBb;
#pragma unroll
            for (int s = 0; s < NLB; ++s) g_load_lds16(tail ? G_OOB : voffB[s], rB, dst + 4096u * s, lw_sgpr(soff));
        };
        // loads of this wave that are younger than the later one of A(tau), B(tau) when tiles up to A(ia), B(ib) have been issued
        constexpr int D = NSA - NSB > 1 ? NSA - NSB : 1;
        constexpr int STEADY = (NSA > NSB ? NSB - 1 : NSA - 2) * NLA + (NSB - 2) * NLB;
        constexpr int PRO = (NSA - 1) * NLA + (NSB - 1) * NLB;      // behind tile 0 in the prologue
        static_assert(PRO < 64, "vmcnt is a 6-bit counter");
        auto wait_tile = [&](int tau, int ia, int ib) {
            const int la = min(ia, NT - 1), lb = min(ib, NT - 1);
            const int ca = max(0, la - (tau + D) + 1), cb = max(0, lb - (tau + 1) + 1);
            lw_vmcnt_le<PRO, STEADY, NLA>(ca * NLA + cb * NLB);
        };
        // prologue: tiles 0 .. NS-1 in the order the schedule keeps afterwards (NSA > NSB: A(k), B(k); else B(k), A(k))
#pragma unroll
        for (int k = 0; k < NSA; ++k) {
            if (NSA > NSB) { issue_a(k); if (k < NSB) issue_b(k); }
            else { issue_b(k); issue_a(k); }
        }
        wait_tile(0, NSA - 1, NSB - 1);
        g_bar();                                   // #0
        LCLK(1);
        for (int t = 0; t < NT; ++t) {
            if (t >= 1) issue_b(t - 1 + NSB);      // I_{2t+1}
            g_bar();                               // #(2t+1)
            if (t >= 1) issue_a(t - 1 + NSA);      // I_{2t+2}
            if (t + 1 < NT) wait_tile(t + 1, t - 1 + NSA, t - 1 + NSB);
            g_bar();                               // #(2t+2)
        }
        g_vmcnt<0>();
        g_bar();                                   // #(2 NT + 1)
    } else {
        // ---------------------------------------------------------------- MFMA waves
        // fragment (16 rows, k-half kh) = one ds_read_b128 per lane: row c, logical chunk g + 4 kh -> physical (g ^ swz) ^ 4 kh
        const int swz = (c >> 1) & 7;
        const uint32_t la = (uint32_t)((grp * (BM / 2) + c) * 128 + ((g ^ swz) << 4));
        const uint32_t lb = (uint32_t)(B0 + (wc * (BN / 4) + c) * 128 + ((g ^ swz) << 4));
        bf16x8 bfr[WNF][2];
        bf16x8 afr[WMF][2];                        // only ~3 pairs are live at a time (streamed)
        constexpr int PRE = WMF < 2 ? WMF : 2;     // A fragment pairs read in the read phase
        auto read_phase = [&](int t) {
            DGX_LDS const unsigned char* sb = lds_opaque((const unsigned char*)lds_raw + (t % NSB) * SBb + lb);
            DGX_LDS const unsigned char* sb1 = lds_opaque((const unsigned char*)lds_raw + (t % NSB) * SBb + (lb ^ 64u));
            DGX_LDS const unsigned char* sa = lds_opaque((const unsigned char*)lds_raw + (t % NSA) * SA + la);
            DGX_LDS const unsigned char* sa1 = lds_opaque((const unsigned char*)lds_raw + (t % NSA) * SA + (la ^ 64u));
#pragma unroll
            for (int j = 0; j < WNF; ++j) {
                bfr[j][0] = *reinterpret_cast<DGX_LDS const bf16x8*>(sb + 2048 * j);
                bfr[j][1] = *reinterpret_cast<DGX_LDS const bf16x8*>(sb1 + 2048 * j);
            }
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                afr[i][0] = *reinterpret_cast<DGX_LDS const bf16x8*>(sa + 2048 * i);
                afr[i][1] = *reinterpret_cast<DGX_LDS const bf16x8*>(sa1 + 2048 * i);
            }
        };
        auto mfma_phase = [&](int t) {
            DGX_LDS const unsigned char* sa = lds_opaque((const unsigned char*)lds_raw + (t % NSA) * SA + la);
            DGX_LDS const unsigned char* sa1 = lds_opaque((const unsigned char*)lds_raw + (t % NSA) * SA + (la ^ 64u));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < WMF; ++i) {
                if (i + PRE < WMF) {               // the pair that is consumed PRE steps from now
                    afr[i + PRE][0] = *reinterpret_cast<DGX_LDS const bf16x8*>(sa + 2048 * (i + PRE));
                    afr[i + PRE][1] = *reinterpret_cast<DGX_LDS const bf16x8*>(sa1 + 2048 * (i + PRE));
                }
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int j = 0; j < WNF; ++j) acc[i][j] = mfma16(bfr[j][kh], afr[i][kh], acc[i][j]);   // D[n][m]: lane = 4 columns of a row
                // pin the order {2 fragment reads, 2 WNF MFMAs} per step: the scheduler would otherwise hoist every read to the top
                if (i + PRE < WMF) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * WNF, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        g_bar();                                   // #0
        LCLK(1);
        if (grp == 1) g_bar();                     // #1: one phase behind group 0
        for (int t = 0; t < NT; ++t) {
            read_phase(t);
            lw_lgkm0();
            g_bar();                               // group 0: #(2t+1), group 1: #(2t+2)
            mfma_phase(t);
            lw_lgkm0();
            g_bar();                               // group 0: #(2t+2), group 1: #(2t+3)
        }
        if (grp == 0) g_bar();                     // #(2 NT + 1)
    }
    LCLK(2);

    // ---- split-K: the raw fp32 accumulators go to this split's slab; gemm_nt's fold kernel finishes the job
    if (P.splits > 1) {
        if (w < 8) {
            float* slab = P.ws + (int64_t)split * P.M * P.N;
            const int colw = n0 + wc * (BN / 4) + 4 * g;
#pragma unroll
            for (int i = 0; i < WMF; ++i) {
                const int gm = m0 + grp * (BM / 2) + 16 * i + c;
#pragma unroll
                for (int j = 0; j < WNF; ++j) {
                    const int gn = colw + 16 * j;
                    if (gm < P.M && gn < P.N) *reinterpret_cast<f32x4*>(slab + (int64_t)gm * P.N + gn) = acc[i][j];
                }
            }
        }
        return;
    }
    // ---- epilogue (gemm_nt.hip): stage the tile as bf16 (bias added) in LDS, stream whole rows out with the fused tail; all 12
    // waves read out.  The tail's extra operands are requested PFN chunks per lane ahead of their use.
    DGX_LDS unsigned char* stg = (DGX_LDS unsigned char*)lds_raw;
    DGX_LDS int64_t* rowtok = reinterpret_cast<DGX_LDS int64_t*>(stg + BM * SROW);
    constexpr int CPR = BN / 8;                    // 16-byte chunks per tile row
    constexpr int NCH = BM * CPR;
    constexpr int ITERS = (NCH + LW_THREADS - 1) / LW_THREADS;
    constexpr int PFN = ITERS % 4 == 0 ? 4 : (ITERS % 3 == 0 ? 3 : (ITERS % 2 == 0 ? 2 : 1));
    if (P.mode == 3) {
        if (tid < BM) {
            int b = 0;
            const int64_t orow = (int64_t)m0 + tid;
            const int64_t tok = orow < P.M ? g_row_token(P.map, orow, b) : -1;
            rowtok[tid] = tok < 0 ? -1 : ((tok << 12) | (int64_t)b);
        }
        __syncthreads();
    }
    u32x4 xa[PFN], xb[PFN];
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    auto locate = [&](int it, int& row, int& ch, int& gm, int& gn, int64_t& tok, float& sc) -> bool {
        const int idx = tid_e + it * LW_THREADS;
        row = idx / CPR;
        ch = idx - row * CPR;
        gm = m0 + row;
        gn = n0 + 8 * ch;
        tok = 0;
        sc = 1.0f;
        bool ok = idx < NCH && gm < P.M && gn < P.N;
        if (P.mode == 3 && ok) {
            const int64_t rt = rowtok[row];
            ok = rt >= 0;
            tok = rt >> 12;
            if (ok && P.scale) sc = P.scale[(int)(rt & 4095)];
        }
        return ok;
    };
#define LW_EPI_PREFETCH(IT0)                                                                                                      \
    if (P.mode == 4 || P.mode == 5) {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < PFN; ++it) {                                                                      \
            const int idx = tid_e + ((IT0) + it) * LW_THREADS, row = idx / CPR, gm = m0 + row, gn = n0 + 8 * (idx - row * CPR);   \
            const int64_t off = (idx < NCH && gm < P.M && gn < P.N) ? (int64_t)gm * P.ldaux + gn : 0;                             \
            xa[it] = *reinterpret_cast<const u32x4*>(P.aux + off);                                                                \
        }                                                                                                                         \
    } else if (P.mode == 3) {                                                                                                     \
        if (P.res_dtype == DGX_BF16) {                                                                                            \
            _Pragma("unroll") for (int it = 0; it < PFN; ++it) {                                                                  \
                const int idx = tid_e + ((IT0) + it) * LW_THREADS, row = idx / CPR, gm = m0 + row, gn = n0 + 8 * (idx - row * CPR); \
                const int64_t rt = idx < NCH ? rowtok[row] : -1;                                                                  \
                const int64_t off = (gm < P.M && gn < P.N && rt >= 0) ? (rt >> 12) * P.N + gn : 0;                                \
                xa[it] = *reinterpret_cast<const u32x4*>((const uint16_t*)P.res + off);                                           \
            }                                                                                                                     \
        } else {                                                                                                                  \
            _Pragma("unroll") for (int it = 0; it < PFN; ++it) {                                                                  \
                const int idx = tid_e + ((IT0) + it) * LW_THREADS, row = idx / CPR, gm = m0 + row, gn = n0 + 8 * (idx - row * CPR); \
                const int64_t rt = idx < NCH ? rowtok[row] : -1;                                                                  \
                const int64_t off = (gm < P.M && gn < P.N && rt >= 0) ? (rt >> 12) * P.N + gn : 0;                                \
                xa[it] = reinterpret_cast<const u32x4*>((const float*)P.res + off)[0];                                            \
                xb[it] = reinterpret_cast<const u32x4*>((const float*)P.res + off)[1];                                            \
            }                                                                                                                     \
        }                                                                                                                         \
    }
    LW_EPI_PREFETCH(0)
    if (w < 8) {
        const int colw = wc * (BN / 4) + 4 * g;    // + 16 j: this lane's 4 consecutive columns
        float bv[WNF][4];
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
            const int n = n0 + colw + 16 * j;
            uint32_t b01 = 0, b23 = 0;
            if (P.bias && n < P.N) {
                const u32x2 raw = *reinterpret_cast<const u32x2*>(P.bias + n);
                b01 = raw[0];
                b23 = raw[1];
            }
            bv[j][0] = __uint_as_float(b01 << 16); bv[j][1] = __uint_as_float(b01 & 0xffff0000u);
            bv[j][2] = __uint_as_float(b23 << 16); bv[j][3] = __uint_as_float(b23 & 0xffff0000u);
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i) {
            const int row = grp * (BM / 2) + 16 * i + c;
#pragma unroll
            for (int j = 0; j < WNF; ++j) {
                const f32x4 a = acc[i][j];
                const u32x2 pk = {pack_bf2(a[0] + bv[j][0], a[1] + bv[j][1]), pack_bf2(a[2] + bv[j][2], a[3] + bv[j][3])};
                *reinterpret_cast<DGX_LDS u32x2*>(stg + row * SROW + (colw + 16 * j) * 2) = pk;
            }
        }
    }
    __syncthreads();
    LCLK(3);
#pragma unroll
    for (int p = 0; p < ITERS / PFN; ++p) {
        if (p > 0) { LW_EPI_PREFETCH(p * PFN) }
#pragma unroll
        for (int it = 0; it < PFN; ++it) {
            int row, ch, gm, gn;
            int64_t tok;
            float sc;
            if (locate(p * PFN + it, row, ch, gm, gn, tok, sc))
                g_epi_finish(P, gm, gn, *reinterpret_cast<DGX_LDS const u32x4*>(stg + row * SROW + ch * 16), tok, sc, xa[it], xb[it]);
        }
    }
#undef LW_EPI_PREFETCH
    LCLK(4);
}

namespace {
template <int BM, int BN, int NSA, int NSB>
int lw_launch_t(GemmP& P, hipStream_t st) {
    using Cfg = LwCfg<BM, BN, NSA, NSB>;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)gemm_lw_kernel<BM, BN, NSA, NSB>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    hipLaunchKernelGGL((gemm_lw_kernel<BM, BN, NSA, NSB>), dim3(8 * P.per_xcd), dim3(LW_THREADS), Cfg::LDS, st, P);
    return DGX_OK;
}
}  // namespace

// Launch the loader-wave kernel for a descriptor whose tiling fields (tiles_n, total, splits, kt_per_split, per_xcd; grp[].tile0 for
// the grouped form) the caller has filled for a bm x bn tile.  Returns DGX_ERR_UNSUPPORTED for a tile that is not instantiated.
int gemm_lw_launch(GemmP& P, int bm, int bn, hipStream_t st) {
    if (bm == 256 && bn == 192) return lw_launch_t<256, 192, 3, 2>(P, st);
    if (bm == 192 && bn == 192) return lw_launch_t<192, 192, 3, 3>(P, st);
    if (bm == 128 && bn == 192) return lw_launch_t<128, 192, 4, 4>(P, st);
    if (bm == 192 && bn == 256) return lw_launch_t<192, 256, 3, 2>(P, st);
    if (bm == 128 && bn == 256) return lw_launch_t<128, 256, 3, 3>(P, st);
    if (bm == 256 && bn == 128) return lw_launch_t<256, 128, 3, 3>(P, st);
    if (bm == 128 && bn == 128) return lw_launch_t<128, 128, 4, 4>(P, st);
    return DGX_ERR_UNSUPPORTED;
}
