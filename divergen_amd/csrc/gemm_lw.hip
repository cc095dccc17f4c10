// bf16 GEMM with fused epilogues for gfx950, loader-wave persistent form:  C[M][N] = epi( sum_k A[m][k] * B[n][k] )  -- the
// contract, the LDS image and the fused tails of gemm_nt.hip (every Linear / 1x1 / 3x3 forward and input gradient of the reference:
// swintransformer.py:133,155,40-46,296, fpn.py:126-154, box_head.py:26-98), with the operand traffic moved off the MFMA waves and
// the workgroup walking a list of output tiles.
//
// Why: gemm_nt's 8 waves issue their own LDS-direct loads.  One `buffer_load_dwordx4 ... lds` (1 KiB) holds its wave for ~60-70
// cycles (the CU's address path moves 64 B per clock), and a wave issues in order, so the 5-7 loads per wave and K-tile sit on the
// critical path of the two-group schedule: measured 2 189 / 1 740 / 1 291 cycles per K-tile for 256x192 / 192x192 / 128x192 tiles
// against 1 536 / 1 152 / 768 of MFMA issue, and exactly  T = (MFMA_group + loads) + max(MFMA_group, reads + loads) + 2 barriers.
// And a tile's first loads (a chip-wide burst: every workgroup starts together) cost 4-6 k cycles before the first MFMA.
// Here a workgroup has 12 waves (3 per SIMD) and one workgroup per CU walks tiles  L, L + W, L + 2 W, ...:
//   * waves 8-11 are LOADERS: each owns a quarter of the row groups (8 rows x 128 B per instruction) of both operand tiles, keeps
//     the rings of K-tiles full and does nothing else; waits are counted per loader (`vmcnt`) and published by the workgroup
//     barrier every wave passes.  When a tile's last K-tile has been read they issue the NEXT tile's first K-tiles straight away:
//     those land while the MFMA waves read the finished tile out, so only a workgroup's first tile pays a prologue;
//   * waves 0-7 are MFMA waves (2 (M) x 4 (N), two groups one phase apart): read phase = all B fragments + the first A fragments of
//     the K-tile, MFMA phase = 2*WMF*WNF MFMAs with the remaining A fragments read from LDS just in time, so a wave needs <= 168
//     registers and three waves fit a SIMD; they never execute a vector-memory instruction in the main loop (1 772 / 1 364 / 1 101
//     cycles per K-tile measured, round 4);
//   * the A and B rings have their own depths (256x192: 3 x 32 KB + 2 x 24 KB = 144 KB): the activations (cold, every column tile
//     of an XCD misses on them together) get 1.5 K-tile periods of flight, the weights (L2 resident) one;
//   * read-out: the LAST A slot plus the spare LDS behind the rings is the staging area (the next tile's prefetch fills the other
//     slots meanwhile): the MFMA waves pass the tile through it in slabs of CH rows (bf16, bias added) and stream whole rows out
//     with the fused tail (bias | bias + exact GELU with both tensors | x GELU'(f1) | x ReLU'(act) | window-reverse + roll + crop +
//     DropPath + residual add) of gemm_common.h; stores are not waited for, they drain under the next tile's main loop.
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {
constexpr int LW_LDS = 160 * 1024;
template <int BM, int BN, int NSA_, int NSB_> struct LwCfg {
    static constexpr int NSA = NSA_, NSB = NSB_;  // ring depths (K-tiles resident or in flight) of the A / B operand
    static_assert(NSA >= NSB && NSA - NSB <= 1 && NSB >= 2, "ring depths: NSA == NSB or NSB + 1");
    static constexpr int WMF = BM / 32;           // 16-row MFMA fragments per wave along M (2 groups)
    static constexpr int WNF = BN / 64;           // 16-column fragments per wave along N (4 waves)
    static constexpr int NLA = BM / 32;           // LDS-direct loads per LOADER wave per K-tile, A rows (BM / 8 row groups over 4 loaders)
    static constexpr int NLB = BN / 32;
    static constexpr int SA = BM * 128;           // bytes per A slot
    static constexpr int SBb = BN * 128;          //           B slot
    // LDS map: A slots 0 .. NSA-2 | B slots | A slot NSA-1 | spare.  Staging = [STG0, LW_LDS - BM * 8), row table behind it.
    static constexpr int B0 = (NSA - 1) * SA;
    static constexpr int STG0 = B0 + NSB * SBb;
    static constexpr int RING = STG0 + SA;
    static_assert(RING <= LW_LDS, "rings do not fit");
    static constexpr int SROW = BN * 2 + 16;      // staging row stride (bytes)
    static constexpr int TOK0 = LW_LDS - BM * 8;  // mode 3: (token << 32 | DropPath factor bits) per tile row
    static constexpr int STGB = TOK0 - STG0;
    static constexpr int HALF = BM / 2;           // rows per MFMA group
    // a slab = HC rows of EACH group (so that every MFMA wave hands over the same share of its accumulators per pass): the largest
    // multiple of 16 that divides HALF and fits the staging area twice
    static constexpr int pick_hc() {
        for (int hc = HALF; hc >= 16; hc -= 16)
            if (HALF % hc == 0 && 2 * hc * SROW <= STGB) return hc;
        return 0;
    }
    static constexpr int HC = pick_hc();
    static_assert(HC >= 16, "staging area too small");
    static constexpr int CH = 2 * HC;             // rows per slab
    static constexpr int NPASS = HALF / HC;
    static constexpr int FPP = HC / 16;           // fragment rows per wave and pass
};
constexpr int LW_THREADS = 768, LW_MFMA_THREADS = 512;

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
// one work item of the launch: output tile + K range
struct LwItem { int m0, n0, kt0, NT, split; };
}  // namespace

// MC: the fused tail, fixed per instantiation so that each pays only for the registers ITS tail needs -- 0: plain | bias (+ ReLU),
// 2: bias + GELU (both tensors), 3: bf16 residual, 6: fp32 residual, 4: x GELU'(f1), 5: x ReLU'(act).  EC = extra 16-byte operands
// per chunk that the tail reads (0, 1 or 2).
template <int BM, int BN, int NSA, int NSB, int MC>
__global__ __launch_bounds__(LW_THREADS) void gemm_lw_kernel(GemmP P0) {
    constexpr int EC = MC == 6 ? 2 : (MC == 3 || MC == 4 || MC == 5) ? 1 : 0;
    if constexpr (MC == 0) { if (P0.mode > 1) __builtin_unreachable(); }
    else if constexpr (MC == 6) { P0.mode = 3; P0.res_dtype = DGX_F32; }
    else if constexpr (MC == 3) { P0.mode = 3; P0.res_dtype = DGX_BF16; }
    else P0.mode = MC;
    using Cfg = LwCfg<BM, BN, NSA, NSB>;
    constexpr int WMF = Cfg::WMF, WNF = Cfg::WNF, NLA = Cfg::NLA, NLB = Cfg::NLB, SA = Cfg::SA, SBb = Cfg::SBb, B0 = Cfg::B0;
    constexpr int SROW = Cfg::SROW, STG0 = Cfg::STG0, CH = Cfg::CH, HC = Cfg::HC, NPASS = Cfg::NPASS, FPP = Cfg::FPP, HALF = Cfg::HALF;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    // work list of this workgroup: items  first, first + stride, ... < bound  (items of an XCD are consecutive tiles: one A row panel)
    const int nwx = (int)(gridDim.x >> 3);
    const int xcd = blockIdx.x & 7;
    const int items = P0.total * P0.splits;
    const int bound = min((xcd + 1) * P0.per_xcd, items);
    const int first = xcd * P0.per_xcd + (int)(blockIdx.x >> 3);
    if (first >= bound) return;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = (w >> 2) & 1, wc = w & 3;      // MFMA waves: group = M half of the tile
    const int c = l & 15, g = l >> 4;
    const int NTK = (P0.K + GBK - 1) / GBK;
    const int ktail = P0.K - (NTK - 1) * GBK;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;
#define LCLK(i) do { if (P0.dbg && threadIdx.x == 0 && L0 == first) P0.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)

    // item L0 -> tile origin, K range and (grouped convolution) the image the tile belongs to
    auto locate_item = [&](int L0, GemmP& P, LwItem& it) {
        int L = L0 / P0.splits;
        it.split = L0 - L * P0.splits;
        if (P0.ngrp > 0) {
            L = __builtin_amdgcn_readfirstlane(L);
            dgxgemm::GemmP::Grp q = P0.grp[0];
#pragma unroll
            for (int k = 1; k < dgxgemm::GEMM_MAXG; ++k)
                if (k < P0.ngrp && L >= P0.grp[k].tile0) q = P0.grp[k];
            P.A = q.A; P.C = q.C; P.M = q.M;
            P.cmap_n = q.cn; P.cmap_h = q.ch; P.cmap_w = q.cw; P.conv_wp = q.wp;
            L -= q.tile0;
        }
        const int tm = L / P0.tiles_n, tn = L - tm * P0.tiles_n;
        it.m0 = tm * BM;
        it.n0 = tn * BN;
        it.kt0 = it.split * P0.kt_per_split;
        it.NT = min(P0.kt_per_split, NTK - it.kt0);
    };
    // A slot s of the LDS map
    auto a_slot = [](int s) -> uint32_t { return s == NSA - 1 ? (uint32_t)STG0 : (uint32_t)s * SA; };

    // Barrier numbering per item (every wave executes #0 .. #(2 NT + 1), then the read-out's); I_k = the interval barrier #k closes.
    // MFMA group 0 reads tile t (B fragments, first A fragments) in I_{2t+1} and multiplies (+ streams the other A fragments) in
    // I_{2t+2}; group 1 one interval later.  So the B slot of tile t is free behind #(2t+2) and its A slot behind #(2t+3); tile t must be
    // visible at #(2t).  Loaders: I_{2u+3} issue B(u + NSB); I_{2u+4} issue A(u + NSA), wait for tile u + 2, barrier.
    if (w >= 8) {
        // ---------------------------------------------------------------- loader waves
        const int lw = w - 8;
        const int rsub = l >> 3;
        const int lc = (l & 7) ^ (((lw & 1) << 2) | (rsub >> 1));     // row group q = lw + 4 s: (q & 1) == (lw & 1)
        const bool kt_ok = lc * 8 < ktail;
        const uint32_t ldsq = lw_sgpr(lds0 + 1024u * lw);
        GemmP P = P0;
        LwItem it;
        uint32_t voffA[NLA], voffB[NLB];
        u32x4 rA;
        const u32x4 rB = g_rsrc(P0.B, (uint32_t)((int64_t)P0.N * P0.ldb * 2));
        auto setup = [&](int L0) {
            locate_item(L0, P, it);
#pragma unroll
            for (int s = 0; s < NLA; ++s) {
                const int m = it.m0 + 8 * (lw + 4 * s) + rsub;
                int64_t arow = m;
                if (P.conv_kc) {                   // output pixel (n, y, x) -> its position in the zero-bordered image
                    const int hw = P.cmap_h * P.cmap_w;
                    const int n = m / hw, r = m - n * hw;
                    const int y = r / P.cmap_w, x = r - y * P.cmap_w;
                    arow = ((int64_t)n * (P.cmap_h + 2) + y + 1) * P.conv_wp + x + 1;
                }
                voffA[s] = m < P.M ? (uint32_t)((arow * P.lda + lc * 8) * 2) : G_OOB;
            }
#pragma unroll
            for (int s = 0; s < NLB; ++s) {
                const int n = it.n0 + 8 * (lw + 4 * s) + rsub;
                voffB[s] = n < P.N ? (uint32_t)(((int64_t)n * P.ldb + lc * 8) * 2) : G_OOB;
            }
            const u32x4 r = g_rsrc(P.A, (uint32_t)((P.conv_kc ? (int64_t)P.cmap_n * (P.cmap_h + 2) * P.conv_wp + 2 * P.conv_wp + 2 : (int64_t)P.M) * P.lda * 2));
            rA = u32x4{lw_sgpr(r[0]), lw_sgpr(r[1]), lw_sgpr(r[2]), lw_sgpr(r[3])};
        };
        auto issue_a = [&](int t) {
            if (t >= it.NT) return;
            const int kta = it.kt0 + t;
            uint32_t soffA = (uint32_t)kta * (GBK * 2);
            if (P.conv_kc) {
                const int tap = kta / P.conv_kc, kc = kta - tap * P.conv_kc;
                soffA = (uint32_t)((tap / 3) * P.conv_wp + tap % 3) * (uint32_t)(P.lda * 2) + (uint32_t)kc * (GBK * 2);
            }
            const bool tail = (kta == NTK - 1) && (ktail != GBK) && !kt_ok;
            const uint32_t dst = ldsq + a_slot(t % NSA);
#pragma unroll
            for (int s = 0; s < NLA; ++s) g_load_lds16(tail ? G_OOB : voffA[s], rA, lw_sgpr(dst + 4096u * s), lw_sgpr(soffA));
        };
        auto issue_b = [&](int t) {
            if (t >= it.NT) return;
            const int kta = it.kt0 + t;
            const uint32_t soff = (uint32_t)kta * (GBK * 2);
            const bool tail = (kta == NTK - 1) && (ktail != GBK) && !kt_ok;
            const uint32_t dst = ldsq + B0 + (uint32_t)(t % NSB) * SBb;
#pragma unroll
            for (int s = 0; s < NLB; ++s) g_load_lds16(tail ? G_OOB : voffB[s], rB, lw_sgpr(dst + 4096u * s), lw_sgpr(soff));
        };
        // loads of this wave that are younger than the later one of A(tau), B(tau) when tiles up to A(ia), B(ib) have been issued
        constexpr int STEADY = (NSA > NSB ? NSB - 1 : NSA - 2) * NLA + (NSB - 2) * NLB;
        constexpr int PRO = (NSA - 1) * NLA + (NSB - 1) * NLB;      // behind tile 0 in the prologue
        static_assert(PRO < 64, "vmcnt is a 6-bit counter");
        auto wait_tile = [&](int tau, int ia, int ib) {
            const int la = min(ia, it.NT - 1), lb = min(ib, it.NT - 1);
            const int ca = max(0, la - (tau + 1) + 1), cb = max(0, lb - (tau + 1) + 1);
            lw_vmcnt_le<PRO, STEADY, NLA>(ca * NLA + cb * NLB);
        };
        // first part of a tile's prologue: everything but A(NSA - 1), whose slot is the staging area of the tile before it; in the
        // order the schedule keeps afterwards (NSA > NSB: A(k), B(k); else B(k), A(k))
        auto prologue_1 = [&]() {
#pragma unroll
            for (int k = 0; k < NSA - 1; ++k) {
                if (NSA > NSB) { issue_a(k); if (k < NSB) issue_b(k); }
                else { issue_b(k); issue_a(k); }
            }
            if (NSA == NSB) issue_b(NSB - 1);
        };
        setup(first);
        prologue_1();
        for (int L0 = first; L0 < bound; L0 += nwx) {
            issue_a(NSA - 1);                      // second part of the prologue: the staging area is free
            wait_tile(0, NSA - 1, NSB - 1);
            g_bar();                               // #0
            const int NT = it.NT;
            for (int t = 0; t < NT; ++t) {
                if (t >= 1) issue_b(t - 1 + NSB);  // I_{2t+1}
                g_bar();                           // #(2t+1)
                if (t >= 1) issue_a(t - 1 + NSA);  // I_{2t+2}
                if (t + 1 < NT) wait_tile(t + 1, t - 1 + NSA, t - 1 + NSB);
                g_bar();                           // #(2t+2)
            }
            g_vmcnt<0>();
            g_bar();                               // #(2 NT + 1): every slot is free
            if (L0 + nwx < bound) {                // the next tile's first K-tiles land during the read-out
                setup(L0 + nwx);
                prologue_1();
            }
            if (P0.splits == 1) {                  // the read-out's barriers (the MFMA waves do the work)
                if (P0.mode == 3) g_bar();
#pragma unroll
                for (int p = 0; p < NPASS; ++p) { g_bar(); g_bar(); }
                if constexpr ((MC == 3 || MC == 6) && Cfg::TOK0 < Cfg::RING) g_bar();      // the row table inside the last A slot has been read
            }
        }
        return;
    }

    // ---------------------------------------------------------------- MFMA waves
    // fragment (16 rows, k-half kh) = one ds_read_b128 per lane: row c, logical chunk g + 4 kh -> physical (g ^ swz) ^ 4 kh
    const int swz = (c >> 1) & 7;
    const uint32_t la = (uint32_t)((grp * (BM / 2) + c) * 128 + ((g ^ swz) << 4));
    const uint32_t lb = (uint32_t)(B0 + (wc * (BN / 4) + c) * 128 + ((g ^ swz) << 4));
    constexpr int PRE = WMF < 2 ? WMF : 2;         // A fragment pairs read in the read phase
    GemmP P = P0;
    for (int L0 = first; L0 < bound; L0 += nwx) {
        LwItem it;
        locate_item(L0, P, it);
        const int m0 = it.m0, n0 = it.n0, NT = it.NT;
        LCLK(0);
        f32x4 acc[WMF][WNF];
#pragma unroll
        for (int i = 0; i < WMF; ++i)
#pragma unroll
            for (int j = 0; j < WNF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 bfr[WNF][2];
        bf16x8 afr[WMF][2];                        // only a few pairs are live at a time (streamed)
        auto read_phase = [&](int t) {
            DGX_LDS const unsigned char* sb = lds_opaque((const unsigned char*)lds_raw + (t % NSB) * SBb + lb);
            DGX_LDS const unsigned char* sb1 = lds_opaque((const unsigned char*)lds_raw + (t % NSB) * SBb + (lb ^ 64u));
            const uint32_t ao = a_slot(t % NSA);
            DGX_LDS const unsigned char* sa = lds_opaque((const unsigned char*)lds_raw + ao + la);
            DGX_LDS const unsigned char* sa1 = lds_opaque((const unsigned char*)lds_raw + ao + (la ^ 64u));
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
            const uint32_t ao = a_slot(t % NSA);
            DGX_LDS const unsigned char* sa = lds_opaque((const unsigned char*)lds_raw + ao + la);
            DGX_LDS const unsigned char* sa1 = lds_opaque((const unsigned char*)lds_raw + ao + (la ^ 64u));
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
        LCLK(2);

        // ---- split-K: the raw fp32 accumulators go to this split's slab; gemm_nt's fold kernel finishes the job
        if (P0.splits > 1) {
            float* slab = P.ws + (int64_t)it.split * P.M * P.N;
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
            continue;
        }
        // ---- read-out in NPASS slabs of CH rows through the staging area; 512 threads stream the slab's rows out with the fused
        // tail.  The tail's extra operands of slab p + 1 are requested before slab p is consumed.
        DGX_LDS unsigned char* stg = (DGX_LDS unsigned char*)lds_raw + STG0;
        DGX_LDS int64_t* rowtok = reinterpret_cast<DGX_LDS int64_t*>((DGX_LDS unsigned char*)lds_raw + Cfg::TOK0);
        constexpr int CPR = BN / 8;                // 16-byte chunks per tile row
        constexpr int NCH = CH * CPR;              // chunks per slab
        constexpr int ITERS = (NCH + LW_MFMA_THREADS - 1) / LW_MFMA_THREADS;
        if (P.mode == 3) {
            if (tid < BM) {
                int b = 0;
                const int64_t orow = (int64_t)m0 + tid;
                const int64_t tok = orow < P.M ? g_row_token(P.map, orow, b) : -1;
                // (token, DropPath factor of its sample): the factor is fetched HERE, once per tile row -- as a load inside `locate` it sat in
                // front of every chunk's store behind an s_waitcnt vmcnt(0) that also waited for the prefetched operands of the next slab and
                // for the stores before it (round 5, tools/isa_wait_scan.py)
                const float scv = (tok >= 0 && P.scale) ? P.scale[b] : 1.0f;
                rowtok[tid] = tok < 0 ? -1 : ((tok << 32) | (int64_t)__float_as_uint(scv));
            }
            lw_lgkm0();
            g_bar();
        }
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));            // opaque: keeps the chunk addresses of the tail from being formed above the main loop
        // slab p, slab row r -> tile row: the slab's first HC rows are rows HC p .. of group 0, the others the same rows of group 1
        auto tile_row = [](int p, int r) -> int { return r < HC ? HC * p + r : HALF + HC * p + (r - HC); };
        // chunk k of this lane in slab p: slab row / 16-byte column, global row / column, token and DropPath factor (mode 3)
        auto locate = [&](int p, int k, int& row, int& ch, int& gm, int& gn, int64_t& tok, float& sc) -> bool {
            const int idx = tid_e + k * LW_MFMA_THREADS;
            row = idx / CPR;
            ch = idx - row * CPR;
            gm = m0 + tile_row(p, row);
            gn = n0 + 8 * ch;
            tok = 0;
            sc = 1.0f;
            bool ok = idx < NCH && gm < P.M && gn < P.N;
            if (P.mode == 3 && ok) {
                const int64_t rt = rowtok[tile_row(p, row)];
                ok = rt >= 0;
                tok = rt >> 32;
                sc = __uint_as_float((uint32_t)rt);
            }
            return ok;
        };
        // extra operands of the tail: two slabs in flight.  (Requesting the WHOLE tile's operands up front -- they are cold in HBM, 2-4 us
        // away -- needs 48 registers next to the 48 of the packed tile and the GELU' arithmetic: the allocator parks them in scratch
        // behind a wait, which serialises exactly the latency the request was meant to hide.  Measured in situ (round 4): the
        // x GELU'(f1) / residual tails of the K <= 768 GEMMs run 1.3-2.1x the time of gemm_nt's two-workgroup read-out here, so the
        // dispatcher keeps those on gemm_nt; this kernel takes the plain / bias tails and the long contractions.)
        constexpr int NX = EC == 0 ? 1 : ITERS;
        constexpr int XD = 2;                      // slabs of extra operands held
        u32x4 xa[XD][NX], xb[EC == 2 ? 2 : 1][EC == 2 ? NX : 1];
        // branch-free per lane (a chunk outside the problem reads element 0 of the operand and is dropped by `locate`)
        auto prefetch = [&](int p) {
            if constexpr (EC == 1) {
                if (P.mode == 3) {
#pragma unroll
                    for (int k = 0; k < ITERS; ++k) {
                        const int idx = tid_e + k * LW_MFMA_THREADS, row = idx / CPR, gm = m0 + tile_row(p, row), gn = n0 + 8 * (idx - row * CPR);
                        const int64_t rt = idx < NCH ? rowtok[tile_row(p, row)] : -1;
                        const int64_t off = (gm < P.M && gn < P.N && rt >= 0) ? (rt >> 32) * P.N + gn : 0;
                        xa[p % XD][k] = *reinterpret_cast<const u32x4*>((const uint16_t*)P.res + off);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < ITERS; ++k) {
                        const int idx = tid_e + k * LW_MFMA_THREADS, row = idx / CPR, gm = m0 + tile_row(p, row), gn = n0 + 8 * (idx - row * CPR);
                        const int64_t off = (idx < NCH && gm < P.M && gn < P.N) ? (int64_t)gm * P.ldaux + gn : 0;
                        xa[p % XD][k] = *reinterpret_cast<const u32x4*>(P.aux + off);
                    }
                }
            } else if constexpr (EC == 2) {
#pragma unroll
                for (int k = 0; k < ITERS; ++k) {
                    const int idx = tid_e + k * LW_MFMA_THREADS, row = idx / CPR, gm = m0 + tile_row(p, row), gn = n0 + 8 * (idx - row * CPR);
                    const int64_t rt = idx < NCH ? rowtok[tile_row(p, row)] : -1;
                    const int64_t off = (gm < P.M && gn < P.N && rt >= 0) ? (rt >> 32) * P.N + gn : 0;
                    xa[p & 1][k] = reinterpret_cast<const u32x4*>((const float*)P.res + off)[0];
                    xb[p & 1][k] = reinterpret_cast<const u32x4*>((const float*)P.res + off)[1];
                }
            }
        };
        // bias added in fp32, ONE rounding to bf16, in registers: the accumulators (and the bias row) are dead from here on and the
        // read-out below works on half the registers
        u32x2 pk[WMF][WNF];
        {
            const int colw = wc * (BN / 4) + 4 * g;
#pragma unroll
            for (int j = 0; j < WNF; ++j) {
                const int n = n0 + colw + 16 * j;
                uint32_t b01 = 0, b23 = 0;
                if (P.bias && n < P.N) {
                    const u32x2 raw = *reinterpret_cast<const u32x2*>(P.bias + n);
                    b01 = raw[0];
                    b23 = raw[1];
                }
                const float bv0 = __uint_as_float(b01 << 16), bv1 = __uint_as_float(b01 & 0xffff0000u);
                const float bv2 = __uint_as_float(b23 << 16), bv3 = __uint_as_float(b23 & 0xffff0000u);
#pragma unroll
                for (int i = 0; i < WMF; ++i) {
                    const f32x4 a = acc[i][j];
                    pk[i][j] = u32x2{pack_bf2(a[0] + bv0, a[1] + bv1), pack_bf2(a[2] + bv2, a[3] + bv3)};
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i)
#pragma unroll
            for (int j = 0; j < WNF; ++j) asm volatile("" : "+v"(pk[i][j]));      // packed HERE, not lazily in front of each slab
        asm volatile("" ::: "memory");             // (and the tail's operand loads below stay below: they need the registers this frees)
        prefetch(0);
        LCLK(3);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (p + 1 < NPASS) prefetch(p + 1);
            {                                      // fragment rows FPP p .. of this wave -> slab rows HC grp + 16 ii + c
                const int colw = wc * (BN / 4) + 4 * g;
#pragma unroll
                for (int ii = 0; ii < FPP; ++ii) {
                    const int i = p * FPP + ii;
                    const int row = HC * grp + 16 * ii + c;
#pragma unroll
                    for (int j = 0; j < WNF; ++j) *reinterpret_cast<DGX_LDS u32x2*>(stg + row * SROW + (colw + 16 * j) * 2) = pk[i][j];
                }
            }
            lw_lgkm0();
            g_bar();                               // slab staged
            u32x4 yv[ITERS];
#pragma unroll
            for (int k = 0; k < ITERS; ++k) {
                const int idx = tid_e + k * LW_MFMA_THREADS, row = idx / CPR, ch = idx - row * CPR;
                yv[k] = *reinterpret_cast<DGX_LDS const u32x4*>(stg + (idx < NCH ? row * SROW + ch * 16 : 0));
            }
            lw_lgkm0();
            g_bar();                               // slab read back: the staging area is free for the next one
#pragma unroll
            for (int k = 0; k < ITERS; ++k) {
                int row, ch, gm, gn;
                int64_t tok;
                float sc;
                if (locate(p, k, row, ch, gm, gn, tok, sc))
                    g_epi_finish(P, gm, gn, yv[k], tok, sc, xa[p % XD][EC == 0 ? 0 : k], xb[EC == 2 ? (p & 1) : 0][EC == 2 ? k : 0]);
                __builtin_amdgcn_sched_barrier(0);  // one chunk's tail at a time: interleaved, the GELU arithmetic of three chunks spills
            }
        }
        // Where the rings fill the whole LDS (128 x 192, 4 + 4 slots) the row table of the residual tails lies INSIDE the last A slot:
        // the loaders' `issue_a(NSA - 1)` for the NEXT item would overwrite it while `locate` above still reads it for the last slab
        // (a garbage token = a store far outside the tensor: found by tests/test_gpu_pins.py with dgx_set_reserved_cus(16), the first
        // configuration that gives a 128 x 192 residual launch a second item per workgroup).  One more barrier per item, this layout only.
        if constexpr ((MC == 3 || MC == 6) && Cfg::TOK0 < Cfg::RING) { lw_lgkm0(); g_bar(); }
        LCLK(4);
    }
}

static int g_reserved_per_xcd = 0;
// CUs (rounded up to one per XCD) that persistent kernels leave to something else running beside them: the RCCL channels of the
// gradient all-reduce that overlaps backward (engine/ddp.ArenaReducer sets it to the channel count when the group has > 1 rank)
extern "C" void dgx_set_reserved_cus(int n) {
    int per = n <= 0 ? 0 : (n + 7) / 8;
    g_reserved_per_xcd = per > 24 ? 24 : per;
}
extern "C" int dgx_get_reserved_cus(void) { return 8 * g_reserved_per_xcd; }

namespace {
template <int BM, int BN, int NSA, int NSB, int MC>
int lw_launch_e(GemmP& P, hipStream_t st) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)gemm_lw_kernel<BM, BN, NSA, NSB, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, LW_LDS) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    // workgroups per XCD: one per CU, minus the CUs left to a concurrent collective (dgx_set_reserved_cus).  A workgroup of this
    // kernel takes a CU's whole LDS and register file, so it cannot share a CU with an RCCL channel: with 256 workgroups launched
    // and 16 CUs held by channels, 16 workgroups would wait for a second round -- the persistent tile loop simply runs on fewer
    const int avail = 32 - g_reserved_per_xcd;
    const int wgx = P.per_xcd < avail ? P.per_xcd : avail;
    hipLaunchKernelGGL((gemm_lw_kernel<BM, BN, NSA, NSB, MC>), dim3(8 * wgx), dim3(LW_THREADS), LW_LDS, st, P);
    return DGX_OK;
}
template <int BM, int BN, int NSA, int NSB>
int lw_launch_t(GemmP& P, hipStream_t st) {
    switch (P.mode) {
        case 2: return lw_launch_e<BM, BN, NSA, NSB, 2>(P, st);
        case 3: return P.res_dtype == DGX_BF16 ? lw_launch_e<BM, BN, NSA, NSB, 3>(P, st) : lw_launch_e<BM, BN, NSA, NSB, 6>(P, st);
        case 4: return lw_launch_e<BM, BN, NSA, NSB, 4>(P, st);
        case 5: return lw_launch_e<BM, BN, NSA, NSB, 5>(P, st);
        default: return lw_launch_e<BM, BN, NSA, NSB, 0>(P, st);
    }
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
