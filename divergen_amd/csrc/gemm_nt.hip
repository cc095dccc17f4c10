// bf16 GEMM with fused epilogues for gfx950:  C[M][N] = epi( sum_k A[m][k] * B[n][k] ),  A (M,K), B (N,K) row-major
// (both operands K-contiguous: y = x W^T of nn.Linear as stored; the input gradient dx = dy W runs on the transposed
// weight image the optimizer step refreshes, so it is the same kernel).  Replaces the hipBLASLt calls behind
// torch.addmm / torch.mm at the Linear sites of the reference: swintransformer.py:133,155 (qkv / proj), :40-46 (Mlp),
// :296 (PatchMerging.reduction), fpn.py:126-154 (1x1 laterals), box_head.py:26-98 (FCs).
//
// Structure (one workgroup = one BM x BN output tile, 8 waves = 2 (M) x 4 (N), K-step 64):
//   * both operand tiles go global -> LDS with LDS-direct loads (`buffer_load_dwordx4 ... lds`, no staging registers):
//     one wave-instruction = 8 tile rows x 128 B = 8 whole cache lines; rows are 128 B in LDS, the 16-byte chunk index
//     is XOR-swizzled with (row >> 1) & 7 (applied to the per-lane SOURCE address, the LDS image of an instruction is
//     lane-linear), which makes every ds_read_b128 of an MFMA fragment (16 rows x one chunk column) bank-conflict free;
//   * two LDS stages; the 8 waves form two groups (waves 0-3 / 4-7 = one wave of each group per SIMD) that run the
//     same per-K-tile program {read all fragments of the tile into registers | 2*WMF*WNF MFMAs} one phase apart: while
//     one wave of a SIMD issues MFMAs back to back, its partner reads LDS and issues the loads of the tile after next.
//     A stage is re-filled as soon as both groups have read it and has ~two phases to land (counted only by vmcnt(0)
//     at the end of the phase before its first read; raw s_barrier, so LDS-direct loads stay in flight across barriers);
//   * MFMA operands are swapped (D = W-fragment x X-fragment) so that a lane holds 4 CONSECUTIVE output columns of one
//     row: the epilogue packs them to bf16, stages the whole tile in LDS (stage buffers are idle by then) and streams
//     it out as whole rows, 16 B per lane, with the fused tail (bias | bias + exact GELU with both tensors written |
//     bias + window_reverse/roll/crop + DropPath + residual add | multiply by GELU'(f1)) applied on the way out;
//   * XCD-aware tile order: workgroups of one XCD (blockIdx % 8) take consecutive tiles = the same A row-panel.
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {
template <int BM, int BN, int NS_> struct GemmCfg {
    static constexpr int NS = NS_;                 // LDS stages (K-tiles resident or in flight)
    static constexpr int WMF = BM / 32;            // 16-row MFMA fragments per wave along M (2 waves)
    static constexpr int WNF = BN / 64;            // 16-column fragments per wave along N (4 waves)
    static constexpr int NA = BM / 64;             // LDS-direct loads per wave per K-tile, A rows
    static constexpr int NB = BN / 64;             //                                      B rows
    static constexpr int SB = (BM + BN) * 128;     // bytes per stage
    static constexpr int SROW = BN * 2 + 16;       // epilogue staging row stride (bytes)
    static constexpr int EPI = BM * SROW + BM * 8; // staged tile + row -> (token, sample) table
    static constexpr int LDS = (NS * SB > EPI ? NS * SB : EPI);
};
}  // namespace

// DIAG (development builds only, -DDGX_GEMM_DEV): ablation bits -- 1 no steady-state loads, 2 no MFMAs, 4 no fragment reads
// MC >= 0: the fused tail is a compile-time constant (2: bias + GELU, 3 / 6: bf16 / fp32 residual, 4: x GELU'(f1)), as in gemm_lw --
// the two-workgroup instantiation that runs the K <= 768 fused-tail GEMMs.  With the mode a run-time field the tail's operand prefetch
// sat under mode branches, and the compiler's wait-count model, merging the path without a prefetch, waited for every outstanding load
// (vmcnt(0)) before the staging pass the prefetch was meant to overlap (round 5, tools/isa_wait_scan.py).  MC = -1: run-time mode.
template <int BM, int BN, int NS, int MINW, int DIAG = 0, int MC = -1>
__global__ __launch_bounds__(512, MINW) void gemm_nt_kernel(GemmP P) {
    if constexpr (MC == 6) { P.mode = 3; P.res_dtype = DGX_F32; }
    else if constexpr (MC == 3) { P.mode = 3; P.res_dtype = DGX_BF16; }
    else if constexpr (MC >= 0) P.mode = MC;
    using Cfg = GemmCfg<BM, BN, NS>;
    constexpr int NL = Cfg::NA + Cfg::NB;          // LDS-direct loads per wave per K-tile
    constexpr int WMF = Cfg::WMF, WNF = Cfg::WNF, NA = Cfg::NA, NB = Cfg::NB, SB = Cfg::SB, SROW = Cfg::SROW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int L0 = (blockIdx.x & 7) * P.per_xcd + (blockIdx.x >> 3);
    if (L0 >= P.total * P.splits) return;
    int L = L0 / P.splits;
    const int split = L0 - L * P.splits;           // the splits of a tile are neighbours on one XCD
    if (P.ngrp > 0) {                              // grouped convolution: this tile's image (P is this kernel's own copy of the descriptor)
        L = __builtin_amdgcn_readfirstlane(L);    // (the division above runs on the vector unit: tell the compiler the value is uniform)
        // constant indices only (a dynamic index into the by-value descriptor would go through private memory and lose uniformity)
        dgxgemm::GemmP::Grp q = P.grp[0];
#pragma unroll
        for (int k = 1; k < dgxgemm::GEMM_MAXG; ++k)
            if (k < P.ngrp && L >= P.grp[k].tile0) q = P.grp[k];
        P.A = q.A; P.C = q.C; P.M = q.M;
        P.cmap_n = q.cn; P.cmap_h = q.ch; P.cmap_w = q.cw; P.conv_wp = q.wp;
        L -= q.tile0;
    }
#define GCLK(i) do { if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#define GCLKR(i) do { if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    GCLK(0);
    GCLKR(5);                                      // constant 100 MHz counter next to the shader-clock one: the ratio is the clock
    const int tm = L / P.tiles_n, tn = L - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, wc = w & 3;            // group = M half of the tile (waves 0-3: rows 0 .. BM/2)
    const int c = l & 15, g = l >> 4;
    const int NTK = (P.K + GBK - 1) / GBK;         // K-tiles of the problem
    const int ktail = P.K - (NTK - 1) * GBK;       // elements of the last one (64 when K % 64 == 0)
    const int kt0 = split * P.kt_per_split;        // this workgroup's K-tiles: kt0 .. kt0 + NT
    const int NT = min(P.kt_per_split, NTK - kt0);

    // ---- loader role: instruction q = w + 8 s of a tile covers tile rows 8q .. 8q+7 (A rows first, then B rows);
    // lane -> row 8q + (l >> 3), physical chunk l & 7 = logical chunk ^ ((row >> 1) & 7), (q & 1) == (w & 1)
    const int rsub = l >> 3;
    const int lc = (l & 7) ^ (((w & 1) << 2) | (rsub >> 1));
    uint32_t voffA[NA], voffB[NB];
#pragma unroll
    for (int s = 0; s < NA; ++s) {
        const int m = m0 + 8 * (w + 8 * s) + rsub;
        int64_t arow = m;
        if (P.conv_kc) {                           // output pixel (n, y, x) -> its position in the zero-bordered image
            const int hw = P.cmap_h * P.cmap_w;
            const int n = m / hw, r = m - n * hw;
            const int y = r / P.cmap_w, x = r - y * P.cmap_w;
            arow = ((int64_t)n * (P.cmap_h + 2) + y + 1) * P.conv_wp + x + 1;
        }
        voffA[s] = m < P.M ? (uint32_t)((arow * P.lda + lc * 8) * 2) : G_OOB;
    }
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        const int n = n0 + 8 * (w + 8 * s) + rsub;
        voffB[s] = n < P.N ? (uint32_t)(((int64_t)n * P.ldb + lc * 8) * 2) : G_OOB;
    }
    const bool kt_ok = lc * 8 < ktail;             // this lane's chunk exists in the last K-tile
    const u32x4 rA = g_rsrc(P.A, (uint32_t)((P.conv_kc ? (int64_t)P.cmap_n * (P.cmap_h + 2) * P.conv_wp + 2 * P.conv_wp + 2 : (int64_t)P.M) * P.lda * 2));
    const u32x4 rB = g_rsrc(P.B, (uint32_t)((int64_t)P.N * P.ldb * 2));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + 1024u * w);
    // one of the NL loads of a tile: k < NA -> A rows, else B rows
    // K rotation (development builds: DGX_GEMM_KROT=1; off in the product): workgroups that share an operand panel walk K from different starting
    // tiles, so a line fetched from HBM for one is in L2 when the others ask.  Measured (tools/gemm_cold_probe2.py): -7..10 %
    // when every operand is cold in HBM, +5..12 % when they sit in L2 / the memory-side cache, nothing on the training step.
    const int rot = (P.krot && NT >= 4) ? ((tm + tn) & 3) * (NT >> 2) : 0;
    auto issue_one = [&](int k, int t, int stage) {
        const int kta = kt0 + (t + rot >= NT ? t + rot - NT : t + rot);
        const uint32_t soff = (uint32_t)kta * (GBK * 2);
        uint32_t soffA = soff;
        if (P.conv_kc) {                           // implicit convolution: tap shift (rows) + channel block of the tap
            const int tap = kta / P.conv_kc, kc = kta - tap * P.conv_kc;
            soffA = (uint32_t)((tap / 3) * P.conv_wp + tap % 3) * (uint32_t)(P.lda * 2) + (uint32_t)kc * (GBK * 2);
        }
        const uint32_t dst = ldsw + (uint32_t)stage * SB;
        const bool tail = (kta == NTK - 1) && (ktail != GBK) && !kt_ok;
        if (k < NA) g_load_lds16(tail ? G_OOB : voffA[k], rA, dst + 8192u * k, soffA);
        else g_load_lds16(tail ? G_OOB : voffB[k - NA], rB, dst + BM * 128 + 8192u * (k - NA), soff);
    };

    // ---- MFMA role: wave tile = rows grp*BM/2 .. (+16 i + c), columns wc*BN/4 .. (+16 j + c); a fragment (16 rows, k-half kh)
    // is one ds_read_b128 per lane: row c, logical chunk g + 4 kh -> physical (g ^ swz) ^ 4 kh, swz = (c >> 1) & 7
    const int swz = (c >> 1) & 7;
    const uint32_t la = (uint32_t)((grp * (BM / 2) + c) * 128 + ((g ^ swz) << 4));
    const uint32_t lb = (uint32_t)((BM + wc * (BN / 4) + c) * 128 + ((g ^ swz) << 4));
    f32x4 acc[WMF][WNF];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[WMF][2], bfr[WNF][2];
    // LOAD phase of group 0: this wave's NL loads of tile `ti` (if >= 0) into stage `istage`, then the 2 (WMF + WNF) fragment
    // reads of the tile in `stage`.  Measured (tools/gemm_phase_probe.py): an LDS-direct load costs its wave ~60 cycles of
    // issue next to MFMAs or plain code and ~80-180 when it sits between ds_reads, so the loads are kept together.
    auto load_phase = [&](int stage, int ti, int istage) {
        if (ti >= 0) {
#pragma unroll
            for (int k = 0; k < NL; ++k) issue_one(k, ti, istage);
        }
        if constexpr ((DIAG & 4) != 0) return;
        DGX_LDS const unsigned char* sa = lds_opaque((const unsigned char*)lds_raw + stage * SB + la);
        DGX_LDS const unsigned char* sa1 = lds_opaque((const unsigned char*)lds_raw + stage * SB + (la ^ 64u));
        DGX_LDS const unsigned char* sb = lds_opaque((const unsigned char*)lds_raw + stage * SB + lb);
        DGX_LDS const unsigned char* sb1 = lds_opaque((const unsigned char*)lds_raw + stage * SB + (lb ^ 64u));
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
            bfr[j][0] = *reinterpret_cast<DGX_LDS const bf16x8*>(sb + 2048 * j);
            bfr[j][1] = *reinterpret_cast<DGX_LDS const bf16x8*>(sb1 + 2048 * j);
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i) {
            af[i][0] = *reinterpret_cast<DGX_LDS const bf16x8*>(sa + 2048 * i);
            af[i][1] = *reinterpret_cast<DGX_LDS const bf16x8*>(sa1 + 2048 * i);
        }
    };
    auto mfmas = [&]() {
        if constexpr ((DIAG & 2) != 0) return;
        if constexpr ((DIAG & 8) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < WNF; ++j) acc[i][j] = mfma16(bfr[j][kh], af[i][kh], acc[i][j]);   // D[n][m]: lane = 4 columns of a row
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- main loop.  Intervals between barriers I_0, I_1, ...: group 0 reads tile t in I_{2t+1} and multiplies in I_{2t+2};
    // group 1 reads in I_{2t+2} and multiplies in I_{2t+3}.  Tile tau lives in stage tau % NS: it replaces tile tau - NS, whose
    // last read (group 1) ends with the barrier that opens I_{2(tau-NS)+3}; both groups issue it in that interval (group 0 in
    // its read phase of tile tau-NS+1, group 1 in its MFMA phase of tile tau-NS) and every wave has waited for its share
    // before the barrier that opens I_{2 tau + 1}: 2 (NS - 1) intervals of flight.
    // wait for this wave's share of tile `tau` with the later tiles it has already issued still in flight
    auto wait_tile = [&](int tau) {
        int later = NT - 1 - tau;
        if (later > NS - 2) later = NS - 2;
        if (NS >= 4 && later >= 2) g_vmcnt<2 * NL>();
        else if (NS >= 3 && later >= 1) g_vmcnt<NL>();
        else g_vmcnt<0>();
    };
#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (t < NT) {
#pragma unroll
            for (int k = 0; k < NL; ++k) issue_one(k, t, t);
        }
    {
        int later = NT - 1;
        if (later > NS - 1) later = NS - 1;
        if (NS >= 4 && later >= 3) g_vmcnt<3 * NL>();
        else if (NS >= 3 && later >= 2) g_vmcnt<2 * NL>();
        else if (later >= 1) g_vmcnt<NL>();
        else g_vmcnt<0>();
    }
    g_bar();                                       // #0: tile 0 visible to everyone
    GCLK(1);
    int rs = 0;                                    // stage of the tile being read
    // Group 0 issues in its LOAD phase, group 1 right AFTER its MFMAs (both in I_{2t+3} for tile t + 2 when NS = 2).
    // Measured per K-tile of a 256x192 tile: loads of group 1 in front of its MFMAs 2525 cycles (their issue time delays the
    // whole interval), behind them 2106; both groups issuing in their LOAD phases (possible from 3 stages on) is no better.
    constexpr bool G1_IN_LOAD = false;
    if (grp == 0 || G1_IN_LOAD) {
        int is = NS - 1;                           // stage of the next tile to issue (tile t + NS - 1 at t >= 1)
        if (grp == 1) g_bar();                     // #1: one phase behind group 0
        for (int t = 0; t < NT; ++t) {
            int ti = -1;
            if (t >= 1) {
                is = is + 1 == NS ? 0 : is + 1;
                if (t + NS - 1 < NT && !(DIAG & 1)) ti = t + NS - 1;
            }
            load_phase(rs, ti, is);
            rs = rs + 1 == NS ? 0 : rs + 1;
            g_lgkm0();
            if (grp == 1 && t + 1 < NT) wait_tile(t + 1);
            g_bar();                               // group 0: #(2t+1), group 1: #(2t+2)
            mfmas();
            if (grp == 0 && t + 1 < NT) wait_tile(t + 1);
            g_bar();                               // group 0: #(2t+2), group 1: #(2t+3)
        }
        if (grp == 0) g_bar();                     // group 1's last phase
    } else {
        int is = 0;                                // tile t + NS goes where tile t was
        g_bar();                                   // #1: one phase behind group 0
        for (int t = 0; t < NT; ++t) {
            load_phase(rs, -1, 0);
            rs = rs + 1 == NS ? 0 : rs + 1;
            g_lgkm0();
            if (t + 1 < NT) wait_tile(t + 1);
            g_bar();                               // #(2t+2)
            mfmas();
            if (t + NS < NT && !(DIAG & 1)) {
#pragma unroll
                for (int k = 0; k < NL; ++k) issue_one(k, t + NS, is);
            }
            is = is + 1 == NS ? 0 : is + 1;
            g_bar();                               // #(2t+3)
        }
    }
    g_vmcnt<0>();
    GCLK(2);

    // ---- split-K: the raw fp32 accumulators go to this split's slab; dgx's fold kernel finishes the job
    if (P.splits > 1) {
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
        return;
    }
    // ---- epilogue: stage the tile as bf16 (bias added) in LDS, then stream whole rows out with the fused tail.  The extra
    // operands of the tail (mode 4 / 5: the saved pre-activation tile; mode 3: the residual rows -- cold in HBM) are requested
    // for the WHOLE tile before the accumulators are staged (ITERS 16-byte loads per lane in flight, hidden behind the staging
    // pass and its barrier); round 2 fetched them in batches of four behind the barrier and the read-out waited on each batch.
    DGX_LDS unsigned char* stg = (DGX_LDS unsigned char*)lds_raw;
    DGX_LDS int64_t* rowtok = reinterpret_cast<DGX_LDS int64_t*>(stg + BM * SROW);   // mode 3: (token << 32 | DropPath factor bits) per tile row
    constexpr int CPR = BN / 8;                    // 16-byte chunks per tile row
    constexpr int ITERS = (BM * CPR + 511) / 512;
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
        __syncthreads();
    }
    // PFN chunks of the tail operands per lane are requested at a time: the whole tile for the one-workgroup-per-CU instantiations,
    // half of it where two workgroups share a CU (128 registers per lane; the partner workgroup covers the second half's latency)
    constexpr int PFN = MINW >= 4 ? ITERS / 2 : ITERS;
    static_assert(PFN > 0 && ITERS % PFN == 0 && ITERS / PFN <= 2, "one or two whole chunks");
    u32x4 xa[PFN], xb[PFN];
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));               // opaque: keeps the chunk addresses of the tail from being formed above the main loop
    // chunk `it` of this lane: tile row / 16-byte column, global row / column, token and DropPath factor (mode 3)
    auto locate = [&](int it, int& row, int& ch, int& gm, int& gn, int64_t& tok, float& sc) -> bool {
        const int idx = tid_e + it * 512;
        row = idx / CPR;
        ch = idx - row * CPR;
        gm = m0 + row;
        gn = n0 + 8 * ch;
        tok = 0;
        sc = 1.0f;
        bool ok = idx < BM * CPR && gm < P.M && gn < P.N;
        if (P.mode == 3 && ok) {
            const int64_t rt = rowtok[row];
            ok = rt >= 0;
            tok = rt >> 32;
            sc = __uint_as_float((uint32_t)rt);
        }
        return ok;
    };
    // branch-free per lane (a chunk outside the problem reads element 0 of the operand and is dropped in the finish loop): with
    // the loads under per-lane conditions the compiler carries both arrays through every join and spills them
    static_assert((BM * CPR) % 512 == 0, "every lane owns exactly ITERS chunks");
    // ONE straight-line set of loads for every tail that reads an operand: the address is chosen per mode, the load is common.  With a
    // load per mode branch (round 4) the values met in different registers at the join and the compiler copied them there -- behind an
    // s_waitcnt vmcnt(0) at the end of each branch, i.e. the prefetch was waited for before the staging pass it was meant to overlap.
#define DGX_EPI_PREFETCH_N(IT0, SLOT0, CNT)                                                                                       \
    if (P.mode >= 3) {                                                                                                            \
        const bool m3_ = P.mode == 3, f32_ = m3_ && P.res_dtype != DGX_BF16;                                                      \
        const char* base_ = m3_ ? (const char*)P.res : (const char*)P.aux;                                                        \
        _Pragma("unroll") for (int pi_ = 0; pi_ < (CNT); ++pi_) {                                                                    \
            const int idx = tid_e + ((IT0) + pi_) * 512, row = idx / CPR, gm = m0 + row, gn = n0 + 8 * (idx - row * CPR);          \
            const bool in_ = gm < P.M && gn < P.N;                                                                                \
            int64_t off = 0;                                                                                                      \
            if (m3_) { const int64_t rt = rowtok[row]; if (in_ && rt >= 0) off = (rt >> 32) * P.N + gn; }                         \
            else if (in_) off = (int64_t)gm * P.ldaux + gn;                                                                       \
            const char* p_ = base_ + off * (f32_ ? 4 : 2);                                                                        \
            xa[(SLOT0) + pi_] = *reinterpret_cast<const u32x4*>(p_);                                                               \
            if (f32_) xb[(SLOT0) + pi_] = *reinterpret_cast<const u32x4*>(p_ + 16);                                                \
        }                                                                                                                         \
    }
#define DGX_EPI_PREFETCH(IT0) DGX_EPI_PREFETCH_N(IT0, 0, PFN)
    {
        const int colw = wc * (BN / 4) + 4 * g;    // + 16 j: this lane's 4 consecutive columns
        // the bias row is requested IN FRONT of the tail's operands: vmcnt retires in order, so waiting for a bias load that was issued
        // behind the prefetch meant waiting for the whole prefetch before the staging pass it was to hide under (round 5)
        u32x2 braw[WNF];
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
            const int n = n0 + colw + 16 * j;
            braw[j] = u32x2{0u, 0u};
            if (P.bias && n < P.N) braw[j] = *reinterpret_cast<const u32x2*>(P.bias + n);
        }
        __builtin_amdgcn_sched_barrier(0);
        DGX_EPI_PREFETCH(0)
        __builtin_amdgcn_sched_barrier(0);
        float bv[WNF][4];
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
            const uint32_t b01 = braw[j][0], b23 = braw[j][1];
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
    GCLK(3);
#pragma unroll
    for (int it = 0; it < PFN; ++it) {
        int row, ch, gm, gn;
        int64_t tok;
        float sc;
        if (locate(it, row, ch, gm, gn, tok, sc))
            g_epi_finish(P, gm, gn, *reinterpret_cast<DGX_LDS const u32x4*>(stg + row * SROW + ch * 16), tok, sc, xa[it], xb[it]);
        // second half (two workgroups per CU: half the tile's operands fit the registers): chunk PFN + it is requested into the
        // registers chunk `it` has just released, so it flies while the rest of the first half is finished
        if constexpr (PFN < ITERS) { DGX_EPI_PREFETCH_N(PFN + it, it, 1) }
    }
    if constexpr (PFN < ITERS) {
#pragma unroll
        for (int it = 0; it < PFN; ++it) {
            int row, ch, gm, gn;
            int64_t tok;
            float sc;
            if (locate(PFN + it, row, ch, gm, gn, tok, sc))
                g_epi_finish(P, gm, gn, *reinterpret_cast<DGX_LDS const u32x4*>(stg + row * SROW + ch * 16), tok, sc, xa[it], xb[it]);
        }
    }
#undef DGX_EPI_PREFETCH
#undef DGX_EPI_PREFETCH_N
    GCLK(4);
    GCLKR(6);
}

// split-K fold: y = bf16(sum_s slab[s] + bias), then the same fused tail; one lane per 8-column chunk
__global__ __launch_bounds__(256) void gemm_splitk_fold_kernel(GemmP P) {
    const int cpr = P.N >> 3;
    const int64_t total = (int64_t)P.M * cpr, slab = (int64_t)P.M * P.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int gm = (int)(i / cpr), gn = (int)(i - (int64_t)gm * cpr) * 8;
        const float* p = P.ws + (int64_t)gm * P.N + gn;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < P.splits; ++s) {
            const float4 a = reinterpret_cast<const float4*>(p + s * slab)[0], b = reinterpret_cast<const float4*>(p + s * slab)[1];
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (P.bias) {
            float bv[8];
            g_unpack8(*reinterpret_cast<const u32x4*>(P.bias + gn), bv);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bv[k];
        }
        int64_t tok = 0;
        float sc = 1.0f;
        if (P.mode == 3) {
            int b = 0;
            tok = g_row_token(P.map, gm, b);
            if (tok < 0) continue;
            if (P.scale) sc = P.scale[b];
        }
        g_epilogue_chunk(P, gm, gn, g_pack8(v), tok, sc);
    }
}

namespace {
struct TileChoice { int bm, bn; };
int64_t g_ws_bytes_cur = 0;     // size of the workspace of the call being planned
// Test / A-B knobs of the dispatch (dgx_dev_set): -1 / 0 = the library's own plan.  The product never sets them; the GEMM tests
// force every tile shape and the split-K path through the one ABI entry with them, the tools compare own form A with own form B.
struct DevKnobs { int lw = -1, two_wg = -1, tile_bm = 0, tile_bn = 0, splitk = 0, k192 = -1; } g_dev;
struct LastForm { int form = -1, bm = 0, bn = 0, splits = 0; } g_last;     // what the most recent dispatch launched (dgx_gemm_last_form)

// split-K plan: few output tiles and a long contraction (box-head FC 1024 x 1024 x 12544, 3x3 convolutions over the small
// FPN levels, stage-3 Linears) leave most CUs idle; S slabs of >= 4 K-tiles each fill them.  Returns 1 when not worth it.
int choose_splits(int64_t tiles, int K, int64_t M, int64_t N, int64_t ws_bytes) {
    const int nt = (K + GBK - 1) / GBK;
    if (g_dev.splitk >= 1) {
        const int v = g_dev.splitk;
        return (v <= nt && (int64_t)v * M * N * 4 <= ws_bytes) ? v : 1;
    }
    if (tiles > 128 || nt < 8) return 1;
    int S = (int)(256 / tiles);
    if (S > nt / 4) S = nt / 4;
    if (S > 16) S = 16;
    while (S > 1 && (int64_t)S * M * N * 4 > ws_bytes) --S;
    return S < 2 ? 1 : S;
}

// Tile selection: BN from the divisibility of N (every Swin width is a multiple of 192), BM from how well the tile count
// fills whole rounds of 256 CUs (one workgroup per CU), weighted by the CU-side efficiency of the smaller tiles.
static bool tile_192x256() { return true; }      // the 192 x 256 tile (2 stages) where it saves a round of the chip
TileChoice choose_tile(int M, int N) {
    if (g_dev.tile_bm) {
        const int bm = g_dev.tile_bm, bn = g_dev.tile_bn;
        if ((bm == 256 || bm == 192 || bm == 128) && (bn == 192 || bn == 128 || bn == 256) && !(bm == 256 && bn == 256) && !(bm == 192 && bn == 128))
            return {bm, bn};
    }
    int bn;
    if (N % 192 == 0) bn = 192;
    else if (N % 256 == 0 || N > 1024) bn = 256;
    else bn = 128;
    const int cand[3] = {256, 192, 128};
    const double eff[3] = {1.0, 0.97, 0.85};
    double best = -1.0;
    int bm = 128;
    for (int i = 0; i < 3; ++i) {
        const int b = cand[i];
        if (bn == 256 && b == 256) continue;       // 256x256 does not fit two waves per SIMD (register file)
        if (bn == 128 && b == 192) continue;       // 192-row tiles: instantiated for BN = 192 and 256
        if (bn == 256 && b == 192 && !tile_192x256()) continue;
        const int64_t tiles = (int64_t)((M + b - 1) / b) * ((N + bn - 1) / bn);
        const int64_t rounds = (tiles + 255) / 256;
        const double fill = (double)M * N / ((double)rounds * 256 * b * bn);
        const double sc = fill * eff[i];
        if (sc > best) { best = sc; bm = b; }
    }
    return {bm, bn};
}

template <int BM, int BN, int NS, int MINW = 2, int DIAG = 0, int MC = -1>
int launch_gemm(GemmP& P, hipStream_t st) {
    using Cfg = GemmCfg<BM, BN, NS>;
    const int tiles_m = (P.M + BM - 1) / BM;
    P.tiles_n = (P.N + BN - 1) / BN;
    P.total = tiles_m * P.tiles_n;
    const int nt = (P.K + GBK - 1) / GBK;
    P.splits = choose_splits(P.total, P.K, P.M, P.N, P.ws ? g_ws_bytes_cur : 0);
    P.kt_per_split = (nt + P.splits - 1) / P.splits;
    P.splits = (nt + P.kt_per_split - 1) / P.kt_per_split;          // no empty split
    P.per_xcd = (P.total * P.splits + 7) / 8;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, NS, MINW, DIAG, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    g_last = {MINW == 4 ? 2 : 0, BM, BN, P.splits};
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, NS, MINW, DIAG, MC>), dim3(8 * P.per_xcd), dim3(512), Cfg::LDS, st, P);
    if (P.splits > 1) {
        const int64_t chunks = (int64_t)P.M * (P.N >> 3);
        const int grid = (int)((chunks + 255) / 256 < 4096 ? (chunks + 255) / 256 : 4096);
        hipLaunchKernelGGL(gemm_splitk_fold_kernel, dim3(grid), dim3(256), 0, st, P);
    }
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
}  // namespace

int gemm_lw_launch(GemmP& P, int bm, int bn, hipStream_t st);     // gemm_lw.hip: the loader-wave form
// Which form runs a problem (measured in situ, profiles/r04_gemm_insitu_*.txt): the loader-wave persistent kernel wins wherever the
// read-out is plain (modes 0 / 1, the implicit convolutions: 0.65-0.97x the time, the skinny K = N = 192 projection excepted) and on
// the long contractions (K > 768) with a residual or GELU tail; the two-workgroup form of gemm_nt keeps the K <= 768 GEMMs whose
// tails read a cold operand or write two tensors (its second workgroup's main loop hides them).  dgx_dev_set("gemm_lw", 0 | 1) forces
// gemm_nt | gemm_lw everywhere (tests and A/B tools only).
static bool use_lw(const GemmP& P) {
    if (g_dev.lw >= 0) return g_dev.lw == 1;
    if (P.mode <= 1) return !(P.N <= 192 && P.K <= 192);
    if (P.mode == 2 || P.mode == 3) return P.K > 768;
    return false;
}
// tiling fields of a bm x bn launch (what launch_gemm<> computes for its instantiation)
static void plan_tiles(GemmP& P, int bm, int bn) {
    const int tiles_m = (P.M + bm - 1) / bm;
    P.tiles_n = (P.N + bn - 1) / bn;
    P.total = tiles_m * P.tiles_n;
    const int nt = (P.K + GBK - 1) / GBK;
    P.splits = choose_splits(P.total, P.K, P.M, P.N, P.ws ? g_ws_bytes_cur : 0);
    P.kt_per_split = (nt + P.splits - 1) / P.splits;
    P.splits = (nt + P.kt_per_split - 1) / P.kt_per_split;
    P.per_xcd = (P.total * P.splits + 7) / 8;
}
static int launch_lw(GemmP& P, int bm, int bn, hipStream_t st) {
    plan_tiles(P, bm, bn);
    const int rc = gemm_lw_launch(P, bm, bn, st);
    if (rc != DGX_OK) return rc;
    g_last = {1, bm, bn, P.splits};
    if (P.splits > 1) {
        const int64_t chunks = (int64_t)P.M * (P.N >> 3);
        const int grid = (int)((chunks + 255) / 256 < 4096 ? (chunks + 255) / 256 : 4096);
        hipLaunchKernelGGL(gemm_splitk_fold_kernel, dim3(grid), dim3(256), 0, st, P);
    }
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
static int dgx_gemm_dispatch(GemmP& P, hipStream_t st);
static bool use_two_wg(const GemmP& P) {          // see dgx_gemm_dispatch
    return g_dev.two_wg != 0 && P.N % 192 == 0 && P.K <= 768 && P.M >= (g_dev.two_wg >= 2 ? g_dev.two_wg : 4096) && !P.conv_kc;
}
static void* g_dbg_buffer = nullptr;
static FILE* g_gemm_log = nullptr;     // one line per launch (dgx_dev_gemm_log), joined with a kernel trace by tools/gemm_insitu.py
static FILE* gemm_log_file() { return g_gemm_log; }
extern "C" void dgx_dev_gemm_set_debug(void* device_buffer) { g_dbg_buffer = device_buffer; }
extern "C" int dgx_dev_gemm_log(const char* path) {
    if (g_gemm_log) { fclose(g_gemm_log); g_gemm_log = nullptr; }
    if (path && *path && !(g_gemm_log = fopen(path, "w"))) return DGX_ERR_BAD_ARG;
    return DGX_OK;
}
extern int g_dgx_dev_wgrad_lw;      // wgrad_lw.hip
extern "C" int dgx_dev_set(const char* key, int value) {
    if (!key) return DGX_ERR_BAD_ARG;
    if (!strcmp(key, "gemm_lw")) g_dev.lw = value;                   // -1 plan, 0 gemm_nt everywhere, 1 gemm_lw everywhere
    else if (!strcmp(key, "gemm_2wg")) g_dev.two_wg = value;         // -1 / 1 plan, 0 never the two-workgroup form, >= 2: its row threshold
    else if (!strcmp(key, "gemm_tile")) { g_dev.tile_bm = value / 1000; g_dev.tile_bn = value % 1000; }     // bm * 1000 + bn, 0 = plan
    else if (!strcmp(key, "gemm_splitk")) g_dev.splitk = value;      // 0 plan, >= 1 forced slab count
    else if (!strcmp(key, "gemm_k192")) g_dev.k192 = value;          // -1 / 1 plan, 0 never the resident-panel kernel (gemm_k192.hip)
    else if (!strcmp(key, "wgrad_lw")) g_dgx_dev_wgrad_lw = value;   // 1 plan, 0 never the loader-wave form, 2 always
    else if (!strcmp(key, "reset")) { g_dev = DevKnobs(); g_dgx_dev_wgrad_lw = 1; }
    else return DGX_ERR_BAD_ARG;
    return DGX_OK;
}
extern "C" int dgx_gemm_last_form(int* bm, int* bn, int* splits) {
    if (bm) *bm = g_last.bm;
    if (bn) *bn = g_last.bn;
    if (splits) *splits = g_last.splits;
    return g_last.form;
}

extern "C" int dgx_gemm_bf16_nt(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb,
                                const dgx_gemm_epilogue* ep, void* stream) {
    if (M <= 0 || N <= 0) return DGX_OK;
    if (!A || !B || !ep || K <= 0 || (K & 7) || (N & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) return DGX_ERR_BAD_ARG;
    if ((int64_t)M * lda * 2 >= (1ll << 31) || (int64_t)N * ldb * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    GemmP P;
    memset((void*)&P, 0, sizeof(P));
    P.A = (const uint16_t*)A; P.B = (const uint16_t*)B;
    P.M = M; P.N = N; P.K = K; P.lda = (int)lda; P.ldb = (int)ldb;
    P.mode = ep->mode;
    P.C = (uint16_t*)ep->c; P.ldc = (int)ep->ldc;
    P.bias = (const uint16_t*)ep->bias;
    P.C2 = (uint16_t*)ep->c2;
    P.aux = (const uint16_t*)ep->aux; P.ldaux = (int)ep->ldaux;
    P.res = ep->residual; P.out = ep->out; P.scale = ep->scale; P.res_dtype = ep->residual_dtype;
    P.ws = (float*)ep->workspace;
    g_ws_bytes_cur = ep->workspace ? ep->workspace_bytes : 0;
    P.relu = (ep->mode <= DGX_EPI_BIAS) ? ep->relu : 0;
#ifdef DGX_GEMM_DEV
    if (const char* kr = getenv("DGX_GEMM_KROT")) P.krot = atoi(kr);     // development build only (tools/gemm_cold_probe2.py)
#endif
    switch (ep->mode) {
        case DGX_EPI_NONE: case DGX_EPI_BIAS:
            if (!ep->c || ep->ldc < N || (ep->ldc & 7)) return DGX_ERR_BAD_ARG;
            if (ep->mode == DGX_EPI_NONE) P.bias = nullptr;
            break;
        case DGX_EPI_BIAS_GELU:
            if (!ep->c || !ep->c2 || ep->ldc < N || (ep->ldc & 7)) return DGX_ERR_BAD_ARG;
            break;
        case DGX_EPI_GELU_GRAD: case DGX_EPI_RELU_GRAD:
            if (!ep->c || !ep->aux || ep->ldc < N || ep->ldaux < N || (ep->ldc & 7) || (ep->ldaux & 7)) return DGX_ERR_BAD_ARG;
            P.bias = nullptr;
            break;
        case DGX_EPI_BIAS_RESIDUAL: {
            // ws < 0: the rows are in COMPACT window order (winmap.h: real tokens only, M = B*H*W)
            const int aws = ep->ws < 0 ? -ep->ws : ep->ws;
            if (!ep->residual || !ep->out || ep->B <= 0 || ep->H <= 0 || ep->W <= 0 || ep->B > 4095 || ep->shift < 0 ||
                (aws > 0 && ep->shift >= aws) || (ep->residual_dtype != DGX_F32 && ep->residual_dtype != DGX_BF16) ||
                (ep->ws < 0 && !wm_compact_ok(ep->H, ep->W, aws, ep->shift)))
                return DGX_ERR_BAD_ARG;
            GMap m = {ep->B, ep->H, ep->W, aws, ep->shift, 0, 0, ep->ws < 0 ? 1 : 0};
            if (aws > 0) { m.nWh = (ep->H + aws - 1) / aws; m.nWw = (ep->W + aws - 1) / aws; }
            const int64_t rows = ep->ws > 0 ? (int64_t)ep->B * m.nWh * m.nWw * aws * aws : (int64_t)ep->B * ep->H * ep->W;
            if (rows != M) return DGX_ERR_BAD_ARG;
            P.map = m;
            break;
        }
        default: return DGX_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const TileChoice tc = choose_tile(M, N);
    P.dbg = (unsigned long long*)g_dbg_buffer;
    // algorithmic traffic: both operands once, every result tensor once (residual mode: residual in, sum out, no C)
    const double mn = (double)M * N, rsz = ep->residual_dtype == DGX_F32 ? 4.0 : 2.0;
    const double obytes = ep->mode == DGX_EPI_BIAS_RESIDUAL ? 2.0 * rsz * mn : (ep->mode >= DGX_EPI_BIAS_GELU ? 4.0 * mn : 2.0 * mn);   // GELU / GELU' / ReLU': two tensors
    DgxProfScope prof(DGX_PROF_GEMM_NT, stream, 2.0 * mn * K, 2.0 * ((double)M * K + (double)N * K) + obytes);
    if (FILE* lf = gemm_log_file()) { fprintf(lf, "%d %d %d %d %d %d\n", M, N, K, ep->mode, (use_two_wg(P) && !use_lw(P)) ? 128 : tc.bm, tc.bn); fflush(lf); }
#ifdef DGX_GEMM_DEV
    if (const char* dg = getenv("DGX_GEMM_DIAG")) {
        switch (atoi(dg)) {
            case 1: return launch_gemm<256, 192, 2, 2, 1>(P, st);
            case 2: return launch_gemm<256, 192, 2, 2, 2>(P, st);
            case 3: return launch_gemm<256, 192, 2, 2, 3>(P, st);
            case 4: return launch_gemm<256, 192, 2, 2, 4>(P, st);
            case 5: return launch_gemm<256, 192, 2, 2, 5>(P, st);
            case 6: return launch_gemm<256, 192, 2, 2, 6>(P, st);
            case 7: return launch_gemm<256, 192, 2, 2, 7>(P, st);
            case 200: return launch_gemm<192, 192, 3, 2, 0>(P, st);
            case 300: return launch_gemm<128, 192, 4, 2, 0>(P, st);
            case 302: return launch_gemm<128, 192, 3, 2, 0>(P, st);
            case 400: return launch_gemm<256, 128, 3, 2, 0>(P, st);
            default: break;
        }
    }
#endif
    return dgx_gemm_dispatch(P, st);
}

bool dgx_gemm_k192_takes(const GemmP& P);              // gemm_k192.hip: the HBM-bound K = 192 problems of Swin stage 0
int dgx_gemm_k192_launch(const GemmP& P, hipStream_t st);
static int dgx_gemm_dispatch(GemmP& P, hipStream_t st) {
    if (g_dev.k192 != 0 && g_dev.lw < 0 && g_dev.tile_bm == 0 && g_dev.splitk == 0 && g_dev.two_wg < 0 && dgx_gemm_k192_takes(P)) {
        const int rc = dgx_gemm_k192_launch(P, st);
        if (rc != DGX_ERR_UNSUPPORTED) {
            g_last = {3, 32, 192, 1};
            return rc;
        }
    }
    const TileChoice tc = choose_tile(P.M, P.N);
    if (use_lw(P)) return launch_lw(P, tc.bm, tc.bn, st);
    if (tc.bn == 192) {
        // contractions of up to 12 K-tiles (K <= 768: every qkv / proj / fc1 / fc2-input-gradient GEMM of the backbone) spend a third of
        // a tile's time in prologue and read-out: TWO workgroups share a CU there (128 x 192 tiles, 2 stages = 80 KB of LDS, 128
        // registers per lane), so one's read-out -- with its GELU / GELU' / residual tail -- runs beside the other's main loop.
        // Round 2 measured this geometry back to back with the bias tail only (-4 % at K = 768, +10..30 % at long K) and dropped it;
        // inside the step, where the tails are the real ones, it wins wherever K <= 768 (round 3, same call: GEMM family
        // 11.80 -> 11.18 ms/step; K <= 384 only: 11.58; every K: 11.70) and loses on the long contractions, which keep the deeper rings.
        // DGX_GEMM_2WG=0 switches it off (A/B).
        if (use_two_wg(P)) {
            switch (P.mode) {            // the tails this form exists for, each with its mode compiled in
                case 2: return launch_gemm<128, 192, 2, 4, 0, 2>(P, st);
                case 3: return P.res_dtype == DGX_BF16 ? launch_gemm<128, 192, 2, 4, 0, 3>(P, st) : launch_gemm<128, 192, 2, 4, 0, 6>(P, st);
                case 4: return launch_gemm<128, 192, 2, 4, 0, 4>(P, st);
                default: return launch_gemm<128, 192, 2, 4>(P, st);
            }
        }
        if (tc.bm == 256) return launch_gemm<256, 192, 2>(P, st);
        if (tc.bm == 192) return launch_gemm<192, 192, 3>(P, st);
        return launch_gemm<128, 192, 4>(P, st);
    }
    if (tc.bn == 256 && tc.bm == 192) return launch_gemm<192, 256, 2>(P, st);
    if (tc.bn == 256) return launch_gemm<128, 256, 3>(P, st);
    if (tc.bm == 256) return launch_gemm<256, 128, 3>(P, st);
    return launch_gemm<128, 128, 4>(P, st);
}

extern "C" int64_t dgx_conv3x3_pad_rows(int N, int H, int W);
static int launch_lw_grouped(GemmP& P, const int* Ms, int n, int bm, int bn, hipStream_t st) {
    P.tiles_n = (P.N + bn - 1) / bn;
    int tot = 0;
    for (int i = 0; i < n; ++i) {
        P.grp[i].tile0 = tot;
        tot += ((Ms[i] + bm - 1) / bm) * P.tiles_n;
    }
    P.total = tot;
    P.splits = 1;
    P.kt_per_split = (P.K + GBK - 1) / GBK;
    P.per_xcd = (P.total + 7) / 8;
    const int rc = gemm_lw_launch(P, bm, bn, st);
    if (rc != DGX_OK) return rc;
    g_last = {1, bm, bn, P.splits};
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
namespace {
template <int BM, int BN, int NS, int MINW = 2>
int launch_gemm_grouped(GemmP& P, const int* Ms, int n, hipStream_t st) {
    using Cfg = GemmCfg<BM, BN, NS>;
    P.tiles_n = (P.N + BN - 1) / BN;
    int tot = 0;
    for (int i = 0; i < n; ++i) {
        P.grp[i].tile0 = tot;
        tot += ((Ms[i] + BM - 1) / BM) * P.tiles_n;
    }
    P.total = tot;
    P.splits = 1;
    P.kt_per_split = (P.K + GBK - 1) / GBK;
    P.per_xcd = (P.total + 7) / 8;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, NS, MINW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, NS, MINW, 0>), dim3(8 * P.per_xcd), dim3(512), Cfg::LDS, st, P);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
}  // namespace

// y_i = conv3x3(x_i, w) (+ bias) for n <= 6 zero-bordered images that share the weights, in ONE launch (see GemmP::grp): the levels of
// a CenterNet tower layer.  Cout <= 256 (one column tile of the instantiations used here).
extern "C" int dgx_conv3x3_gemm_multi(const dgx_conv_item* items, int n, const void* w, const void* bias, int Cin, int Cout, int relu,
                                      void* stream) {
    if (n <= 0) return DGX_OK;
    if (!items || !w || Cin <= 0 || Cout <= 0 || (Cin & 63) || (Cout & 7)) return DGX_ERR_BAD_ARG;
    if (n > dgxgemm::GEMM_MAXG || (int64_t)Cout * 9 * Cin * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    GemmP P;
    memset((void*)&P, 0, sizeof(P));
    P.B = (const uint16_t*)w;
    P.N = Cout; P.K = 9 * Cin; P.lda = Cin; P.ldb = 9 * Cin;
    P.mode = bias ? DGX_EPI_BIAS : DGX_EPI_NONE;
    P.ldc = Cout;
    P.bias = (const uint16_t*)bias;
    P.conv_kc = Cin / GBK;
    P.relu = relu;
    P.ngrp = n;
    int Ms[dgxgemm::GEMM_MAXG];
    double fl = 0.0, by = 2.0 * 9.0 * Cin * Cout;
    for (int i = 0; i < n; ++i) {
        const dgx_conv_item& a = items[i];
        if (!a.xpad || !a.y || a.N <= 0 || a.H <= 0 || a.W <= 0) return DGX_ERR_BAD_ARG;
        if (dgx_conv3x3_pad_rows(a.N, a.H, a.W) * Cin * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
        Ms[i] = a.N * a.H * a.W;
        P.grp[i].A = (const uint16_t*)a.xpad; P.grp[i].C = (uint16_t*)a.y; P.grp[i].M = Ms[i];
        P.grp[i].cn = a.N; P.grp[i].ch = a.H; P.grp[i].cw = a.W; P.grp[i].wp = a.W + 2;
        const double pix = (double)a.N * a.H * a.W;
        fl += 2.0 * pix * Cout * 9.0 * Cin;
        by += 2.0 * ((double)a.N * (a.H + 2) * (a.W + 2) * Cin + pix * Cout);
    }
    for (int i = n; i < dgxgemm::GEMM_MAXG; ++i) P.grp[i] = P.grp[0];
    // first group's fields also in the single-image slots (the kernel overwrites them per tile)
    P.A = P.grp[0].A; P.C = P.grp[0].C; P.M = Ms[0];
    P.cmap_n = P.grp[0].cn; P.cmap_h = P.grp[0].ch; P.cmap_w = P.grp[0].cw; P.conv_wp = P.grp[0].wp;
    if (FILE* lf = gemm_log_file()) {              // one line per LAUNCH (tools/gemm_insitu.py joins lines with dispatches): M = all images' rows
        int msum = 0;
        for (int i = 0; i < n; ++i) msum += Ms[i];
        int t128 = 0, t192 = 0;
        for (int i = 0; i < n; ++i) { t128 += (Ms[i] + 127) / 128; t192 += (Ms[i] + 191) / 192; }
        const bool big = Cout > 128 && tile_192x256() && ((t192 + 255) / 256) * 192 < ((t128 + 255) / 256) * 128;
        fprintf(lf, "%d %d %d %d %d %d\n", msum, P.N, P.K, 9, big ? 192 : 128, Cout > 128 ? 256 : 128);
        fflush(lf);
    }
    DgxProfScope prof(DGX_PROF_GEMM_NT, stream, fl, by);
    hipStream_t st = (hipStream_t)stream;
    if (Cout > 256) return DGX_ERR_UNSUPPORTED;
    if (Cout > 128) {
        // 192-row tiles when they save a round of the chip (the five tower levels at 1024^2 x 2 images: 341 tiles of 128 rows =
        // two rounds, 229 tiles of 192 rows = one)
        int t128 = 0, t192 = 0;
        for (int i = 0; i < n; ++i) { t128 += (Ms[i] + 127) / 128; t192 += (Ms[i] + 191) / 192; }
        const bool big = tile_192x256() && ((t192 + 255) / 256) * 192 < ((t128 + 255) / 256) * 128;
        if (g_dev.lw != 0) return launch_lw_grouped(P, Ms, n, big ? 192 : 128, 256, st);
        if (big) return launch_gemm_grouped<192, 256, 2>(P, Ms, n, st);
        return launch_gemm_grouped<128, 256, 3>(P, Ms, n, st);
    }
    if (g_dev.lw != 0) return launch_lw_grouped(P, Ms, n, 128, 128, st);
    return launch_gemm_grouped<128, 128, 4>(P, Ms, n, st);
}

extern "C" int64_t dgx_conv3x3_pad_rows(int N, int H, int W) {
    return (N <= 0 || H <= 0 || W <= 0) ? 0 : (int64_t)N * (H + 2) * (W + 2) + 2 * (int64_t)(W + 3);
}

extern "C" int dgx_conv3x3_gemm(const void* xpad, const void* w, const void* bias, void* y, int N, int H, int W, int Cin, int Cout,
                                int relu, void* workspace, int64_t workspace_bytes, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!xpad || !w || !y || Cin <= 0 || Cout <= 0 || (Cin & 63) || (Cout & 7)) return DGX_ERR_BAD_ARG;
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    if (dgx_conv3x3_pad_rows(N, H, W) * Cin * 2 >= (1ll << 31) || (int64_t)Cout * 9 * Cin * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    GemmP P;
    memset((void*)&P, 0, sizeof(P));
    P.A = (const uint16_t*)xpad;                   // row 0 of the GEMM = padded position 0 minus (Wp + 1): the leading slack
    P.B = (const uint16_t*)w;
    P.M = N * H * W; P.N = Cout; P.K = 9 * Cin; P.lda = Cin; P.ldb = 9 * Cin;
    P.mode = bias ? DGX_EPI_BIAS : DGX_EPI_NONE;
    P.C = (uint16_t*)y; P.ldc = Cout;
    P.bias = (const uint16_t*)bias;
    P.conv_kc = Cin / GBK; P.conv_wp = W + 2;
    P.cmap_n = N; P.cmap_h = H; P.cmap_w = W;
    P.relu = relu;
    P.ws = (float*)workspace;                      // small FPN levels: few tiles, 9 Cin / 64 K-tiles -> split-K slabs + fold
    g_ws_bytes_cur = workspace ? workspace_bytes : 0;
    if (FILE* lf = gemm_log_file()) {
        const TileChoice tc = choose_tile(P.M, P.N);
        fprintf(lf, "%d %d %d %d %d %d\n", P.M, P.N, P.K, 9, tc.bm, tc.bn); fflush(lf);
    }
    const double pix = (double)N * H * W;       // useful work: the H x W interior (border rows of the padded grid are overhead)
    DgxProfScope prof(DGX_PROF_GEMM_NT, stream, 2.0 * pix * Cout * 9.0 * Cin,
                      2.0 * ((double)Mp * Cin + 9.0 * Cin * Cout + pix * Cout));
    return dgx_gemm_dispatch(P, (hipStream_t)stream);
}

// Development timing hook (not part of include/divergen_hip.h): `iters` back-to-back launches of the same GEMM between two
// HIP events on `stream`; returns the average kernel time in microseconds (host launch cost excluded), < 0 on error.
extern "C" float dgx_dev_gemm_time_us(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb,
                                      const dgx_gemm_epilogue* ep, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 3; ++i)
        if (dgx_gemm_bf16_nt(A, B, M, N, K, lda, ldb, ep, stream) != DGX_OK) return -1.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2.f;
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) (void)dgx_gemm_bf16_nt(A, B, M, N, K, lda, ldb, ep, stream);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms * 1000.f / (float)iters;
}
