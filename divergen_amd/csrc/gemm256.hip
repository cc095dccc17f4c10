// bf16 GEMM for gfx950, large-tile form:  C[M][N] = epi( sum_k A[m][k] * B[n][k] )  -- same contract and fused tails as
// gemm_nt.hip (every Linear / 1x1 forward and input gradient of the reference: swintransformer.py:133,155,40-46,296), for the
// shapes where a 256 x BN tile per compute unit fills the chip (N >= 1152 or very long M).
//
// Why a second kernel: gemm_nt's 8-wave, 2-stage LDS-direct main loop was measured load-latency bound (one 56 KB K-tile in
// flight per CU against ~1 900 cycles of latency in situ, 2 190 cycles per K-tile against 1 536 of MFMA issue) and LDS-bandwidth
// bound (232 KB of LDS traffic per K-tile).  Here:
//   * 4 waves (one per SIMD), wave tile 128 x BN/2 = 8 x BN/32 MFMA 16x16x32 tiles, accumulators pinned to the AGPR half of the
//     register file (256 registers at BN = 256), operands in the 256 architectural VGPRs: fragment reads per K-tile drop from
//     176 KB to 128 KB (BN = 256: 64 MFMAs per 16 fragment reads) -- the main loop is MFMA-bound on the LDS side;
//   * the A operand (activations: cold in HBM / the memory-side cache, every tile column of an XCD misses on it together) goes
//     global -> VGPR -> LDS (`buffer_load_dwordx4` + `ds_write_b128`) with TWO K-tiles of register staging per wave: a load is
//     issued ~1.5 K-tile periods (~3 000 cycles) before its data is written to LDS -- the flight time LDS capacity cannot buy
//     (3 stages of a 256-row tile do not fit 160 KB); the B operand (weights: L2 resident) stays on LDS-direct loads with one
//     period of flight.  Both land in the same XOR-swizzled 128-byte-row image gemm_nt uses (conflict-free ds_read_b128);
//   * one barrier per K-tile: fragments of sub-step (t, kh) are read into the spare fragment set under the MFMAs of the
//     previous sub-step, so a stage buffer is free as soon as the second half of its tile has been READ, not multiplied;
//   * per-group issue order {1 memory op, 1-2 fragment reads, 4 MFMAs} pinned by scheduling barriers (as wgrad256.hip);
//   * read-out: each wave passes its 16-row strips through a private LDS staging area (bf16, bias added) and streams whole
//     row segments (256 / 192 B) out with the fused tails of gemm_common.h.
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {
__device__ __forceinline__ void mfma16_acc(f32x4& c, bf16x8 a, bf16x8 b) {      // accumulator in an AGPR quad
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int V> using IC = std::integral_constant<int, V>;
}  // namespace

// DIAG (development builds, -DDGX_GEMM_DEV): ablation bits -- 1 no A register loads, 2 no ds_write of A, 4 no LDS-direct B loads,
// 8 no fragment reads (results are then wrong; only the timing stamps are read)
template <int BN, int DIAG = 0>
__global__ __launch_bounds__(256) void gemm256_kernel(GemmP P) {
    constexpr int BM = 256;
    constexpr int WNF = BN / 32;                   // 16-column fragments per wave along N (2 waves): 8 | 6
    constexpr int NBL = BN / 32;                   // LDS-direct B loads per wave per K-tile (BN / 8 row groups over 4 waves)
    constexpr int SB = (BM + BN) * 128;            // bytes per stage (A rows, then B rows; 128-byte rows)
    constexpr int BOFF = BM * 128;
    constexpr int SROW = BN + 16;                  // staging row stride (bytes): BN / 2 bf16 columns + 16 B
    constexpr int STG0 = 2 * SB;                   // staging: 4 waves x 16 rows
    constexpr int TOK0 = STG0 + 4 * 16 * SROW;     // mode 3: (token << 12 | sample) per tile row
    constexpr int NMF = 8 * WNF;                   // MFMAs per sub-step (K = 32)
    constexpr int NG = NMF / 4;                    // issue groups per sub-step
    constexpr int NFR = 8 + WNF;                   // fragment reads per sub-step
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int L = (blockIdx.x & 7) * P.per_xcd + (blockIdx.x >> 3);      // XCD-aware order: an XCD's workgroups share A row panels
    if (L >= P.total) return;
#define G2CLK(i) do { if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    G2CLK(0);
    const int tm = L / P.tiles_n, tn = L - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int c = l & 15, g = l >> 4;
    const int NT = (P.K + GBK - 1) / GBK;
    const int ktail = P.K - (NT - 1) * GBK;

    // ---- loader role.  Row group q = w + 4 s of an operand tile = rows 8q .. 8q + 7; lane -> row 8q + (l >> 3), physical chunk
    // l & 7 = logical chunk ^ ((row >> 1) & 7), (q & 1) == (w & 1); LDS image of a group: 1 KiB, lane-linear
    const int rsub = l >> 3;
    const int lc = (l & 7) ^ (((w & 1) << 2) | (rsub >> 1));
    const bool kt_ok = lc * 8 < ktail;
    uint32_t voffA[8], voffB[NBL];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int m = m0 + 8 * (w + 4 * s) + rsub;
        voffA[s] = m < P.M ? (uint32_t)(((int64_t)m * P.lda + lc * 8) * 2) : G_OOB;
    }
#pragma unroll
    for (int s = 0; s < NBL; ++s) {
        const int n = n0 + 8 * (w + 4 * s) + rsub;
        voffB[s] = n < P.N ? (uint32_t)(((int64_t)n * P.ldb + lc * 8) * 2) : G_OOB;
    }
    const uint32_t bytesA = (uint32_t)((int64_t)P.M * P.lda * 2), bytesB = (uint32_t)((int64_t)P.N * P.ldb * 2);
    auto srsrc = [](u32x4 r) __attribute__((always_inline)) {
        return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)r[0]), (uint32_t)__builtin_amdgcn_readfirstlane((int)r[1]),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)r[2]), (uint32_t)__builtin_amdgcn_readfirstlane((int)r[3])};
    };
    const u32x4 rB = srsrc(g_rsrc(P.B, bytesB));
    const u32x4 rAl = srsrc(g_rsrc(P.A, bytesA));
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)bytesA, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + 1024u * w);
    DGX_LDS unsigned char* const lane_wr = (DGX_LDS unsigned char*)lds_raw + 1024 * w + 16 * l;     // this lane's slot in a row group

    // K-tiles beyond the problem (the look-ahead of the last tiles) and chunks beyond K in the last tile read as zeros: every load
    // of the main loop is issued unconditionally, so there is ONE loop body and no branch around a load
    auto tail_mask = [&](int kt, uint32_t v) __attribute__((always_inline)) {
        return (kt >= NT || ((kt == NT - 1) && (ktail != GBK) && !kt_ok)) ? G_OOB : v;
    };
    auto sgpr = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };     // wave-uniform values, pinned scalar
    auto issue_b = [&](int s, int kt, int stage) __attribute__((always_inline)) {       // LDS-direct: B row group w + 4 s of K-tile kt
        g_load_lds16(tail_mask(kt, voffB[s]), rB, sgpr(ldsw + (uint32_t)stage * SB + BOFF + 4096u * s), sgpr((uint32_t)kt * (GBK * 2)));
    };
    auto issue_a_lds = [&](int s, int kt, int stage) __attribute__((always_inline)) {   // LDS-direct form of the A loads (prologue only)
        g_load_lds16(tail_mask(kt, voffA[s]), rAl, sgpr(ldsw + (uint32_t)stage * SB + 4096u * s), sgpr((uint32_t)kt * (GBK * 2)));
    };
    auto load_a_reg = [&](int s, int kt) __attribute__((always_inline)) -> u32x4 {      // register form: returned value is tracked by the compiler's vmcnt
        return __builtin_amdgcn_raw_buffer_load_b128(rA, (int)tail_mask(kt, voffA[s]), kt * (GBK * 2), 0);
    };
    auto store_a = [&](int s, int stage, u32x4 v) __attribute__((always_inline)) {
        *reinterpret_cast<DGX_LDS u32x4*>(lane_wr + stage * SB + 4096 * s) = v;
    };

    // ---- MFMA role: fragment = 16 rows x one k-half, one ds_read_b128 per lane: row c, logical chunk g + 4 kh
    const int swz = (c >> 1) & 7;
    const uint32_t la = (uint32_t)((wm * 128 + c) * 128 + ((g ^ swz) << 4));
    const uint32_t lb = (uint32_t)(BOFF + (wn * (BN / 2) + c) * 128 + ((g ^ swz) << 4));
    f32x4 acc[8][WNF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fx[NFR], fy[NFR];                       // fragment sets: [0..7] A rows 16 i.., [8..] B columns 16 j..
    u32x4 ra0[8], ra1[8];                          // A register staging: K-tiles of even / odd parity
    // fragment f of (stage, kh): f < 8 -> A fragment f, else B fragment f - 8
    auto read_frag = [&](DGX_LDS const unsigned char* pa, DGX_LDS const unsigned char* pb, int f, bf16x8 (&fr)[NFR]) __attribute__((always_inline)) {
        fr[f] = f < 8 ? *reinterpret_cast<DGX_LDS const bf16x8*>(pa + 2048 * f) : *reinterpret_cast<DGX_LDS const bf16x8*>(pb + 2048 * (f - 8));
    };
    // one sub-step: MFMAs of `cur`, fragment reads into `nxt` (if RD), one memory op per group chosen by MODE -- all compile-time,
    // so that no load sits behind a branch (the compiler's vmcnt bookkeeping turns conservative -- vmcnt(0) -- across branches):
    //   MODE 1: A register loads of K-tile `kt` into `ra` (8 groups);  MODE 2: ds_write of `ra` + LDS-direct B of K-tile kt
    // PH (0 | 1, by the wave's N half): the two wave classes place their memory operations in DIFFERENT issue groups -- all four
    // waves run in lock-step behind the barrier, and a memory instruction whose issue waits for the address path behind three
    // others idles this SIMD's matrix pipe (one wave per SIMD: nothing else can issue).  Measured (tools/gemm_phase_probe.py,
    // ablations): the 22-24 memory instructions of a K-tile cost ~700 cycles of issue on top of ~1 840 (BN 192) / ~2 350 (BN 256).
    auto substep = [&](const bf16x8 (&cur)[NFR], bf16x8 (&nxt)[NFR], auto rdc, auto rstagec, auto rkhc, auto modec, int kt, auto stagec,
                       u32x4 (&ra)[8], auto phc) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value;
        constexpr bool RD = decltype(rdc)::value != 0;
        constexpr int RSTAGE = decltype(rstagec)::value, RKH = decltype(rkhc)::value, MODE = decltype(modec)::value;
        constexpr int STAGE = decltype(stagec)::value;
        // lane pointers of the fragment reads: everything else is an immediate offset (2 KiB per fragment)
        DGX_LDS const unsigned char* pa = lds_opaque((const unsigned char*)lds_raw + RSTAGE * SB + (la ^ (RKH ? 64u : 0u)));
        DGX_LDS const unsigned char* pb = lds_opaque((const unsigned char*)lds_raw + RSTAGE * SB + (lb ^ (RKH ? 64u : 0u)));
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            // slot of this group in the wave class's schedule: 16 groups (BN 256): class 0 uses the even groups for its first
            // kind of operation and the odd ones for the second, class 1 the other way round; 12 groups (BN 192): one kind from the
            // front, the other from the back, swapped between the classes
            if constexpr (MODE == 1 && !(DIAG & 1)) {
                if constexpr (NG == 16) {
                    if ((gq & 1) == PH) ra[gq >> 1] = load_a_reg(gq >> 1, kt);
                } else {
                    if (gq >= 4 * PH && gq < 4 * PH + 8) ra[gq - 4 * PH] = load_a_reg(gq - 4 * PH, kt);
                }
            }
            if constexpr (MODE == 2) {
                if constexpr (NG == 16) {
                    if ((gq & 1) == PH && !(DIAG & 2)) store_a(gq >> 1, STAGE, ra[gq >> 1]);
                    if ((gq & 1) != PH && !(DIAG & 4)) issue_b(gq >> 1, kt, STAGE);
                } else if constexpr (PH == 0) {
                    if (gq < 8 && !(DIAG & 2)) store_a(gq, STAGE, ra[gq]);
                    if (gq >= NG - NBL && !(DIAG & 4)) issue_b(gq - (NG - NBL), kt, STAGE);
                } else {
                    if (gq < NBL && !(DIAG & 4)) issue_b(gq, kt, STAGE);
                    if (gq >= NG - 8 && !(DIAG & 2)) store_a(gq - (NG - 8), STAGE, ra[gq - (NG - 8)]);
                }
            }
            if constexpr (RD && !(DIAG & 8)) {
                read_frag(pa, pb, gq, nxt);
                if (gq + NG < NFR) read_frag(pa, pb, gq + NG, nxt);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * gq + u, i = e / WNF, j = e - i * WNF;
                mfma16_acc(acc[i][j], cur[8 + j], cur[i]);          // D[n][m]: a lane holds 4 consecutive columns of one row
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: tiles 0, 1 straight into the stages (LDS-direct), tile 2 into registers
#pragma unroll
    for (int s = 0; s < 8; ++s) issue_a_lds(s, 0, 0);
#pragma unroll
    for (int s = 0; s < NBL; ++s) issue_b(s, 0, 0);
#pragma unroll
    for (int s = 0; s < 8; ++s) issue_a_lds(s, 1, 1);
#pragma unroll
    for (int s = 0; s < NBL; ++s) issue_b(s, 1, 1);
    // (always issued: a K-tile beyond the problem reads as zeros and is never written to LDS -- keeps the loads branch-free)
#pragma unroll
    for (int s = 0; s < 8; ++s) ra0[s] = load_a_reg(s, 2);
    g_vmcnt<8>();
    g_bar();
    G2CLK(1);
    {
        DGX_LDS const unsigned char* pa = lds_opaque((const unsigned char*)lds_raw + la);
        DGX_LDS const unsigned char* pb = lds_opaque((const unsigned char*)lds_raw + lb);
#pragma unroll
        for (int f = 0; f < NFR; ++f) read_frag(pa, pb, f, fx);
    }

    // ---- main loop: K-tile t = MFMAs of (t, kh 0) out of fx, then of (t, kh 1) out of fy.  Tile t lives in stage t & 1.
    //   (t, 0): read fy <- (t, kh 1); register loads of A(t + 3) -> ra[(t + 1) & 1]
    //   boundary: own B(t + 1) landed (only the A(t + 3) loads are younger), own LDS ops retired, barrier:
    //             every wave has read tile t, and A(t + 1) / B(t + 1) are visible
    //   (t, 1): read fx <- (t + 1, kh 0); A(t + 2): ra[t & 1] -> stage t & 1 (ds_write), then B(t + 2) LDS-direct -> stage t & 1
    auto ktile = [&](int t, auto parc, auto phc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        if constexpr (PAR == 0) substep(fx, fy, IC<1>{}, IC<PAR>{}, IC<1>{}, IC<1>{}, t + 3, IC<0>{}, ra1, phc);
        else substep(fx, fy, IC<1>{}, IC<PAR>{}, IC<1>{}, IC<1>{}, t + 3, IC<0>{}, ra0, phc);
        // s_waitcnt through the builtin: the compiler's own counter model then knows that every LDS read is back and that at
        // most the 8 register loads just issued are in flight (it would otherwise re-wait in front of the first MFMA / ds_write)
        __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0)
        __builtin_amdgcn_s_waitcnt(0x0F78);        // vmcnt(8)
        g_bar();
        if constexpr (DIAG & 16) {                 // dev: the waves of one N half start each K-tile ~64 cycles late (de-correlated issue)
            if (wn) __builtin_amdgcn_s_sleep(1);
        }
        if constexpr (PAR == 0) substep(fy, fx, IC<1>{}, IC<(PAR ^ 1)>{}, IC<0>{}, IC<2>{}, t + 2, IC<PAR>{}, ra0, phc);
        else substep(fy, fx, IC<1>{}, IC<(PAR ^ 1)>{}, IC<0>{}, IC<2>{}, t + 2, IC<PAR>{}, ra1, phc);
    };
    auto mainloop = [&](auto phc) __attribute__((always_inline)) {
        int t = 0;
        for (; t + 2 <= NT; t += 2) {
            ktile(t, IC<0>{}, phc);
            ktile(t + 1, IC<1>{}, phc);
        }
        if (t < NT) ktile(t, IC<0>{}, phc);
    };
    mainloop(IC<0>{});
    g_vmcnt<0>();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA -> accumulator read-out hazard
    G2CLK(2);

    // ---- read-out.  Mode 3: row -> (token, sample) table for the whole tile first (one thread per row)
    DGX_LDS int64_t* rowtok = reinterpret_cast<DGX_LDS int64_t*>((DGX_LDS unsigned char*)lds_raw + TOK0);
    if (P.mode == 3) {
        int b = 0;
        const int64_t orow = (int64_t)m0 + tid;
        const int64_t tok = orow < P.M ? g_row_token(P.map, orow, b) : -1;
        rowtok[tid] = tok < 0 ? -1 : ((tok << 12) | (int64_t)b);
        __syncthreads();
    }
    DGX_LDS unsigned char* stg = (DGX_LDS unsigned char*)lds_raw + STG0 + w * (16 * SROW);
    float bv[WNF][4];
#pragma unroll
    for (int j = 0; j < WNF; ++j) {
        const int n = n0 + wn * (BN / 2) + 16 * j + 4 * g;
        uint32_t b01 = 0, b23 = 0;
        if (P.bias && n < P.N) {
            const u32x2 raw = *reinterpret_cast<const u32x2*>(P.bias + n);
            b01 = raw[0];
            b23 = raw[1];
        }
        bv[j][0] = __uint_as_float(b01 << 16); bv[j][1] = __uint_as_float(b01 & 0xffff0000u);
        bv[j][2] = __uint_as_float(b23 << 16); bv[j][3] = __uint_as_float(b23 & 0xffff0000u);
    }
    constexpr int CPR = BN / 16;                   // 16-byte chunks per strip row (BN / 2 columns)
    constexpr int NCH = 16 * CPR / 64;             // chunks per lane per strip: 4 | 3
    // one instantiation per fused tail: with the mode a compile-time constant the strip loop stays small enough to be unrolled
    // completely (a rolled loop would index the accumulators dynamically and push all of them to scratch memory)
    auto readout = [&](auto modec) __attribute__((always_inline)) {
        GemmP Q = P;
        Q.mode = decltype(modec)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < WNF; ++j) {
                const f32x4 a = acc[i][j];
                const u32x2 pk = {pack_bf2(a[0] + bv[j][0], a[1] + bv[j][1]), pack_bf2(a[2] + bv[j][2], a[3] + bv[j][3])};
                *reinterpret_cast<DGX_LDS u32x2*>(stg + c * SROW + (16 * j + 4 * g) * 2) = pk;
            }
            // wave-private staging: the wave's own LDS writes are ordered before its reads by lgkmcnt (no barrier)
            int gm[NCH], gn[NCH];
            bool ok[NCH];
            int64_t tok[NCH];
            float sc[NCH];
            u32x4 y[NCH], xa[NCH], xb[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = k * 64 + l, row = idx / CPR, ch = idx - row * CPR;
                const int trow = wm * 128 + 16 * i + row;
                gm[k] = m0 + trow;
                gn[k] = n0 + wn * (BN / 2) + 8 * ch;
                ok[k] = gm[k] < P.M && gn[k] < P.N;
                tok[k] = 0;
                sc[k] = 1.0f;
                xa[k] = xb[k] = u32x4{0u, 0u, 0u, 0u};
                if (Q.mode == 3 && ok[k]) {
                    const int64_t rt = rowtok[trow];
                    ok[k] = rt >= 0;
                    tok[k] = rt >> 12;
                    if (ok[k] && P.scale) sc[k] = P.scale[(int)(rt & 4095)];
                }
                if (ok[k]) g_epi_prefetch(Q, gm[k], gn[k], tok[k], xa[k], xb[k]);
                y[k] = *reinterpret_cast<DGX_LDS const u32x4*>(stg + row * SROW + ch * 16);
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (ok[k]) g_epi_finish(Q, gm[k], gn[k], y[k], tok[k], sc[k], xa[k], xb[k]);
        }
    };
    G2CLK(3);
    switch (P.mode) {
        case 2: readout(IC<2>{}); break;
        case 3: readout(IC<3>{}); break;
        case 4: readout(IC<4>{}); break;
        case 5: readout(IC<5>{}); break;
        default: readout(IC<1>{}); break;          // plain / bias (+ ReLU): bias already added to the staged tile
    }
    G2CLK(4);
}

bool gemm256_supported(const GemmP& P) {
    return P.conv_kc == 0 && P.M >= 256 && P.N >= 192 && P.K >= 64;
}

template <int BN, int DIAG = 0>
static int launch256(GemmP& P, hipStream_t st) {
    constexpr int LDS = 2 * (256 + BN) * 128 + 4 * 16 * (BN + 16) + 256 * 8;
    const int tiles_m = (P.M + 255) / 256;
    P.tiles_n = (P.N + BN - 1) / BN;
    P.total = tiles_m * P.tiles_n;
    P.splits = 1;
    P.per_xcd = (P.total + 7) / 8;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)gemm256_kernel<BN, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<BN, DIAG>), dim3(8 * P.per_xcd), dim3(256), LDS, st, P);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

int gemm256_launch(GemmP& P, int bn, hipStream_t st) {
#ifdef DGX_GEMM_DEV
    if (const char* dg = getenv("DGX_GEMM_DIAG")) {
        switch (atoi(dg)) {
            case 1: return bn == 256 ? launch256<256, 1>(P, st) : launch256<192, 1>(P, st);
            case 3: return bn == 256 ? launch256<256, 3>(P, st) : launch256<192, 3>(P, st);
            case 4: return bn == 256 ? launch256<256, 4>(P, st) : launch256<192, 4>(P, st);
            case 7: return bn == 256 ? launch256<256, 7>(P, st) : launch256<192, 7>(P, st);
            case 15: return bn == 256 ? launch256<256, 15>(P, st) : launch256<192, 15>(P, st);
            case 16: return bn == 256 ? launch256<256, 16>(P, st) : launch256<192, 16>(P, st);
            default: break;
        }
    }
#endif
    return bn == 256 ? launch256<256>(P, st) : launch256<192>(P, st);
}
