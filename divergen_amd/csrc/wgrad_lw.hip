// Grouped weight-gradient GEMM for gfx950, loader-wave persistent form:  for each problem p
//     C_p[Nn][Kk] = beta * C_p + sum_m A_p[m][Nn] * B_p[m][Kk]        (A = dY, B = X, row-major bf16; C fp32 in the gradient arena)
// -- the contract of wgrad256.hip (the weight gradients autograd forms for nn.Linear at swintransformer.py:36-46,101-108), for
// launches that hold MANY output tiles (the Linears of several Swin blocks of stages 2 / 3: 144 / 576 tiles of 256 x 192 per block).
//
// Why a second kernel.  wgrad256's four waves (256 x 256 tile, 256 AGPR accumulators each) issue their own LDS-direct loads: 8 per
// wave and 32-row stage at ~60 cycles of in-order issue each, next to 64 MFMAs (1 024 cycles) and 32 transpose reads -- 0.9-1.2
// PFLOP/s in situ.  With accumulators that size there is no room for a second wave per SIMD to take the loads.  The structure that
// gave the forward GEMM 84-87 % MFMA issue in its main loop (gemm_lw.hip: 1 772-1 836 cycles per K-tile of a 256 x 192 tile against
// 1 536 of MFMA issue) fits here as well, because the main loop is ALL there is to a weight gradient (M = 8 192 .. 10 368 rows = 128 ..
// 162 K-tiles per tile, the read-out is 2 % of it):
//   * 12 waves: waves 8-11 only load (LDS-direct, counted vmcnt, published by the workgroup barrier), waves 0-7 only multiply
//     (2 groups x 4: wave tile 128 (Nn) x 48 (Kk), 96 accumulator registers, <= 168 registers per lane);
//   * operand images as in memory, [m][n]: 64 rows x 512 B per K-tile and operand (the X image uses 384 of its 512 bytes per row, so
//     that both operands share one layout: 16-byte chunk c of row r at chunk c ^ ((r & 3) << 1) ^ (((r >> 3) & 1) << 3) -- the first term spreads
//     the four rows a 16-lane group reads, the second puts the rows of the odd lane groups into the other 128 bytes of the 256-byte bank
//     period, without it the two 16-lane groups of a 32-lane LDS pass meet on the same 32 banks (SQ_LDS_BANK_CONFLICT = half of all LDS
//     cycles, measured); fragments by ds_read_b64_tr_b16); rings of 3 (dY) + 2 (X) K-tiles = 160 KB;
//   * swapped MFMA operands (D = X-fragment^T x dY-fragment^T): a lane holds 4 consecutive Kk entries of one Nn row, the read-out is
//     a 16-byte fp32 read-modify-write per lane straight from the accumulators -- no LDS staging, so the loaders fill the rings with the
//     NEXT tile's first K-tiles while the finished tile is folded into the arena;
//   * one workgroup per CU walks a list of (problem, tile) items; NO M-split (every tile contracts its whole M: nothing to reduce,
//     no fp32 partial slabs); the bias gradient (column sums of dY) comes from the dY fragments of a problem's first tile column
//     through v_dot2 (8 registers instead of the 32 an MFMA-with-ones accumulator would take).
// The caller (layers/swin_block.py) queues the problems of ~7 blocks per launch so that the item count is a multiple of the CU count
// to within a few percent (7 stage-2 blocks = 1 008 items = 3.94 rounds).
#include "dgx_common.h"
#include <stdlib.h>
#include <string.h>

namespace {
constexpr int WL_TN = 256, WL_TK = 192;          // tile: Nn rows x Kk columns of the weight gradient
constexpr int WL_BM = 64;                        // m-depth of a K-tile (two K = 32 MFMA steps)
constexpr int WL_SLOT = WL_BM * 512;             // bytes of one operand image per K-tile (512-byte rows)
constexpr int WL_NSA = 3, WL_NSB = 2;
constexpr int WL_B0 = WL_NSA * WL_SLOT;
constexpr int WL_LDS = (WL_NSA + WL_NSB) * WL_SLOT;      // 160 KB
constexpr int WL_MAXP = 32;
constexpr int WL_THREADS = 768;
constexpr int WL_NLA = 8, WL_NLB = 8;            // LDS-direct loads per loader wave per K-tile and operand (32 instructions of 2 rows over 4 loaders)

struct WlProb {
    const uint16_t* A;
    const uint16_t* B;
    float* C;
    float* gb;
    int M, Nn, Kk, ldc, tiles_k, item0;
};
struct WlParams {
    WlProb p[WL_MAXP];               // first member: the kernel reads the table through the kernarg segment pointer (dynamic index, scalar loads)
    int item0[WL_MAXP];              // first item of problem i (searched with constant indices)
    int n, total, per_xcd;
    float beta;
    int diag;                        // DGX_WGRAD_LW_DIAG=1 (timing experiments only, results wrong): every workgroup streams the panels of item 0
};

__device__ __forceinline__ void wl_load_lds16(uint32_t voff, u32x4 rsrc, uint32_t lds_addr, uint32_t soff) {
#ifndef WL_NO_DMA              // (timing experiment without the loads)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
                 : "memory");
#endif
}
__device__ __forceinline__ u32x4 wl_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
template <int N> __device__ __forceinline__ void wl_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wl_bar() {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wl_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }
__device__ __forceinline__ uint32_t wl_sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
typedef __bf16 wl_bf2 __attribute__((ext_vector_type(2)));
}  // namespace

#ifdef WL_NO_STREAM            // timing experiment: the streamed dY fragments are not read (stale registers)
#define WL_STREAM(a, f)
#else
#define WL_STREAM(a, f) frag2(a, f)
#endif
#ifdef DGX_GEMM_DEV       // phase time stamps of workgroup 0 (development build only): [wave 0 | wave 4 | wave 8][K-tile][4 stamps], plain stores
constexpr int WL_CLK_KT = 1024;
__device__ unsigned long long wl_clk[3 * WL_CLK_KT * 4 + 1];
#define WLCLK(base, i) do { if (blockIdx.x == 0 && l == 0 && kt < WL_CLK_KT) wl_clk[(((base) >> 2) * WL_CLK_KT + kt) * 4 + (i)] = clock64(); } while (0)
#define WLCLK0() int kt = 0, itn = 0
#define WLCLKN() ++kt
__device__ unsigned long long wl_clk2[3 * 8 * 4];      // per wave, item, {after the last barrier, after the read-out, after barrier #0 of the next item, -}
#define WLCLKI(base, i) do { if (blockIdx.x == 0 && l == 0 && itn < 8) wl_clk2[(((base) >> 2) * 8 + itn) * 4 + (i)] = clock64(); } while (0)
#define WLCLKIN() ++itn
#else
#define WLCLK(base, i)
#define WLCLK0()
#define WLCLKN()
#define WLCLKI(base, i)
#define WLCLKIN()
#endif
__global__ __launch_bounds__(WL_THREADS) void wgrad_lw_kernel(WlParams P) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int nwx = (int)(gridDim.x >> 3);
    const int xcd = blockIdx.x & 7;
    const int bound = min((xcd + 1) * P.per_xcd, P.total);
    const int first = xcd * P.per_xcd + (int)(blockIdx.x >> 3);
    if (first >= bound) return;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;

    // item -> problem (the last one whose first item is <= L) and tile.  The search runs over the compact item0 array with constant
    // indices; the descriptor itself comes from the kernarg segment by scalar loads with a dynamic offset (indexing the by-value
    // struct dynamically would make the compiler copy it to private memory)
    typedef const __attribute__((address_space(4))) WlProb* TabPtr;
    const TabPtr tab = (TabPtr)__builtin_amdgcn_kernarg_segment_ptr();
    auto locate = [&](int L, WlProb& q, int& n0, int& k0) {
        L = __builtin_amdgcn_readfirstlane(L);
        int idx = 0;
#pragma unroll
        for (int i = 1; i < WL_MAXP; ++i)
            if (L >= P.item0[i]) idx = i;              // unused entries hold INT_MAX
        q.A = tab[idx].A; q.B = tab[idx].B; q.C = tab[idx].C; q.gb = tab[idx].gb;
        q.M = tab[idx].M; q.Nn = tab[idx].Nn; q.Kk = tab[idx].Kk; q.ldc = tab[idx].ldc; q.tiles_k = tab[idx].tiles_k; q.item0 = tab[idx].item0;
        const int t = L - q.item0;
        const int tn = t / q.tiles_k;
        n0 = tn * WL_TN;
        k0 = (t - tn * q.tiles_k) * WL_TK;
    };
    auto a_slot = [](int s) -> uint32_t { return (uint32_t)s * WL_SLOT; };

    // Barrier numbering per item as in gemm_lw.hip: #0 .. #(2 NT + 1); MFMA group 0 reads K-tile t (X fragments, first dY fragments)
    // in I_{2t+1} and multiplies (streaming the other dY fragments) in I_{2t+2}, group 1 one interval later; the X slot of tile t is
    // free behind #(2t+2), its dY slot behind #(2t+3); tile t must be visible at #(2t).
    if (w >= 8) {
        // ---------------------------------------------------------------- loader waves
        const int lw = w - 8;
        const int rip = l >> 5;                                       // row inside the 2-row instruction
        // logical 16-byte chunk of this lane in instruction s: row = 8 s + 2 lw + rip, row & 3 = 2 (lw & 1) + rip, (row >> 3) & 1 = s & 1
        const int lc0 = (l & 31) ^ (((2 * (lw & 1) + rip) & 3) << 1), lc1 = lc0 ^ 8;
        const uint32_t ldsq = wl_sgpr(lds0 + 1024u * lw);
        const uint32_t OOB = 0x80000000u;
        WlProb q;
        int n0, k0, NT = 0;
        uint32_t vA[2] = {OOB, OOB}, vB[2] = {OOB, OOB};       // even / odd instructions
        u32x4 rA, rB;
        auto setup = [&](int L) {
            locate((P.diag & 1) ? 0 : L, q, n0, k0);
            if (P.diag & 1) { WlProb q1; int a1, b1; locate(L, q1, a1, b1); q.M = q1.M; }
            NT = (q.M + WL_BM - 1) / WL_BM;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int lc = o ? lc1 : lc0;
                vA[o] = (n0 + 8 * lc < q.Nn) ? (uint32_t)(((2 * lw + rip + 8 * o) * q.Nn + n0 + 8 * lc) * 2) : OOB;
                vB[o] = (8 * lc < WL_TK && k0 + 8 * lc < q.Kk) ? (uint32_t)(((2 * lw + rip + 8 * o) * q.Kk + k0 + 8 * lc) * 2) : OOB;
            }
            rA = wl_rsrc(q.A, (uint32_t)((int64_t)q.M * q.Nn * 2));   // rows >= M are out of range: zeros
            rB = wl_rsrc(q.B, (uint32_t)((int64_t)q.M * q.Kk * 2));
        };
        // instruction s of this loader covers image rows 2 (lw + 4 s), +1 = tile rows 8 s + 2 lw (+1).  The row offset goes through the
        // VECTOR offset (an add per load, the loaders have nothing else to do) so that rows >= M fall under the descriptor's range check
        // and arrive as zeros; lanes beyond the matrix width keep bit 31 set through the adds (M * width * 2 < 2^31)
        auto issue_a = [&](int t) {
            if (t >= NT) return;
            const uint32_t dst = ldsq + a_slot(t % WL_NSA);
            const uint32_t step = (uint32_t)(16 * q.Nn * 2), at = (uint32_t)(t * WL_BM) * (uint32_t)(q.Nn * 2);
            uint32_t v0 = vA[0] + at, v1 = vA[1] + at;
#pragma unroll
            for (int s = 0; s < WL_NLA; s += 2, v0 += step, v1 += step) {
                wl_load_lds16(v0, rA, wl_sgpr(dst + 4096u * s), 0u);
                wl_load_lds16(v1, rA, wl_sgpr(dst + 4096u * (s + 1)), 0u);
            }
        };
        auto issue_b = [&](int t) {
            if (t >= NT) return;
            const uint32_t dst = ldsq + WL_B0 + (uint32_t)(t % WL_NSB) * WL_SLOT;
            const uint32_t step = (uint32_t)(16 * q.Kk * 2), at = (uint32_t)(t * WL_BM) * (uint32_t)(q.Kk * 2);
            uint32_t v0 = vB[0] + at, v1 = vB[1] + at;
#pragma unroll
            for (int s = 0; s < WL_NLB; s += 2, v0 += step, v1 += step) {
                wl_load_lds16(v0, rB, wl_sgpr(dst + 4096u * s), 0u);
                wl_load_lds16(v1, rB, wl_sgpr(dst + 4096u * (s + 1)), 0u);
            }
        };
        // loads of this wave younger than B(tau) (the later one of the pair: NSA > NSB) when tiles up to A(ia), B(ib) have been issued
        auto wait_tile = [&](int tau, int ia, int ib) {
            const int la = min(ia, NT - 1), lb = min(ib, NT - 1);
            const int n = max(0, la - tau) * WL_NLA + max(0, lb - tau) * WL_NLB;
            if (n >= 2 * WL_NLA + WL_NLB) wl_vmcnt<2 * WL_NLA + WL_NLB>();
            else if (n >= WL_NLA) wl_vmcnt<WL_NLA>();
            else wl_vmcnt<0>();
        };
        // Bias gradient (column sums of dY) of the items of a problem's first tile column, on the loader waves (they idle between their
        // loads; the MFMA waves have no register to spare): wave lw sums the 64 columns 64 lw .. of the dY image as it lands -- lane
        // (rg = l >> 4, cq = l & 15) the 4 columns 4 cq .. of rows = rg (mod 4), 8 rows per half tile, by 8-byte reads at the swizzled place
        const int rg = l >> 4, cq = l & 15;
        const uint32_t boff = (uint32_t)(rg * 512 + (((8 * lw + (cq >> 1)) ^ (rg << 1)) << 4) + (cq & 1) * 8);
        float bs[4] = {0.f, 0.f, 0.f, 0.f};
        auto bias_rows = [&](int t, int half) {
            DGX_LDS const unsigned char* sl = (DGX_LDS const unsigned char*)lds_raw + a_slot(t % WL_NSA) + half * (32 * 512);
            DGX_LDS const u32x2* pr0 = reinterpret_cast<DGX_LDS const u32x2*>(sl + boff);            // rows with (row >> 3) & 1 = 0
            DGX_LDS const u32x2* pr1 = reinterpret_cast<DGX_LDS const u32x2*>(sl + (boff ^ 128u));   // ... = 1: the other 128 bytes
#pragma unroll
            for (int k = 0; k < 8; ++k) {                      // row 4 k + rg
                const u32x2 v = ((k >> 1) & 1 ? pr1 : pr0)[k * (4 * 512 / 8)];
                bs[0] += __uint_as_float(v[0] << 16); bs[1] += __uint_as_float(v[0] & 0xffff0000u);
                bs[2] += __uint_as_float(v[1] << 16); bs[3] += __uint_as_float(v[1] & 0xffff0000u);
            }
        };
        setup(first);
        WLCLK0();
        for (int L = first; L < bound; L += nwx) {
            const bool do_bias = q.gb != nullptr && k0 == 0;
            // prologue in the order the schedule keeps: A(0), B(0), A(1), B(1), A(2).  Every slot is free here (the read-out uses no LDS)
            issue_a(0); issue_b(0); issue_a(1); issue_b(1); issue_a(2);
            wait_tile(0, WL_NSA - 1, WL_NSB - 1);
            wl_bar();                              // #0
            for (int t = 0; t < NT; ++t) {
                if (t >= 1) issue_b(t - 1 + WL_NSB);   // I_{2t+1}
                if (do_bias) bias_rows(t, 0);          // tile t is visible since #(2t); its dY slot is rewritten behind #(2t+3)
                if (lw == 0) WLCLK(8, 0);
                wl_bar();                          // #(2t+1)
                if (lw == 0) WLCLK(8, 1);
                if (t >= 1) issue_a(t - 1 + WL_NSA);   // I_{2t+2}
                if (do_bias) bias_rows(t, 1);
                if (t + 1 < NT) wait_tile(t + 1, t - 1 + WL_NSA, t - 1 + WL_NSB);
                if (lw == 0) WLCLK(8, 2);
                wl_bar();                          // #(2t+2)
                if (lw == 0) WLCLK(8, 3);
                WLCLKN();
            }
            wl_vmcnt<0>();
            wl_bar();                              // #(2 NT + 1)
            if (do_bias) {                         // fold the four row classes, then lane cq of row class 0 owns 4 entries
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = bs[e];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    const int n = n0 + 64 * lw + 4 * cq + e;
                    if (rg == 0 && n < q.Nn) q.gb[n] = P.beta != 0.f ? P.beta * q.gb[n] + v : v;
                    bs[e] = 0.f;
                }
            }
            if (L + nwx < bound) setup(L + nwx);
        }
        return;
    }

    // ---------------------------------------------------------------- MFMA waves
    const int grp = (w >> 2) & 1, wc = w & 3;
    const int g = l >> 4, c16 = l & 15;
    // transpose-read lane pointers (wgrad256.hip): lane p = c16 supplies row 8g + (p >> 2) (+ 4), the 8-byte piece p & 3 of a fragment's 16
    // columns; fragment I of a row (columns 16 I ..) sits at byte 32 (I ^ x) + 16 hi + 8 lo under the swizzle, x = p >> 2
    const int x = c16 >> 2, hi = (c16 >> 1) & 1, lo = c16 & 1;
    const uint32_t rowb = (uint32_t)((8 * g + x) * 512 + 16 * hi + 8 * lo);
    uint32_t fa[4], fb[3];
    // fragment I of row r: byte 32 (I ^ (r & 3) ^ 4 ((r >> 3) & 1)); r & 3 = x and (r >> 3) & 1 = g & 1 for every row this lane reads
    const int gb = g & 1;
#pragma unroll
    for (int b = 0; b < 4; ++b) fa[b] = rowb + 32u * (uint32_t)(b ^ x) + 128u * (uint32_t)gb + (uint32_t)(grp * 128) * 2u;   // dY fragment i = 4 a + b, a = 0
    const uint32_t fa1 = gb ? (uint32_t)-128 : 128u;                                                        // a = 1: the other 128 bytes
#pragma unroll
    for (int j = 0; j < 3; ++j) fb[j] = (uint32_t)WL_B0 + rowb + 32u * (uint32_t)((3 * wc + j) ^ x ^ (gb << 2));      // X fragment j of this wave
    WLCLK0();
#ifdef DGX_GEMM_DEV
    if (blockIdx.x == 0 && tid == 0) { wl_clk2[92] = clock64(); wl_clk2[93] = wall_clock64(); }
#endif
    for (int L = first; L < bound; L += nwx) {
        WlProb q;
        int n0, k0;
        locate(L, q, n0, k0);
        const int NT = (q.M + WL_BM - 1) / WL_BM;
        f32x4 acc[8][3];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 bfr[3][2], afr[8][2];
        // both k-halves of a fragment from one lane address (made opaque: the four reads then differ by immediate offsets only)
        auto frag2 = [&](uint32_t addr, bf16x8 (&f)[2]) {
            DGX_LDS const uint16_t* p = reinterpret_cast<DGX_LDS const uint16_t*>((DGX_LDS const unsigned char*)lds_raw + addr);
            asm volatile("" : "+v"(p));
            f[0] = tr_frag(p, 0, 4 * 256);
            f[1] = tr_frag(p, 32 * 256, 32 * 256 + 4 * 256);
        };
        constexpr int PRE = 2;
        auto read_phase = [&](int t) {
            const uint32_t bo = (uint32_t)(t % WL_NSB) * WL_SLOT, ao = a_slot(t % WL_NSA);
#pragma unroll
            for (int j = 0; j < 3; ++j) frag2(bo + fb[j], bfr[j]);
#pragma unroll
            for (int i = 0; i < PRE; ++i) frag2(ao + fa[i & 3] + ((i >> 2) ? fa1 : 0u), afr[i]);
        };
        auto mfma_phase = [&](int t) {
            const uint32_t ao = a_slot(t % WL_NSA);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + PRE < 8) WL_STREAM(ao + fa[(i + PRE) & 3] + (((i + PRE) >> 2) ? fa1 : 0u), afr[i + PRE]);
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(bfr[j][kh], afr[i][kh], acc[i][j]);     // D[k'][n]^T: lane = 4 Kk columns of one Nn row
                if (i + PRE < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        wl_bar();                                  // #0
        if (wc == 0) WLCLKI(4 * grp, 2);
        WLCLKIN();
        if (grp == 1) wl_bar();                    // #1: one phase behind group 0
        for (int t = 0; t < NT; ++t) {
            read_phase(t);
            wl_lgkm0();
            if (wc == 0) WLCLK(4 * grp, 0);
            wl_bar();
            if (wc == 0) WLCLK(4 * grp, 1);
            mfma_phase(t);
            wl_lgkm0();
            if (wc == 0) WLCLK(4 * grp, 2);
            wl_bar();
            if (wc == 0) WLCLK(4 * grp, 3);
            WLCLKN();
        }
        if (grp == 0) wl_bar();                    // #(2 NT + 1)
        if (wc == 0) WLCLKI(4 * grp, 0);
        // ---- read-out: fp32 read-modify-write of the gradient straight from the accumulators (16 bytes per lane, 64-byte runs per row)
        const float beta = P.beta;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + grp * 128 + 16 * i + c16;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = k0 + wc * 48 + 16 * j + 4 * g;
                if (n < q.Nn && k < q.Kk) {
                    f32x4* o = reinterpret_cast<f32x4*>(q.C + (int64_t)n * q.ldc + k);
                    f32x4 v = acc[i][j];
                    if (beta != 0.f) { const f32x4 c = *o; v[0] += beta * c[0]; v[1] += beta * c[1]; v[2] += beta * c[2]; v[3] += beta * c[3]; }
                    *o = v;
                }
            }
        }
        if (wc == 0) WLCLKI(4 * grp, 1);
    }
#ifdef DGX_GEMM_DEV
    if (blockIdx.x == 0 && tid == 0) { wl_clk2[94] = clock64(); wl_clk2[95] = wall_clock64(); }
#endif
}

// The grouped launch.  Returns DGX_ERR_UNSUPPORTED when the group is not this kernel's kind (the caller falls back to wgrad256).
int wgrad_lw_launch(const dgx_wgrad_problem* pr, int n, float beta, hipStream_t st) {
    if (n <= 0 || n > WL_MAXP) return DGX_ERR_UNSUPPORTED;
    WlParams P;
    memset((void*)&P, 0, sizeof(P));
    int items = 0;
    for (int i = 0; i < n; ++i) {
        const dgx_wgrad_problem& p = pr[i];
        if (!p.dy || !p.x || !p.gw || p.M <= 0 || (p.Nn & 7) || (p.Kk & 7)) return DGX_ERR_BAD_ARG;
        if ((int64_t)p.M * p.Nn * 2 >= (1ll << 31) || (int64_t)p.M * p.Kk * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
        WlProb& q = P.p[i];
        q.A = (const uint16_t*)p.dy; q.B = (const uint16_t*)p.x; q.C = p.gw; q.gb = p.gb;
        q.M = p.M; q.Nn = p.Nn; q.Kk = p.Kk; q.ldc = p.Kk;
        q.tiles_k = (p.Kk + WL_TK - 1) / WL_TK;
        q.item0 = items;
        items += ((p.Nn + WL_TN - 1) / WL_TN) * q.tiles_k;
    }
    for (int i = 0; i < WL_MAXP; ++i) P.item0[i] = i < n ? P.p[i].item0 : 0x7fffffff;
    P.n = n; P.total = items; P.beta = beta;
    static const int diag = getenv("DGX_WGRAD_LW_DIAG") ? atoi(getenv("DGX_WGRAD_LW_DIAG")) : 0;
    P.diag = diag;
    P.per_xcd = (items + 7) / 8;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)wgrad_lw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS) != hipSuccess) return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    extern int dgx_get_reserved_cus(void);
    const int avail = 32 - dgx_get_reserved_cus() / 8;
    const int wgx = P.per_xcd < avail ? P.per_xcd : avail;
    hipLaunchKernelGGL(wgrad_lw_kernel, dim3(8 * wgx), dim3(WL_THREADS), WL_LDS, st, P);
    return DGX_OK;
}

// Is this group worth the persistent form?  Many tiles of the loader-wave kernel (>= 3/4 of a round of the chip) whose contraction is
// long enough for the main loop to dominate, and whose item count fills whole rounds to >= 74 % (4 stage-2 blocks = 2.25 rounds still win).
bool wgrad_lw_wants(const dgx_wgrad_problem* pr, int n) {
    static const int mode = getenv("DGX_WGRAD_LW") ? atoi(getenv("DGX_WGRAD_LW")) : 1;
    if (!mode || n <= 0 || n > WL_MAXP) return false;
    int items = 0;
    for (int i = 0; i < n; ++i) {
        if (pr[i].M < 1024) return false;
        items += ((pr[i].Nn + WL_TN - 1) / WL_TN) * ((pr[i].Kk + WL_TK - 1) / WL_TK);
    }
    if (mode == 2) return true;
    if (items < 192) return false;
    const int rounds = (items + 255) / 256;
    return (double)items / (256.0 * rounds) >= 0.74;
}

#ifdef DGX_GEMM_DEV
extern "C" int dgx_dev_wl_clocks(unsigned long long* out, int reset) {      // out: 3 * 1024 * 4 stamps
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(wl_clk), sizeof(unsigned long long) * 3 * WL_CLK_KT * 4) != hipSuccess) return -1;
    if (reset && hipMemcpyFromSymbol(out, HIP_SYMBOL(wl_clk2), sizeof(unsigned long long) * 3 * 8 * 4) != hipSuccess) return -1;     // reset = 1: the item stamps instead
    return 0;
}
#endif
