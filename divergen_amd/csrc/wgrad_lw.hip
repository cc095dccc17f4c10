// Grouped weight-gradient GEMM for gfx950, loader-wave persistent form:  for each problem p
//     C_p[Nn][Kk] = beta * C_p + sum_m A_p[m][Nn] * B_p[m][Kk]        (A = dY, B = X, row-major bf16; C fp32 in the gradient arena)
// -- the contract of wgrad256.hip (the weight gradients autograd forms for nn.Linear at swintransformer.py:36-46,101-108), for
// launches that hold MANY output tiles (the Linears of several Swin blocks of stages 2 / 3: 144 / 576 tiles of 256 x 192 per block).
//
// Why a second kernel.  wgrad256's four waves (256 x 256 tile, 256 AGPR accumulators each) issue their own LDS-direct loads next to
// 64 MFMAs and 32 transpose reads per 32-row stage: a lone wave per SIMD loses MFMA issue to every instruction it interleaves, and the
// M-split that fills the chip costs fp32 partial slabs and a reduce launch.  Here (round 4):
//   * 12 waves: waves 8-11 only load (LDS-direct, counted vmcnt, published by the workgroup barrier), waves 0-7 only multiply -- two
//     per SIMD, both free-running through the same loop so that one's LDS reads and waits hide behind the other's MFMAs (wave tile
//     128 (Nn) x 48 (Kk), 96 accumulator registers, 168 registers per lane = 3 waves per SIMD);
//   * operand images as in memory, [m][n], in rings of 32-row GRANULES (one K = 32 MFMA step): 6 granules of dY + 4 of X, 16 KB each =
//     160 KB (the X image uses 384 of its 512 bytes per row, so that both operands share one layout: 16-byte chunk c of row r at chunk
//     c ^ ((r & 3) << 1) ^ (((r >> 3) & 1) << 3) -- the first term spreads the four rows a 16-lane group reads, the second puts the rows
//     of the odd lane groups into the other 128 bytes of the 256-byte bank period; without it the two 16-lane groups of a 32-lane LDS
//     pass meet on the same 32 banks: SQ_LDS_BANK_CONFLICT was half of all LDS cycles); fragments by ds_read_b64_tr_b16;
//   * ONE barrier per half K-tile; dY is loaded 4 halves and X 3 halves ahead of its use;
//   * NO VALU instruction in the MFMA waves' loop: it is unrolled over the rings' common period (12 halves), every LDS address is a
//     lane pointer formed once plus an immediate.  Measured (tools/probes/mfma_mix_probe.hip, tools/wgrad_lw_clocks.py): with this
//     instruction mix (22 transpose reads per 24 MFMAs and wave) one v_add per fragment costs 15 % of the MFMA rate -- 2 360 -> 2 000
//     cycles per K-tile here; a VALU instruction of a loader wave waits ~20 cycles for an issue slot, which is why the bias gradient
//     (column sums of dY on a problem's first tile column) is ones^T x fragment on the matrix pipe, 4 MFMAs per loader wave and half;
//   * swapped MFMA operands (D = X-fragment^T x dY-fragment^T): a lane holds 4 consecutive Kk entries of one Nn row, the read-out is
//     a 16-byte fp32 read-modify-write per lane straight from the accumulators -- no LDS staging, so the loaders fill the rings with the
//     NEXT tile's first granules while the finished tile is folded into the arena;
//   * one workgroup per CU walks a list of (problem, tile) items; NO M-split (every tile contracts its whole M: nothing to reduce,
//     no fp32 partial slabs).
// What bounds it: power.  The shader clock under this kernel is 1.5-1.7 GHz (2.38 under MFMAs alone, 1.9 with L2-resident operands):
// fewer cycles per K-tile came back as a lower clock (2 360 cycles at 1.63 GHz -> 2 000 at 1.51), 0.93-0.99 PFLOP/s on 7 stage-2
// blocks against 0.75 for the split-M form with its bias sums (profiles/r04_wgrad_lw_*).
// The caller (layers/swin_block.py) queues the problems of ~7 blocks per launch so that the item count is a multiple of the CU count
// to within a few percent (7 stage-2 blocks = 1 008 items = 3.94 rounds).
#include "dgx_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {
constexpr int WL_TN = 256, WL_TK = 192;          // tile: Nn rows x Kk columns of the weight gradient
constexpr int WL_BM = 64;                        // m-depth of a K-tile (two K = 32 MFMA steps)
constexpr int WL_GR = 32;                        // rows of a granule (half a K-tile: one K = 32 MFMA step)
constexpr int WL_GB = WL_GR * 512;               // bytes of one granule image (512-byte rows)
constexpr int WL_NGA = 6, WL_NGB = 4;            // ring depths (granules)
constexpr int WL_B0 = WL_NGA * WL_GB;
constexpr int WL_LDS = (WL_NGA + WL_NGB) * WL_GB;        // 160 KB
constexpr int WL_MAXP = 32;
constexpr int WL_THREADS = 768;

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
    int diag;                        // development builds, DGX_WGRAD_LW_DIAG=1 (timing experiments only, results wrong): every workgroup streams the panels of item 0
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
template <int N> using wl_ic = std::integral_constant<int, N>;
template <int N, int I = 0, typename F> __device__ __forceinline__ void wl_static_for(F&& f) {
    if constexpr (I < N) { f(wl_ic<I>{}); wl_static_for<N, I + 1>(f); }
}
}  // namespace

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
        // the tile column is rotated per tile row (and once more every 8 rows) so that the items that also sum the bias gradient (k0 == 0)
        // do not all fall to the same workgroups (item = workgroup + rounds * 32 keeps its residue mod tiles_k otherwise)
        k0 = ((t - tn * q.tiles_k + tn + (tn >> 3)) % q.tiles_k) * WL_TK;
    };
    // Rings of 32-row granules (half a K-tile): dY granule g in slot g % 6, X granule g in slot g % 4 (16 KB each).  ONE barrier per half:
    // at B_h the MFMA waves are about to multiply half h -- its X fragments they already hold -- and will read the rest of dY(h), the first
    // fragments of dY(h+1) and all of X(h+1).  The loaders guarantee dY(h+1), X(h+1) landed, and may overwrite dY(h-1) and X(h): every read of
    // those has returned, because LDS reads return in order and a YOUNGER read of each wave (a dY(h-1) fragment) fed an MFMA that was issued
    // before the barrier.  So dY(h+5) and X(h+4) are issued behind B_h, 4 resp. 3 halves ahead of their use.
    // Both MFMA groups run the same free-running loop (no read / multiply phases): the two MFMA waves of a SIMD hide each other's LDS
    // reads and waits -- a lone wave loses ~6 cycles of MFMA issue per instruction it interleaves (measured: 48 MFMAs took 1 084 cycles in
    // the two-phase form of this kernel against 778 back to back).
    if (w >= 8) {
        // ---------------------------------------------------------------- loader waves: the workgroup's pace-maker next to the MFMA waves
        // (8 loads per half at ~65 cycles of issue each), so everything per load is a running register: no multiply, no modulo
        const int lw = w - 8;
        const int rip = l >> 5;                                       // row inside the 2-row instruction
        // logical 16-byte chunk of this lane in instruction s: row = 8 s + 2 lw + rip, row & 3 = 2 (lw & 1) + rip, (row >> 3) & 1 = s & 1
        const int lc0 = (l & 31) ^ (((2 * (lw & 1) + rip) & 3) << 1), lc1 = lc0 ^ 8;
        const uint32_t ldsq = wl_sgpr(lds0 + 1024u * lw);
        const uint32_t OOB = 0x80000000u;
        WlProb q;
        int n0, k0, NH = 0;
        uint32_t va[4], vb[4];                     // byte offsets of this lane in the 4 instructions of the NEXT dY / X granule to issue
        uint32_t stepA = 0, stepB = 0, sa = 0, sb = 0;     // bytes per granule; ring slots (byte offsets) of those granules
        u32x4 rA, rB;
        auto setup = [&](int L) {
            locate((P.diag & 1) ? 0 : L, q, n0, k0);
            if (P.diag & 1) { WlProb q1; int a1, b1; locate(L, q1, a1, b1); q.M = q1.M; }
            NH = 2 * ((q.M + WL_BM - 1) / WL_BM);
#pragma unroll
            for (int s = 0; s < 4; ++s) {          // instruction s: rows 8 s + 2 lw + rip
                const int lc = (s & 1) ? lc1 : lc0, row = 8 * s + 2 * lw + rip;
                va[s] = (n0 + 8 * lc < q.Nn) ? (uint32_t)((row * q.Nn + n0 + 8 * lc) * 2) : OOB;
                vb[s] = (8 * lc < WL_TK && k0 + 8 * lc < q.Kk) ? (uint32_t)((row * q.Kk + k0 + 8 * lc) * 2) : OOB;
            }
            stepA = (uint32_t)(WL_GR * q.Nn * 2); stepB = (uint32_t)(WL_GR * q.Kk * 2);
            sa = 0; sb = 0;
            rA = wl_rsrc(q.A, (uint32_t)((int64_t)q.M * q.Nn * 2));   // rows >= M are out of range: zeros
            rB = wl_rsrc(q.B, (uint32_t)((int64_t)q.M * q.Kk * 2));
        };
        // One granule = 4 instructions of this wave.  The row offset lives in the VECTOR offset so that rows >= M fall under the descriptor's
        // range check and arrive as zeros (granules behind the end of M are issued all the same: no memory traffic, and the load counts
        // the waits rely on stay fixed); lanes beyond the matrix width keep bit 31 set through the adds (M * width * 2 < 2^31)
        auto issue_a1 = [&](int s) { wl_load_lds16(va[s], rA, wl_sgpr(ldsq + sa + 4096u * s), 0u); va[s] += stepA; };
        auto issue_b1 = [&](int s) { wl_load_lds16(vb[s], rB, wl_sgpr(ldsq + WL_B0 + sb + 4096u * s), 0u); vb[s] += stepB; };
        auto next_a = [&]() { sa = sa + WL_GB == WL_NGA * WL_GB ? 0u : sa + WL_GB; };
        auto next_b = [&]() { sb = (sb + WL_GB) & (WL_NGB * WL_GB - 1); };
        auto issue_a = [&]() {
#pragma unroll
            for (int s = 0; s < 4; ++s) issue_a1(s);
            next_a();
        };
        auto issue_b = [&]() {
#pragma unroll
            for (int s = 0; s < 4; ++s) issue_b1(s);
            next_b();
        };
        // Bias gradient (column sums of dY) of the items of a problem's first tile column, on the loader waves -- and on the MATRIX pipe:
        // any VALU instruction in this kernel waits ~20 cycles for an issue slot between the MFMA waves' instructions (32 adds per half
        // cost 650 cycles when tried), while ones^T x fragment is 4 MFMAs per loader wave and half (+8 % pipe time on these items only).
        // Wave lw owns the dY fragments 4 lw + f (columns 64 lw + 16 f ..): their transpose reads go out a half ahead (LDS latency under the
        // MFMA waves' traffic is several hundred cycles), the MFMAs follow behind the next half's loads.
        const int g = l >> 4, c16 = l & 15, x = c16 >> 2;
        DGX_LDS const uint16_t* pf[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
            pf[f] = lds_opaque(reinterpret_cast<const uint16_t*>(lds_raw + (8 * g + x) * 512 + 16 * ((c16 >> 1) & 1) + 8 * (c16 & 1) + 32 * ((4 * lw + f) ^ x ^ ((g & 1) << 2))));
        const bf16x8 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};
        f32x4 bacc[4];
        bf16x8 bfrag[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) bacc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint32_t sbias = 0;                        // slot (in elements) of the dY granule whose fragments are read next
        auto bias_read = [&]() {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                DGX_LDS const uint16_t* p = pf[f] + sbias;
                asm volatile("" : "+v"(p));
                bfrag[f] = tr_frag(p, 0, 4 * 256);
            }
            sbias = sbias + WL_GB / 2 == WL_NGA * WL_GB / 2 ? 0u : sbias + WL_GB / 2;
        };
        auto bias_fold = [&](int f) { bacc[f] = mfma16(ones, bfrag[f], bacc[f]); };
        setup(first);
        WLCLK0();
        for (int L = first; L < bound; L += nwx) {
            const bool do_bias = q.gb != nullptr && k0 == 0;
            // every slot is free here (barrier E of the previous item); issue order A0 X0 A1 X1 A2 X2 A3 X3 A4, then A(h+5) X(h+4) behind B_h
            issue_a(); issue_b(); issue_a(); issue_b(); issue_a(); issue_b(); issue_a(); issue_b(); issue_a();
            wl_vmcnt<28>();                        // A0, X0 (7 granules younger)
            wl_bar();                              // P: the MFMA waves read X(0) and the first dY(0) fragments
            sbias = 0;
            if (do_bias) bias_read();
            for (int h = 0; h < NH; ++h) {
                // before B_h: dY(h+1), X(h+1).  Younger than X(h+1): h = 0: A2 X2 A3 X3 A4; h = 1: A3 X3 A4 A5 X4; h = 2: A4 A5 X4 A6 X5; later:
                // the four granules issued behind B_(h-2) and B_(h-1)
                if (h < 3) wl_vmcnt<20>(); else wl_vmcnt<16>();
                if (lw == 0) WLCLK(8, 0);
                wl_bar();                          // B_h
                if (lw == 0) WLCLK(8, 1);
                if (do_bias) {                     // dY(h)'s rows were read a half ago (their LDS latency hides behind the barrier)
#pragma unroll
                    for (int s = 0; s < 4; ++s) { issue_a1(s); bias_fold(s); }
                    issue_b();
                    next_a();
                    bias_read();                   // dY(h+1): visible since this barrier, stays until B_(h+2); behind the end of M: zeros / unused
                } else {
                    issue_a();
                    issue_b();
                }
                if (lw == 0) WLCLK(8, 2);
                if (lw == 0) WLCLK(8, 3);
                WLCLKN();
            }
            wl_bar();                              // E: every read of this item's granules has fed its MFMA
            if (do_bias) {                         // every row of the 16 x 16 result holds the column sums: lanes 0-15 own one entry per fragment
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int n = n0 + 64 * lw + 16 * f + c16;
                    if (g == 0 && n < q.Nn) q.gb[n] = P.beta != 0.f ? P.beta * q.gb[n] + bacc[f][0] : bacc[f][0];
                    bacc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (L + nwx < bound) setup(L + nwx);
        }
        return;
    }

    // ---------------------------------------------------------------- MFMA waves
    // Group grp owns the dY fragments I = 8 c + 4 grp + b (c = 0, 1; b = 0..3): the two 64-column blocks 64 grp .. and 128 + 64 grp ..; wave wc
    // the X fragments 3 wc + j.  NO address arithmetic in the loop (a per-fragment v_add costs this instruction mix ~15 % of its MFMA rate,
    // tools/probes/mfma_mix_probe.hip): transpose-read lane pointers -- lane p = c16 supplies row 8g + (p >> 2) (+ 4), the 8-byte piece p & 3
    // of a fragment's 16 columns; fragment I of row r sits at byte 32 (I ^ (r & 3) ^ 4 ((r >> 3) & 1)), r & 3 = x = p >> 2 and (r >> 3) & 1 =
    // g & 1 for every row this lane reads -- are formed once, four per 64 KB window of the dY ring (the DS offset field is 16 bits) and three
    // for the X ring; granule slot, c and the second 4-row block are immediates because the loop is unrolled over the rings' common period
    // of 12 halves.
    const int grp = (w >> 2) & 1, wc = w & 3;
    const int g = l >> 4, c16 = l & 15;
    const int x = c16 >> 2, hi = (c16 >> 1) & 1, lo = c16 & 1;
    const uint32_t rowb = (uint32_t)((8 * g + x) * 512 + 16 * hi + 8 * lo);
    const int gb = g & 1;
    DGX_LDS const uint16_t* pa[2][4];
    DGX_LDS const uint16_t* pb[3];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t o = rowb + 32u * (uint32_t)(b ^ x) + 128u * (uint32_t)(grp ^ gb);
        pa[0][b] = lds_opaque(reinterpret_cast<const uint16_t*>(lds_raw + o));
        pa[1][b] = lds_opaque(reinterpret_cast<const uint16_t*>(lds_raw + o + 4 * WL_GB));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) pb[j] = lds_opaque(reinterpret_cast<const uint16_t*>(lds_raw + WL_B0 + rowb + 32u * (uint32_t)((3 * wc + j) ^ x ^ (gb << 2))));
    WLCLK0();
#ifdef DGX_GEMM_DEV
    if (blockIdx.x == 0 && tid == 0) { wl_clk2[92] = clock64(); wl_clk2[93] = wall_clock64(); }
#endif
    for (int L = first; L < bound; L += nwx) {
        WlProb q;
        int n0, k0;
        locate(L, q, n0, k0);
        const int NH = 2 * ((q.M + WL_BM - 1) / WL_BM);
        f32x4 acc[8][3];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 b0[3], b1[3], afr[4];               // X fragments of the current / the next half; ring of streamed dY fragments
        constexpr int PRE = 3;                     // dY fragments in flight ahead of their MFMAs (ring of PRE + 1)
        // dY fragment i = 4 c + b of the granule in ring slot s (8 k-slots = the lane group's 8 rows): two reads, immediates only
        auto fragA = [&](auto S, auto I) -> bf16x8 {
            constexpr int s_ = decltype(S)::value, i_ = decltype(I)::value;
            constexpr int off = (s_ & 3) * (WL_GB / 2) + 128 * (i_ >> 2);           // elements
            return tr_frag(pa[s_ >> 2][i_ & 3], off, off + 4 * 256);
        };
        auto fragB = [&](auto S, auto J) -> bf16x8 {
            constexpr int off = decltype(S)::value * (WL_GB / 2);
            return tr_frag(pb[decltype(J)::value], off, off + 4 * 256);
        };
        // half at ring position U (granule h = U mod 12): bc = X(h) fragments (held), bn <- X(h+1)
        auto half = [&](auto U, const bf16x8 (&bc)[3], bf16x8 (&bn)[3]) {
            constexpr int u = decltype(U)::value, sa = u % WL_NGA, sa1 = (u + 1) % WL_NGA, sb1 = (u + 1) % WL_NGB;
            wl_bar();                              // B_h
            __builtin_amdgcn_s_setprio(1);
            wl_static_for<8>([&](auto I) {
                constexpr int i = decltype(I)::value, ip = i + PRE;
                if constexpr (ip < 8) afr[ip & 3] = fragA(wl_ic<sa>{}, wl_ic<ip>{});
                else afr[ip & 3] = fragA(wl_ic<sa1>{}, wl_ic<ip - 8>{});
                if constexpr (i < 3) bn[i] = fragB(wl_ic<sb1>{}, wl_ic<i>{});
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(bc[j], afr[i & 3], acc[i][j]);     // D[k'][n]^T: lane = 4 Kk columns of one Nn row
            });
            // pin the order: the compiler otherwise sinks the reads to just above their use (prefetch distance 1 instead of PRE)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        wl_bar();                                  // P: dY(0), X(0) landed
        if (wc == 0) WLCLKI(4 * grp, 2);
        WLCLKIN();
        wl_static_for<3>([&](auto J) { b0[decltype(J)::value] = fragB(wl_ic<0>{}, J); });
        wl_static_for<PRE>([&](auto I) { afr[decltype(I)::value] = fragA(wl_ic<0>{}, I); });
        for (int h = 0; h < NH; h += 12) {         // NH is even; both rings are back at slot 0 after 12 halves
            if (wc == 0) WLCLK(4 * grp, 0);
            half(wl_ic<0>{}, b0, b1); half(wl_ic<1>{}, b1, b0);
            if (wc == 0) WLCLK(4 * grp, 1);
            if (h + 2 >= NH) break;
            half(wl_ic<2>{}, b0, b1); half(wl_ic<3>{}, b1, b0);
            if (h + 4 >= NH) break;
            half(wl_ic<4>{}, b0, b1); half(wl_ic<5>{}, b1, b0);
            if (h + 6 >= NH) break;
            half(wl_ic<6>{}, b0, b1); half(wl_ic<7>{}, b1, b0);
            if (h + 8 >= NH) break;
            half(wl_ic<8>{}, b0, b1); half(wl_ic<9>{}, b1, b0);
            if (h + 10 >= NH) break;
            half(wl_ic<10>{}, b0, b1); half(wl_ic<11>{}, b1, b0);
            if (wc == 0) WLCLK(4 * grp, 2);
            WLCLKN();
        }
        wl_bar();                                  // E
        if (wc == 0) WLCLKI(4 * grp, 0);
        // ---- read-out: fp32 read-modify-write of the gradient straight from the accumulators (16 bytes per lane, 64-byte runs per row)
        const float beta = P.beta;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + 128 * (i >> 2) + 64 * grp + 16 * (i & 3) + c16;
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
#ifdef DGX_GEMM_DEV
    static const int diag = getenv("DGX_WGRAD_LW_DIAG") ? atoi(getenv("DGX_WGRAD_LW_DIAG")) : 0;     // development build only (tools/wgrad_lw_clocks.py)
    P.diag = diag;
#endif
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
int g_dgx_dev_wgrad_lw = 1;      // dgx_dev_set("wgrad_lw", v): 1 = the plan below, 0 = never this form, 2 = always (tests / A-B tools)
bool wgrad_lw_wants(const dgx_wgrad_problem* pr, int n) {
    const int mode = g_dgx_dev_wgrad_lw;
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
