// Swin window attention core for gfx950: one workgroup per (window, head), one wave per 16-row
// strip, MFMA 16x16x32 bf16, fp32 softmax.  head_dim is 32 in every Swin size, so one MFMA spans
// the whole QK^T contraction.  Reference semantics: DG/divergen/modeling/backbone/swintransformer.py:133-154.
//
// Forward ("swapped" orientation): S^T = K Q^T so that a lane owns ONE query (col = lane&15) and
// 4 keys per tile; row max/sum are 2 xor-shuffles and P feeds the P*V MFMA straight from registers
// (the k-slot permutation it implies is applied to the V^T image read from LDS).
// Backward: a wave owns a 16-KEY strip (dK, dV and the rel-pos-bias gradient are wave-local and
// the bias gradient accumulates in registers across the windows of a chunk); dS goes through LDS
// once for dQ = dS K.
#include "dgx_common.h"
#include <type_traits>
#include <map>
#include <mutex>
#include <utility>
#include "winmap.h"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "window_attention.hip orders the bias-gradient slot hand-over by store COMPLETION (sc1 stores acknowledged at the device coherence point; see flush_head): verified for gfx950 only"
#endif
typedef float wa_f32x2 __attribute__((ext_vector_type(2)));

template <int WS> struct WinCfg;
template <> struct WinCfg<12> { static constexpr int N = 144, NT = 9, NTK = 10, RS = 168, TBL = 529; };
template <> struct WinCfg<7>  { static constexpr int N = 49,  NT = 4, NTK = 4,  RS = 72,  TBL = 169; };
// NT  = ceil(N/16) strips (= waves per workgroup); NTK = NT rounded up to even (K=32 steps);
// RS  = row stride (elements) of the transposed [32][NTK*16] LDS images; chosen so that the
//       16 rows x 4 lane-groups of a ds_read_b64 hit 64 distinct banks (RS*2/4 = 4*odd mod 64).

__device__ __forceinline__ bf16x8 ld_frag_global(const uint16_t* p, bool ok) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return ok ? *reinterpret_cast<const bf16x8*>(p) : z;
}

// transposed staging: src rows r (r < N valid) of 32 bf16 at src + r*row_stride -> T[d][r]
template <int N, int NP, int RS>
__device__ __forceinline__ void stage_transposed(uint16_t* T, const uint16_t* src, int64_t row_stride,
                                                 int tid, int nthreads) {
    constexpr int HALF = NP / 2;
    for (int u = tid; u < HALF * 4; u += nthreads) {
        const int c = u / HALF, m = u - c * HALF;
        const int r0 = 2 * m, r1 = r0 + 1;
        bf16x8 v0 = ld_frag_global(src + (int64_t)r0 * row_stride + 8 * c, r0 < N);
        bf16x8 v1 = ld_frag_global(src + (int64_t)r1 * row_stride + 8 * c, r1 < N);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t w = (uint32_t)(uint16_t)v0[i] | ((uint32_t)(uint16_t)v1[i] << 16);
            *reinterpret_cast<uint32_t*>(&T[(8 * c + i) * RS + r0]) = w;
        }
    }
}

// B/A fragment from a transposed image: slots j<4 -> cols 32t+4g+j, j>=4 -> cols 32t+16+4g+(j-4)
template <int RS>
__device__ __forceinline__ bf16x8 ld_frag_T(const uint16_t* T, int row, int t, int g) {
    const uint2 a = *reinterpret_cast<const uint2*>(&T[row * RS + 32 * t + 4 * g]);
    const uint2 b = *reinterpret_cast<const uint2*>(&T[row * RS + 32 * t + 16 + 4 * g]);
    u32x4 v = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(bf16x8, v);
}

// row-major staging of NP rows x 32 bf16 (rows >= N zero-filled) with 16-byte stores; row stride RR elems
template <int N, int NP, int RR>
__device__ __forceinline__ void stage_rows(uint16_t* img, const uint16_t* src, int64_t row_stride, int tid, int nthreads) {
    for (int u = tid; u < NP * 4; u += nthreads) {
        const int r = u >> 2, c = u & 3;
        *reinterpret_cast<bf16x8*>(&img[r * RR + 8 * c]) = ld_frag_global(src + (int64_t)r * row_stride + 8 * c, r < N);
    }
}

// XCD-aware block -> (window, head): blocks that share an XCD (id % 8) walk the heads of one
// window back to back, so the 64-byte head slices of a qkv row are served by one L2.
__device__ __forceinline__ void block_to_window_head(int bid, int nH, int& b, int& h) {
    const int xcd = bid & 7, slot = bid >> 3;
    b = (slot / nH) * 8 + xcd;
    h = slot % nH;
}

// COMPACT window order (winmap.h): the qkv / out / dout / dqkv rows of the REAL tokens only, M = B*H*W, in (image, window, token) order;
// a padding token's q, k, v are the qkv BIAS (its LayerNorm-ed input is 0: swintransformer.py:216-221 pads after norm1), its attention
// output is never stored (cropped at :248-251), its dO is 0; the backward still writes its dk / dv (and a zero dq) to row T + its rank
// among the padding tokens, for the qkv bias gradient.  H == 0: classic order, every token has a row.
struct AttnGeom {
    int H, W, shift;                 // token grid and cyclic shift of this block
    int T, P;                        // B*H*W real rows; padding tokens per image
    const uint16_t* bias;            // qkv bias, bf16 (3 * C)
    const uint16_t* zeros;           // >= 64 bytes of zeros (dO / O of a padding token)
};
// row of token n of window (image img, wr, wc): real -> its compact row, padding -> -(1 + its padding rank in the whole batch)
template <int WS>
__device__ __forceinline__ int attn_row(const AttnGeom& G, const WmGeom& g, const WmWindow& w, int img, int wr, int wc, int n) {
    const int i = n / WS, j = n - i * WS;
    int before;
    const bool real = wm_token_real(g, w, wr, wc, i, j, before);
    const int wi = wr * g.nWw + wc;
    return real ? img * (G.H * G.W) + w.base + before : -(1 + img * G.P + (wi * (WS * WS) - w.base) + (n - before));
}

#ifdef DIAG_CLOCK
__device__ unsigned long long dgx_clk[16];
#define CLK(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == DIAG_WAVE) { const unsigned long long t__ = clock64(); dgx_clk[i] += t__ - tprev; tprev = t__; } } while (0)
// forward: the time stamps of wave DIAG_WAVE of EVERY workgroup, kept in registers and stored once at the end (slot 0: start, 1 + i: FCLK(i));
// an atomic per phase onto one counter made the kernel five times slower
__device__ unsigned long long dgx_fclk[8192 * 8];
#define FCLK_START unsigned long long fts[8]; fts[0] = clock64()
#define FCLK(i) do { fts[1 + (i)] = clock64(); } while (0)
#define FCLK_END(n) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == DIAG_WAVE && blockIdx.x < 8192) { for (int i__ = 0; i__ <= (n); ++i__) dgx_fclk[blockIdx.x * 8 + i__] = fts[i__]; } } while (0)
#else
#define CLK(i)
#define FCLK(i)
#define FCLK_START
#define FCLK_END(n)
#endif

// Softmax runs in the log2 domain: the bias row is pre-multiplied by log2(e) when it is staged in LDS and the
// score is one FMA, s2 = qk * (scale*log2e) + bias2 (the -100 of the shift mask becomes -100*log2e), so that
// p = exp2(s2 - max2) is a bare v_exp_f32.  MASKED = false (W-MSA blocks) drops the region compare entirely.
#define DGX_LOG2E 1.4426950408889634f
#define DGX_LN2 0.6931471805599453f
template <int WS, bool MASKED, bool COMPACT = false>
__global__ __launch_bounds__(WinCfg<WS>::NT * 64) void win_attn_fwd_kernel(
    const uint16_t* __restrict__ qkv, const float* __restrict__ table, const int8_t* __restrict__ region,
    uint16_t* __restrict__ out, float* __restrict__ lse, int B_, int nW, int nH, float scale, int64_t t_sh, int64_t t_si,
    AttnGeom G = AttnGeom{0, 0, 0, 0, 0, nullptr, nullptr}) {
    using Cf = WinCfg<WS>;
    constexpr int N = Cf::N, NT = Cf::NT, NTK = Cf::NTK, NP = NTK * 16, RS = Cf::RS, TBL = Cf::TBL;
    // row stride (elements) of the row-major K / V images: 48 = 24 banks.  The 16-byte fragment reads (8 consecutive rows per pass, 4
    // banks each) AND the transpose reads (4 consecutive rows x 8 banks per 16-lane pass) are both conflict-free at 24 banks; at the
    // former 40 (20 banks) the fourth row of a transpose read wrapped onto the first (60 + 8 > 64): a 2-way conflict on every one
    constexpr int RR = WS == 12 ? 48 : 40;
    __shared__ __attribute__((aligned(16))) uint16_t Vs[NP * RR];
    __shared__ __attribute__((aligned(16))) uint16_t Ks[NP * RR];   // K rows too: one coalesced load per workgroup instead of
                                                                    // nine per-wave passes over K with their global latencies
    __shared__ float tbl[TBL];
    __shared__ __attribute__((aligned(16))) int koff_s[NP];   // rel-pos offset of key k: yk*(2WS-1) + xk
    __shared__ __attribute__((aligned(16))) int kreg_s[NP];   // region id of key k (this window); padded keys: -1
    static_assert(TBL <= NT * 64 && NP <= NT * 64 && NP * 4 <= 2 * NT * 64, "one prologue load per thread and array");

    int b, h;
    block_to_window_head(blockIdx.x, nH, b, h);
    if (b >= B_) return;
    FCLK_START;
    const int C = nH * 32;
    const int64_t rowst = 3 * (int64_t)C;
    const uint16_t* base = qkv + (int64_t)b * N * rowst + h * 32;
    const int tid = threadIdx.x, nthreads = NT * 64;
    const int w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    const int qi = 16 * w + c16;
    const bool qok = qi < N;
    // COMPACT: the address of (token n, part: 0 q / 1 k / 2 v, 16-byte chunk c) -- a real token's qkv row or the bias
    WmGeom wg;
    WmWindow ww;
    int img = 0, wr = 0, wc = 0;
    if (COMPACT) {
        wg = wm_geom(G.H, G.W, WS, G.shift);
        img = b / nW;
        const int wi = b - img * nW;
        wr = wi / wg.nWw;
        wc = wi - wr * wg.nWw;
        ww = wm_window(wg, wr, wc);
    }
    // (a window without padding tokens -- most of them -- takes the uniform short cut: its rows are consecutive)
    const bool wfull = COMPACT && ww.rh == WS && ww.rw == WS;
    auto row_of = [&](int n) -> int {
        if (wfull) return img * (G.H * G.W) + ww.base + n;
        return attn_row<WS>(G, wg, ww, img, wr, wc, n);
    };
    auto row_ptr = [&](int n) -> const uint16_t* {          // token n's q | k | v row: its qkv row, or the bias for a padding token
        if (!COMPACT) return base + (int64_t)n * rowst;
        const int row = row_of(n < N ? n : 0);
        return (row >= 0 ? qkv + (int64_t)row * rowst : G.bias) + h * 32;
    };

    // ---- every global load of the workgroup is issued here, back to back (ONE memory latency), then parked in LDS
    const int8_t* reg = region + (int64_t)(b % nW) * N;
    const float tv = tid < TBL ? table[h * t_sh + tid * t_si] : 0.f;
    const int kr_in = (MASKED && tid < N) ? (int)reg[tid] : -1;
    const int rq = (MASKED && qok) ? (int)reg[qi] : 0;
    const int u0 = tid, u1 = tid + nthreads;                       // 16-byte chunks of the K / V images: row u >> 2, chunk u & 3
    const uint16_t* r0 = row_ptr(u0 >> 2);
    const uint16_t* r1 = row_ptr(u1 >> 2);
    const bf16x8 k0 = ld_frag_global(r0 + C + 8 * (u0 & 3), (u0 >> 2) < N);
    const bf16x8 v0 = ld_frag_global(r0 + 2 * C + 8 * (u0 & 3), (u0 >> 2) < N);
    const bool two = u1 < NP * 4;
    const bf16x8 k1 = ld_frag_global(r1 + C + 8 * (u1 & 3), two && (u1 >> 2) < N);
    const bf16x8 v1 = ld_frag_global(r1 + 2 * C + 8 * (u1 & 3), two && (u1 >> 2) < N);
    const int q_row = COMPACT ? row_of(qok ? qi : 0) : 0;
    const bf16x8 qf = ld_frag_global(COMPACT ? (q_row >= 0 ? qkv + (int64_t)q_row * rowst : G.bias) + h * 32 + 8 * g : base + (int64_t)qi * rowst + 8 * g, qok);
    FCLK(0);            // address arithmetic + issue
    if (tid < TBL) tbl[tid] = tv * DGX_LOG2E;
    if (tid < NP) {
        const int yk = tid / WS;
        koff_s[tid] = tid < N ? yk * (2 * WS - 1) + (tid - yk * WS) : 0;
        kreg_s[tid] = kr_in;
    }
    if (u0 < NP * 4) {
        *reinterpret_cast<bf16x8*>(&Ks[(u0 >> 2) * RR + 8 * (u0 & 3)]) = k0;
        *reinterpret_cast<bf16x8*>(&Vs[(u0 >> 2) * RR + 8 * (u0 & 3)]) = v0;
    }
    if (two) {
        *reinterpret_cast<bf16x8*>(&Ks[(u1 >> 2) * RR + 8 * (u1 & 3)]) = k1;
        *reinterpret_cast<bf16x8*>(&Vs[(u1 >> 2) * RR + 8 * (u1 & 3)]) = v1;
    }
    FCLK(1);            // the loads' latency + parking in LDS
    __syncthreads();
    FCLK(2);            // barrier

    const int yq = qi / WS, xq = qi - yq * WS;
    const int base_q = qok ? (yq + WS - 1) * (2 * WS - 1) + (xq + WS - 1) : (WS - 1) * (2 * WS - 1) + (WS - 1);
    const float scale2 = scale * DGX_LOG2E;
    DGX_LDS const float* tbl_q = lds_opaque(tbl + base_q);
    DGX_LDS const int* koff_g = lds_opaque(koff_s + 4 * g);
    DGX_LDS const int* kreg_g = lds_opaque(kreg_s + 4 * g);

    float p[NTK][4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(16 * kt + c16) * RR + 8 * g]);   // rows >= N are zero
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 acc = mfma16(kf, qf, z);  // acc[r] = S^T[key 16kt+4g+r][query c16]
        const i32x4 ko = *reinterpret_cast<DGX_LDS const i32x4*>(koff_g + 16 * kt);
        i32x4 kr4 = {0, 0, 0, 0};
        if (MASKED) kr4 = *reinterpret_cast<DGX_LDS const i32x4*>(kreg_g + 16 * kt);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = __builtin_fmaf(acc[r], scale2, tbl_q[-ko[r]]);
            if (MASKED) s += kr4[r] != rq ? -100.0f * DGX_LOG2E : 0.0f;
            if (NP != N && 16 * kt + 4 * g + r >= N) s = -INFINITY;   // padded key columns (NTK rounds the key tiles up to even)
            p[kt][r] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(p[kt][r] - mx);
            p[kt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (g == 0 && qok) lse[((int64_t)b * nH + h) * N + qi] = mx * DGX_LN2 + __logf(sum);   // natural-log LSE, as before
    const float inv = 1.0f / sum;
    FCLK(3);            // scores + softmax

    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    DGX_LDS const uint16_t* v_l4 = tr_lane_ptr(Vs, RR, 4 * g, 0, c16);
#pragma unroll
    for (int t = 0; t < NTK / 2; ++t) {
        u32x4 pk = {pack_bf2(p[2 * t][0], p[2 * t][1]), pack_bf2(p[2 * t][2], p[2 * t][3]),
                    pack_bf2(p[2 * t + 1][0], p[2 * t + 1][1]), pack_bf2(p[2 * t + 1][2], p[2 * t + 1][3])};
        // A = P^T-slot fragment: row (l&15) = query, slots (g, j) = keys {32t+4g+j, 32t+16+4g+j-4}
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
        // B = V with the same k-slot order as P: keys {32t+4g+j, 32t+16+4g+j}, straight from the row-major image
        // swapped operands (as the backward's dV / dK / dQ): D = V^T P^T = O^T, so a lane ends up with 4 CONSECUTIVE head-dim entries of ITS
        // query -- two 8-byte stores per lane and the lane's own 1 / sum, instead of eight 2-byte stores and four shuffles (rounds 1-4)
        o[0] = mfma16(tr_frag(v_l4, 32 * t * RR, (32 * t + 16) * RR), pf, o[0]);  // o[dt][r] = O[query 16w+c16][d 16dt+4g+r]
        o[1] = mfma16(tr_frag(v_l4, 32 * t * RR + 16, (32 * t + 16) * RR + 16), pf, o[1]);
    }
    int64_t out_row = (int64_t)b * N + qi;
    if (COMPACT) out_row = qok ? q_row : -1;       // a padding query's output is cropped: not stored
    if (qok && out_row >= 0) {
        uint16_t* orow = out + out_row * C + h * 32 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            *reinterpret_cast<u32x2*>(orow + 16 * dt) = u32x2{pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv)};
    }
    FCLK(4);            // P V + stores issued
    FCLK_END(5);
}

// ------------------------------------------------------------------------------------ backward

// value of lane (4 * (lane / 4) + R) of every quad: v_mov_b32_dpp quad_perm:[R,R,R,R] -- no LDS traffic (the backward kernel is
// bound by the LDS pipe: profiles/r05_attn_bwd_*.txt)
template <int R>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), R * 0x55, 0xf, 0xf, true));
}
template <int R>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, R * 0x55, 0xf, 0xf, true); }

// One window's worth of global loads for one thread, held in registers while the previous window computes.
struct BwdPrefetch {
    bf16x8 q, d, o, k;   // this thread's 16-byte chunk (row tid>>2, chunk tid&3) of Q, dO, O, K
    bf16x8 v;            // ... and of V (staged like K since round 5: the fragment a wave needs is read from the LDS image)
    float lse;           // chunk-0 threads: lse / region id of the row
    int reg;
};

// HELP (window 12 only): 12 waves instead of 9.  The hardware deals the waves of a workgroup round-robin over the four SIMDs, so with nine
// strip waves SIMD 0 runs three of them (strips 0, 4, 8) and sets the pace of phase 1 (27 of the 81 query-tile steps against 18 on the
// other SIMDs: the other waves waited ~1 800 of ~14 000 cycles per window at the barrier behind it, profiles/r03_attn_bwd_phases.txt).
// Waves 9-11 land on SIMDs 1-3 and take the query tiles 6-8 of strips 0 / 4 / 8 (21 + 20 + 20 + 20 steps): they write their rows of the
// dS^T image and their own bias-gradient terms like any strip wave and hand their partial dK / dV to the strip's owner through LDS.
template <int WS, bool MASKED, bool HELP = false, bool COMPACT = false>
__global__ __launch_bounds__((WinCfg<WS>::NT + (HELP ? 3 : 0)) * 64) void win_attn_bwd_kernel(
    const uint16_t* __restrict__ qkv, const float* __restrict__ table, const int8_t* __restrict__ region,
    const uint16_t* __restrict__ out, const float* __restrict__ lse, const uint16_t* __restrict__ dout,
    uint16_t* __restrict__ dqkv, float* __restrict__ dtable, int B_, int nW, int nH, float scale, int upw, int upl, float* __restrict__ part_ws, int* __restrict__ head_cnt,
    int64_t dt_sh, int64_t dt_si, AttnGeom G = AttnGeom{0, 0, 0, 0, 0, nullptr, nullptr}) {
    using Cf = WinCfg<WS>;
    constexpr int N = Cf::N, NT = Cf::NT, NTK = Cf::NTK, NP = NTK * 16, TBL = Cf::TBL;
    // row strides (elements): the [NP][32] images 48 (24 banks: 16-byte fragment reads over 8 rows and transpose reads over 4 rows x 8
    // banks both conflict-free; 40 = 20 banks put the fourth row of every transpose read onto the first), the dS^T image [key][query]
    // 184 (92 = 28 mod 64 banks: the 8-byte writes of 16 key rows hit 16 distinct bank pairs AND the four rows of a transpose read
    // 0 / 28 / 56 / 20 do not overlap; the former 168 = 20 mod 64 had the same wrap)
    constexpr int RR = WS == 12 ? 48 : 40;
    constexpr int RD = WS == 12 ? 184 : NP + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* Qs = reinterpret_cast<uint16_t*>(smem);       // [NP][RR] row-major Q
    uint16_t* dOs = Qs + NP * RR;                            // [NP][RR] row-major dO
    uint16_t* Ks = dOs + NP * RR;                            // [NP][RR] row-major K
    uint16_t* dSt = Ks + NP * RR;                            // [NP keys][RD] dS^T (bf16)
    float* lse_s = reinterpret_cast<float*>(dSt + NP * RD);  // [NP] (16-B aligned: read as float4)
    float* delta_s = lse_s + NP;                             // [NP]
    int* qoff_s = reinterpret_cast<int*>(delta_s + NP);      // [NP] rel-pos offset of query q
    int* reg_s = qoff_s + NP;                                // [NP] region id of query q (this window)
    float* tbl = reinterpret_cast<float*>(reg_s + NP);       // [TBL]
    float* tblacc = tbl + TBL;                               // [TBL]
    float* part_s = tblacc + TBL + ((4 - (2 * TBL) % 4) % 4);    // HELP: [3 helpers][4 vectors][64 lanes] f32x4 partial dV / dK (16-B aligned)
    // PK (window 12): per-query terms and bias rows in the form that costs the LDS pipe least.  The kernel is LDS-bound (~1 MB of LDS
    // traffic per (window, head); 36 % of it the four 16-byte BROADCAST reads of lse / delta / bias offset / region per query tile, in
    // which 16 lanes fetch the same 16 bytes):  * (lse, delta) interleaved, one 8-byte read per lane (its quad's query r = lane & 3),
    // spread over the quad by DPP;  * the bias row: the four queries 4 j .. 4 j + 3 of a lane never straddle a window row (12 = 3 x 4),
    // so their table entries are CONSECUTIVE -- four copies of the table shifted by 0..3 entries make that run a 16-byte ALIGNED read
    // at a per-lane address that is constant for the whole kernel (9 registers) instead of an offset vector + four gathers
    constexpr bool PK = WS == 12;
    constexpr int TBLS = (TBL + 3 + 3) / 4 * 4;                   // stride (floats) of a shifted copy
    float* tbl4 = part_s + (HELP ? 3 * 4 * 64 * 4 : 0);           // [4][TBLS]
    uint16_t* Vs = reinterpret_cast<uint16_t*>(tbl4 + (PK ? 4 * TBLS : 0));     // [NP][RR] row-major V

    // Runs of `upw` (window, head) units per workgroup.  Each head's B_ windows are cut into q = B_ / upw full runs, served by the first
    // q * nH workgroups -- neighbouring workgroups take the SAME windows of different heads (together they read whole qkv rows; a head-major
    // order, every workgroup on its own windows, ran 30-45 % slower) -- and the B_ % upw windows left over per head are strung together
    // head-major into further runs of `upl` units, which cross a head boundary every B_ % upw units: there the bias-gradient row is flushed
    // and the next head's bias row loaded (~10 us: these runs are kept SHORT -- upl fills the CUs the full runs leave -- so that they end
    // with the full runs in spite of it).  Binding every workgroup to one head (round 4) left Swin-L stage 2 (72 windows x 24 heads) at 216
    // workgroups of 8 units on 256 CUs; this gives 240 runs of 7 + 16 of 3.
    const int q_runs = B_ / upw, main_wgs = q_runs * nH, left = B_ - q_runs * upw;
    int h, b, lo, hi, count;             // current unit, the window range [lo, hi) the run walks per head, units in the run
    if ((int)blockIdx.x < main_wgs) {
        h = (int)blockIdx.x % nH;
        lo = ((int)blockIdx.x / nH) * upw; hi = lo + upw;
        b = lo; count = upw;
    } else {
        const int j0 = ((int)blockIdx.x - main_wgs) * upl;           // first leftover unit of this run; unit j = (head j / left, window lo + j % left)
        lo = q_runs * upw; hi = B_;
        h = j0 / left; b = lo + (j0 - h * left);
        count = min(upl, nH * left - j0);
    }
    const int C = nH * 32;
    const int64_t rowst = 3 * (int64_t)C;
    const int tid = threadIdx.x, nthreads = (NT + (HELP ? 3 : 0)) * 64;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = HELP && wv >= NT;      // waves 9..11: strips 0 / 4 / 8, query tiles 6..8
    const int w = helper ? 4 * (wv - NT) : wv; // the key strip this wave works on
    const bool shared = HELP && !helper && (w & 3) == 0;      // a strip owner that has a helper: query tiles 0..5 only
    const int l = tid & 63, g = l >> 4, c16 = l & 15;
    const int srow = tid >> 2, sc = tid & 3;   // staging role: row, 16-byte chunk
    const bool stager = tid < N * 4;
    const int key = 16 * w + c16;              // this lane's key in phase 1
    const bool kok = key < N;
    // transpose-read lane pointers: phase 1 k-slots are rows {32t+4g+j, 32t+16+4g+j}; phase 2 rows 32t+8g+j
    DGX_LDS const uint16_t* q_l4 = tr_lane_ptr(Qs, RR, 4 * g, 0, c16);       // dOs / Ks images are NP*RR / 2*NP*RR further on
    DGX_LDS const uint16_t* do_l4 = q_l4 + NP * RR;
    DGX_LDS const uint16_t* k_l8 = tr_lane_ptr(Ks, RR, 8 * g, 0, c16);
    DGX_LDS const uint16_t* ds_l8 = tr_lane_ptr(dSt, RD, 8 * g, 16 * w, c16);
    DGX_LDS uint16_t* img_st = lds_opaque(Qs + srow * RR + 8 * sc);         // staging destinations
    DGX_LDS uint16_t* v_st = lds_opaque(Vs + srow * RR + 8 * sc);
    DGX_LDS float* meta_st = lds_opaque(lse_s + (PK ? 2 * srow : srow));    // lse_s / delta_s / qoff_s / reg_s are NP apart (PK: pairs)
    // phase-1 lane pointers (everything in the unrolled loop is one of these + a constant)
    DGX_LDS const uint16_t* q_row = lds_opaque(Qs + c16 * RR + 8 * g);      // A fragment of query tile qt: + 16*qt*RR
    DGX_LDS const uint16_t* do_row = q_row + NP * RR;
    DGX_LDS const float* ld_lane = lds_opaque(lse_s + 2 * (4 * g + (c16 & 3)));   // PK: {-lse/scale, -delta} of query 16 qt + 4 g + (lane & 3): + 32 qt
    DGX_LDS const int* reg_lane = lds_opaque(reg_s + 4 * g + (c16 & 3));          // PK: its region id: + 16 qt
    DGX_LDS const float* lse_g = lds_opaque(lse_s + 4 * g);                 // rows 16*qt + 4g .. +3; the four arrays are NP apart
    DGX_LDS const float* delta_g = lse_g + NP;
    DGX_LDS const int* qoff_g = reinterpret_cast<DGX_LDS const int*>(lse_g) + 2 * NP;
    DGX_LDS const int* reg_g = reinterpret_cast<DGX_LDS const int*>(lse_g) + 3 * NP;
    DGX_LDS uint16_t* ds_w = lds_opaque(dSt + key * RD + 4 * g);            // dS^T row `key`, queries 16*qt + 4g ..

    // global addressing = uniform (SGPR) window base + one 32-bit lane offset per access pattern
    const uint32_t st_qk_c = (uint32_t)(srow * (int)rowst + 8 * sc);     // staging chunk inside this window's qkv rows
    const uint32_t st_o_c = (uint32_t)(srow * C + 8 * sc);               // ... inside out / dout rows
    const uint32_t row4_c = (uint32_t)((kok ? key : 0) * (int)rowst + 4 * g);   // output row = this lane's key / query, entries 4g ..
    // COMPACT: the row of token n of window b -- a real token's compact row, or -(1 + padding rank) (attn_row)
    auto compact_row = [&](int b, int n) -> int {
        const WmGeom wg = wm_geom(G.H, G.W, WS, G.shift);
        const int img = b / nW, wi = b - img * nW, wr = wi / wg.nWw, wc = wi - wr * wg.nWw;
        const WmWindow ww = wm_window(wg, wr, wc);
        if (ww.rh == WS && ww.rw == WS) return img * (G.H * G.W) + ww.base + n;      // no padding token in this window (uniform branch)
        return attn_row<WS>(G, wg, ww, img, wr, wc, n);
    };
    auto issue = [&](int b, int h, BwdPrefetch& P) {
        if (COMPACT) {
            // q / k / v of a padding token = the qkv bias, its dO and O = 0 (its output is cropped: no gradient reaches it)
            const int row = compact_row(b, stager ? srow : 0);
            const uint16_t* qp = (row >= 0 ? qkv + (int64_t)row * rowst : G.bias) + h * 32 + 8 * sc;
            const uint16_t* dp = row >= 0 ? dout + (int64_t)row * C + h * 32 + 8 * sc : G.zeros + 8 * sc;
            const uint16_t* op = row >= 0 ? out + (int64_t)row * C + h * 32 + 8 * sc : G.zeros + 8 * sc;
            if (stager) {
                P.q = *reinterpret_cast<const bf16x8*>(qp);
                P.k = *reinterpret_cast<const bf16x8*>(qp + C);
                P.v = *reinterpret_cast<const bf16x8*>(qp + 2 * C);
                P.d = *reinterpret_cast<const bf16x8*>(dp);
                P.o = *reinterpret_cast<const bf16x8*>(op);
                P.lse = (lse + ((int64_t)b * nH + h) * N)[(uint32_t)srow];
                P.reg = (int)(region + (MASKED ? (int64_t)(b % nW) * N : 0))[(uint32_t)srow];      // W-MSA: the one zero row, whatever nW is
            }
            return;
        }
        const uint16_t* base = qkv + (int64_t)b * N * rowst + h * 32;
        const uint16_t* dob = dout + (int64_t)b * N * C + h * 32;
        const uint16_t* ob = out + (int64_t)b * N * C + h * 32;
        // (the registers of lanes that stage nothing are initialised ONCE in front of the loop: re-writing them here made the compiler
        // order the writes behind every outstanding memory operation of the previous window -- its six stores included)
        // opaque copies: the 64-bit addresses are rebuilt per window (2 VALU each) instead of living in
        // ~20 registers across the whole loop
        uint32_t st_qk = st_qk_c, st_o = st_o_c;
        asm volatile("" : "+v"(st_qk), "+v"(st_o));
        if (stager) {
            P.q = *reinterpret_cast<const bf16x8*>(base + st_qk);
            P.k = *reinterpret_cast<const bf16x8*>(base + C + st_qk);
            P.v = *reinterpret_cast<const bf16x8*>(base + 2 * C + st_qk);
            P.d = *reinterpret_cast<const bf16x8*>(dob + st_o);
            P.o = *reinterpret_cast<const bf16x8*>(ob + st_o);
            P.lse = (lse + ((int64_t)b * nH + h) * N)[(uint32_t)srow];
            P.reg = (int)(region + (int64_t)(b % nW) * N)[(uint32_t)srow];
        }
        // (no select on the loaded value here: `if (!kok) P.v = z` made the compiler wait for EVERY load of this prefetch -- vmcnt(0)
        // -- right behind their issue, at the top of phase 1: the whole memory latency of the next window's operands, ~half of the
        // kernel's time, was exposed in front of the math it was meant to fly under (round 5, tools/r05_attn_variants.sh: removing
        // the exponentials, the metadata reads, the transpose reads or the S / dP MFMAs changed the phase's duration by < 3 % each).
        // Rows past the window (window 7 only) are zeroed where the fragment is consumed.)
    };

    // ---- LDS setup: the head's bias row (again at a head boundary inside the run), once: zero padding rows/columns, rel-pos offsets
    // The row is read from memory ONCE, every lane's entries requested before the first is used, and the four shifted copies are made
    // from the LDS copy: as five loops of `load, wait, write` per entry (round 4) this prologue cost every workgroup 4-5 memory round trips
    // in series -- ~20 us of a 56-94 us launch (the per-window clocks of r05_attn_bwd_phases.txt add up to 28-40 us).
    BwdPrefetch P;
    {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        P.q = P.d = P.o = P.k = P.v = z;
        P.lse = INFINITY;
        P.reg = 0;
    }
    if (count > 0) issue(b, h, P);       // the first window's operands fly under the whole prologue
    auto load_head = [&](int hh) {
        constexpr int NTHR = (NT + (HELP ? 3 : 0)) * 64, KT = (TBL + NTHR - 1) / NTHR;
        float tv[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int i = tid + k * NTHR;
            tv[k] = table[hh * dt_sh + (i < TBL ? i : 0) * dt_si];
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int i = tid + k * NTHR;
            if (i < TBL) { tbl[i] = tv[k] * DGX_LOG2E; tblacc[i] = 0.f; }      // log2 domain
        }
        if (PK) {
            __syncthreads();
            for (int i = tid; i < 4 * TBLS; i += nthreads) {
                const int sft = i / TBLS, j = i - sft * TBLS + sft;
                tbl4[i] = j < TBL ? tbl[j] : 0.f;
            }
        }
    };
    load_head(h);
    auto zero_pads = [&]() {             // (window 7: rows N .. NP of the images; window 12 has none)
        for (int i = tid; i < (NP - N) * RR; i += nthreads) { Qs[N * RR + i] = 0; dOs[N * RR + i] = 0; Ks[N * RR + i] = 0; Vs[N * RR + i] = 0; }
        for (int i = tid; i < (NP - N) * RD; i += nthreads) dSt[N * RD + i] = 0;   // padded key rows feed the last K=32 step of dQ
    };
    zero_pads();
    for (int i = tid; i < NP; i += nthreads) {
        const int yq = i / WS;
        qoff_s[i] = i < N ? yq * (2 * WS - 1) + (i - yq * WS) : 0;
        if (i >= N) {                                                           // padded queries: p = 0
            if (PK) { lse_s[2 * i] = -INFINITY; lse_s[2 * i + 1] = 0.f; }
            else { lse_s[i] = -INFINITY; delta_s[i] = 0.f; }
            reg_s[i] = 0;
        }
    }
    const int yk = key / WS, xk = key - yk * WS;
    const int kbase = kok ? (WS - 1) * (2 * WS - 1) + (WS - 1) - (yk * (2 * WS - 1) + xk) : 0;
    const float kneg = kok ? 0.0f : -INFINITY;   // padded key columns: p = 0
    const float scale2 = scale * DGX_LOG2E, inv_scale = 1.0f / scale;
    DGX_LDS const float* tbl_k = lds_opaque(tbl + kbase);
    DGX_LDS const float* bptr[NT];       // PK: this lane's bias run of query tile qt (queries 16 qt + 4 g .. + 3, key `key`)
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int q0 = 16 * qt + 4 * g, y0 = q0 / WS;
        const int idx0 = kbase + y0 * (2 * WS - 1) + (q0 - y0 * WS);
        bptr[qt] = lds_opaque(tbl4 + (idx0 & 3) * TBLS + (idx0 & ~3));
    }
    float dbias[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbias[i][r] = 0.f;

    // The window loop is ROTATED: staging of window b + 1 closes iteration b.  With the staging at the top, the compiler's wait-count
    // model merged the pre-loop path (prefetch in flight, nothing behind it) into the loop header and waited `vmcnt(3..0)` there -- in
    // steady state that is "all loads AND the previous window's six dQ / dK / dV stores complete": the stores' latency was exposed in
    // front of every staging.  Here the wait sits behind a unique path (7 loads, then 6 stores) and only covers the loads.
    auto stage = [&]() {
        if (stager) {
            *reinterpret_cast<DGX_LDS bf16x8*>(img_st) = P.q;                   // Qs, dOs, Ks are NP*RR apart
            *reinterpret_cast<DGX_LDS bf16x8*>(img_st + NP * RR) = P.d;
            *reinterpret_cast<DGX_LDS bf16x8*>(img_st + 2 * NP * RR) = P.k;
            *reinterpret_cast<DGX_LDS bf16x8*>(v_st) = P.v;
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d += bf2f((uint16_t)P.d[i]) * bf2f((uint16_t)P.o[i]);
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            // stored negated and pre-divided: they are the INITIAL accumulators of the two MFMAs of phase 1, so that
            // S*scale2 - lse2 and dP - delta cost no VALU instruction (lse_s = -lse/scale, delta_s = -delta)
            if (sc == 0) {
                if (PK) {
                    *reinterpret_cast<DGX_LDS wa_f32x2*>(meta_st) = wa_f32x2{-P.lse * inv_scale, -d};
                    reinterpret_cast<DGX_LDS int*>(lds_opaque(reg_s + srow))[0] = P.reg;
                } else {
                    meta_st[NP] = -d; meta_st[0] = -P.lse * inv_scale; reinterpret_cast<DGX_LDS int*>(meta_st)[3 * NP] = P.reg;
                }
            }
        }
    };
    // Bias gradient of head hh.  A lane's 36 registers are the entries dS_sum[q][key] of the run's summed dS for its key and its queries;
    // entry (dy, dx) of the bias row is the sum of dS_sum over the pairs with yq - yk = dy, xq - xk = dx (1 .. N pairs).  The registers
    // are WRITTEN as the N x N matrix over the staging images (free between two windows), and one lane per table entry adds its pairs
    // up in a fixed order.  Rounds 2-4 reduced them with `ds_add_f32` into an LDS table: LDS float atomics whose lanes share
    // addresses run at about one LANE per two cycles -- 27 600 of them cost every workgroup 24-27 us behind its last window, a third of the
    // stage-2 launch (45 -> 73 us at 72 x 24; tools/r05_attn_sw.sh ablations; the global float atomics behind them cost 2 us).
    // The entries then go to this run's slot of the workspace, and the LAST of the runs that hold a part of head hh (a device counter per
    // head tells which one that is) adds the slots up in slot order onto dtable: one atomic per run and head, and a sum that no longer
    // depends on the order of arrival.  The registers start again at zero.
    constexpr int TBLP = (TBL + 3) / 4 * 4;          // stride of a slot
    constexpr int LDM = N + 1;                       // row stride of the matrix (odd: the 16 keys of a wave's store land in 16 banks)
    static_assert((size_t)N * LDM * 4 <= (size_t)(3 * NP * RR + NP * RD) * 2, "the pair matrix lies over the Q / dO / K / dS^T images");
    const int slots_per_head = q_runs + (left > 0 ? (left + upl - 1) / upl + 1 : 0);      // the head's full runs + the runs over its left-over windows
    __shared__ int last_s;
    auto flush_head = [&](int hh) {
        constexpr int TSPLIT_Q = 3;      // = TSPLIT of the window loop
#ifdef ABL_NOFLUSH      // (tools/r05_attn_sw.sh: no table-gradient reduction at all -- which also lets the compiler drop the 36 accumulations
        return;         //  per window from the VALU-bound phase 1, so the difference to the kernel as built overstates the flush)
#endif
        DGX_LDS float* M = reinterpret_cast<DGX_LDS float*>(lds_opaque(Qs));
        int krow = key * LDM + 4 * g;
        asm volatile("" : "+v"(krow));   // opaque: keeps the 36 store addresses from being formed above the window loop (registers)
        if (kok) {
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                if (HELP && (qt < 2 * TSPLIT_Q ? helper : shared)) continue;      // the other wave of the strip holds these terms
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * g + r;
                    if (q < N) M[krow + 16 * qt + r] = dbias[qt][r];
                    dbias[qt][r] = 0.f;
                }
            }
        }
        __syncthreads();
        // the runs over head hh's left-over windows: units [hh * left, (hh + 1) * left) of the left-over list, run = unit / upl
        const int lrun0 = left > 0 ? (hh * left) / upl : 0;
        const int nlrun = left > 0 ? ((hh + 1) * left - 1) / upl - lrun0 + 1 : 0;
        const int slot = (int)blockIdx.x < main_wgs ? (int)blockIdx.x / nH : q_runs + ((int)blockIdx.x - main_wgs - lrun0);
        float* mine = part_ws + ((int64_t)hh * slots_per_head + slot) * TBLP;
        // Slots and counter are DEVICE-scope ATOMIC accesses (write-through stores, L2-bypassing loads: the runs of a head sit on different
        // XCDs, whose L2s do not snoop each other); no location is touched by a plain access of two workgroups.  Their ORDER is by completion,
        // which is a gfx942 / gfx950 property, not a guarantee of the HIP memory model for relaxed atomics: an agent-scope (sc1) store is
        // acknowledged -- vmcnt counts it down -- only once it is visible at the device's coherence point, so `s_waitcnt vmcnt(0)` + the
        // workgroup barrier put every slot store before the counter's fetch_add, and the last arriver's slot loads are issued behind the
        // fetch_add's RETURN value (control dependence through `last_s` and a barrier).  The model-conforming form -- release on the
        // fetch_add, acquire in the last arriver -- makes the compiler emit `buffer_wbl2 sc1` (release) and `buffer_inv sc1` (acquire):
        // a write-back of the whole L2, full of this kernel's dq / dk / dv rows, measured at 40 us per flush (x 24 heads per launch).
        // The file refuses to build for any other architecture (top of the file).
        for (int i = tid; i < TBL; i += nthreads) {
            const int iy = i / (2 * WS - 1), dy = iy - (WS - 1), dx = i - iy * (2 * WS - 1) - (WS - 1);
            const int yk0 = dy < 0 ? -dy : 0, yk1 = dy > 0 ? WS - dy : WS, xk0 = dx < 0 ? -dx : 0, xk1 = dx > 0 ? WS - dx : WS;
            float sum = 0.f;
            for (int yk = yk0; yk < yk1; ++yk) {
                DGX_LDS const float* row = M + (yk * WS) * LDM + (yk + dy) * WS + dx;        // + xk * (LDM + 1)
                for (int xk = xk0; xk < xk1; ++xk) sum += row[xk * (LDM + 1)];
            }
            __hip_atomic_store(mine + i, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef ABL_NOCOUNT      // (breakdown: the flush without the slot hand-over)
        __syncthreads();
        return;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // (the matrix has been read: the images may be staged again)
        if (tid == 0) last_s = __hip_atomic_fetch_add(&head_cnt[hh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == q_runs + nlrun - 1;
        __syncthreads();
        if (last_s) {
            const float* p0 = part_ws + (int64_t)hh * slots_per_head * TBLP;
            const int ns = q_runs + nlrun;
            for (int i = tid; i < TBL; i += nthreads) {
                float sum = dtable[hh * dt_sh + i * dt_si];
                for (int s0 = 0; s0 < ns; s0 += 16) {          // sixteen slots requested before the first is added: one round trip per batch
                    float v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        v[k] = __hip_atomic_load(p0 + (s0 + k < ns ? s0 + k : s0) * TBLP + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < 16; ++k) sum += s0 + k < ns ? v[k] : 0.f;
                }
                dtable[hh * dt_sh + i * dt_si] = sum;
            }
            if (tid == 0) __hip_atomic_store(&head_cnt[hh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
        }
    };
    if (count > 0) stage();
    // The run as segments of one head each: the window loop inside, the change of head (flush, next bias row) between two segments --
    // written INTO the window loop it lengthened every live range of the loop and the masked kernel spilled 12 registers, whose reloads
    // wait for the prefetch in flight.
    int done = 0;
    while (done < count) {
    const int seg = min(count - done, hi - b);       // units up to the end of the run or of this head's windows
    for (int u = 0; u < seg; ++u) {
        const bool more = done + u + 1 < count;      // another unit in the run: its operands are requested under this one's math
        const bool wrap = u + 1 == seg;              // ... and it is the first unit of the next segment
        const int bn = b + 1 == hi ? lo : b + 1, hn = b + 1 == hi ? h + 1 : h;
#ifdef DIAG_CLOCK
        unsigned long long tprev = clock64();
#endif
        CLK(0);
        __syncthreads();
        CLK(1);
#if !defined(PREFETCH_LATE)
        if (more) issue(bn, hn, P);   // next window's loads fly under this window's math
#endif
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(kok ? key : 0) * RR + 8 * g]);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vs[(kok ? key : 0) * RR + 8 * g]);
        const int rk = MASKED ? reg_s[kok ? key : 0] : 0;

        // ---- phase 1: this wave's 16 keys x its query tiles, two query tiles (one K=32 step) at a time
        f32x4 dV[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 dK[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // ONE unrolled copy of the five steps (every register index below is static); a wave skips the steps of the other wave of its
        // strip with a scalar branch: the owner of a shared strip runs steps 0..2, its helper steps 3..4
        constexpr int TSPLIT = 3;
        // Shifted windows: most windows of a shifted layer lie inside ONE region (25 of the 36 of a 6 x 6 grid) and need no mask; the
        // phase is VALU-bound with three waves per SIMD, so the 16 instructions per query tile that fetch, spread and compare the region
        // ids cost it 60 % (6 900 against 4 200 cycles per window on the harness).  Such a window skips them;
        // in the others a masked pair's probability is SET to zero behind the exponential (the reference's exp(s - 100 - lse) is below
        // 4e-44 times the unmasked value: zero in the bf16 operands it feeds, and nothing in an fp32 sum next to real terms).
        bool mixed = false;
        if (MASKED) {
            const int r0 = reg_s[0];
            bool diff = false;
            for (int i = l; i < N; i += 64) diff = diff || reg_s[i] != r0;
            mixed = __builtin_amdgcn_readfirstlane((int)(__ballot(diff) != 0ull)) != 0;
        }
        // (ONE copy of the phase with scalar branches around the mask's instructions: two copies behind one branch spilled 58 registers)
        const bool MK = MASKED && mixed;
#pragma unroll
        for (int t = 0; t < NTK / 2; ++t) {
            if (HELP && (t < TSPLIT ? helper : shared)) continue;
            uint32_t ppk[2][2], dpk[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int qt = 2 * t + hh;
                float pv[4] = {0.f, 0.f, 0.f, 0.f}, dsv[4] = {0.f, 0.f, 0.f, 0.f};
                if (qt < NT) {
                    const bf16x8 qa = *reinterpret_cast<DGX_LDS const bf16x8*>(q_row + 16 * qt * RR);
                    const bf16x8 da = *reinterpret_cast<DGX_LDS const bf16x8*>(do_row + 16 * qt * RR);
                    f32x4 lv, dl, bv = {0.f, 0.f, 0.f, 0.f};
                    i32x4 ov = {0, 0, 0, 0}, rv = {0, 0, 0, 0};
#ifdef ABL_META
                    if constexpr (PK) { lv = f32x4{scale2, scale2, inv_scale, 0.f}; dl = lv; bv = lv; }
                    else
#endif
                    if constexpr (PK) {
                        const wa_f32x2 ld = *reinterpret_cast<DGX_LDS const wa_f32x2*>(ld_lane + 32 * qt);
                        lv = f32x4{quad_bcast<0>(ld.x), quad_bcast<1>(ld.x), quad_bcast<2>(ld.x), quad_bcast<3>(ld.x)};
                        dl = f32x4{quad_bcast<0>(ld.y), quad_bcast<1>(ld.y), quad_bcast<2>(ld.y), quad_bcast<3>(ld.y)};
                        bv = *reinterpret_cast<DGX_LDS const f32x4*>(bptr[qt]);
                        if (MK) {
                            const int rq1 = reg_lane[16 * qt];
                            rv = i32x4{quad_bcast<0>(rq1), quad_bcast<1>(rq1), quad_bcast<2>(rq1), quad_bcast<3>(rq1)};
                        }
                    } else {
                        lv = *reinterpret_cast<DGX_LDS const f32x4*>(lse_g + 16 * qt);     // -lse[q] / scale
                        dl = *reinterpret_cast<DGX_LDS const f32x4*>(delta_g + 16 * qt);   // -delta[q]
                        ov = *reinterpret_cast<DGX_LDS const i32x4*>(qoff_g + 16 * qt);
                        if (MK) rv = *reinterpret_cast<DGX_LDS const i32x4*>(reg_g + 16 * qt);
                    }
#ifdef ABL_MFMA1
                    const f32x4 s = lv + f32x4{bf2f(qa[0]), bf2f(qa[1]), bf2f(kf[0]), bf2f(kf[1])}, dp = dl + f32x4{bf2f(da[0]), bf2f(da[1]), bf2f(vf[0]), bf2f(vf[1])};
#else
                    const f32x4 s = mfma16(qa, kf, lv);    // s[r]  = S[q 16qt+4g+r][key] - lse[q]/scale
                    const f32x4 dp = mfma16(da, vf, dl);   // dp[r] = dP[q][key] - delta[q]
#endif
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sv = __builtin_fmaf(s[r], scale2, PK ? bv[r] : tbl_k[ov[r]]);       // log2 domain (bias row, lse pre-scaled)
                        if (N % 16 != 0) sv += kneg;
#ifdef ABL_EXP
                        pv[r] = sv * 0.001f;
#else
                        pv[r] = __builtin_amdgcn_exp2f(sv);
#endif
                        if (MK) pv[r] = rv[r] != rk ? 0.0f : pv[r];
                        dsv[r] = pv[r] * dp[r];
                        dbias[qt < NT ? qt : 0][r] += dsv[r];
                    }
                }
                ppk[hh][0] = pack_bf2(pv[0], pv[1]);
                ppk[hh][1] = pack_bf2(pv[2], pv[3]);
                dpk[hh][0] = pack_bf2(dsv[0], dsv[1]);
                dpk[hh][1] = pack_bf2(dsv[2], dsv[3]);
                // dS^T image row = key, 4 consecutive queries: one 8-byte store
#ifndef ABL_DSWRITE
                if (qt < NT)
                    *reinterpret_cast<DGX_LDS u32x2*>(ds_w + 16 * qt) = u32x2{dpk[hh][0], dpk[hh][1]};
#endif
#ifndef NO_HH_BARRIER
                __builtin_amdgcn_sched_barrier(0);   // one query tile at a time: its 40-odd temporaries die before the next starts
#endif
            }
            u32x4 a = {ppk[0][0], ppk[0][1], ppk[1][0], ppk[1][1]};
            u32x4 d = {dpk[0][0], dpk[0][1], dpk[1][0], dpk[1][1]};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, a);   // A = P^T : row = key, slots = queries
            const bf16x8 df = __builtin_bit_cast(bf16x8, d);   // A = dS^T
            // B operands = dO / Q rows {32t+4g+j, 32t+16+4g+j} (the k-slot order of P^T / dS^T), by transpose reads
            // swapped operands: D = (dO^T) (P) = dV^T, so a lane ends up with 4 CONSECUTIVE head-dim entries of ONE key
            // (8-byte stores; with P^T as the A operand it held one entry of 4 keys: 2-byte stores, 4 x 32-byte pieces each)
#ifdef ABL_TR
            dV[0] = mfma16(pf, pf, dV[0]); dV[1] = mfma16(df, pf, dV[1]); dK[0] = mfma16(pf, df, dK[0]); dK[1] = mfma16(df, df, dK[1]);
#else
            dV[0] = mfma16(tr_frag(do_l4, 32 * t * RR, (32 * t + 16) * RR), pf, dV[0]);
            dV[1] = mfma16(tr_frag(do_l4, 32 * t * RR + 16, (32 * t + 16) * RR + 16), pf, dV[1]);
            dK[0] = mfma16(tr_frag(q_l4, 32 * t * RR, (32 * t + 16) * RR), df, dK[0]);
            dK[1] = mfma16(tr_frag(q_l4, 32 * t * RR + 16, (32 * t + 16) * RR + 16), df, dK[1]);
#endif
#ifndef NO_T_BARRIER
            __builtin_amdgcn_sched_barrier(0);   // keep the t-steps apart: shorter live ranges, no spills
#endif
        }
        CLK(2);
        uint16_t* dqb = dqkv + (COMPACT ? (int64_t)0 : (int64_t)b * N * rowst) + h * 32;
        uint32_t row4 = row4_c;
        if (COMPACT) {                   // this lane's key / query: its compact row, or its padding row behind the T real ones
            const int row = compact_row(b, kok ? key : 0);
            row4 = (uint32_t)(row >= 0 ? row : G.T - 1 - row) * (uint32_t)rowst + 4 * g;
        }
        asm volatile("" : "+v"(row4));   // same: store addresses are per-window temporaries
        auto store_dkdv = [&]() {
            if (kok) {                   // this lane: key 16w + c16, head-dim entries 16 dt + 4g .. + 3
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    *reinterpret_cast<u32x2*>(dqb + C + row4 + 16 * dt) =
                        u32x2{pack_bf2(dK[dt][0] * scale, dK[dt][1] * scale), pack_bf2(dK[dt][2] * scale, dK[dt][3] * scale)};
                    *reinterpret_cast<u32x2*>(dqb + 2 * C + row4 + 16 * dt) = u32x2{pack_bf2(dV[dt][0], dV[dt][1]), pack_bf2(dV[dt][2], dV[dt][3])};
                }
            }
        };
        DGX_LDS f32x4* part_l = reinterpret_cast<DGX_LDS f32x4*>(lds_opaque(part_s + ((w >> 2) * 4 * 64 + l) * 4));   // + 64 f32x4 per vector
        if (helper) {                    // partial dV / dK of query tiles 6..8 -> the strip's owner
            part_l[0] = dV[0]; part_l[64] = dV[1]; part_l[128] = dK[0]; part_l[192] = dK[1];
        } else if (!shared) {
            store_dkdv();
        }
        CLK(3);
        __syncthreads();  // dS^T image (and the helpers' partial sums) complete
        CLK(4);
        if (shared) {
            dV[0] += part_l[0]; dV[1] += part_l[64]; dK[0] += part_l[128]; dK[1] += part_l[192];
            store_dkdv();
        }
        u32x2 dqpk[2] = {{0u, 0u}, {0u, 0u}};
        if (!helper) {                   // phase 2 belongs to the nine query strips
#if defined(PREFETCH_LATE)
        if (more) issue(bn, hn, P);   // next window's loads fly under phase 2 (phase 1 has no registers to spare)
#endif
        // ---- phase 2: dQ strip w = dS[16w.., :] K ; A = dS rows (queries 16w + c16) out of the key-major image
        f32x4 dQ[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < NTK / 2; ++t) {
            const bf16x8 sa = tr_frag(ds_l8, 32 * t * RD, (32 * t + 4) * RD);   // k-slots = keys 32t+8g .. +7 for A and B
            dQ[0] = mfma16(tr_frag(k_l8, 32 * t * RR, (32 * t + 4) * RR), sa, dQ[0]);            // swapped: dQ^T, see dV / dK
            dQ[1] = mfma16(tr_frag(k_l8, 32 * t * RR + 16, (32 * t + 4) * RR + 16), sa, dQ[1]);
        }
        CLK(5);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            dqpk[dt] = u32x2{pack_bf2(dQ[dt][0] * scale, dQ[dt][1] * scale), pack_bf2(dQ[dt][2] * scale, dQ[dt][3] * scale)};
        CLK(6);
        }
        __syncthreads();                 // this window's LDS consumers are done
        if (more && !wrap) stage();
        // the dQ rows leave BEHIND the staging of the next window: the staging waits for its prefetch with vmcnt counts that also
        // cover every store issued before it (one in-order counter), and stores issued a few cycles earlier would put their whole
        // latency in front of it; the dK / dV stores are a phase 2 old by then
        if (!helper && kok) {            // query 16w + c16
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<u32x2*>(dqb + row4 + 16 * dt) = dqpk[dt];
        }
        if (!wrap) { b = bn; h = hn; }
    }
    done += seg;
    flush_head(h);                       // nobody reads the bias row or the images between the loop's last barrier and the next staging
    if (done < count) {                  // head boundary inside the run
        if (NP > N) zero_pads();         // (the flush's matrix lay over the images' padding rows)
        h = b + 1 == hi ? h + 1 : h;
        b = b + 1 == hi ? lo : b + 1;
        load_head(h);
        stage();
    }
    }
}

template <int WS>
static size_t bwd_smem_bytes(bool help = false) {
    using Cf = WinCfg<WS>;
    constexpr int NP = Cf::NTK * 16;
    return (size_t)(3 * NP * (WS == 12 ? 48 : 40) + NP * (WS == 12 ? 184 : NP + 8)) * 2 + (size_t)(4 * NP + 2 * Cf::TBL) * 4 + 64 + (help ? 16 + 3 * 4 * 64 * 16 : 0) +
           (WS == 12 ? 16 + 4 * ((Cf::TBL + 6) / 4 * 4) * 4 : 0) + (size_t)NP * (WS == 12 ? 48 : 40) * 2;
}

// Workspace of the backward kernel's bias-gradient reduction: one slot of TBL floats per (head, run) and a counter per head, zero between
// launches (the last run of a head resets it).  One per (device, stream): launches on ONE stream are ordered, launches on different streams
// must not share slots, and the null stream of two devices is the same handle.  Allocated at the first call on a stream; inside a graph
// capture the allocation runs with the thread's capture mode relaxed (hipMalloc is not a stream operation) and the counters' zero-fill is
// recorded into the graph (a memset of 256 bytes per replay: the counters are zero between launches anyway).  `dirty` = the previous launch
// on this stream was refused by the runtime: its counters may not have been reset, so the next call zero-fills them first.
struct BwdTableWs { float* part; int* cnt; int64_t floats; int heads; bool dirty; };
static BwdTableWs* bwd_table_ws(hipStream_t st, int64_t floats, int heads) {
    static std::map<std::pair<int, hipStream_t>, BwdTableWs> pool;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    BwdTableWs& w = pool[std::make_pair(device, st)];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (w.part && w.floats >= floats && w.heads >= heads) {
        // A launch that is being CAPTURED always carries its own zero-fill of the counters (a 256-byte memset node in front of it): the
        // workspace may have been allocated inside an earlier capture on a stream handle the framework has since handed out again (torch
        // draws its streams from a pool of 32 per device) -- then the only zero-fill its counters ever got sits in THAT graph, and a graph
        // captured now would replay onto whatever hipMalloc returned.  Found by running the one-rank RCCL test behind the loader test in
        // one process: every relative-position table gradient of the first replay was 0 (the flush condition never met).
        if (w.dirty || capturing) {
            if (hipMemsetAsync(w.cnt, 0, (size_t)w.heads * 4, st) != hipSuccess) return nullptr;
            if (!capturing) w.dirty = false;
        }
        return &w;
    }
    if (w.part && capturing) return nullptr;                  // growing means freeing what earlier captured launches point at
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (capturing && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) return nullptr;
    if (w.part) { (void)hipStreamSynchronize(st); (void)hipFree(w.part); (void)hipFree(w.cnt); w.part = nullptr; w.cnt = nullptr; }
    const int64_t nf = floats > (int64_t)300 * 532 ? floats : (int64_t)300 * 532;     // every Swin-L / Swin-T launch shape fits the first allocation
    const int nh = heads > 64 ? heads : 64;
    bool ok = hipMalloc((void**)&w.part, (size_t)nf * 4) == hipSuccess;
    if (!ok) w.part = nullptr;
    if (ok && hipMalloc((void**)&w.cnt, (size_t)nh * 4) != hipSuccess) { (void)hipFree(w.part); w.part = nullptr; w.cnt = nullptr; ok = false; }
    if (capturing) (void)hipThreadExchangeStreamCaptureMode(&mode);       // back to what the capture was started with
    if (!ok) return nullptr;
    if (hipMemsetAsync(w.cnt, 0, (size_t)nh * 4, st) != hipSuccess) return nullptr;      // ordered before the launch that follows on `st`
    w.floats = nf; w.heads = nh;
    w.dirty = capturing;      // allocated inside a capture: the zero-fill above is a graph node, the memory itself is still what hipMalloc returned
    return &w;
}

// NULL region (W-MSA) is served by a process-lifetime all-zero row: one code path in the kernels
// (the compiler's null/non-null loop versions differ 4x in speed for the backward kernel).
static const int8_t* zero_region() {
    static int8_t* z = nullptr;
    if (!z) {
        if (hipMalloc((void**)&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return z;
}

static int attn_geom(AttnGeom& G, int B_, int& nW, int ws, int B, int H, int W, int shift, const void* qkv_bias, int nH) {
    // compact order: B images of H x W tokens, window ws, cyclic shift `shift`; B_ = B * nW windows
    if (B <= 0 || H <= 0 || W <= 0 || shift < 0 || shift >= ws || !qkv_bias || !wm_compact_ok(H, W, ws, shift)) return DGX_ERR_BAD_ARG;
    const WmGeom g = wm_geom(H, W, ws, shift);
    nW = g.nWh * g.nWw;
    if ((int64_t)B * nW != B_) return DGX_ERR_BAD_ARG;
    const int64_t Tw = (int64_t)B_ * ws * ws;
    if (Tw * 3 * nH * 32 >= ((int64_t)1 << 32)) return DGX_ERR_UNSUPPORTED;      // 32-bit element offsets of the row addressing
    G.H = H; G.W = W; G.shift = shift;
    G.T = B * H * W;
    G.P = nW * ws * ws - H * W;
    G.bias = (const uint16_t*)qkv_bias;
    return DGX_OK;
}

static int attn_fwd_launch(const void* qkv, const float* table, int64_t table_stride_head, int64_t table_stride_index, const int8_t* region,
                           void* out, float* lse, int B_, int nW, int nH, int ws, float scale, const AttnGeom* geom, void* stream) {
    if (B_ <= 0) return DGX_OK;
    if (!qkv || !table || !out || !lse || nH <= 0 || nW <= 0 || (region && B_ % nW)) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool masked = region != nullptr;
    if (!region) { region = zero_region(); if (!geom) nW = 1; if (!region) return DGX_ERR_BAD_ARG; }
    const int grid = ((B_ + 7) / 8) * 8 * nH;
    // head_dim 32: QK^T and PV (2 x 2 N^2 32 FLOP per head), q, k, v in + out (bf16) + lse
    const double heads = (double)B_ * nH, ntok = (double)ws * ws;
    DgxProfScope prof(DGX_PROF_ATTN_FWD, stream, heads * 2.0 * 2.0 * ntok * ntok * 32.0, heads * ntok * (32.0 * 2.0 * 4.0 + 4.0));
    AttnGeom G = geom ? *geom : AttnGeom{0, 0, 0, 0, 0, nullptr, nullptr};
#define FWD_LAUNCH(WSV, MK, CP) hipLaunchKernelGGL((win_attn_fwd_kernel<WSV, MK, CP>), dim3(grid), dim3(WinCfg<WSV>::NT * 64), 0, st, \
                                                  (const uint16_t*)qkv, table, region, (uint16_t*)out, lse, B_, nW, nH, scale, \
                                                  table_stride_head, table_stride_index, G)
#define FWD_LAUNCH_C(WSV, MK) do { if (geom) FWD_LAUNCH(WSV, MK, true); else FWD_LAUNCH(WSV, MK, false); } while (0)
    if (ws == 12) { if (masked) FWD_LAUNCH_C(12, true); else FWD_LAUNCH_C(12, false); }
    else if (ws == 7) { if (masked) FWD_LAUNCH_C(7, true); else FWD_LAUNCH_C(7, false); }
#undef FWD_LAUNCH_C
#undef FWD_LAUNCH
    else
        return DGX_ERR_UNSUPPORTED;
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_window_attention_fwd(const void* qkv, const float* table, int64_t table_stride_head,
                                        int64_t table_stride_index, const int8_t* region, void* out, float* lse, int B_,
                                        int nW, int nH, int ws, float scale, void* stream) {
    return attn_fwd_launch(qkv, table, table_stride_head, table_stride_index, region, out, lse, B_, nW, nH, ws, scale, nullptr, stream);
}

extern "C" int dgx_window_attention_fwd_compact(const void* qkv, const void* qkv_bias, const float* table, int64_t table_stride_head,
                                                int64_t table_stride_index, const int8_t* region, void* out, float* lse, int B, int H, int W,
                                                int nH, int ws, int shift, float scale, void* stream) {
    if (ws != 12 && ws != 7) return DGX_ERR_UNSUPPORTED;
    AttnGeom G = {0, 0, 0, 0, 0, nullptr, nullptr};
    int nW = 0;
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
    const int B_ = B * nWh * nWw;
    if (int rc = attn_geom(G, B_, nW, ws, B, H, W, shift, qkv_bias, nH)) return rc;
    return attn_fwd_launch(qkv, table, table_stride_head, table_stride_index, region, out, lse, B_, nW, nH, ws, scale, &G, stream);
}

static int attn_bwd_launch(const void* qkv, const float* table, const int8_t* region, const void* out,
                           const float* lse, const void* dout, void* dqkv, float* dtable,
                           int64_t dtable_stride_head, int64_t dtable_stride_index, int B_, int nW, int nH,
                           int ws, float scale, const AttnGeom* geom, void* stream) {
    if (B_ <= 0) return DGX_OK;
    if (!qkv || !table || !out || !lse || !dout || !dqkv || !dtable || nH <= 0 || nW <= 0 || (region && B_ % nW))
        return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool masked = region != nullptr;
    if (!region) { region = zero_region(); if (!geom) nW = 1; if (!region) return DGX_ERR_BAD_ARG; }
    AttnGeom G = geom ? *geom : AttnGeom{0, 0, 0, 0, 0, nullptr, nullptr};
    if (geom) { G.zeros = (const uint16_t*)zero_region(); if (!G.zeros) return DGX_ERR_BAD_ARG; }
    // One workgroup per CU (LDS-limited) in a single round: per-workgroup setup (bias row, flush of the bias-gradient row) is amortised
    // over the run (measured best among 256/512/768/2048 workgroup targets), runs of equal length over the CUs the persistent kernels'
    // reservation leaves (engine/ddp.py: RCCL's channels at N > 1); the kernel's header says how the runs are laid over windows and heads
    extern int dgx_get_reserved_cus(void);
    const int cus = 256 - dgx_get_reserved_cus();
    const int64_t total = (int64_t)B_ * nH;
    if (total > 0x7fffffff) return DGX_ERR_UNSUPPORTED;
    int upw = (int)((total + cus - 1) / cus);
    if (upw > B_) upw = B_;                                            // more heads than CUs: one run per head
    const int left = B_ % upw, main_wgs = (B_ / upw) * nH;
    int upl = upw;                                                     // length of the runs over the windows left per head: what fills
    if (left > 0 && cus > main_wgs) {                                  // the CUs the full runs leave (see the kernel)
        upl = (nH * left + (cus - main_wgs) - 1) / (cus - main_wgs);
        if (upl > upw) upl = upw;
    }
    const int grid = main_wgs + (nH * left + upl - 1) / upl;
    // S, dP, dV, dK, dQ (5 x 2 N^2 32 FLOP per head); q, k, v, out, dout in + dq, dk, dv out (bf16) + lse
    const double heads = (double)B_ * nH, ntok = (double)ws * ws;
    DgxProfScope prof(DGX_PROF_ATTN_BWD, stream, heads * 5.0 * 2.0 * ntok * ntok * 32.0, heads * ntok * (32.0 * 2.0 * 8.0 + 4.0));
    const int64_t tblp = ((2 * ws - 1) * (2 * ws - 1) + 3) / 4 * 4;
    BwdTableWs* tw = bwd_table_ws(st, (int64_t)nH * (B_ / upw + (left > 0 ? (left + upl - 1) / upl + 1 : 0)) * tblp, nH);
    if (!tw) return DGX_ERR_UNSUPPORTED;
#define BWD_ARGS (const uint16_t*)qkv, table, region, (const uint16_t*)out, lse, (const uint16_t*)dout, (uint16_t*)dqkv, dtable, B_, nW, nH, \
                 scale, upw, upl, tw->part, tw->cnt, dtable_stride_head, dtable_stride_index, G
    if (ws == 12) {
        static bool once = false;
        const size_t sm = bwd_smem_bytes<12>(true);
        if (!once) {
            (void)hipFuncSetAttribute((const void*)win_attn_bwd_kernel<12, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            (void)hipFuncSetAttribute((const void*)win_attn_bwd_kernel<12, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            (void)hipFuncSetAttribute((const void*)win_attn_bwd_kernel<12, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            (void)hipFuncSetAttribute((const void*)win_attn_bwd_kernel<12, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            once = true;
        }
#define BWD12(MK, CP) hipLaunchKernelGGL((win_attn_bwd_kernel<12, MK, true, CP>), dim3(grid), dim3((WinCfg<12>::NT + 3) * 64), sm, st, BWD_ARGS)
        if (masked) { if (geom) BWD12(true, true); else BWD12(true, false); }
        else { if (geom) BWD12(false, true); else BWD12(false, false); }
#undef BWD12
    } else if (ws == 7) {
        const size_t sm = bwd_smem_bytes<7>();
#define BWD7(MK, CP) hipLaunchKernelGGL((win_attn_bwd_kernel<7, MK, false, CP>), dim3(grid), dim3(WinCfg<7>::NT * 64), sm, st, BWD_ARGS)
        if (masked) { if (geom) BWD7(true, true); else BWD7(true, false); }
        else { if (geom) BWD7(false, true); else BWD7(false, false); }
#undef BWD7
    } else {
        return DGX_ERR_UNSUPPORTED;
    }
#undef BWD_ARGS
    if (hipError_t e = hipGetLastError(); e != hipSuccess) {
        tw->dirty = true;                // a refused launch: whatever state the counters are in, the next call starts from zero
        return -(int)e - 1000;
    }
    return DGX_OK;
}

extern "C" int dgx_window_attention_bwd(const void* qkv, const float* table, const int8_t* region, const void* out,
                                        const float* lse, const void* dout, void* dqkv, float* dtable,
                                        int64_t dtable_stride_head, int64_t dtable_stride_index, int B_, int nW, int nH,
                                        int ws, float scale, void* stream) {
    return attn_bwd_launch(qkv, table, region, out, lse, dout, dqkv, dtable, dtable_stride_head, dtable_stride_index, B_, nW, nH, ws, scale,
                           nullptr, stream);
}

extern "C" int dgx_window_attention_bwd_compact(const void* qkv, const void* qkv_bias, const float* table, const int8_t* region, const void* out,
                                                const float* lse, const void* dout, void* dqkv, float* dtable, int64_t dtable_stride_head,
                                                int64_t dtable_stride_index, int B, int H, int W, int nH, int ws, int shift, float scale,
                                                void* stream) {
    if (ws != 12 && ws != 7) return DGX_ERR_UNSUPPORTED;
    AttnGeom G = {0, 0, 0, 0, 0, nullptr, nullptr};
    int nW = 0;
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
    const int B_ = B * nWh * nWw;
    if (int rc = attn_geom(G, B_, nW, ws, B, H, W, shift, qkv_bias, nH)) return rc;
    return attn_bwd_launch(qkv, table, region, out, lse, dout, dqkv, dtable, dtable_stride_head, dtable_stride_index, B_, nW, nH, ws, scale,
                           &G, stream);
}
