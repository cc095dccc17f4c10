// Fused LayerNorm for the Swin blocks (gfx950).  HBM-bound: one wave per token row, 16-byte lanes.
//   forward : x fp32 (T,C) -> y bf16, optionally written straight in WINDOW order (zero rows for the
//             padding tokens): LayerNorm + bf16 cast + pad + roll + window_partition in one pass
//             (reads 4 B, writes 2 B per element) instead of LN (4+4) + cast (4+2) + gather (2+2).
//   backward: dy bf16 (window order when gathered) + x + (mean, rstd) -> dx fp32, and per-block partial
//             sums of dgamma / dbeta (a second tiny pass adds them into the gradient arena).
// Reference semantics: nn.LayerNorm at swintransformer.py:213,255 followed by :216-233.
#include "dgx_common.h"
#include "winmap.h"

// ws == 0: identity (no gather).  compact: the window order without the padding tokens (winmap.h): rows 0 .. B*H*W-1 are the real
// tokens, whatever rows a buffer has beyond them are padding rows (source -1).  The entry points take it as ws < 0.
struct WinMap { int B, H, W, ws, shift, nWh, nWw, compact; };

// output row index (window order) -> source token index, or -1 for a padding token
__device__ __forceinline__ int64_t win_src(const WinMap& m, int64_t orow) {
    if (m.compact) {
        if (orow >= (int64_t)m.B * m.H * m.W) return -1;
        int b;
        return wm_token_of_row(wm_geom(m.H, m.W, m.ws, m.shift), (int)orow, b);
    }
    const int N = m.ws * m.ws;
    const int n = (int)(orow % N);
    int64_t t = orow / N;
    const int wc = (int)(t % m.nWw);
    t /= m.nWw;
    const int wr = (int)(t % m.nWh);
    const int b = (int)(t / m.nWh);
    int hh = wr * m.ws + n / m.ws + m.shift, ww = wc * m.ws + n % m.ws + m.shift;
    const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
    if (hh >= Hp) hh -= Hp;
    if (ww >= Wp) ww -= Wp;
    if (hh >= m.H || ww >= m.W) return -1;
    return ((int64_t)b * m.H + hh) * m.W + ww;
}

// the same map in 32-bit arithmetic (every row / token count on the path is far below 2^31; the launchers check it): the 64-bit
// divisions of win_src / win_dst are software sequences of ~150 VALU instructions each -- a large share of a kernel that handles two
// to four rows per wave
__device__ __forceinline__ int win_src32(const WinMap& m, int orow) {
    if (m.compact) {
        if (orow >= m.B * m.H * m.W) return -1;
        int b;
        return wm_token_of_row(wm_geom(m.H, m.W, m.ws, m.shift), orow, b);
    }
    const int N = m.ws * m.ws;
    const int t0 = orow / N, n = orow - t0 * N;
    const int t1 = t0 / m.nWw, wc = t0 - t1 * m.nWw;
    const int b = t1 / m.nWh, wr = t1 - b * m.nWh;
    const int nr = n / m.ws;
    int hh = wr * m.ws + nr + m.shift, ww = wc * m.ws + (n - nr * m.ws) + m.shift;
    const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
    if (hh >= Hp) hh -= Hp;
    if (ww >= Wp) ww -= Wp;
    if (hh >= m.H || ww >= m.W) return -1;
    return (b * m.H + hh) * m.W + ww;
}
__device__ __forceinline__ int win_dst32(const WinMap& m, int tok) {
    const int t = tok / m.W, ww0 = tok - t * m.W;
    const int b = t / m.H, hh0 = t - b * m.H;
    if (m.compact) return wm_row_of_token(wm_geom(m.H, m.W, m.ws, m.shift), b, hh0, ww0);
    const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
    int hs = hh0 - m.shift, wsx = ww0 - m.shift;
    if (hs < 0) hs += Hp;
    if (wsx < 0) wsx += Wp;
    const int wr = hs / m.ws, wc = wsx / m.ws;
    const int n = (hs - wr * m.ws) * m.ws + (wsx - wc * m.ws);
    return ((b * m.nWh + wr) * m.nWw + wc) * (m.ws * m.ws) + n;
}

// source token -> its row in window order
__device__ __forceinline__ int64_t win_dst(const WinMap& m, int64_t tok) {
    const int ww0 = (int)(tok % m.W);
    int64_t t = tok / m.W;
    const int hh0 = (int)(t % m.H);
    const int b = (int)(t / m.H);
    if (m.compact) return wm_row_of_token(wm_geom(m.H, m.W, m.ws, m.shift), b, hh0, ww0);
    const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
    int hs = hh0 - m.shift, wsx = ww0 - m.shift;
    if (hs < 0) hs += Hp;
    if (wsx < 0) wsx += Wp;
    const int wr = hs / m.ws, wc = wsx / m.ws;
    const int n = (hs - wr * m.ws) * m.ws + (wsx - wc * m.ws);
    return (((int64_t)b * m.nWh + wr) * m.nWw + wc) * (m.ws * m.ws) + n;
}

// 4 consecutive elements of a row as fp32, for fp32 or bf16 storage
template <typename T> __device__ __forceinline__ float4 ld4(const T* row, int i);
template <> __device__ __forceinline__ float4 ld4<float>(const float* row, int i) { return reinterpret_cast<const float4*>(row)[i]; }
template <> __device__ __forceinline__ float4 ld4<uint16_t>(const uint16_t* row, int i) {
    const uint2 d = reinterpret_cast<const uint2*>(row)[i];
    return make_float4(__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16),
                       __uint_as_float(d.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* row, int i, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* row, int i, float4 v) { reinterpret_cast<float4*>(row)[i] = v; }
template <> __device__ __forceinline__ void st4<uint16_t>(uint16_t* row, int i, float4 v) {
    reinterpret_cast<uint2*>(row)[i] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
}

// a row quad as loaded (kept packed in registers: 2 registers for bf16, 4 for fp32) and its fp32 expansion
template <typename T> struct Raw4;
template <> struct Raw4<float> {
    typedef float4 type;
    static __device__ __forceinline__ float4 ld(const float* row, int i) { return reinterpret_cast<const float4*>(row)[i]; }
    static __device__ __forceinline__ float4 f(float4 r) { return r; }
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct Raw4<uint16_t> {
    typedef uint2 type;
    static __device__ __forceinline__ uint2 ld(const uint16_t* row, int i) { return reinterpret_cast<const uint2*>(row)[i]; }
    static __device__ __forceinline__ float4 f(uint2 d) {
        return make_float4(__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16), __uint_as_float(d.y & 0xffff0000u));
    }
    static __device__ __forceinline__ uint2 zero() { return make_uint2(0u, 0u); }
};

// v as it reads back after a store in type T
template <typename T> __device__ __forceinline__ float4 rnd4(float4 v);
template <> __device__ __forceinline__ float4 rnd4<float>(float4 v) { return v; }
template <> __device__ __forceinline__ float4 rnd4<uint16_t>(float4 v) {
    const uint32_t a = pack_bf2(v.x, v.y), b = pack_bf2(v.z, v.w);
    return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// NJ = ceil(C/256) float4 columns per lane: the row is read ONCE into registers (mean, centred variance and the output all
// come from them); NJ == 0 is the generic three-pass form for rows wider than 1536.
template <typename XT, typename YT = uint16_t, int NJ = 0>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, YT* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t T_out,
                                                     int C, float eps, WinMap m) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    if constexpr (NJ > 0) {
        // Two rows per trip, EVERY load of the trip in flight before anything is consumed: the rows as raw bits (no conversion, no
        // select next to the load: round 5 found each `i < C/4 ? ld4(...) : 0` compiled to load + s_waitcnt vmcnt(0) in its own
        // predicated block -- six memory latencies in series per trip on a kernel that should be ONE latency long), gamma and beta
        // with them; out-of-row lanes read the row's last quad (clamped index) and are masked where the value is used.
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int wave32 = (int)blockIdx.x * 4 + wv, nwaves32 = (int)gridDim.x * 4, T32 = (int)T_out, nq = C / 4;
        constexpr int NJ1 = NJ > 0 ? NJ : 1;
        float4 gv[NJ1], bv[NJ1];
        bool in[NJ1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = lane + 64 * j;
            in[j] = i < nq;
            gv[j] = g4[in[j] ? i : nq - 1];
            bv[j] = b4[in[j] ? i : nq - 1];
        }
        for (int orow0 = wave32; orow0 < T32; orow0 += 2 * nwaves32) {
            typename Raw4<XT>::type raw[2][NJ1];
            int toks[2];
            const int orows[2] = {orow0, orow0 + nwaves32};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                toks[u] = orows[u] < T32 ? (m.ws ? win_src32(m, orows[u]) : orows[u]) : -2;      // -1: padding row, -2: no row
                toks[u] = __builtin_amdgcn_readfirstlane(toks[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) raw[u][j] = Raw4<XT>::zero();
                if (toks[u] >= 0) {
                    const XT* xr = x + (int64_t)toks[u] * C;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) raw[u][j] = Raw4<XT>::ld(xr, in[j] ? lane + 64 * j : nq - 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // the loads stay together, in front of every use
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (toks[u] == -2) continue;
                YT* yo = y + (int64_t)orows[u] * C;
                if (toks[u] < 0) {
                    for (int i = lane; i < nq; i += 64) st4<YT>(yo, i, make_float4(0.f, 0.f, 0.f, 0.f));
                    continue;
                }
                float4 v[NJ1];
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    v[j] = Raw4<XT>::f(raw[u][j]);
                    if (!in[j]) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
                }
                const float mu = wave_sum(s) / (float)C;
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (in[j]) {
                        const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
                        q += (a * a + b * b) + (c * c + d * d);
                    }
                const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
                if (lane == 0) { mean[toks[u]] = mu; rstd[toks[u]] = rs; }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int i = lane + 64 * j;
                    if (in[j]) {
                        const float4 g = gv[j], b = bv[j];
                        st4<YT>(yo, i, make_float4((v[j].x - mu) * rs * g.x + b.x, (v[j].y - mu) * rs * g.y + b.y,
                                                   (v[j].z - mu) * rs * g.z + b.z, (v[j].w - mu) * rs * g.w + b.w));
                    }
                }
            }
        }
        return;
    }
    for (int64_t orow = wave; orow < T_out; orow += nwaves) {
        const int64_t tok = m.ws ? win_src(m, orow) : orow;
        YT* yo = y + orow * C;
        if (tok < 0) {
            for (int i = lane; i < C / 4; i += 64) st4<YT>(yo, i, make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        const XT* xr = x + tok * C;
        float s = 0.f;
        for (int i = lane; i < C / 4; i += 64) { const float4 v = ld4<XT>(xr, i); s += (v.x + v.y) + (v.z + v.w); }
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
        for (int i = lane; i < C / 4; i += 64) {
            const float4 v = ld4<XT>(xr, i);
            const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
            q += (a * a + b * b) + (c * c + d * d);
        }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) { mean[tok] = mu; rstd[tok] = rs; }
        for (int i = lane; i < C / 4; i += 64) {
            const float4 v = ld4<XT>(xr, i), g = g4[i], b = b4[i];
            st4<YT>(yo, i, make_float4((v.x - mu) * rs * g.x + b.x, (v.y - mu) * rs * g.y + b.y,
                                       (v.z - mu) * rs * g.z + b.z, (v.w - mu) * rs * g.w + b.w));
        }
    }
}

// NJ = ceil(C/256): float4 columns per lane
constexpr int LNB_WAVES = 8;          // waves per backward workgroup: one partial row of (dgamma | dbeta) per workgroup
// EMIT (dgx_layernorm_bwd_emit): the kernel also writes  emit[row(tok)] = bf16(escale[b] * dx[tok])  -- the operand of the NEXT
// GEMM of the backward pass -- in token order (em.ws == 0) or in window order (zero rows for the padding tokens), i.e. what a
// separate dgx_residual_bwd pass over dx produced before (one read of dx and one launch less per use).  The value is formed from
// dx AS STORED (rounded to its dtype first), so the two-kernel path gives the same bits.
template <int NJ, typename XT, typename DT = uint16_t, bool EMIT = false>
__global__ __launch_bounds__(64 * LNB_WAVES, ((NJ <= 2 || (NJ == 3 && sizeof(XT) == 2)) ? 4 : 2)) void ln_bwd_kernel(const DT* __restrict__ dy, const XT* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const XT* dres, XT* dx,
                                                     float* __restrict__ part, int64_t T, int C, WinMap m,
                                                     uint16_t* __restrict__ emit = nullptr, const float* __restrict__ escale = nullptr,
                                                     WinMap em = WinMap{0, 0, 0, 0, 0, 0, 0, 0}) {
    __shared__ float red[2][NJ * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * LNB_WAVES + w;
    const int64_t nwaves = (int64_t)gridDim.x * LNB_WAVES;
    float dg[NJ][4], db[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) dg[j][k] = db[j][k] = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    // two rows per trip: both rows' loads are in flight before the first row's reductions (T / waves = 2 rows per wave at
    // Swin-L stage 2: one memory latency per launch instead of two on a 16 us kernel)
    // Row / token indices are wave-uniform and far below 2^31: 32-bit scalar arithmetic (the 64-bit divisions of the window map were
    // ~150 VALU instructions per row and map).
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave32 = (int)blockIdx.x * LNB_WAVES + wv, nwaves32 = (int)gridDim.x * LNB_WAVES, T32 = (int)T, nq = C / 4;
    bool in[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) in[j] = lane + 64 * j < nq;
    for (int tok0 = wave32; tok0 < T32; tok0 += 2 * nwaves32) {
        // The two rows stay in registers AS LOADED (packed: 6 registers per bf16 quad for x, dy and the residual-branch gradient).
        // EVERY load of the trip -- both rows' statistics, their 3 x NJ quads each, gamma -- is issued before anything is consumed
        // (round 5: the compiler had waited for row 0 before requesting row 1 and re-requested gamma in front of each use: ~8 exposed
        // L2 / HBM latencies per trip on a kernel that is one trip long).  x_hat and dy * gamma are formed twice, for the sums and for
        // the result, by the same operations.
        typename Raw4<XT>::type xr_[2][NJ], rr_[2][NJ];
        typename Raw4<DT>::type dr_[2][NJ];
        float4 gq[NJ];
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rsv[2] = {0.f, 0.f}, muv[2] = {0.f, 0.f};
        const int toks[2] = {tok0, tok0 + nwaves32};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) { xr_[u][j] = Raw4<XT>::zero(); dr_[u][j] = Raw4<DT>::zero(); rr_[u][j] = Raw4<XT>::zero(); }
            const int tok = toks[u];
            if (tok >= T32) continue;
            const int drow = __builtin_amdgcn_readfirstlane(m.ws ? win_dst32(m, tok) : tok);
            const DT* dyr = dy + (int64_t)drow * C;
            const XT* xr = x + (int64_t)tok * C;
            const XT* drr0 = dres ? dres + (int64_t)tok * C : nullptr;
            muv[u] = mean[tok];
            rsv[u] = rstd[tok];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int i = in[j] ? lane + 64 * j : nq - 1;        // out-of-row lanes re-read the last quad and are masked at the uses
                xr_[u][j] = Raw4<XT>::ld(xr, i);
                dr_[u][j] = Raw4<DT>::ld(dyr, i);
                if (drr0) rr_[u][j] = Raw4<XT>::ld(drr0, i);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // (gamma behind the rows: an L2 hit, it lands first anyway)
#pragma unroll
        for (int j = 0; j < NJ; ++j) gq[j] = g4[in[j] ? lane + 64 * j : nq - 1];
        __builtin_amdgcn_sched_barrier(0);      // the loads stay together, in front of every use
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (toks[u] >= T32) continue;
            const float mu = muv[u], rs = rsv[u];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (in[j]) {
                    const float4 v = Raw4<XT>::f(xr_[u][j]), g = gq[j], d4 = Raw4<DT>::f(dr_[u][j]);
                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
                    const float xv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xh = (xv[k] - mu) * rs, gv = dv[k] * gg[k];
                        s1[u] += gv;
                        s2[u] += gv * xh;
                        dg[j][k] += dv[k] * xh;
                        db[j][k] += dv[k];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tok = toks[u];
            if (tok >= T32) continue;
            const float a1 = wave_sum(s1[u]) / (float)C, a2 = wave_sum(s2[u]) / (float)C, rs = rsv[u], mu = muv[u];
            XT* dxr = dx + (int64_t)tok * C;
            uint16_t* er = nullptr;
            float es = 1.0f;
            if (EMIT) {
                er = emit + (int64_t)__builtin_amdgcn_readfirstlane(em.ws ? win_dst32(em, tok) : tok) * C;
                if (escale) es = escale[tok / (em.H * em.W)];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int i = lane + 64 * j;
                if (in[j]) {
                    const float4 v = Raw4<XT>::f(xr_[u][j]), g = gq[j], d4 = Raw4<DT>::f(dr_[u][j]);
                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
                    const float xv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
                    float ov[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xh = (xv[k] - mu) * rs, gv = dv[k] * gg[k];
                        ov[k] = rs * (gv - a1 - xh * a2);
                    }
                    float4 o = make_float4(ov[0], ov[1], ov[2], ov[3]);
                    if (dres) {                // gradient arriving on the residual branch (may alias dx: read above, before this store)
                        const float4 a = Raw4<XT>::f(rr_[u][j]);
                        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                    }
                    st4<XT>(dxr, i, o);
                    if (EMIT) {
                        const float4 r = rnd4<XT>(o);     // dx as stored
                        st4<uint16_t>(er, i, make_float4(es * r.x, es * r.y, es * r.z, es * r.w));
                    }
                }
            }
        }
    }
    if (EMIT && em.ws && !em.compact) {        // classic window order: the rows of the padding tokens are zero (the compact emit buffer has none)
        const int64_t rows = (int64_t)em.B * em.nWh * em.nWw * em.ws * em.ws;
        if (rows != T)
            for (int64_t orow = wave; orow < rows; orow += nwaves)
                if (win_src(em, orow) < 0)
                    for (int i = lane; i < C / 4; i += 64) st4<uint16_t>(emit + orow * C, i, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    // block partials: the 8 waves fold into one LDS row in turn -> one row of `part` per block: [block][2][C]
    // (tried: every wave into its own LDS row + one barrier -- 49 KB of LDS per workgroup cost more occupancy than the seven barriers:
    // 16.1 -> 22.3 us)
    for (int turn = 0; turn < LNB_WAVES; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = (lane + 64 * j) * 4 + k;
                    red[0][c] = turn ? red[0][c] + dg[j][k] : dg[j][k];
                    red[1][c] = turn ? red[1][c] + db[j][k] : db[j][k];
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 64 * LNB_WAVES) {
        part[((int64_t)blockIdx.x * 2 + 0) * C + c] = red[0][c];
        part[((int64_t)blockIdx.x * 2 + 1) * C + c] = red[1][c];
    }
}

// dgamma += sum_blocks part[b][0], dbeta += sum_blocks part[b][1].
// block = LNR_QB float4 column-quads x LNR_RG row groups (1024 threads): 256 B contiguous per row read.  Round 2 used 64 quads x 16
// row groups: 6 workgroups for a C = 768 norm -- the whole second stage ran on 6 (12 for two norms) of the 256 CUs, 10.7 us for
// 6 MB; 16 quads x 64 row groups are 24 (48) workgroups.  Summation order (both kernels, so the one- and two-norm forms stay
// bit-identical): row group rg takes rows rg, rg + 64, ...; eight threads fold eight groups each, the first folds those eight.
constexpr int LNR_QB = 16, LNR_RG = 64;
__device__ __forceinline__ void ln_reduce_rows(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, int nblk,
                                               int C, int qblock, float4 (*red)[LNR_QB]) {
    const int lane = threadIdx.x % LNR_QB, rg = threadIdx.x / LNR_QB;
    const int q = qblock * LNR_QB + lane;          // float4 index inside the 2*C-wide partial row
    const int nq = 2 * C / 4;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (q < nq) {
        const float4* p4 = reinterpret_cast<const float4*>(part);
#pragma unroll 4
        for (int b = rg; b < nblk; b += LNR_RG) {
            const float4 v = p4[(int64_t)b * nq + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg < 8) {
        float4 a = red[8 * rg][lane];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = red[8 * rg + r][lane]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        s = a;
    }
    __syncthreads();
    if (rg < 8) red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && q < nq) {
        float4 a = red[0][lane];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = red[r][lane]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        // the partial row is [dgamma(C) | dbeta(C)]; C % 4 == 0 so a quad never straddles the two
        float* dst = (4 * q < C) ? dgamma + 4 * q : dbeta + (4 * q - C);
        dst[0] += a.x; dst[1] += a.y; dst[2] += a.z; dst[3] += a.w;
    }
}
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int nblk, int C) {
    __shared__ float4 red[LNR_RG][LNR_QB];
    ln_reduce_rows(part, dgamma, dbeta, nblk, C, blockIdx.x, red);
}

static WinMap make_map(int B, int H, int W, int ws, int shift) {      // ws < 0: compact window order
    WinMap m = {B, H, W, ws < 0 ? -ws : ws, shift, 0, 0, ws < 0 ? 1 : 0};
    if (m.ws > 0) { m.nWh = (H + m.ws - 1) / m.ws; m.nWw = (W + m.ws - 1) / m.ws; }
    return m;
}
static bool map_args_ok(int B, int H, int W, int ws, int shift, int64_t T) {
    if (ws == 0) return true;
    const int a = ws < 0 ? -ws : ws;
    if ((int64_t)B * H * W != T || shift < 0 || shift >= a) return false;
    return ws > 0 || wm_compact_ok(H, W, a, shift);
}

extern "C" int dgx_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                 float* rstd, int64_t T, int C, float eps, int B, int H, int W, int ws, int shift,
                                 int x_dtype, void* stream) {
    if (T <= 0) return DGX_OK;
    if (!x || !gamma || !beta || !y_bf16 || !mean || !rstd || (C & 3) || !map_args_ok(B, H, W, ws, shift, T))
        return DGX_ERR_BAD_ARG;
    const WinMap m = make_map(B, H, W, ws, shift);
    // (compact as well: the output keeps the padding rows, all zero, behind the real ones -- the qkv weight gradient's operand)
    const int64_t T_out = m.ws > 0 ? (int64_t)B * m.nWh * m.nWw * m.ws * m.ws : T;
    if (T_out + 16384 >= ((int64_t)1 << 31)) return DGX_ERR_UNSUPPORTED;      // 32-bit row arithmetic in the kernels
    const int nj = (C + 255) / 256;
    const int64_t per_wg = nj <= 6 ? 8 : 4;       // rows per workgroup and trip: 4 waves x 2 rows (register-resident rows)
    const int grid = (int)((T_out + per_wg - 1) / per_wg < 8192 ? (T_out + per_wg - 1) / per_wg : 8192);
#define LN_FWD(XT, YT, YP, NJ) hipLaunchKernelGGL((ln_fwd_kernel<XT, YT, NJ>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const XT*)x, \
                                                  gamma, beta, (YT*)YP, mean, rstd, T_out, C, eps, m)
#define LN_FWD_NJ(XT, YT, YP)                                                                                          \
    do {                                                                                                               \
        if (nj <= 1) LN_FWD(XT, YT, YP, 1);                                                                            \
        else if (nj <= 2) LN_FWD(XT, YT, YP, 2);                                                                       \
        else if (nj <= 3) LN_FWD(XT, YT, YP, 3);                                                                       \
        else if (nj <= 6) LN_FWD(XT, YT, YP, 6);                                                                       \
        else LN_FWD(XT, YT, YP, 0);                                                                                    \
    } while (0)
    if (x_dtype == DGX_BF16) LN_FWD_NJ(uint16_t, uint16_t, y_bf16);
    else LN_FWD_NJ(float, uint16_t, y_bf16);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_layernorm_bwd_blocks(int64_t T) {
    int64_t b = (T + LNB_WAVES - 1) / LNB_WAVES;
    return (int)(b < 512 ? (b < 1 ? 1 : b) : 512);
}

static int ln_bwd_launch(const void* dy_bf16, const void* x, const float* mean, const float* rstd, const float* gamma, const void* dres,
                         void* dx, float* dgamma, float* dbeta, float* part, int64_t T, int C, int B, int H, int W, int ws, int shift,
                         int x_dtype, void* emit, const float* escale, int eB, int eH, int eW, int ews, int eshift, void* stream) {
    if (T <= 0) return DGX_OK;
    if (!dy_bf16 || !x || !mean || !rstd || !gamma || !dx || ((dgamma == nullptr) != (dbeta == nullptr)) || !part || (C & 3) || C > 1536 ||
        !map_args_ok(B, H, W, ws, shift, T))
        return DGX_ERR_BAD_ARG;
    if (emit && ((int64_t)eB * eH * eW != T || !map_args_ok(eB, eH, eW, ews, eshift, T))) return DGX_ERR_BAD_ARG;
    if (T + 16384 >= ((int64_t)1 << 31)) return DGX_ERR_UNSUPPORTED;          // 32-bit row / token arithmetic in the kernels
    const WinMap m = make_map(B, H, W, ws, shift);
    const WinMap em = emit ? make_map(eB, eH, eW, ews, eshift) : WinMap{0, 0, 0, 0, 0, 0, 0, 0};
    const int grid = dgx_layernorm_bwd_blocks(T);
    hipStream_t st = (hipStream_t)stream;
    const int nj = (C + 255) / 256;
#define LN_BWD_E(NJ, E)                                                                                                                \
    do {                                                                                                                               \
        if (x_dtype == DGX_BF16)                                                                                                       \
            hipLaunchKernelGGL((ln_bwd_kernel<NJ, uint16_t, uint16_t, E>), dim3(grid), dim3(64 * LNB_WAVES), 0, st, (const uint16_t*)dy_bf16, \
                               (const uint16_t*)x, mean, rstd, gamma, (const uint16_t*)dres, (uint16_t*)dx, part, T, C, m, (uint16_t*)emit,   \
                               escale, em);                                                                                            \
        else                                                                                                                           \
            hipLaunchKernelGGL((ln_bwd_kernel<NJ, float, uint16_t, E>), dim3(grid), dim3(64 * LNB_WAVES), 0, st, (const uint16_t*)dy_bf16,    \
                               (const float*)x, mean, rstd, gamma, (const float*)dres, (float*)dx, part, T, C, m, (uint16_t*)emit, escale,    \
                               em);                                                                                                    \
    } while (0)
#define LN_BWD(NJ) do { if (emit) LN_BWD_E(NJ, true); else LN_BWD_E(NJ, false); } while (0)
    if (nj <= 1) LN_BWD(1);
    else if (nj <= 2) LN_BWD(2);
    else if (nj <= 3) LN_BWD(3);
    else LN_BWD(6);
#undef LN_BWD
#undef LN_BWD_E
    // dgamma == dbeta == NULL: the per-block partial rows stay in `part` for dgx_layernorm_param_reduce2 (two norms, one launch)
    if (dgamma) hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C / 4 + LNR_QB - 1) / LNR_QB), dim3(1024), 0, st, part, dgamma, dbeta, grid, C);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_layernorm_bwd(const void* dy_bf16, const void* x, const float* mean, const float* rstd,
                                 const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta, float* part,
                                 int64_t T, int C, int B, int H, int W, int ws, int shift, int x_dtype, void* stream) {
    return ln_bwd_launch(dy_bf16, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, part, T, C, B, H, W, ws, shift, x_dtype, nullptr, nullptr,
                         0, 0, 0, 0, 0, stream);
}

extern "C" int dgx_layernorm_bwd_emit(const void* dy_bf16, const void* x, const float* mean, const float* rstd, const float* gamma,
                                      const void* dres, void* dx, float* dgamma, float* dbeta, float* part, int64_t T, int C, int B,
                                      int H, int W, int ws, int shift, int x_dtype, void* emit_bf16, const float* emit_scale, int eB,
                                      int eH, int eW, int ews, int eshift, void* stream) {
    if (!emit_bf16) return DGX_ERR_BAD_ARG;
    return ln_bwd_launch(dy_bf16, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, part, T, C, B, H, W, ws, shift, x_dtype, emit_bf16,
                         emit_scale, eB, eH, eW, ews, eshift, stream);
}

struct LnRed2 { const float* part[2]; float* dgamma[2]; float* dbeta[2]; };
__global__ __launch_bounds__(1024) void ln_param_reduce2_kernel(LnRed2 R, int nblk, int C) {
    __shared__ float4 red[LNR_RG][LNR_QB];
    const int which = blockIdx.y;
    ln_reduce_rows(R.part[which], R.dgamma[which], R.dbeta[which], nblk, C, blockIdx.x, red);
}

// The second stage of TWO dgx_layernorm_bwd calls made with dgamma = dbeta = NULL (same T, hence the same number of partial
// rows, and the same C): dgamma_x += sum of part_x's rows, in the fixed order of the single-norm kernel (bit-identical).
extern "C" int dgx_layernorm_param_reduce2(const float* part_a, float* dgamma_a, float* dbeta_a, const float* part_b, float* dgamma_b,
                                           float* dbeta_b, int64_t T, int C, void* stream) {
    if (T <= 0) return DGX_OK;
    if (!part_a || !dgamma_a || !dbeta_a || !part_b || !dgamma_b || !dbeta_b || (C & 3) || C > 1536) return DGX_ERR_BAD_ARG;
    LnRed2 R = {{part_a, part_b}, {dgamma_a, dgamma_b}, {dbeta_a, dbeta_b}};
    hipLaunchKernelGGL(ln_param_reduce2_kernel, dim3((2 * C / 4 + LNR_QB - 1) / LNR_QB, 2), dim3(1024), 0, (hipStream_t)stream, R,
                       dgx_layernorm_bwd_blocks(T), C);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// The second stage of up to 16 norms in ONE launch (the two norms of up to eight consecutive Swin blocks of a stage: same T and C): per
// norm the fixed summation order of the single-norm kernel (bit-identical); entries: part / dgamma / dbeta pointer triples.
struct LnRedN { const float* part[16]; float* dgamma[16]; float* dbeta[16]; };
__global__ __launch_bounds__(1024) void ln_param_reduce_n_kernel(LnRedN R, int nblk, int C) {
    __shared__ float4 red[LNR_RG][LNR_QB];
    const float* part = R.part[0];
    float* dg = R.dgamma[0];
    float* db = R.dbeta[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if ((int)blockIdx.y == k) { part = R.part[k]; dg = R.dgamma[k]; db = R.dbeta[k]; }      // constant indices into the by-value table
    ln_reduce_rows(part, dg, db, nblk, C, blockIdx.x, red);
}
extern "C" int dgx_layernorm_param_reduce_n(const float* const* parts, float* const* dgammas, float* const* dbetas, int n, int64_t T, int C,
                                            void* stream) {
    if (T <= 0 || n <= 0) return DGX_OK;
    if (!parts || !dgammas || !dbetas || n > 16 || (C & 3) || C > 1536) return DGX_ERR_BAD_ARG;
    LnRedN R;
    for (int k = 0; k < 16; ++k) {
        const int j = k < n ? k : 0;
        if (!parts[j] || !dgammas[j] || !dbetas[j]) return DGX_ERR_BAD_ARG;
        R.part[k] = parts[j]; R.dgamma[k] = dgammas[j]; R.dbeta[k] = dbetas[j];
    }
    hipLaunchKernelGGL(ln_param_reduce_n_kernel, dim3((2 * C / 4 + LNR_QB - 1) / LNR_QB, n), dim3(1024), 0, (hipStream_t)stream, R,
                       dgx_layernorm_bwd_blocks(T), C);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// ------------------------------------------------------------------------------------------------
// PatchMerging front half (swintransformer.py:272-298): zero-pad to even H/W, gather the 2x2 neighbourhood
// [x(2i,2j) | x(2i+1,2j) | x(2i,2j+1) | x(2i+1,2j+1)] into one 4*C0-wide row and LayerNorm it -- in ONE pass over x,
// bf16 out (the operand of the reduction GEMM).  The reference materialises four strided slices, their concatenation and
// an fp32 LayerNorm; its backward zero-fills and adds four full-size tensors.  Here every element of x is read once in
// the forward and every element of dx written once in the backward.
struct MergeMap { int B, H, W, H2, W2, C0; };

// float4 column i (of 4*C0/4) of merged row (b, i2, j2) -> source float4 pointer index, or -1 in the zero padding
__device__ __forceinline__ int64_t merge_src4(const MergeMap& m, int b, int i2, int j2, int i, int q0) {
    const int seg = i / q0, wi = i - seg * q0;
    const int y = 2 * i2 + (seg & 1), x = 2 * j2 + (seg >> 1);
    if (y >= m.H || x >= m.W) return -1;
    return (((int64_t)b * m.H + y) * m.W + x) * q0 + wi;
}
template <typename T> __device__ __forceinline__ float4 ld4q(const T* base, int64_t q) {
    if (q < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
    return ld4<T>(base + q * 4, 0);
}

template <typename XT>
__global__ __launch_bounds__(256) void pm_ln_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd, float eps,
                                                        MergeMap m) {
    const int lane = threadIdx.x & 63;
    const int C = 4 * m.C0, q0 = m.C0 / 4;
    const int64_t T2 = (int64_t)m.B * m.H2 * m.W2;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < T2; row += nwaves) {
        const int j2 = (int)(row % m.W2), i2 = (int)((row / m.W2) % m.H2), b = (int)(row / ((int64_t)m.W2 * m.H2));
        float s = 0.f;
        for (int i = lane; i < C / 4; i += 64) {
            const float4 v = ld4q<XT>(x, merge_src4(m, b, i2, j2, i, q0));
            s += (v.x + v.y) + (v.z + v.w);
        }
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
        for (int i = lane; i < C / 4; i += 64) {
            const float4 v = ld4q<XT>(x, merge_src4(m, b, i2, j2, i, q0));
            const float a = v.x - mu, bb = v.y - mu, c = v.z - mu, d = v.w - mu;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        const float4* g4 = reinterpret_cast<const float4*>(gamma);
        const float4* b4 = reinterpret_cast<const float4*>(beta);
        uint2* yo = reinterpret_cast<uint2*>(y + row * C);
        for (int i = lane; i < C / 4; i += 64) {
            const float4 v = ld4q<XT>(x, merge_src4(m, b, i2, j2, i, q0)), g = g4[i], be = b4[i];
            yo[i] = make_uint2(pack_bf2((v.x - mu) * rs * g.x + be.x, (v.y - mu) * rs * g.y + be.y),
                               pack_bf2((v.z - mu) * rs * g.z + be.z, (v.w - mu) * rs * g.w + be.w));
        }
    }
}

// NJ = ceil(4*C0/256) float4 columns per lane; same reduction layout as ln_bwd_kernel (part: [block][2][C])
// WAVES: waves per workgroup (4 for the 3072-wide case, whose 200 live registers do not fit an 8-wave workgroup)
template <int NJ, typename XT, int WAVES = LNB_WAVES>
__global__ __launch_bounds__(64 * WAVES) void pm_ln_bwd_kernel(const uint16_t* __restrict__ dy, const XT* __restrict__ x,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, XT* __restrict__ dx,
                                                        float* __restrict__ part, MergeMap m) {
    __shared__ float red[2][NJ * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int C = 4 * m.C0, q0 = m.C0 / 4;
    const int64_t T2 = (int64_t)m.B * m.H2 * m.W2;
    const int64_t wave = (int64_t)blockIdx.x * WAVES + w, nwaves = (int64_t)gridDim.x * WAVES;
    float dg[NJ][4], db[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) dg[j][k] = db[j][k] = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    for (int64_t row = wave; row < T2; row += nwaves) {
        const int j2 = (int)(row % m.W2), i2 = (int)((row / m.W2) % m.H2), b = (int)(row / ((int64_t)m.W2 * m.H2));
        const uint2* dyr = reinterpret_cast<const uint2*>(dy + row * C);
        const float mu = mean[row], rs = rstd[row];
        float xh[NJ][4], gv[NJ][4];
        int64_t src[NJ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = lane + 64 * j;
            src[j] = -1;
            if (i < C / 4) {
                src[j] = merge_src4(m, b, i2, j2, i, q0);
                const float4 v = ld4q<XT>(x, src[j]), g = g4[i];
                const uint2 d = dyr[i];
                const float dv[4] = {__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u),
                                     __uint_as_float(d.y << 16), __uint_as_float(d.y & 0xffff0000u)};
                const float xv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xh[j][k] = (xv[k] - mu) * rs;
                    gv[j][k] = dv[k] * gg[k];
                    s1 += gv[j][k];
                    s2 += gv[j][k] * xh[j][k];
                    dg[j][k] += dv[k] * xh[j][k];
                    db[j][k] += dv[k];
                }
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (src[j] >= 0)
                st4<XT>(dx + src[j] * 4, 0,
                        make_float4(rs * (gv[j][0] - s1 - xh[j][0] * s2), rs * (gv[j][1] - s1 - xh[j][1] * s2),
                                    rs * (gv[j][2] - s1 - xh[j][2] * s2), rs * (gv[j][3] - s1 - xh[j][3] * s2)));
    }
    for (int turn = 0; turn < WAVES; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = (lane + 64 * j) * 4 + k;
                    red[0][c] = turn ? red[0][c] + dg[j][k] : dg[j][k];
                    red[1][c] = turn ? red[1][c] + db[j][k] : db[j][k];
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 64 * WAVES) {
        part[((int64_t)blockIdx.x * 2 + 0) * C + c] = red[0][c];
        part[((int64_t)blockIdx.x * 2 + 1) * C + c] = red[1][c];
    }
}

static bool merge_args_ok(int B, int H, int W, int C0) { return B > 0 && H > 0 && W > 0 && C0 > 0 && (C0 & 3) == 0 && 4 * C0 <= 3072; }

extern "C" int dgx_patch_merge_ln_fwd(const void* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                      float* rstd, int B, int H, int W, int C0, float eps, int x_dtype, void* stream) {
    if (!x || !gamma || !beta || !y_bf16 || !mean || !rstd || !merge_args_ok(B, H, W, C0)) return DGX_ERR_BAD_ARG;
    const MergeMap m = {B, H, W, (H + 1) / 2, (W + 1) / 2, C0};
    const int64_t T2 = (int64_t)B * m.H2 * m.W2;
    const int grid = (int)((T2 + 3) / 4 < 8192 ? (T2 + 3) / 4 : 8192);
    if (x_dtype == DGX_BF16)
        hipLaunchKernelGGL(pm_ln_fwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, gamma,
                           beta, (uint16_t*)y_bf16, mean, rstd, eps, m);
    else
        hipLaunchKernelGGL(pm_ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, gamma, beta,
                           (uint16_t*)y_bf16, mean, rstd, eps, m);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_patch_merge_ln_bwd(const void* dy_bf16, const void* x, const float* mean, const float* rstd,
                                      const float* gamma, void* dx, float* dgamma, float* dbeta, float* part, int B, int H,
                                      int W, int C0, int x_dtype, void* stream) {
    if (!dy_bf16 || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !part || !merge_args_ok(B, H, W, C0))
        return DGX_ERR_BAD_ARG;
    const MergeMap m = {B, H, W, (H + 1) / 2, (W + 1) / 2, C0};
    const int64_t T2 = (int64_t)B * m.H2 * m.W2;
    const int C = 4 * C0, grid = dgx_layernorm_bwd_blocks(T2), nj = (C + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
#define PM_BWD(NJ)                                                                                                          \
    do {                                                                                                                    \
        if (x_dtype == DGX_BF16)                                                                                            \
            hipLaunchKernelGGL((pm_ln_bwd_kernel<NJ, uint16_t>), dim3(grid), dim3(64 * LNB_WAVES), 0, st, (const uint16_t*)dy_bf16,     \
                               (const uint16_t*)x, mean, rstd, gamma, (uint16_t*)dx, part, m);                              \
        else                                                                                                                \
            hipLaunchKernelGGL((pm_ln_bwd_kernel<NJ, float>), dim3(grid), dim3(64 * LNB_WAVES), 0, st, (const uint16_t*)dy_bf16,        \
                               (const float*)x, mean, rstd, gamma, (float*)dx, part, m);                                    \
    } while (0)
    if (nj <= 2) PM_BWD(2);
    else if (nj <= 3) PM_BWD(3);
    else if (nj <= 6) PM_BWD(6);
    else if (x_dtype == DGX_BF16)
        hipLaunchKernelGGL((pm_ln_bwd_kernel<12, uint16_t, 4>), dim3(grid), dim3(256), 0, st, (const uint16_t*)dy_bf16,
                           (const uint16_t*)x, mean, rstd, gamma, (uint16_t*)dx, part, m);
    else
        hipLaunchKernelGGL((pm_ln_bwd_kernel<12, float, 4>), dim3(grid), dim3(256), 0, st, (const uint16_t*)dy_bf16,
                           (const float*)x, mean, rstd, gamma, (float*)dx, part, m);
#undef PM_BWD
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C / 4 + LNR_QB - 1) / LNR_QB), dim3(1024), 0, st, part, dgamma, dbeta, grid, C);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}


// LayerNorm with an fp32 result (PatchEmbed.norm, swintransformer.py:440-442: under autocast the reference's LayerNorm
// returns fp32 and that tensor IS the stage-0 residual stream).  x f32|bf16 (T,C) -> y f32; backward takes dy f32.
extern "C" int dgx_layernorm_f32out_fwd(const void* x, const float* gamma, const float* beta, float* y, float* mean,
                                        float* rstd, int64_t T, int C, float eps, int x_dtype, void* stream) {
    if (T <= 0) return DGX_OK;
    if (!x || !gamma || !beta || !y || !mean || !rstd || (C & 3)) return DGX_ERR_BAD_ARG;
    const WinMap m = make_map(0, 0, 0, 0, 0);
    const int grid = (int)((T + 3) / 4 < 8192 ? (T + 3) / 4 : 8192);
    const int nj = (C + 255) / 256;
    const int64_t T_out = T;
    if (x_dtype == DGX_BF16) LN_FWD_NJ(uint16_t, float, y);
    else LN_FWD_NJ(float, float, y);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_layernorm_f32out_bwd(const float* dy, const void* x, const float* mean, const float* rstd,
                                        const float* gamma, void* dx, float* dgamma, float* dbeta, float* part, int64_t T,
                                        int C, int x_dtype, void* stream) {
    if (T <= 0) return DGX_OK;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !part || (C & 3) || C > 768) return DGX_ERR_BAD_ARG;
    const WinMap m = make_map(0, 0, 0, 0, 0);
    const int grid = dgx_layernorm_bwd_blocks(T);
    hipStream_t st = (hipStream_t)stream;
    const int nj = (C + 255) / 256;
#define LNF_BWD(NJ)                                                                                                        \
    do {                                                                                                                   \
        if (x_dtype == DGX_BF16)                                                                                           \
            hipLaunchKernelGGL((ln_bwd_kernel<NJ, uint16_t, float>), dim3(grid), dim3(64 * LNB_WAVES), 0, st, dy, (const uint16_t*)x,  \
                               mean, rstd, gamma, (const uint16_t*)nullptr, (uint16_t*)dx, part, T, C, m);                  \
        else                                                                                                               \
            hipLaunchKernelGGL((ln_bwd_kernel<NJ, float, float>), dim3(grid), dim3(64 * LNB_WAVES), 0, st, dy, (const float*)x, mean,  \
                               rstd, gamma, (const float*)nullptr, (float*)dx, part, T, C, m);                              \
    } while (0)
    if (nj <= 1) LNF_BWD(1);
    else if (nj <= 2) LNF_BWD(2);
    else LNF_BWD(3);
#undef LNF_BWD
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C / 4 + LNR_QB - 1) / LNR_QB), dim3(1024), 0, st, part, dgamma, dbeta, grid, C);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
