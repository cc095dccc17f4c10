// Shared pieces of libdgx's forward / input-gradient GEMM kernels (gemm_nt.hip: 8 waves that load their own operands, two workgroups
// per CU for the short contractions; gemm_lw.hip: 8 MFMA waves + 4 loader waves, persistent): problem descriptor, LDS-direct load
// helper, exact-GELU arithmetic and the fused tails of the read-out (bias | bias + GELU with both tensors | x GELU'(f1) |
// x ReLU'(act) | window-reverse + DropPath + residual).
#pragma once
#include "dgx_common.h"
#include "winmap.h"

constexpr int GBK = 64;                 // K-step (elements): 128-byte tile rows
static constexpr uint32_t G_OOB = 0x80000000u; // voffset beyond num_records: the lane's 16 bytes land in LDS as zeros

namespace dgxgemm {
struct GMap { int B, H, W, ws, shift, nWh, nWw, compact; };   // residual.hip's RMap; compact: winmap.h

struct GemmP {
    const uint16_t* A;
    const uint16_t* B;
    int M, N, K, lda, ldb;
    int tiles_n, total, per_xcd;
    int splits, kt_per_split;  // split-K: workgroup (tile, s) contracts K-tiles [s * kt_per_split, ...) into ws[s] (fp32)
    float* ws;
    int mode;
    unsigned long long* dbg; // development: per-workgroup phase timestamps (s_memtime), 8 per workgroup, or null
    uint16_t* C;            // bf16 (M, ldc): modes 0, 1, 2 (pre-activation), 4
    int ldc;
    const uint16_t* bias;   // bf16 (N) or null
    uint16_t* C2;           // mode 2: GELU(C)
    const uint16_t* aux;    // mode 4: f1 (M, ldaux)
    int ldaux;
    const void* res;        // mode 3: residual stream (tokens, N) fp32 | bf16
    void* out;              //         out = res + scale[b] * y
    const float* scale;     //         per-sample DropPath factor or null
    int res_dtype;
    GMap map;
    // implicit 3x3 convolution (pad 1, stride 1) over a zero-bordered NHWC image (dgx_conv3x3_*): K-tile kt of the A operand
    // is tap kt / conv_kc, channels 64 (kt % conv_kc) ..: the SAME rows shifted by (tap / 3) * conv_wp + tap % 3 pixels;
    // GEMM rows are the N*H*W OUTPUT pixels (cmap: n, h, w); row m reads the padded position of its pixel (per-lane offset), so
    // no border rows are computed and M tiles exactly when N*H*W does (P3 level, 2 x 128 x 128 = 256 tiles of 128 rows: one round
    // of the chip; over the padded grid it was 265 tiles = two rounds)
    int conv_kc, conv_wp;
    int cmap_n, cmap_h, cmap_w;
    int relu;
    int krot;
    // grouped implicit convolution (dgx_conv3x3_gemm_multi): ngrp > 0 images that share B / bias / N / K / the epilogue -- the FPN
    // levels under one tower layer -- in ONE launch: tile L belongs to group g = the last one with tile0 <= L; the kernel patches
    // A / C / M and the image geometry from the group's record and goes on as for a single image (no split-K in this form)
    int ngrp;
    struct Grp { const uint16_t* A; uint16_t* C; int M, cn, ch, cw, wp, tile0; } grp[6];
};
constexpr int GEMM_MAXG = 6;
}  // namespace dgxgemm
using dgxgemm::GMap;
using dgxgemm::GemmP;

namespace {

__device__ __forceinline__ void g_load_lds16(uint32_t voff, u32x4 rsrc, uint32_t lds_addr, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32x4 g_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
template <int N> __device__ __forceinline__ void g_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void g_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void g_bar() {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

constexpr float kGInvSqrt2 = 0.70710678118654752440f;
constexpr float kGInvSqrt2Pi = 0.39894228040143267794f;
__device__ __forceinline__ void g_unpack8(const u32x4 r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 g_pack8(const float (&v)[8]) {
    return u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}

// window-order output row -> token index of the (B, H*W) stream, -1 for a padding row (residual.hip: residual_bwd_kernel)
__device__ __forceinline__ int64_t g_row_token(const GMap& m, int64_t orow, int& b) {
    if (m.ws == 0) {
        b = (int)(orow / ((int64_t)m.H * m.W));
        return orow;
    }
    if (m.compact) return wm_token_of_row(wm_geom(m.H, m.W, m.ws, m.shift), (int)orow, b);      // every compact row is a real token
    const int Nw = m.ws * m.ws;
    const int n = (int)(orow % Nw);
    int64_t t = orow / Nw;
    const int wc = (int)(t % m.nWw);
    t /= m.nWw;
    const int wr = (int)(t % m.nWh);
    b = (int)(t / m.nWh);
    int hh = wr * m.ws + n / m.ws + m.shift, ww = wc * m.ws + n % m.ws + m.shift;
    const int Hp = m.nWh * m.ws, Wp = m.nWw * m.ws;
    if (hh >= Hp) hh -= Hp;
    if (ww >= Wp) ww -= Wp;
    return (hh < m.H && ww < m.W) ? ((int64_t)b * m.H + hh) * m.W + ww : -1;
}

// Exact-GELU pieces  cdf(x) = (1 + erf(x / sqrt 2)) / 2  and  pdf(x) = exp(-x^2 / 2) / sqrt(2 pi)  from ONE exponential:
// erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), z >= 0 (Abramowitz & Stegun 7.1.26,
// |error| <= 1.5e-7 absolute, i.e. ~2 ulp of an fp32 erf near 1 and 4 orders below the bf16 quantum of the results it
// feeds); the negative side uses erfc directly, so the tail keeps its relative accuracy.  ~16 VALU operations per
// element against ~37 for the two-branch erff of the device library -- the epilogue is VALU-bound on it.
__device__ __forceinline__ void g_gelu_terms(float x, float& cdf, float& pdf) {
    const float z = fabsf(x) * kGInvSqrt2;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __expf(-z * z);
    float p = 1.061405429f;                        // explicit fma: the library is built with -ffp-contract=off
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float h = 0.5f * (p * t) * e;            // erfc(|z|) / 2
    cdf = x < 0.f ? h : 1.0f - h;
    pdf = e * kGInvSqrt2Pi;
}
// The same arithmetic on PAIRS of elements: every plain operation is a packed-fp32 instruction (v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32: two results per issue slot, the same IEEE roundings as the scalar forms, so results are bit-identical); only the
// reciprocal, the exponential and the sign select stay per element.  Halves the VALU slots of the GELU read-outs; measured
// in situ the read-out is bound by its loads and stores, not by these (fc1 forward 70.8 -> 69.6 us).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void g_gelu_terms2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * kGInvSqrt2;
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 nz = -z;
    const f32x2 zz = nz * z;
    const f32x2 e = {__expf(zz[0]), __expf(zz[1])};
    f32x2 p = {1.061405429f, 1.061405429f};
    p = __builtin_elementwise_fma(p, t, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
    const f32x2 h = (p * t) * 0.5f * e;           // scalar form: 0.5f * (p * t) * e -- multiplication commutes exactly
    const f32x2 o = f32x2{1.0f, 1.0f} - h;
    cdf = f32x2{x[0] < 0.f ? h[0] : o[0], x[1] < 0.f ? h[1] : o[1]};
    pdf = e * kGInvSqrt2Pi;
}
__device__ __forceinline__ f32x2 g_gelu2(f32x2 x) {
    f32x2 cdf, pdf;
    g_gelu_terms2(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ f32x2 g_gelu_grad2(f32x2 x) {
    f32x2 cdf, pdf;
    g_gelu_terms2(x, cdf, pdf);
    return cdf + x * pdf;                          // mul then add, as the scalar form (no contraction)
}
__device__ __forceinline__ float g_gelu(float x) {
    float cdf, pdf;
    g_gelu_terms(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float g_gelu_grad(float x) {
    float cdf, pdf;
    g_gelu_terms(x, cdf, pdf);
    return cdf + x * pdf;
}

// The fused tail of one 8-column chunk y = bf16(acc + bias) of output row gm (shared by the GEMM read-out and the split-K
// fold).  tok / sc: mode 3 token index (>= 0) and DropPath factor of the row.
// The fused tail of one 8-column chunk y = bf16(acc + bias) of output row gm, in two halves so that callers can put many
// chunks' extra operands (saved pre-activation of mode 4, residual of mode 3) in flight before consuming any:
// g_epi_prefetch issues the loads, g_epi_finish does the arithmetic and the stores.  tok / sc: mode 3 token index (>= 0)
// and DropPath factor of the row.
__device__ __forceinline__ void g_epi_prefetch(const GemmP& P, int gm, int gn, int64_t tok, u32x4& xa, u32x4& xb) {
    if (P.mode == 4 || P.mode == 5) {
        xa = *reinterpret_cast<const u32x4*>(P.aux + (int64_t)gm * P.ldaux + gn);
    } else if (P.mode == 3) {
        const int64_t o = tok * P.N + gn;
        if (P.res_dtype == DGX_BF16) {
            xa = *reinterpret_cast<const u32x4*>((const uint16_t*)P.res + o);
        } else {
            xa = reinterpret_cast<const u32x4*>((const float*)P.res + o)[0];
            xb = reinterpret_cast<const u32x4*>((const float*)P.res + o)[1];
        }
    }
}
__device__ __forceinline__ void g_epi_finish(const GemmP& P, int gm, int gn, const u32x4 y, int64_t tok, float sc, const u32x4 xa,
                                             const u32x4 xb) {
    if (P.mode <= 1) {
        u32x4 o = y;
        if (P.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ((o[k] & 0x8000u) ? 0u : (o[k] & 0xffffu)) | ((o[k] & 0x80000000u) ? 0u : (o[k] & 0xffff0000u));
        }
        *reinterpret_cast<u32x4*>(P.C + (int64_t)gm * P.ldc + gn) = o;
    } else if (P.mode == 2) {
        *reinterpret_cast<u32x4*>(P.C + (int64_t)gm * P.ldc + gn) = y;
        float v[8], o[8];
        g_unpack8(y, v);
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const f32x2 r = g_gelu2(f32x2{v[k], v[k + 1]});
            o[k] = r[0]; o[k + 1] = r[1];
        }
        *reinterpret_cast<u32x4*>(P.C2 + (int64_t)gm * P.ldc + gn) = g_pack8(o);
    } else if (P.mode == 4) {
        float gq[8], v[8], d[8];
        g_unpack8(y, gq);
        g_unpack8(xa, v);
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const f32x2 r = f32x2{gq[k], gq[k + 1]} * g_gelu_grad2(f32x2{v[k], v[k + 1]});
            d[k] = r[0]; d[k + 1] = r[1];
        }
        *reinterpret_cast<u32x4*>(P.C + (int64_t)gm * P.ldc + gn) = g_pack8(d);
    } else if (P.mode == 5) {                      // ReLU': pass y where the saved activation is positive
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = xa[k];
            const uint32_t lo = ((a & 0x8000u) || !(a & 0x7fffu)) ? 0u : 0xffffu;
            const uint32_t hi = ((a & 0x80000000u) || !(a & 0x7fff0000u)) ? 0u : 0xffff0000u;
            o[k] = y[k] & (lo | hi);
        }
        *reinterpret_cast<u32x4*>(P.C + (int64_t)gm * P.ldc + gn) = o;
    } else {                                       // mode 3
        float yv[8], xv[8];
        g_unpack8(y, yv);
        const int64_t o = tok * P.N + gn;
        if (P.res_dtype == DGX_BF16) {
            g_unpack8(xa, xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] += sc * yv[k];
            *reinterpret_cast<u32x4*>((uint16_t*)P.out + o) = g_pack8(xv);
        } else {
            f32x4 x0 = __builtin_bit_cast(f32x4, xa), x1 = __builtin_bit_cast(f32x4, xb);
#pragma unroll
            for (int k = 0; k < 4; ++k) { x0[k] += sc * yv[k]; x1[k] += sc * yv[4 + k]; }
            reinterpret_cast<f32x4*>((float*)P.out + o)[0] = x0;
            reinterpret_cast<f32x4*>((float*)P.out + o)[1] = x1;
        }
    }
}
__device__ __forceinline__ void g_epilogue_chunk(const GemmP& P, int gm, int gn, const u32x4 y, int64_t tok, float sc) {
    u32x4 xa = {0u, 0u, 0u, 0u}, xb = {0u, 0u, 0u, 0u};
    g_epi_prefetch(P, gm, gn, tok, xa, xb);
    g_epi_finish(P, gm, gn, y, tok, sc, xa, xb);
}
}  // namespace
