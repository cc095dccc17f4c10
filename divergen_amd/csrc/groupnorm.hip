// GroupNorm (+ optional ReLU) over channels-last activations for the CenterNet tower
// (CN/modeling/dense_heads/centernet_head.py:52-75: Conv3x3 -> GroupNorm(32, 256) -> ReLU, x4 per level).
// x (N, HW, C) bf16, 8 channels per group (C = 8 G): one 16-byte vector = one group at one pixel, so every
// pass is a fully coalesced stream.  HBM-bound: forward reads x twice + writes y once; backward reads x, dy twice.
//   stats   : slab partial sums per (n, g) in fp32 (shifted sums: no cancellation)
//   apply   : folds the partials into mean / rstd per workgroup, then y = relu?((x - mean) * rstd * gamma + beta)
//   bwd red : per (n, g): s1 = sum dyh*gamma, s2 = sum dyh*gamma*xhat and the per-channel sums for dgamma / dbeta
//   bwd dx  : dx = rstd * (dyh*gamma - (s1 + xhat*s2)/m),  dyh = dy * (y > 0)
#include "dgx_common.h"

namespace {
__device__ __forceinline__ void unpack8(const u32x4 r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    return u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}
__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
}  // namespace

// grid (N*G, S): slab s of the pixels of one (n, g).  Sums are taken of x - x0 (x0 = the group's first element) so that
// var = E[d^2] - E[d]^2 does not cancel when the mean is large against the spread.  part[(ng*S + s)*2 + {0,1}].
__device__ __forceinline__ void gn_stats_partial_body(const uint16_t* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                               int G, int S, int bx, int by) {
    __shared__ float red[4];
    const int n = bx / G, g = bx % G, s = by;
    const uint16_t* base = x + (int64_t)n * HW * C + 8 * g;
    const float x0 = bf2f(base[0]);
    const int per = (HW + S - 1) / S, p0 = s * per, p1 = min(HW, p0 + per);
    float a = 0.f, q = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(base + (int64_t)p * C), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i] - x0; a += d; q += d * d; }
    }
    a = block_sum(a, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) { part[((int64_t)bx * S + s) * 2] = a; part[((int64_t)bx * S + s) * 2 + 1] = q; }
}

// grid (blocks per sample, N): every workgroup first folds the slab partials of its sample's G groups into mean / rstd
// (LDS; workgroup 0 of the sample also stores them for the backward), then normalises its share of the pixels.
__device__ __forceinline__ void gn_apply_body(const uint16_t* __restrict__ x, const float* __restrict__ part,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       uint16_t* __restrict__ y, int HW, int C, int G, int S, float eps, int relu, int bx, int by, int gdx) {
    __shared__ float mu_s[256], rs_s[256];
    const int n = by;
    // eight lanes per group fold the slab partials (lane u: slabs u, u + 8, ..; then across the eight by shuffles) -- with one lane per
    // group this was a serial chain of 2 S dependent loads in front of every workgroup's work
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + (threadIdx.x >> 3), u = threadIdx.x & 7;
        float a = 0.f, q = 0.f;
        const int ng = n * G + (g < G ? g : 0);
        if (g < G)
            for (int s = u; s < S; s += 8) { a += part[((int64_t)ng * S + s) * 2]; q += part[((int64_t)ng * S + s) * 2 + 1]; }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
        if (g < G && u == 0) {
            const float x0 = bf2f(x[(int64_t)n * HW * C + 8 * g]);
            const float m = (float)HW * 8.0f;
            const float md = a / m;
            const float var = fmaxf(q / m - md * md, 0.0f);
            const float mu = x0 + md, rs = rsqrtf(var + eps);
            mu_s[g] = mu;
            rs_s[g] = rs;
            if (bx == 0) { mean[ng] = mu; rstd[ng] = rs; }
        }
    }
    __syncthreads();
    const int64_t per_n = (int64_t)HW * G;
    const u32x4* xv = reinterpret_cast<const u32x4*>(x) + (int64_t)n * per_n;
    u32x4* yv = reinterpret_cast<u32x4*>(y) + (int64_t)n * per_n;
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < per_n; i += (int64_t)gdx * 256) {
        const int g = (int)(i % G);
        const float mu = mu_s[g], rs = rs_s[g];
        float v[8], o[8];
        unpack8(xv[i], v);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 8 * g), g1 = *reinterpret_cast<const f32x4*>(gamma + 8 * g + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + 8 * g), b1 = *reinterpret_cast<const f32x4*>(beta + 8 * g + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gm = k < 4 ? g0[k] : g1[k - 4], bt = k < 4 ? b0[k] : b1[k - 4];
            const float t = (v[k] - mu) * rs * gm + bt;
            o[k] = relu ? fmaxf(t, 0.f) : t;
        }
        yv[i] = pack8(o);
    }
}

// part2: [N*G][S][18] = s1, s2, dgamma[8], dbeta[8] of slab s (folded by the dx and parameter kernels)
__device__ __forceinline__ void gn_bwd_reduce_body(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ part2, int HW, int C, int G, int relu, int S, int bx, int by) {
    __shared__ float red[4];
    const int n = bx / G, g = bx % G;
    const int64_t off = (int64_t)n * HW * C + 8 * g;
    const float mu = mean[bx], rs = rstd[bx];
    float gm[8], bt[8], dg[8], db[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gm[k] = gamma[8 * g + k]; bt[k] = beta[8 * g + k]; dg[k] = db[k] = 0.f; }
    float s1 = 0.f, s2 = 0.f;
    const int per = (HW + S - 1) / S, p0 = by * per, p1 = min(HW, p0 + per);
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        float v[8], d[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + off + (int64_t)p * C), v);
        unpack8(*reinterpret_cast<const u32x4*>(dy + off + (int64_t)p * C), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (v[k] - mu) * rs;
            const float dh = (relu && !(xh * gm[k] + bt[k] > 0.f)) ? 0.f : d[k];
            dg[k] += dh * xh;
            db[k] += dh;
            s1 += dh * gm[k];
            s2 += dh * gm[k] * xh;
        }
    }
    float* out = part2 + ((int64_t)bx * S + by) * 18;
    float r = block_sum(s1, red);
    if (threadIdx.x == 0) out[0] = r;
    r = block_sum(s2, red);
    if (threadIdx.x == 0) out[1] = r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r = block_sum(dg[k], red);
        if (threadIdx.x == 0) out[2 + k] = r;
        r = block_sum(db[k], red);
        if (threadIdx.x == 0) out[10 + k] = r;
    }
}

// ---- G = 32 (C = 256: every GroupNorm of the shipped heads): the two reduction passes PIXEL-major.  The forms above give a
// workgroup one (sample, group): 16 bytes out of every 512-byte pixel row -- each 128-byte line is fetched by eight workgroups at
// different times (bwd reduce: 44 MB in 58 us).  Here a workgroup owns a slab of whole pixels of one sample: lane t reads group
// t & 31 of pixel t >> 5 (a wave = two whole rows, 1 KB contiguous), keeps its sums in registers, and the eight lanes of a group
// (t, t + 32, ..: lane pairs through a shuffle, waves through LDS) fold into the same [n*G + g][slab] partials as before
// (58 -> 33 us, statistics pass 27 -> 14 us).
constexpr int GN_RG = 32;
__device__ __forceinline__ void gn_fold32(float v, int k, float (*red)[GN_RG][18]) {   // value k of this lane's group -> red[wave][g][k]
    v += __shfl_xor(v, 32);
    if ((threadIdx.x & 63) < 32) red[threadIdx.x >> 6][threadIdx.x & 31][k] = v;
}
__device__ __forceinline__ void gn_stats_rows_body(const uint16_t* __restrict__ x, float* __restrict__ part, int HW, int C, int S, int s,
                                                   int n) {
    __shared__ float red[4][GN_RG][18];
    const int g = threadIdx.x & 31, j = threadIdx.x >> 5;
    const uint16_t* base = x + (int64_t)n * HW * C + 8 * g;
    const float x0 = bf2f(base[0]);
    const int per = (HW + S - 1) / S, p0 = s * per, p1 = min(HW, p0 + per);
    float a = 0.f, q = 0.f;
    for (int p = p0 + j; p < p1; p += 8) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(base + (int64_t)p * C), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i] - x0; a += d; q += d * d; }
    }
    gn_fold32(a, 0, red);
    gn_fold32(q, 1, red);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int gg = threadIdx.x >> 1, k = threadIdx.x & 1;
        part[(((int64_t)n * GN_RG + gg) * S + s) * 2 + k] = (red[0][gg][k] + red[1][gg][k]) + (red[2][gg][k] + red[3][gg][k]);
    }
}
__device__ __forceinline__ void gn_bwd_reduce_rows_body(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ part2, int HW, int C, int relu, int S, int s, int n) {
    __shared__ float red[4][GN_RG][18];
    const int g = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int64_t off = (int64_t)n * HW * C + 8 * g;
    const float mu = mean[n * GN_RG + g], rs = rstd[n * GN_RG + g];
    float gm[8], bt[8], dg[8], db[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gm[k] = gamma[8 * g + k]; bt[k] = beta[8 * g + k]; dg[k] = db[k] = 0.f; }
    float s1 = 0.f, s2 = 0.f;
    const int per = (HW + S - 1) / S, p0 = s * per, p1 = min(HW, p0 + per);
    for (int p = p0 + j; p < p1; p += 8) {
        float v[8], d[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + off + (int64_t)p * C), v);
        unpack8(*reinterpret_cast<const u32x4*>(dy + off + (int64_t)p * C), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (v[k] - mu) * rs;
            const float dh = (relu && !(xh * gm[k] + bt[k] > 0.f)) ? 0.f : d[k];
            dg[k] += dh * xh;
            db[k] += dh;
            s1 += dh * gm[k];
            s2 += dh * gm[k] * xh;
        }
    }
    gn_fold32(s1, 0, red);
    gn_fold32(s2, 1, red);
#pragma unroll
    for (int k = 0; k < 8; ++k) { gn_fold32(dg[k], 2 + k, red); gn_fold32(db[k], 10 + k, red); }
    __syncthreads();
    for (int e = threadIdx.x; e < GN_RG * 18; e += 256) {
        const int gg = e / 18, k = e - gg * 18;
        part2[(((int64_t)n * GN_RG + gg) * S + s) * 18 + k] = (red[0][gg][k] + red[1][gg][k]) + (red[2][gg][k] + red[3][gg][k]);
    }
}

// grid (blocks per sample, N): s1 / s2 of the sample's groups folded from the slab partials into LDS, then dx
__device__ __forceinline__ void gn_bwd_dx_body(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ part2, uint16_t* __restrict__ dx, int HW, int C,
                                                        int G, int S, int relu, int bx, int by, int gdx) {
    __shared__ float s1_s[256], s2_s[256];
    const int n = by;
    for (int g0 = 0; g0 < G; g0 += 32) {           // eight lanes per group: see gn_apply_body
        const int g = g0 + (threadIdx.x >> 3), u = threadIdx.x & 7;
        float a = 0.f, b = 0.f;
        if (g < G)
            for (int s = u; s < S; s += 8) {
                a += part2[(((int64_t)n * G + g) * S + s) * 18];
                b += part2[(((int64_t)n * G + g) * S + s) * 18 + 1];
            }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (g < G && u == 0) { s1_s[g] = a; s2_s[g] = b; }
    }
    __syncthreads();
    const float inv_m = 1.0f / ((float)HW * 8.0f);
    const int64_t per_n = (int64_t)HW * G;
    const u32x4* xv = reinterpret_cast<const u32x4*>(x) + (int64_t)n * per_n;
    const u32x4* dv = reinterpret_cast<const u32x4*>(dy) + (int64_t)n * per_n;
    u32x4* ov = reinterpret_cast<u32x4*>(dx) + (int64_t)n * per_n;
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < per_n; i += (int64_t)gdx * 256) {
        const int g = (int)(i % G);
        const int ng = n * G + g;
        const float mu = mean[ng], rs = rstd[ng], s1 = s1_s[g], s2 = s2_s[g];
        float v[8], d[8], o[8];
        unpack8(xv[i], v);
        unpack8(dv[i], d);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 8 * g), g1 = *reinterpret_cast<const f32x4*>(gamma + 8 * g + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + 8 * g), b1 = *reinterpret_cast<const f32x4*>(beta + 8 * g + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gm = k < 4 ? g0[k] : g1[k - 4], bt = k < 4 ? b0[k] : b1[k - 4];
            const float xh = (v[k] - mu) * rs;
            const float dh = (relu && !(xh * gm + bt > 0.f)) ? 0.f : d[k];
            o[k] = rs * (dh * gm - (s1 + xh * s2) * inv_m);
        }
        ov[i] = pack8(o);
    }
}

// dgamma[c] += sum over samples and slabs of part2[n][g][s][2 + k], dbeta likewise (c = 8 g + k); fixed order.  One wave per group:
// lane l takes the (sample, slab) entries l, l + 64, .. and the 16 sums go through a shuffle tree (one lane per channel walking
// every entry was 14 us on a single workgroup, 35 us with the 64 slabs per sample of the pixel-major passes)
__device__ __forceinline__ void gn_param_wave_store(float (&acc)[16], float* __restrict__ dgamma, float* __restrict__ dbeta, int g) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { dgamma[8 * g + k] += acc[k]; dbeta[8 * g + k] += acc[8 + k]; }
    }
}
__global__ __launch_bounds__(64) void gn_bwd_param_kernel(const float* __restrict__ part2, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int N, int C, int G, int S) {
    const int g = blockIdx.x;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int e = threadIdx.x; e < N * S; e += 64) {
        const int n = e / S, s = e - n * S;
        const float* p = part2 + (((int64_t)n * G + g) * S + s) * 18 + 2;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] += p[k];
    }
    gn_param_wave_store(acc, dgamma, dbeta, g);
}

// ---- launchable forms: one tensor, or up to GN_MAXI tensors (the FPN levels of one tower layer: same weights, same C / G, different
// N / HW) in ONE launch, blockIdx.z = the tensor -- the small levels' workgroups then run beside the large level's instead of
// in latency-bound launches of their own
constexpr int GN_MAXI = 8;
struct GnItem {
    const uint16_t* x; const uint16_t* dy; uint16_t* out; float* mean; float* rstd; float* scratch;
    int N, HW, S, bpn;
};
struct GnMulti { GnItem it[GN_MAXI]; int n; };

__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const uint16_t* x, float* part, int HW, int C, int G, int S) {
    if (G == GN_RG) gn_stats_rows_body(x, part, HW, C, S, blockIdx.x, blockIdx.y);       // grid (slabs, samples)
    else gn_stats_partial_body(x, part, HW, C, G, S, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* x, const float* part, float* mean, float* rstd, const float* gamma,
                                                       const float* beta, uint16_t* y, int HW, int C, int G, int S, float eps, int relu) {
    gn_apply_body(x, part, mean, rstd, gamma, beta, y, HW, C, G, S, eps, relu, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const uint16_t* x, const uint16_t* dy, const float* mean, const float* rstd,
                                                            const float* gamma, const float* beta, float* part2, int HW, int C, int G, int relu,
                                                            int S) {
    if (G == GN_RG) gn_bwd_reduce_rows_body(x, dy, mean, rstd, gamma, beta, part2, HW, C, relu, S, blockIdx.x, blockIdx.y);
    else gn_bwd_reduce_body(x, dy, mean, rstd, gamma, beta, part2, HW, C, G, relu, S, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const uint16_t* x, const uint16_t* dy, const float* mean, const float* rstd,
                                                        const float* gamma, const float* beta, const float* part2, uint16_t* dx, int HW, int C,
                                                        int G, int S, int relu) {
    gn_bwd_dx_body(x, dy, mean, rstd, gamma, beta, part2, dx, HW, C, G, S, relu, blockIdx.x, blockIdx.y, gridDim.x);
}

__global__ __launch_bounds__(256) void gn_stats_partial_multi_kernel(GnMulti M, int C, int G) {
    const GnItem& t = M.it[blockIdx.z];
    if (G == GN_RG) {                              // grid (slabs, samples, tensors)
        if ((int)blockIdx.x < t.S && (int)blockIdx.y < t.N) gn_stats_rows_body(t.x, t.scratch, t.HW, C, t.S, blockIdx.x, blockIdx.y);
        return;
    }
    if ((int)blockIdx.x >= t.N * G || (int)blockIdx.y >= t.S) return;
    gn_stats_partial_body(t.x, t.scratch, t.HW, C, G, t.S, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void gn_apply_multi_kernel(GnMulti M, const float* gamma, const float* beta, int C, int G, float eps,
                                                             int relu) {
    const GnItem& t = M.it[blockIdx.z];
    if ((int)blockIdx.y >= t.N || (int)blockIdx.x >= t.bpn) return;
    gn_apply_body(t.x, t.scratch, t.mean, t.rstd, gamma, beta, t.out, t.HW, C, G, t.S, eps, relu, blockIdx.x, blockIdx.y, t.bpn);
}
__global__ __launch_bounds__(256) void gn_bwd_reduce_multi_kernel(GnMulti M, const float* gamma, const float* beta, int C, int G, int relu) {
    const GnItem& t = M.it[blockIdx.z];
    if (G == GN_RG) {
        if ((int)blockIdx.x < t.S && (int)blockIdx.y < t.N)
            gn_bwd_reduce_rows_body(t.x, t.dy, t.mean, t.rstd, gamma, beta, t.scratch + (int64_t)t.N * G * 18, t.HW, C, relu, t.S, blockIdx.x, blockIdx.y);
        return;
    }
    if ((int)blockIdx.x >= t.N * G || (int)blockIdx.y >= t.S) return;
    gn_bwd_reduce_body(t.x, t.dy, t.mean, t.rstd, gamma, beta, t.scratch + (int64_t)t.N * G * 18, t.HW, C, G, relu, t.S, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void gn_bwd_dx_multi_kernel(GnMulti M, const float* gamma, const float* beta, int C, int G, int relu) {
    const GnItem& t = M.it[blockIdx.z];
    if ((int)blockIdx.y >= t.N || (int)blockIdx.x >= t.bpn) return;
    gn_bwd_dx_body(t.x, t.dy, t.mean, t.rstd, gamma, beta, t.scratch + (int64_t)t.N * G * 18, t.out, t.HW, C, G, t.S, relu, blockIdx.x, blockIdx.y,
                   t.bpn);
}
// dgamma[c] += sum over the tensors (in list order), their samples and slabs -- one wave per group, fixed order
__global__ __launch_bounds__(64) void gn_bwd_param_multi_kernel(GnMulti M, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int G) {
    const int g = blockIdx.x;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int i = 0; i < M.n; ++i) {
        const GnItem& t = M.it[i];
        const float* part2 = t.scratch + (int64_t)t.N * G * 18;
        for (int e = threadIdx.x; e < t.N * t.S; e += 64) {
            const int n = e / t.S, s = e - n * t.S;
            const float* p = part2 + (((int64_t)n * G + g) * t.S + s) * 18 + 2;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] += p[k];
        }
    }
    gn_param_wave_store(acc, dgamma, dbeta, g);
}

// pixel slabs per (n, g) so that the two reduction kernels fill the GPU (N*G alone is 64 workgroups for a batch of 2)
static int gn_slabs(int NG, int HW, int G) {
    if (G == GN_RG) {                              // pixel-major passes: slabs of whole pixels per SAMPLE, >= 128 pixels each
        const int S = (HW + 127) / 128;
        return S < 1 ? 1 : (S > 64 ? 64 : S);
    }
    int S = (1024 + NG - 1) / NG;
    const int mx = (HW + 255) / 256;
    if (S > mx) S = mx;
    return S < 1 ? 1 : (S > 64 ? 64 : S);
}

extern "C" int64_t dgx_groupnorm_scratch_floats(int N, int HW, int G) { return (int64_t)N * G * (gn_slabs(N * G, HW, G) * 18 + 18); }

extern "C" int dgx_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 float* scratch, int N, int HW, int C, int G, float eps, int relu, void* stream) {
    if (N <= 0 || HW <= 0) return DGX_OK;
    if (!x || !gamma || !beta || !y || !mean || !rstd || C != 8 * G) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int S = gn_slabs(N * G, HW, G);
    if (!scratch) return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(gn_stats_partial_kernel, G == GN_RG ? dim3(S, N) : dim3(N * G, S), dim3(256), 0, st, (const uint16_t*)x, scratch, HW, C, G, S);
    if (G > 256) return DGX_ERR_UNSUPPORTED;
    const int64_t per_n = (int64_t)HW * G;
    int bpn = (int)((per_n + 255) / 256);
    const int cap = (4096 + N - 1) / N;
    if (bpn > cap) bpn = cap;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(bpn, N), dim3(256), 0, st, (const uint16_t*)x, scratch, mean, rstd, gamma, beta,
                       (uint16_t*)y, HW, C, G, S, eps, relu);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_groupnorm_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, void* dx, float* dgamma, float* dbeta, float* part, int N, int HW, int C, int G,
                                 int relu, void* stream) {
    if (N <= 0 || HW <= 0) return DGX_OK;
    if (!x || !dy || !mean || !rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !part || C != 8 * G) return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int S = gn_slabs(N * G, HW, G);
    float* part2 = part + (int64_t)N * G * 18;      // scratch layout: [N*G][18] folded, then [N*G][S][18]
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, G == GN_RG ? dim3(S, N) : dim3(N * G, S), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)dy, mean, rstd, gamma,
                       beta, part2, HW, C, G, relu, S);
    if (G > 256) return DGX_ERR_UNSUPPORTED;
    const int64_t per_n = (int64_t)HW * G;
    int bpn = (int)((per_n + 255) / 256);
    const int cap = (4096 + N - 1) / N;
    if (bpn > cap) bpn = cap;
    hipLaunchKernelGGL(gn_bwd_dx_kernel, dim3(bpn, N), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)dy, mean, rstd, gamma, beta,
                       part2, (uint16_t*)dx, HW, C, G, S, relu);
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(G), dim3(64), 0, st, part2, dgamma, dbeta, N, C, G, S);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// Several tensors through the same GroupNorm (+ ReLU) in one launch per pass (see GnMulti): items[i] = {x, y, mean, rstd, scratch,
// N, HW}; scratch as for the single form (dgx_groupnorm_scratch_floats(N, HW, G) floats each).
static int gn_fill(GnMulti& M, const dgx_gn_item* items, int n, int G, bool bwd, int& maxNG, int& maxS, int& maxN, int& maxbpn) {
    M.n = n;
    maxNG = maxS = maxN = maxbpn = 0;
    for (int i = 0; i < n; ++i) {
        const dgx_gn_item& a = items[i];
        if (a.N <= 0 || a.HW <= 0 || !a.x || !a.out || !a.mean || !a.rstd || !a.scratch || (bwd && !a.dy)) return DGX_ERR_BAD_ARG;
        GnItem& t = M.it[i];
        t.x = (const uint16_t*)a.x; t.dy = (const uint16_t*)a.dy; t.out = (uint16_t*)a.out;
        t.mean = a.mean; t.rstd = a.rstd; t.scratch = a.scratch;
        t.N = a.N; t.HW = a.HW; t.S = gn_slabs(a.N * G, a.HW, G);
        const int64_t per_n = (int64_t)a.HW * G;
        int bpn = (int)((per_n + 255) / 256);
        const int cap = (4096 + a.N - 1) / a.N;
        t.bpn = bpn > cap ? cap : bpn;
        maxNG = a.N * G > maxNG ? a.N * G : maxNG;
        maxS = t.S > maxS ? t.S : maxS;
        maxN = a.N > maxN ? a.N : maxN;
        maxbpn = t.bpn > maxbpn ? t.bpn : maxbpn;
    }
    for (int i = n; i < GN_MAXI; ++i) M.it[i] = M.it[0];
    return DGX_OK;
}

extern "C" int dgx_groupnorm_fwd_multi(const dgx_gn_item* items, int n, const float* gamma, const float* beta, int C, int G, float eps, int relu,
                                       void* stream) {
    if (n <= 0) return DGX_OK;
    if (!items || n > GN_MAXI || !gamma || !beta || C != 8 * G || G > 256) return n > GN_MAXI ? DGX_ERR_UNSUPPORTED : DGX_ERR_BAD_ARG;
    GnMulti M;
    int maxNG, maxS, maxN, maxbpn;
    const int rc = gn_fill(M, items, n, G, false, maxNG, maxS, maxN, maxbpn);
    if (rc != DGX_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_stats_partial_multi_kernel, G == GN_RG ? dim3(maxS, maxN, n) : dim3(maxNG, maxS, n), dim3(256), 0, st, M, C, G);
    hipLaunchKernelGGL(gn_apply_multi_kernel, dim3(maxbpn, maxN, n), dim3(256), 0, st, M, gamma, beta, C, G, eps, relu);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_groupnorm_bwd_multi(const dgx_gn_item* items, int n, const float* gamma, const float* beta, float* dgamma, float* dbeta, int C,
                                       int G, int relu, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!items || n > GN_MAXI || !gamma || !beta || !dgamma || !dbeta || C != 8 * G || G > 256) return n > GN_MAXI ? DGX_ERR_UNSUPPORTED : DGX_ERR_BAD_ARG;
    GnMulti M;
    int maxNG, maxS, maxN, maxbpn;
    const int rc = gn_fill(M, items, n, G, true, maxNG, maxS, maxN, maxbpn);
    if (rc != DGX_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_reduce_multi_kernel, G == GN_RG ? dim3(maxS, maxN, n) : dim3(maxNG, maxS, n), dim3(256), 0, st, M, gamma, beta, C, G, relu);
    hipLaunchKernelGGL(gn_bwd_dx_multi_kernel, dim3(maxbpn, maxN, n), dim3(256), 0, st, M, gamma, beta, C, G, relu);
    hipLaunchKernelGGL(gn_bwd_param_multi_kernel, dim3(G), dim3(64), 0, st, M, dgamma, dbeta, C, G);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
