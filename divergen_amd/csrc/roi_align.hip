// ROIAlign / multi-level ROIPooler / GT-mask crop for gfx950.
//
// HBM-bound gathers.  Features are channels-last (N,H,W,C): one workgroup per RoI, one lane per
// channel, so every bilinear tap is a contiguous C*esize-byte read shared by the whole wave and the
// sample geometry is wave-uniform (scalar registers).  The float sequence of the sample positions
// and weights is the oracle's (oracle/roi_nms.c, torchvision's published algorithm) and this file
// is compiled with -ffp-contract=off, so bin geometry and the boolean mask targets are bit-exact.
// Backward scatters with hardware fp32 atomics into an fp32 (N,H,W,C) gradient.
// (The GT-mask crop keeps the per-sample form: its thresholded output must be bit-exact.)
#include "dgx_common.h"
#include <stdlib.h>

#define MAX_LEVELS 4
struct PoolLevels {
    const void* feat[MAX_LEVELS];
    float* grad[MAX_LEVELS];
    int H[MAX_LEVELS], W[MAX_LEVELS];
    float scale[MAX_LEVELS];
    int num_levels, min_level;
};

struct Taps { int pos[4]; float w[4]; };

__device__ __forceinline__ bool bilinear_taps(int H, int W, float y, float x, Taps& t) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - y_low, lx = x - x_low, hy = 1.0f - ly, hx = 1.0f - lx;
    t.pos[0] = y_low * W + x_low;   t.w[0] = hy * hx;
    t.pos[1] = y_low * W + x_high;  t.w[1] = hy * lx;
    t.pos[2] = y_high * W + x_low;  t.w[2] = ly * hx;
    t.pos[3] = y_high * W + x_high; t.w[3] = ly * lx;
    return true;
}

struct RoiGeom { float sw, sh, bin_w, bin_h; int gw, gh; float count; int b; };

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int ph, int pw, int sampling_ratio, bool aligned) {
    RoiGeom G;
    G.b = (int)roi[0];
    const float off = aligned ? 0.5f : 0.0f;
    G.sw = roi[1] * scale - off;
    G.sh = roi[2] * scale - off;
    const float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
    float rw = ew - G.sw, rh = eh - G.sh;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    G.bin_h = rh / (float)ph;
    G.bin_w = rw / (float)pw;
    G.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    G.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    G.count = (float)max(G.gh * G.gw, 1);
    return G;
}

// level = floor(canonical_level + log2(sqrt(area)/224 + 1e-8)) clamped (poolers.py:50-58); evaluated
// with exact power-of-two comparisons instead of log2 so it cannot straddle an integer.
__device__ __forceinline__ int assign_level(const float* roi, int min_level, int num_levels) {
    const float area = (roi[3] - roi[1]) * (roi[4] - roi[2]);
    float t = sqrtf(area) / 224.0f + 1e-8f;
    int lv = 4;
    if (!(t > 0.0f)) return 0;  // degenerate / NaN: lowest level
    while (t >= 2.0f && lv < 64) { t *= 0.5f; ++lv; }
    while (t < 1.0f && lv > -64) { t *= 2.0f; --lv; }
    lv -= min_level;
    return lv < 0 ? 0 : (lv >= num_levels ? num_levels - 1 : lv);
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }

// Separable form.  The bilinear weight of a sample on a pixel is hat(y)*hat(x) and a bin's samples
// form a tensor grid, so the total weight a bin puts on pixel (r,q) is WY[i][r]*WX[j][q] with 1-D
// tables built once per RoI in LDS (same clamping / validity rules as the per-sample form).  A bin
// then touches each pixel of its footprint once -- (bin_h+1)(bin_w+1) taps instead of
// 4*ceil(bin_h)*ceil(bin_w) -- and backward issues one atomic per footprint pixel per bin.
struct Span { int base, n; };

__device__ __forceinline__ void build_axis_table(float* Wt, Span* sp, int nb, int stride_tbl, float start, float bin,
                                                 int g, int extent, int tid, int tid0) {
    const int i = tid - tid0;
    if (i < 0 || i >= nb) return;
    float* w = Wt + i * stride_tbl;
    for (int k = 0; k < stride_tbl; ++k) w[k] = 0.0f;
    int base = -1, hi = -1;
    for (int s = 0; s < g; ++s) {
        float y = start + i * bin + ((float)s + 0.5f) * bin / (float)g;
        if (y < -1.0f || y > (float)extent) continue;
        if (y <= 0) y = 0;
        int lo = (int)y, up;
        if (lo >= extent - 1) { up = lo = extent - 1; y = (float)lo; } else up = lo + 1;
        const float l = y - lo, h = 1.0f - l;
        if (base < 0) base = lo;
        w[lo - base] += h;
        w[up - base] += l;
        hi = up;
    }
    sp[i].base = base < 0 ? 0 : base;
    sp[i].n = base < 0 ? 0 : hi - base + 1;
}

// grid (R, nsplit): block handles bins [split*bpb, ...).  NCHW output goes through an LDS
// transpose so that the (C, ph*pw) tile of a RoI is written as one contiguous run.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roi_align_sep_kernel(PoolLevels L, const float* __restrict__ rois, T* __restrict__ io,
                                                            int32_t* __restrict__ levels_out, int C, int ph, int pw,
                                                            int sampling_ratio, int aligned, int out_nhwc, int bins_per_block,
                                                            int spy, int spx) {
    extern __shared__ float smem[];
    float* WY = smem;                      // [ph][spy]
    float* WX = WY + ph * spy;             // [pw][spx]
    Span* SY = reinterpret_cast<Span*>(WX + pw * spx);
    Span* SX = SY + ph;
    float* tile = reinterpret_cast<float*>(SX + pw);  // [bins_per_block][C] when !out_nhwc
    const int r = blockIdx.x;
    const float* roi = rois + 5 * (int64_t)r;
    const int lvl = L.num_levels > 1 ? assign_level(roi, L.min_level, L.num_levels) : 0;
    if (!BWD && levels_out && blockIdx.y == 0 && threadIdx.x == 0) levels_out[r] = lvl;
    const int H = L.H[lvl], W = L.W[lvl];
    const RoiGeom G = roi_geom(roi, L.scale[lvl], ph, pw, sampling_ratio, aligned != 0);
    build_axis_table(WY, SY, ph, spy, G.sh, G.bin_h, G.gh, H, threadIdx.x, 0);
    build_axis_table(WX, SX, pw, spx, G.sw, G.bin_w, G.gw, W, threadIdx.x, 64);
    const int bins = ph * pw;
    const int bin0 = blockIdx.y * bins_per_block, bin1 = min(bins, bin0 + bins_per_block);
    const int nb = bin1 - bin0;
    if (BWD && !out_nhwc) {
        for (int idx = threadIdx.x; idx < nb * C; idx += blockDim.x) {
            const int c = idx / nb, bb = idx - c * nb;
            tile[bb * C + c] = ldf(io + ((int64_t)r * C + c) * bins + bin0 + bb);
        }
    }
    __syncthreads();
    const T* fb = BWD ? nullptr : (const T*)L.feat[lvl] + (int64_t)G.b * H * W * C;
    float* gb = BWD ? L.grad[lvl] + (int64_t)G.b * H * W * C : nullptr;
    for (int bin = bin0; bin < bin1; ++bin) {
        const int i = bin / pw, j = bin - i * pw;
        const Span sy = SY[i], sx = SX[j];
        const float* wy = WY + i * spy;
        const float* wx = WX + j * spx;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (BWD) {
                const float go = out_nhwc ? ldf(io + ((int64_t)r * bins + bin) * C + c) : tile[(bin - bin0) * C + c];
                const float g = go / G.count;
                for (int ky = 0; ky < sy.n; ++ky) {
                    const float gy = g * wy[ky];
                    float* row = gb + ((int64_t)(sy.base + ky) * W + sx.base) * C + c;
                    for (int kx = 0; kx < sx.n; ++kx) atomicAdd(row + (int64_t)kx * C, gy * wx[kx]);
                }
            } else {
                float acc = 0.0f;
                for (int ky = 0; ky < sy.n; ++ky) {
                    const T* row = fb + ((int64_t)(sy.base + ky) * W + sx.base) * C + c;
                    float racc = 0.0f;
                    for (int kx = 0; kx < sx.n; ++kx) racc += wx[kx] * ldf(row + (int64_t)kx * C);
                    acc += wy[ky] * racc;
                }
                const float v = acc / G.count;
                if (out_nhwc) stf(io + ((int64_t)r * bins + bin) * C + c, v);
                else tile[(bin - bin0) * C + c] = v;
            }
        }
    }
    if (!BWD && !out_nhwc) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < nb * C; idx += blockDim.x) {
            const int c = idx / nb, bb = idx - c * nb;
            stf(io + ((int64_t)r * C + c) * bins + bin0 + bb, tile[bb * C + c]);
        }
    }
}

// Vectorised form of the kernel above for C % 8 == 0 and channels-last OUTPUT: a lane owns 8 consecutive channels
// (one 16-byte load per tap for bf16 features), C/8 lanes cover a pixel and the 256/(C/8) lane groups of the
// workgroup walk different bins at the same time.  Same tables, same summation order per channel.
template <typename T> struct Vec8;
template <> struct Vec8<uint16_t> {
    static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[8]) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[8]) {
        *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    }
    // the load and its conversion apart: a batch of loads is requested before the first value is touched
    struct Raw { u32x4 r; };
    static __device__ __forceinline__ Raw load_raw(const uint16_t* p) { return Raw{*reinterpret_cast<const u32x4*>(p)}; }
    static __device__ __forceinline__ void unpack(const Raw& w, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w.r[i] << 16); v[2 * i + 1] = __uint_as_float(w.r[i] & 0xffff0000u); }
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    struct Raw { f32x4 a, b; };
    static __device__ __forceinline__ Raw load_raw(const float* p) { return Raw{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)}; }
    static __device__ __forceinline__ void unpack(const Raw& w, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = w.a[i]; v[4 + i] = w.b[i]; }
    }
};

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roi_align_vec_kernel(PoolLevels L, const float* __restrict__ rois, T* __restrict__ io,
                                                            int32_t* __restrict__ levels_out, int C, int ph, int pw,
                                                            int sampling_ratio, int aligned, int bins_per_block, int spy, int spx) {
    extern __shared__ float smem[];
    float* WY = smem;                      // [ph][spy]
    float* WX = WY + ph * spy;             // [pw][spx]
    Span* SY = reinterpret_cast<Span*>(WX + pw * spx);
    Span* SX = SY + ph;
    const int r = blockIdx.x;
    const float* roi = rois + 5 * (int64_t)r;
    const int lvl = L.num_levels > 1 ? assign_level(roi, L.min_level, L.num_levels) : 0;
    if (!BWD && levels_out && blockIdx.y == 0 && threadIdx.x == 0) levels_out[r] = lvl;
    const int H = L.H[lvl], W = L.W[lvl];
    const RoiGeom G = roi_geom(roi, L.scale[lvl], ph, pw, sampling_ratio, aligned != 0);
    build_axis_table(WY, SY, ph, spy, G.sh, G.bin_h, G.gh, H, threadIdx.x, 0);
    build_axis_table(WX, SX, pw, spx, G.sw, G.bin_w, G.gw, W, threadIdx.x, 64);
    __syncthreads();
    const int bins = ph * pw;
    const int bin0 = blockIdx.y * bins_per_block, bin1 = min(bins, bin0 + bins_per_block);
    const int CV = C >> 3, nslots = 256 / CV;
    const int cv = threadIdx.x % CV, slot = threadIdx.x / CV;
    if (slot >= nslots) return;
    const T* fb = BWD ? nullptr : (const T*)L.feat[lvl] + (int64_t)G.b * H * W * C + 8 * cv;
    // backward: lane owns channels cv, cv + CV, ..., so that one atomic instruction covers CV consecutive floats
    float* gb = BWD ? L.grad[lvl] + (int64_t)G.b * H * W * C + cv : nullptr;
    for (int bin = bin0 + slot; bin < bin1; bin += nslots) {
        const int i = bin / pw, j = bin - i * pw;
        const Span sy = SY[i], sx = SX[j];
        const float* wy = WY + i * spy;
        const float* wx = WX + j * spx;
        T* iop = io + ((int64_t)r * bins + bin) * C + 8 * cv;
        if (BWD) {
            float g[8];
            const T* gop = io + ((int64_t)r * bins + bin) * C + cv;
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = ldf(gop + CV * e) / G.count;
            for (int ky = 0; ky < sy.n; ++ky) {
                float* row = gb + ((int64_t)(sy.base + ky) * W + sx.base) * C;
                const float wyk = wy[ky];
                for (int kx = 0; kx < sx.n; ++kx) {
                    const float wxk = wx[kx];
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(row + (int64_t)kx * C + CV * e, (g[e] * wyk) * wxk);
                }
            }
        } else {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int ky = 0; ky < sy.n; ++ky) {
                const T* row = fb + ((int64_t)(sy.base + ky) * W + sx.base) * C;
                float racc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // four taps of a row requested together (a tap past the span repeats the last one with weight 0): one memory round trip
                // per four taps instead of one per tap (see the gather backward)
                for (int kx0 = 0; kx0 < sx.n; kx0 += 4) {
                    typename Vec8<T>::Raw raw[4];
                    float wxk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int kx = kx0 + k < sx.n ? kx0 + k : sx.n - 1;
                        wxk[k] = kx0 + k < sx.n ? wx[kx] : 0.0f;
                        raw[k] = Vec8<T>::load_raw(row + (int64_t)kx * C);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v[8];
                        Vec8<T>::unpack(raw[k], v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) racc[e] += wxk[k] * v[e];
                    }
                }
                const float wyk = wy[ky];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wyk * racc[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = acc[e] / G.count;
            Vec8<T>::store(iop, acc);
        }
    }
}

// ---- backward as an output-stationary gather ------------------------------------------------------------------------
// One workgroup owns a GT_TH x TW pixel tile of one (level, image) gradient map and every lane 8 channels of one tile
// column: it walks the RoIs in index order, keeps those of its image and level whose sample grid reaches the tile, rebuilds
// the two 1-D weight tables of each (the float sequence of build_axis_table, restricted to the tile) and accumulates
//   d feat[y][x][c] = sum_r sum_i WY_r[i][y] * ( sum_j WX_r[j][x] / count_r * d out[r][i][j][c] )
// in registers.  Every gradient pixel is written exactly once, in the output dtype: no atomics, no zero fill, no fp32
// staging map, and the summation order is fixed by the RoI order (bit-reproducible from run to run).
#define GT_TH 8
#define GT_QB 8
#define GT_PMAX 16
#define GT_TWMAX 32
struct GatherP {
    void* out[MAX_LEVELS];
    int H[MAX_LEVELS], W[MAX_LEVELS];
    float scale[MAX_LEVELS];
    int tile0[MAX_LEVELS + 1], tiles_x[MAX_LEVELS], tiles_y[MAX_LEVELS];
    int num_levels, min_level, N, C, R, ph, pw, sampling_ratio, aligned, total;
    int accumulate;      // the maps already hold a gradient (another pooling of the same features): add to it, one rounding
    float gscale;        // factor on this pooling's contribution (the cascade's _ScaleGradient on the pooled features), folded into the x table
};
struct GatherMeta { int r, i_lo, i_hi; float sw, bin_w; };

__device__ __forceinline__ void gather_axis_rows(float* w, int nw, int i, float start, float bin, int g, int extent, int p0) {
    for (int k = 0; k < nw; ++k) w[k] = 0.0f;
    for (int s = 0; s < g; ++s) {
        float y = start + i * bin + ((float)s + 0.5f) * bin / (float)g;
        if (y < -1.0f || y > (float)extent) continue;
        if (y <= 0) y = 0;
        int lo = (int)y, up;
        if (lo >= extent - 1) { up = lo = extent - 1; y = (float)lo; } else up = lo + 1;
        const float l = y - lo, h = 1.0f - l;
        if ((unsigned)(lo - p0) < (unsigned)nw) w[lo - p0] += h;
        if ((unsigned)(up - p0) < (unsigned)nw) w[up - p0] += l;
    }
}

__device__ __forceinline__ int clamp_bin(float v, int hi) { return (int)fminf(fmaxf(v, 0.0f), (float)hi); }

template <typename T, int TH>
__global__ __launch_bounds__(256) void roi_pool_bwd_gather_kernel(GatherP P, const float* __restrict__ rois, const T* __restrict__ go) {
    __shared__ __attribute__((aligned(16))) float WY[GT_QB][GT_PMAX][TH];
    __shared__ float WX[GT_QB][GT_PMAX][GT_TWMAX];
    __shared__ int list[256];
    __shared__ int wcount[4];
    __shared__ GatherMeta meta[GT_QB];
    const int tid = threadIdx.x;
    // coarsest level first: its tiles see the most RoIs each
    int t = P.total - 1 - (int)blockIdx.x, l = 0;
    while (l + 1 < P.num_levels && t >= P.tile0[l + 1]) ++l;
    int local = t - P.tile0[l];
    const int H = P.H[l], W = P.W[l], txn = P.tiles_x[l], tyn = P.tiles_y[l];
    const int C = P.C, CV = C >> 3, TW = 256 / CV, ph = P.ph, pw = P.pw, bins = ph * pw;
    const int n = local / (txn * tyn);
    local -= n * txn * tyn;
    const int y0 = (local / txn) * TH, x0 = (local % txn) * TW;
    const int cv = tid % CV, tx = tid / CV;
    const float scale = P.scale[l];
    float acc[TH][8];
#pragma unroll
    for (int a = 0; a < TH; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[a][e] = 0.0f;

    for (int base = 0; base < P.R; base += 256) {
        const int r = base + tid;
        bool hit = false;
        if (r < P.R) {
            const float* roi = rois + 5 * (int64_t)r;
            const int lvl = P.num_levels > 1 ? assign_level(roi, P.min_level, P.num_levels) : 0;
            if ((int)roi[0] == n && lvl == l) {
                const RoiGeom G = roi_geom(roi, scale, ph, pw, P.sampling_ratio, P.aligned != 0);
                // a sample at y weighs rows floor(y) and floor(y)+1 (after clamping into the map): rows within [y-1, y+1]
                hit = G.gh > 0 && G.gw > 0 && G.sh - 1.0f <= (float)(y0 + TH - 1) && G.sh + G.bin_h * ph + 1.0f >= (float)y0 &&
                      G.sw - 1.0f <= (float)(x0 + TW - 1) && G.sw + G.bin_w * pw + 1.0f >= (float)x0;
            }
        }
        const unsigned long long m = __ballot(hit);
        const int wave = tid >> 6, lane = tid & 63;
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int off = 0, hits = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int cnt = wcount[w]; if (w < wave) off += cnt; hits += cnt; }
        if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        for (int q0 = 0; q0 < hits; q0 += GT_QB) {
            const int nq = min(GT_QB, hits - q0);
            {
                const int q = tid >> 5, t5 = tid & 31;
                if (q < nq) {
                    const int rr = list[q0 + q];
                    const RoiGeom G = roi_geom(rois + 5 * (int64_t)rr, scale, ph, pw, P.sampling_ratio, P.aligned != 0);
                    if (t5 < 16) {
                        if (t5 < ph) gather_axis_rows(&WY[q][t5][0], TH, t5, G.sh, G.bin_h, G.gh, H, y0);
                    } else if (t5 - 16 < pw) {
                        float* w = &WX[q][t5 - 16][0];
                        gather_axis_rows(w, TW, t5 - 16, G.sw, G.bin_w, G.gw, W, x0);
                        for (int k = 0; k < TW; ++k) w[k] = w[k] / G.count * P.gscale;
                    }
                    if (t5 == 0) {
                        GatherMeta mt;
                        mt.r = rr; mt.sw = G.sw; mt.bin_w = G.bin_w;
                        mt.i_lo = 0; mt.i_hi = ph;
                        if (G.bin_h > 0.0f) {   // bins whose interval meets [y0 - 1, y0 + TH], one bin of slack either side
                            mt.i_lo = clamp_bin(floorf(((float)(y0 - 1) - G.sh) / G.bin_h) - 1.0f, ph);
                            mt.i_hi = clamp_bin(floorf(((float)(y0 + TH) - G.sh) / G.bin_h) + 2.0f, ph);
                        }
                        meta[q] = mt;
                    }
                }
            }
            __syncthreads();
            for (int q = 0; q < nq; ++q) {
                const GatherMeta mt = meta[q];
                int j_lo = 0, j_hi = pw;
                if (mt.bin_w > 0.0f) {
                    const float xf = (float)(x0 + tx);
                    j_lo = clamp_bin(floorf((xf - 1.0f - mt.sw) / mt.bin_w) - 1.0f, pw);
                    j_hi = clamp_bin(floorf((xf + 1.0f - mt.sw) / mt.bin_w) + 2.0f, pw);
                }
                const T* gr = go + (int64_t)mt.r * bins * C + 8 * cv;
                for (int i = mt.i_lo; i < mt.i_hi; ++i) {
                    float wy[TH];
#pragma unroll
                    for (int a4 = 0; a4 < TH; a4 += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(&WY[q][i][a4]);
                        wy[a4] = wv[0]; wy[a4 + 1] = wv[1]; wy[a4 + 2] = wv[2]; wy[a4 + 3] = wv[3];
                    }
                    bool any = false;
#pragma unroll
                    for (int a = 0; a < TH; ++a) any = any || wy[a] != 0.0f;
                    if (!any) continue;
                    float gx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const T* gi = gr + (int64_t)i * pw * C;
                    // The four columns of a chunk are requested TOGETHER and unconditionally (a column past the range repeats the last valid
                    // address with weight 0; a zero weight multiplies what it loaded): as `if (wx != 0) load` per column (round 3) the loads
                    // sat under lane conditions, each behind its own wait -- a box of 7 x 7 bins inside one pixel cost its tile 14 memory
                    // round trips in series (tools/roi_gather_probe.py: 1 024 such boxes 91 -> 61 us, in the step 115 -> 102 us per
                    // launch; eight columns, or two bin rows, per request lose again: more loads than the weights use).
                    for (int j0 = j_lo; j0 < j_hi; j0 += 4) {
                        typename Vec8<T>::Raw raw[4];
                        float wx[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int j = j0 + k < j_hi ? j0 + k : j_hi - 1;
                            wx[k] = j0 + k < j_hi ? WX[q][j][tx] : 0.0f;
                            raw[k] = Vec8<T>::load_raw(gi + (int64_t)j * C);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float v[8];
                            Vec8<T>::unpack(raw[k], v);
#pragma unroll
                            for (int e = 0; e < 8; ++e) gx[e] += wx[k] * v[e];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < TH; ++a)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[a][e] += wy[a] * gx[e];
                }
            }
            __syncthreads();
        }
    }
    T* ob = (T*)P.out[l] + (int64_t)n * H * W * C + 8 * cv;
    const int x = x0 + tx;
    if (x < W) {
        if (P.accumulate) {
            float prev[TH][8];
#pragma unroll
            for (int a = 0; a < TH; ++a)
                if (y0 + a < H) Vec8<T>::load(ob + ((int64_t)(y0 + a) * W + x) * C, prev[a]);
#pragma unroll
            for (int a = 0; a < TH; ++a)
                if (y0 + a < H) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[a][e] += prev[a][e];
                }
        }
#pragma unroll
        for (int a = 0; a < TH; ++a)
            if (y0 + a < H) Vec8<T>::store(ob + ((int64_t)(y0 + a) * W + x) * C, acc[a]);
    }
}

static int launch_gather(GatherP& P, const float* rois, const void* go, int dtype, void* stream) {
    if (P.N <= 0) return DGX_OK;
    if (!rois && P.R > 0) return DGX_ERR_BAD_ARG;
    if (!go && P.R > 0) return DGX_ERR_BAD_ARG;
    const int C = P.C;
    if (C <= 0 || (C & 7) || 256 % (C >> 3) || 256 / (C >> 3) > GT_TWMAX || P.ph <= 0 || P.pw <= 0 || P.ph > GT_PMAX || P.pw > GT_PMAX)
        return DGX_ERR_UNSUPPORTED;
    if (((uintptr_t)go & 15) != 0) return DGX_ERR_UNSUPPORTED;
    const int TW = 256 / (C >> 3);
    // tile height: 8 rows, or 4 when that is what it takes to give every CU a tile (a tile's workgroup walks the RoIs that reach it
    // one after the other: with the RoI heads' three levels at 1024^2 x 2 images the 32 tiles of the coarsest level take most of the
    // boxes of an untrained model each, while 650 small-level tiles finish early -- 4-row tiles halve the longest walk's work)
    int th = GT_TH;
    {
        int64_t t8 = 0;
        for (int l = 0; l < P.num_levels; ++l) t8 += (int64_t)P.N * ((P.W[l] + TW - 1) / TW) * ((P.H[l] + 7) / 8);
        if (t8 < 4 * 256) th = 4;
    }
    int total = 0;
    for (int l = 0; l < P.num_levels; ++l) {
        if (!P.out[l] || ((uintptr_t)P.out[l] & 15)) return P.out[l] ? DGX_ERR_UNSUPPORTED : DGX_ERR_BAD_ARG;
        P.tile0[l] = total;
        P.tiles_x[l] = (P.W[l] + TW - 1) / TW;
        P.tiles_y[l] = (P.H[l] + th - 1) / th;
        total += P.N * P.tiles_x[l] * P.tiles_y[l];
    }
    P.tile0[P.num_levels] = total;
    P.total = total;
    if (total <= 0) return DGX_OK;
    if (dtype == DGX_BF16) {
        if (th == 4) hipLaunchKernelGGL((roi_pool_bwd_gather_kernel<uint16_t, 4>), dim3(total), dim3(256), 0, (hipStream_t)stream, P, rois, (const uint16_t*)go);
        else hipLaunchKernelGGL((roi_pool_bwd_gather_kernel<uint16_t, 8>), dim3(total), dim3(256), 0, (hipStream_t)stream, P, rois, (const uint16_t*)go);
    } else {
        if (th == 4) hipLaunchKernelGGL((roi_pool_bwd_gather_kernel<float, 4>), dim3(total), dim3(256), 0, (hipStream_t)stream, P, rois, (const float*)go);
        else hipLaunchKernelGGL((roi_pool_bwd_gather_kernel<float, 8>), dim3(total), dim3(256), 0, (hipStream_t)stream, P, rois, (const float*)go);
    }
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

static int launch_pool(bool fwd, const PoolLevels& L, const float* rois, const void* io, int32_t* levels_out, int C,
                       int R, int ph, int pw, int sampling_ratio, int aligned, int out_nhwc, int dtype, void* stream) {
    if (R <= 0) return DGX_OK;
    if (!rois || !io || C <= 0 || ph <= 0 || pw <= 0 || ph > 64 || pw > 64) return DGX_ERR_BAD_ARG;
    const int bins = ph * pw;
    int spy = 0, spx = 0;
    for (int l = 0; l < L.num_levels; ++l) { spy = L.H[l] > spy ? L.H[l] : spy; spx = L.W[l] > spx ? L.W[l] : spx; }
    spy += 1; spx += 1;
    const size_t tbl = (size_t)(ph * spy + pw * spx) * 4 + (size_t)(ph + pw) * sizeof(Span);
    int bpb, nsplit;
    if (out_nhwc) {
        nsplit = R >= 2048 ? 1 : (R >= 512 ? 2 : 4);
        if (nsplit > bins) nsplit = bins;
        bpb = (bins + nsplit - 1) / nsplit;
    } else {
        if (tbl + (size_t)C * 4 > 64 * 1024) return DGX_ERR_UNSUPPORTED;
        bpb = (int)((64 * 1024 - tbl) / ((size_t)C * 4));
        if (bpb > bins) bpb = bins;
    }
    nsplit = (bins + bpb - 1) / bpb;
    const size_t sm = tbl + (out_nhwc ? 0 : (size_t)bpb * C * 4);
    if (sm > 64 * 1024) return DGX_ERR_UNSUPPORTED;
    dim3 grid(R, nsplit), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (out_nhwc && (C & 7) == 0 && C <= 2048 && 256 % (C >> 3) == 0) {   // 8 channels per lane, several bins in flight
        bool aligned16 = true;
        for (int l = 0; l < L.num_levels; ++l)
            aligned16 = aligned16 && (((uintptr_t)(fwd ? L.feat[l] : (const void*)L.grad[l]) & 15) == 0);
        if (aligned16 && ((uintptr_t)io & 15) == 0) {
#define VEC_LAUNCH(TT, BW) hipLaunchKernelGGL((roi_align_vec_kernel<TT, BW>), grid, block, tbl, st, L, rois, (TT*)io, fwd ? levels_out : nullptr, C, ph, pw, sampling_ratio, aligned, bpb, spy, spx)
            if (fwd) { if (dtype == DGX_BF16) VEC_LAUNCH(uint16_t, false); else VEC_LAUNCH(float, false); }
            else { if (dtype == DGX_BF16) VEC_LAUNCH(uint16_t, true); else VEC_LAUNCH(float, true); }
#undef VEC_LAUNCH
            DGX_LAUNCH_CHECK();
            return DGX_OK;
        }
    }
    if (fwd) {
        if (dtype == DGX_BF16)
            hipLaunchKernelGGL((roi_align_sep_kernel<uint16_t, false>), grid, block, sm, st, L, rois, (uint16_t*)io, levels_out,
                               C, ph, pw, sampling_ratio, aligned, out_nhwc, bpb, spy, spx);
        else
            hipLaunchKernelGGL((roi_align_sep_kernel<float, false>), grid, block, sm, st, L, rois, (float*)io, levels_out, C,
                               ph, pw, sampling_ratio, aligned, out_nhwc, bpb, spy, spx);
    } else {
        if (dtype == DGX_BF16)
            hipLaunchKernelGGL((roi_align_sep_kernel<uint16_t, true>), grid, block, sm, st, L, rois, (uint16_t*)io, nullptr, C,
                               ph, pw, sampling_ratio, aligned, out_nhwc, bpb, spy, spx);
        else
            hipLaunchKernelGGL((roi_align_sep_kernel<float, true>), grid, block, sm, st, L, rois, (float*)io, nullptr, C, ph,
                               pw, sampling_ratio, aligned, out_nhwc, bpb, spy, spx);
    }
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_roi_align_fwd(const void* feat, const float* rois, void* out, int N, int H, int W, int C, int R,
                                 float spatial_scale, int ph, int pw, int sampling_ratio, int aligned, int out_nhwc,
                                 int dtype, void* stream) {
    (void)N;
    PoolLevels L = {};
    L.feat[0] = feat; L.H[0] = H; L.W[0] = W; L.scale[0] = spatial_scale; L.num_levels = 1; L.min_level = 0;
    if (R > 0 && !feat) return DGX_ERR_BAD_ARG;
    return launch_pool(true, L, rois, out, nullptr, C, R, ph, pw, sampling_ratio, aligned, out_nhwc, dtype, stream);
}

extern "C" int dgx_roi_align_bwd(const void* grad_out, const float* rois, float* grad_feat, int N, int H, int W, int C,
                                 int R, float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                                 int out_nhwc, int dtype, void* stream) {
    (void)N;
    PoolLevels L = {};
    L.grad[0] = grad_feat; L.H[0] = H; L.W[0] = W; L.scale[0] = spatial_scale; L.num_levels = 1; L.min_level = 0;
    if (R > 0 && !grad_feat) return DGX_ERR_BAD_ARG;
    return launch_pool(false, L, rois, grad_out, nullptr, C, R, ph, pw, sampling_ratio, aligned, out_nhwc, dtype, stream);
}

static int fill_levels(PoolLevels& L, const void* const* feats, float* const* grads, const int* Hs, const int* Ws,
                       int num_levels, int min_level) {
    if (num_levels < 1 || num_levels > MAX_LEVELS || !Hs || !Ws) return DGX_ERR_BAD_ARG;
    L.num_levels = num_levels;
    L.min_level = min_level;
    for (int l = 0; l < num_levels; ++l) {
        L.feat[l] = feats ? feats[l] : nullptr;
        L.grad[l] = grads ? grads[l] : nullptr;
        L.H[l] = Hs[l];
        L.W[l] = Ws[l];
        L.scale[l] = 1.0f / (float)(1 << (min_level + l));
    }
    return DGX_OK;
}

extern "C" int dgx_roi_pooler_fwd(const void* const* feats, const int* Hs, const int* Ws, int num_levels, int min_level,
                                  const float* rois, void* out, int32_t* levels_out, int N, int C, int R, int ph, int pw,
                                  int sampling_ratio, int out_nhwc, int dtype, void* stream) {
    (void)N;
    PoolLevels L = {};
    if (!feats) return DGX_ERR_BAD_ARG;
    int e = fill_levels(L, feats, nullptr, Hs, Ws, num_levels, min_level);
    if (e) return e;
    return launch_pool(true, L, rois, out, levels_out, C, R, ph, pw, sampling_ratio, 1, out_nhwc, dtype, stream);
}

extern "C" int dgx_roi_pooler_bwd(const void* grad_out, float* const* grad_feats, const int* Hs, const int* Ws,
                                  int num_levels, int min_level, const float* rois, int N, int C, int R, int ph, int pw,
                                  int sampling_ratio, int out_nhwc, int dtype, void* stream) {
    (void)N;
    PoolLevels L = {};
    if (!grad_feats) return DGX_ERR_BAD_ARG;
    int e = fill_levels(L, nullptr, grad_feats, Hs, Ws, num_levels, min_level);
    if (e) return e;
    return launch_pool(false, L, rois, grad_out, nullptr, C, R, ph, pw, sampling_ratio, 1, out_nhwc, dtype, stream);
}

// Backward of dgx_roi_pooler_fwd / dgx_roi_align_fwd (channels-last pooled gradient) as a gather: grad_feats[l] is an
// (N,H_l,W_l,C) map in the dtype of grad_out and is OVERWRITTEN (no zero fill needed, no atomics; deterministic).
// num_levels == 1: plain ROIAlign with `spatial_scale` / `aligned`; otherwise the ROIPooler level rule with scales
// 2^-(min_level + l).  DGX_ERR_UNSUPPORTED when C % 8 != 0, 256 % (C/8) != 0, C < 64, the output side exceeds 16 or a
// pointer is not 16-byte aligned: the caller then uses the scatter forms above.
// _accum: accumulate != 0 ADDS to the maps (fp32 sum of the map's value and this pooling's contribution, rounded once): the
// gradients of several poolings of the same features (three cascade stages + the mask head) then need no separate additions.
// grad_scale multiplies this pooling's contribution (cascade_rcnn.py:20-28 _ScaleGradient sits between pooler and box head).
extern "C" int dgx_roi_pooler_bwd_gather_accum(const void* grad_out, void* const* grad_feats, const int* Hs, const int* Ws, int num_levels,
                                               int min_level, float spatial_scale, int aligned, const float* rois, int N, int C, int R,
                                               int ph, int pw, int sampling_ratio, int accumulate, float grad_scale, int dtype,
                                               void* stream) {
    if (!grad_feats || !Hs || !Ws || num_levels < 1 || num_levels > MAX_LEVELS) return DGX_ERR_BAD_ARG;
    if (dtype != DGX_BF16 && dtype != DGX_F32) return DGX_ERR_BAD_ARG;
    GatherP P = {};
    P.num_levels = num_levels; P.min_level = min_level; P.N = N; P.C = C; P.R = R < 0 ? 0 : R; P.ph = ph; P.pw = pw;
    P.sampling_ratio = sampling_ratio; P.aligned = num_levels > 1 ? 1 : aligned;
    for (int l = 0; l < num_levels; ++l) {
        P.out[l] = grad_feats[l]; P.H[l] = Hs[l]; P.W[l] = Ws[l];
        P.scale[l] = num_levels > 1 ? 1.0f / (float)(1 << (min_level + l)) : spatial_scale;
    }
    P.accumulate = accumulate != 0;
    P.gscale = grad_scale;
    return launch_gather(P, rois, grad_out, dtype, stream);
}
extern "C" int dgx_roi_pooler_bwd_gather(const void* grad_out, void* const* grad_feats, const int* Hs, const int* Ws, int num_levels,
                                         int min_level, float spatial_scale, int aligned, const float* rois, int N, int C, int R,
                                         int ph, int pw, int sampling_ratio, int dtype, void* stream) {
    return dgx_roi_pooler_bwd_gather_accum(grad_out, grad_feats, Hs, Ws, num_levels, min_level, spatial_scale, aligned, rois, N, C, R, ph, pw,
                                           sampling_ratio, 0, 1.0f, dtype, stream);
}

// ---- GT mask crop: one workgroup per box, one lane per output pixel, byte taps --------------
__global__ __launch_bounds__(256) void mask_crop_kernel(const uint8_t* __restrict__ masks, const float* __restrict__ boxes,
                                                        const int32_t* __restrict__ mask_idx, uint8_t* __restrict__ out,
                                                        int H, int W, int S) {
    // blockIdx.y = slice of the box's bins: a box with a large adaptive sample grid (ceil(extent / S) samples per bin and axis: up to
    // 37 x 37 at 1024 px) is minutes of byte taps for ONE workgroup while ~100 boxes leave more than half of the CUs without work
    const int r = blockIdx.x;
    const float roi[5] = {0.0f, boxes[4 * r], boxes[4 * r + 1], boxes[4 * r + 2], boxes[4 * r + 3]};
    const RoiGeom G = roi_geom(roi, 1.0f, S, S, 0, true);
    const uint8_t* m = masks + (int64_t)mask_idx[r] * H * W;
    for (int bin = blockIdx.y * blockDim.x + threadIdx.x; bin < S * S; bin += gridDim.y * blockDim.x) {
        const int i = bin / S, j = bin - i * S;
        float acc = 0.0f;
        for (int iy = 0; iy < G.gh; ++iy) {
            const float y = G.sh + i * G.bin_h + ((float)iy + 0.5f) * G.bin_h / (float)G.gh;
            for (int ix = 0; ix < G.gw; ++ix) {
                const float x = G.sw + j * G.bin_w + ((float)ix + 0.5f) * G.bin_w / (float)G.gw;
                Taps t;
                if (!bilinear_taps(H, W, y, x, t)) continue;
                acc += t.w[0] * (float)m[t.pos[0]] + t.w[1] * (float)m[t.pos[1]] + t.w[2] * (float)m[t.pos[2]] +
                       t.w[3] * (float)m[t.pos[3]];
            }
        }
        out[(int64_t)r * S * S + bin] = (acc / G.count) >= 0.5f ? 1 : 0;
    }
}

extern "C" int dgx_mask_crop(const uint8_t* masks, const float* boxes, const int32_t* mask_idx, uint8_t* out, int M,
                             int H, int W, int R, int S, void* stream) {
    (void)M;
    if (R <= 0) return DGX_OK;
    if (!masks || !boxes || !mask_idx || !out || S <= 0) return DGX_ERR_BAD_ARG;
    const int slices = (S * S + 255) / 256 < 8 ? (S * S + 255) / 256 : 8;
    hipLaunchKernelGGL(mask_crop_kernel, dim3(R, slices), dim3(256), 0, (hipStream_t)stream, masks, boxes, mask_idx, out, H, W, S);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
