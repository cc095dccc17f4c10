// Evaluation post-processing: paste the SxS mask probabilities of each detection into the image
// (D2/layers/mask_ops.py:17-150 `_do_paste_mask` / `paste_masks_in_image`: F.grid_sample bilinear, align_corners=False, zero
// padding, then >= threshold) and run-length encode the result in COCO order (column-major, first run = zeros), WITHOUT ever
// materialising the N x H x W probability tensor the reference builds (1.3 GB per 300 detections at 800x1333).
//   dgx_paste_masks : (N,S,S) f32 + boxes -> u8 (N,H,W) binary masks      (API parity with paste_masks_in_image)
//   dgx_paste_rle   : the same bits evaluated on the fly, emitted as run lengths per detection (what LVIS results need)
// The float sequence of the sample coordinates and of grid_sample's bilinear weights is torch's (compiled with
// -ffp-contract=off); pixels whose interpolated value sits within an ulp of the threshold may still differ from a CPU build
// that contracts multiply-adds.
#include "dgx_common.h"

namespace {
struct PasteBox { float x0, y0, x1, y1; };

// One axis of the sample: the two taps (index of the lower one, their weights) for pixel centre p + 0.5 through a box side
// [lo, hi) onto S cells.  The float sequence is the reference's (mask_ops.py:51-54) followed by ATen's
// grid_sampler_unnormalize (align_corners=False) and the bilinear corner weights.
struct AxisTap { float w0, w1; int c; };
__device__ __forceinline__ AxisTap axis_tap(int p, float lo, float hi, int S) {
    const float g = ((float)p + 0.5f - lo) / (hi - lo) * 2.0f - 1.0f;
    const float i = ((g + 1.0f) * (float)S - 1.0f) / 2.0f;
    const float f = floorf(i);
    AxisTap t;
    t.w1 = i - f;                  // weight of the upper tap  (ix - ix_nw)
    t.w0 = (f + 1.0f) - i;         // weight of the lower tap  (ix_se - ix)
    // saturate: anything outside [-1, S) has no tap inside the map (NaN -> no tap either)
    t.c = (f >= -1.0f && f < (float)S) ? (int)f : -2;
    return t;
}

// value of the pasted probability map at one pixel from its two axis taps; m = the S x S map (LDS)
__device__ __forceinline__ float paste_value(const float* m, int S, AxisTap tx, AxisTap ty) {
    const bool xin0 = tx.c >= 0 && tx.c < S, xin1 = tx.c + 1 >= 0 && tx.c + 1 < S;
    const bool yin0 = ty.c >= 0 && ty.c < S, yin1 = ty.c + 1 >= 0 && ty.c + 1 < S;
    const float nw = tx.w0 * ty.w0, ne = tx.w1 * ty.w0, sw = tx.w0 * ty.w1, se = tx.w1 * ty.w1;
    float out = 0.0f;
    if (yin0 && xin0) out += m[ty.c * S + tx.c] * nw;
    if (yin0 && xin1) out += m[ty.c * S + tx.c + 1] * ne;
    if (yin1 && xin0) out += m[(ty.c + 1) * S + tx.c] * sw;
    if (yin1 && xin1) out += m[(ty.c + 1) * S + tx.c + 1] * se;
    return out;
}

// pixels [a, b) along one axis that can see the map: the box side widened by one map cell (+1 px of rounding slack)
__device__ __forceinline__ void live_range(float lo, float hi, int S, int size, int& a, int& b) {
    const float cell = fabsf(hi - lo) / (float)S;
    a = (int)fmaxf(floorf(fminf(lo, hi) - cell) - 1.0f, 0.0f);
    b = (int)fminf(ceilf(fmaxf(lo, hi) + cell) + 1.0f, (float)size);
    if (!(hi - lo == hi - lo) || !(cell < 1e30f)) { a = 0; b = size; }      // NaN / inf box: no pruning
    if (b < a) b = a;
}
constexpr int PASTE_ROWS = 16;
}  // namespace

// grid (ceil(W/256), ceil(H/16), N): a 256-column x 16-row tile of detection n per workgroup.  Tiles that cannot see the map
// are all zero: written as such, or skipped when the caller has already cleared `out` (prezeroed).
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                          uint8_t* __restrict__ out, int S, int H, int W, float thr,
                                                          int prezeroed) {
    extern __shared__ float ms[];
    __shared__ AxisTap ytap[PASTE_ROWS];
    const int n = blockIdx.z, y0 = blockIdx.y * PASTE_ROWS, x = blockIdx.x * 256 + threadIdx.x;
    const int rows = min(PASTE_ROWS, H - y0);
    const PasteBox b = {boxes[4 * n], boxes[4 * n + 1], boxes[4 * n + 2], boxes[4 * n + 3]};
    int xa, xb, ya, yb;
    live_range(b.x0, b.x1, S, W, xa, xb);
    live_range(b.y0, b.y1, S, H, ya, yb);
    uint8_t* o = out + ((int64_t)n * H + y0) * W + x;
    const bool dead = (int)(blockIdx.x * 256) >= xb || (int)(blockIdx.x * 256 + 256) <= xa || y0 >= yb || y0 + rows <= ya;
    if (dead) {
        if (!prezeroed && x < W)
            for (int r = 0; r < rows; ++r) o[(int64_t)r * W] = 0;
        return;
    }
    for (int i = threadIdx.x; i < S * S; i += 256) ms[i] = masks[(int64_t)n * S * S + i];
    if (threadIdx.x < rows) ytap[threadIdx.x] = axis_tap(y0 + threadIdx.x, b.y0, b.y1, S);
    __syncthreads();
    if (x >= W) return;
    const AxisTap tx = axis_tap(x, b.x0, b.x1, S);
    for (int r = 0; r < rows; ++r) o[(int64_t)r * W] = paste_value(ms, S, tx, ytap[r]) >= thr ? 1 : 0;
}

// One workgroup (1024 threads) per detection.  The sequence b[i], i = x*H + y (column-major) is scanned in tiles of 8192
// over the columns that can see the map; positions of value changes are compacted in order, then differenced into run
// lengths in place.
// FROM_BITS: the sequence is read from an existing (N,H,W) u8 bitmask instead (dgx_rle_encode); `masks` then points at it.
template <bool FROM_BITS>
__global__ __launch_bounds__(1024) void paste_rle_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ nruns, int S,
                                                         int H, int W, float thr, int cap) {
    extern __shared__ float ms[];
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!FROM_BITS)
        for (int i = tid; i < S * S; i += 1024) ms[i] = masks[(int64_t)n * S * S + i];
    if (tid == 0) base_s = 0;
    PasteBox b = {0.f, 0.f, 0.f, 0.f};
    int xa = 0, xb = W;
    if (!FROM_BITS) {
        b = {boxes[4 * n], boxes[4 * n + 1], boxes[4 * n + 2], boxes[4 * n + 3]};
        live_range(b.x0, b.x1, S, W, xa, xb);
    }
    const uint8_t* bits = reinterpret_cast<const uint8_t*>(masks) + (int64_t)n * H * W;
    int32_t* pos = counts + (int64_t)n * cap;       // first: positions of changes, then turned into run lengths
    __syncthreads();
    const int64_t i0 = (int64_t)xa * H, i1 = (int64_t)xb * H, total = (int64_t)H * W;
    int curx = -1;
    AxisTap tx = {0.f, 0.f, -2};
    auto bit_xy = [&](int x, int y) -> int {
        if (FROM_BITS) return bits[(int64_t)y * W + x] != 0;
        if (x != curx) { tx = axis_tap(x, b.x0, b.x1, S); curx = x; }
        return paste_value(ms, S, tx, axis_tap(y, b.y0, b.y1, S)) >= thr ? 1 : 0;
    };
    for (int64_t t0 = i0; t0 < i1; t0 += 8192) {
        const int64_t s = t0 + 8 * tid;
        int flags = 0, cnt = 0;
        if (s < i1) {
            int x = (int)(s / H), y = (int)(s - (int64_t)x * H);
            int prev = 0;
            if (s > i0) prev = y > 0 ? bit_xy(x, y - 1) : bit_xy(x - 1, H - 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int v = (s + k < i1) ? bit_xy(x, y) : prev;
                if (v != prev) { flags |= 1 << k; ++cnt; }
                prev = v;
                if (++y == H) { y = 0; ++x; }
            }
        }
        int inc = cnt;                               // inclusive scan inside the wave, then across the 16 waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int w = base_s + inc - cnt;
        for (int j = 0; j < wv; ++j) w += wave_tot[j];
        for (int k = 0; k < 8; ++k)
            if (flags & (1 << k)) { if (w < cap) pos[w] = (int32_t)(s + k); ++w; }
        __syncthreads();
        if (tid == 1023) base_s = w;
        __syncthreads();
    }
    // a set bit at the very end of the scanned range closes there (bits outside the range are zero); at the end of the
    // image the last run simply ends with it
    if (tid == 0 && i1 > i0 && i1 < total && bit_xy(xb - 1, H - 1)) {
        if (base_s < cap) pos[base_s] = (int32_t)i1;
        base_s += 1;
    }
    __syncthreads();
    const int T = base_s;                   // number of value changes; runs = T + 1 (COCO: first run counts zeros)
    const int Tc = T < cap - 1 ? T : cap - 1;
    // difference in place: chunks of 1024 from the back, so the pos[k-1] a chunk reads is still a position
    for (int c0 = (Tc / 1024) * 1024; c0 >= 0; c0 -= 1024) {
        const int k = c0 + tid;
        int32_t v = 0;
        if (k <= Tc) {
            const int64_t hi = k < T ? pos[k] : total, lo = k > 0 ? pos[k - 1] : 0;
            v = (int32_t)(hi - lo);
        }
        __syncthreads();
        if (k <= Tc) pos[k] = v;
        __syncthreads();
    }
    if (tid == 0) nruns[n] = T + 1 <= cap ? T + 1 : -(T + 1);       // negative: did not fit `cap`
}

extern "C" int dgx_paste_masks(const float* masks, const float* boxes, uint8_t* out, int N, int S, int H, int W,
                               float threshold, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!masks || !boxes || !out || S <= 0 || S > 112 || H > 65535) return DGX_ERR_BAD_ARG;
    // clearing 1 byte/pixel at copy-engine speed and then touching only the tiles a box can reach beats computing zeros
    const hipError_t me = hipMemsetAsync(out, 0, (size_t)N * H * W, (hipStream_t)stream);
    if (me != hipSuccess) return -(int)me - 1000;
    hipLaunchKernelGGL(paste_masks_kernel, dim3((W + 255) / 256, (H + PASTE_ROWS - 1) / PASTE_ROWS, N), dim3(256),
                       (size_t)S * S * 4, (hipStream_t)stream, masks, boxes, out, S, H, W, threshold, 1);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_paste_rle(const float* masks, const float* boxes, int32_t* counts, int32_t* nruns, int N, int S, int H,
                             int W, float threshold, int cap, void* stream) {
    if (N <= 0) return DGX_OK;
    if (!masks || !boxes || !counts || !nruns || S <= 0 || S > 112 || H <= 0 || W <= 0 || cap < 2 ||
        (int64_t)H * W >= (1ll << 31))
        return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(paste_rle_kernel<false>, dim3(N), dim3(1024), (size_t)S * S * 4, (hipStream_t)stream, masks, boxes, counts,
                       nruns, S, H, W, threshold, cap);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_rle_encode(const uint8_t* bits, int32_t* counts, int32_t* nruns, int N, int H, int W, int cap,
                              void* stream) {
    if (N <= 0) return DGX_OK;
    if (!bits || !counts || !nruns || H <= 0 || W <= 0 || cap < 2 || (int64_t)H * W >= (1ll << 31)) return DGX_ERR_BAD_ARG;
    hipLaunchKernelGGL(paste_rle_kernel<true>, dim3(N), dim3(1024), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(bits), (const float*)nullptr, counts, nruns, 0, H, W, 0.5f, cap);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// Host side of the results writer: the COCO "compressed RLE" string of one run-length list (what pycocotools' rleToString
// emits: 5 bits per character + continuation bit, counts from the 4th on stored as a difference against two back).
// Returns the string length, or -(needed) when `cap` is too small.
extern "C" int64_t dgx_rle_to_string(const int32_t* counts, int64_t n, char* out, int64_t cap) {
    int64_t p = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t x = counts[i];
        if (i > 2) x -= counts[i - 2];
        bool more = true;
        while (more) {
            int c = (int)(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            if (p < cap) out[p] = (char)(c + 48);
            ++p;
        }
    }
    return p <= cap ? p : -p;
}
