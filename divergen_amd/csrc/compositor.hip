// Instance copy-paste compositor for gfx950 ('basic' blend), all-integer, bit-exact with the
// sequential reference (custom_build_copypaste_mapper.py:488-566, :79-92; custom_cp_method.py:5-9).
//
// The reference touches the whole (n,H,W) mask stack and the image once PER PASTE.  Here the K
// pastes are resolved in one sweep: per pixel a K-bit "cover" word says which pastes have alpha>0
// there; the image takes the top-most paste; an object's pixel dies at the first later paste that
// covers it, so per (object, death step) pixel count and extents are enough to replay the
// reference's per-step occlusion filter exactly (boxes are small integers in fp32).
//   k1 cover+blend  : HBM-bound, reads image once, writes image + cover (4 B/pixel)
//   k2 stats        : per object rows -> LDS histogram over death step -> global atomics
//   k3 resolve      : one lane per object, sequential replay of the filter over K steps
//   k4 masks        : final masks of all objects
#include "dgx_common.h"

#define CP_MAX_K 31

__global__ __launch_bounds__(256) void cp_cover_blend_kernel(uint8_t* __restrict__ image, int H, int W,
                                                             const uint8_t* __restrict__ rgba,
                                                             const int32_t* __restrict__ desc, int K,
                                                             uint32_t* __restrict__ cover) {
    __shared__ int32_t d[CP_MAX_K * 5];
    for (int i = threadIdx.x; i < K * 5; i += blockDim.x) d[i] = desc[i];
    __syncthreads();
    const int64_t HW = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        uint32_t bits = 0;
        int top = -1;
        const uint8_t* topp = nullptr;
        for (int k = 0; k < K; ++k) {
            const int h = d[5 * k + 1], w = d[5 * k + 2], sx = x - d[5 * k + 3], sy = y - d[5 * k + 4];
            if (sx >= 0 && sy >= 0 && sx < w && sy < h) {
                const uint8_t* px = rgba + d[5 * k] + 4 * ((int64_t)sy * w + sx);
                if (px[3] > 0) { bits |= 1u << k; top = k; topp = px; }
            }
        }
        cover[p] = bits;
        if (top >= 0) {
            image[p] = topp[0];
            image[HW + p] = topp[1];
            image[2 * HW + p] = topp[2];
        }
    }
}

// stats[obj][t][5] = {count, minx, maxx, miny, maxy} over the object's pixels that die at step t
// (t == K: never).  obj < n0: original mask, born before paste 0; obj = n0 + j: paste j's footprint.
__global__ __launch_bounds__(256) void cp_stats_kernel(const uint8_t* __restrict__ masks, const uint32_t* __restrict__ cover,
                                                       int n0, int H, int W, int K, int rows_per_block,
                                                       int32_t* __restrict__ stats) {
    __shared__ int32_t s[(CP_MAX_K + 1) * 5];
    const int obj = blockIdx.x;
    for (int i = threadIdx.x; i < (K + 1) * 5; i += blockDim.x) {
        const int f = i % 5;
        s[i] = (f == 0 || f == 2 || f == 4) ? (f == 0 ? 0 : -1) : 0x7fffffff;
    }
    __syncthreads();
    const int y0 = blockIdx.y * rows_per_block, y1 = min(H, y0 + rows_per_block);
    const int born = obj < n0 ? -1 : obj - n0;
    const uint32_t later = born + 1 >= 32 ? 0u : (0xffffffffu << (born + 1));
    for (int64_t p = (int64_t)y0 * W + threadIdx.x; p < (int64_t)y1 * W; p += blockDim.x) {
        const uint32_t cv = cover[p];
        const bool in = obj < n0 ? masks[(int64_t)obj * H * W + p] != 0 : ((cv >> born) & 1u) != 0;
        if (!in) continue;
        const uint32_t lb = cv & later;
        const int t = lb ? __ffs((int)lb) - 1 : K;
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        atomicAdd(&s[5 * t], 1);
        atomicMin(&s[5 * t + 1], x);
        atomicMax(&s[5 * t + 2], x);
        atomicMin(&s[5 * t + 3], y);
        atomicMax(&s[5 * t + 4], y);
    }
    __syncthreads();
    int32_t* g = stats + (int64_t)obj * (K + 1) * 5;
    for (int i = threadIdx.x; i < (K + 1) * 5; i += blockDim.x) {
        const int f = i % 5;
        if (s[5 * (i / 5)] == 0) continue;
        if (f == 0) atomicAdd(&g[i], s[i]);
        else if (f == 1 || f == 3) atomicMin(&g[i], s[i]);
        else atomicMax(&g[i], s[i]);
    }
}

__global__ void cp_stats_init_kernel(int32_t* stats, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % 5);
        stats[i] = f == 0 ? 0 : ((f == 1 || f == 3) ? 0x7fffffff : -1);
    }
}

// Replay of _copy_paste's filter: at step k an object present in the list gets the box of its
// remaining mask (zeros if empty); it stays iff all |box - previous box| <= 10 or remaining area > 300.
__global__ void cp_resolve_kernel(const int32_t* __restrict__ stats, const float* __restrict__ boxes0, int n0, int K,
                                  float* __restrict__ out_boxes, uint8_t* __restrict__ out_valid) {
    const int obj = blockIdx.x * blockDim.x + threadIdx.x;
    if (obj >= n0 + K) return;
    const int32_t* s = stats + (int64_t)obj * (K + 1) * 5;
    const int born = obj < n0 ? -1 : obj - n0;
    float prev[4];
    if (obj < n0) {
        for (int i = 0; i < 4; ++i) prev[i] = boxes0[4 * obj + i];
    } else {  // tight box of the paste's footprint at birth (get_bboxes of the translated mask)
        int cnt = 0, x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1;
        for (int t = 0; t <= K; ++t) {
            cnt += s[5 * t];
            x0 = min(x0, s[5 * t + 1]); x1 = max(x1, s[5 * t + 2]);
            y0 = min(y0, s[5 * t + 3]); y1 = max(y1, s[5 * t + 4]);
        }
        if (cnt > 0) { prev[0] = (float)x0; prev[1] = (float)y0; prev[2] = (float)(x1 + 1); prev[3] = (float)(y1 + 1); }
        else prev[0] = prev[1] = prev[2] = prev[3] = 0.0f;
    }
    bool valid = true;
    for (int k = born + 1; k < K && valid; ++k) {
        int cnt = 0, x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1;
        for (int t = k + 1; t <= K; ++t) {
            cnt += s[5 * t];
            x0 = min(x0, s[5 * t + 1]); x1 = max(x1, s[5 * t + 2]);
            y0 = min(y0, s[5 * t + 3]); y1 = max(y1, s[5 * t + 4]);
        }
        float cur[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (cnt > 0) { cur[0] = (float)x0; cur[1] = (float)y0; cur[2] = (float)(x1 + 1); cur[3] = (float)(y1 + 1); }
        bool box_ok = true;
        for (int i = 0; i < 4; ++i) box_ok = box_ok && fabsf(cur[i] - prev[i]) <= 10.0f;
        valid = box_ok || cnt > 300;
        for (int i = 0; i < 4; ++i) prev[i] = cur[i];
    }
    for (int i = 0; i < 4; ++i) out_boxes[4 * obj + i] = prev[i];
    out_valid[obj] = valid ? 1 : 0;
}

__global__ __launch_bounds__(256) void cp_masks_kernel(const uint8_t* __restrict__ masks, const uint32_t* __restrict__ cover,
                                                       int n0, int H, int W, int K, uint8_t* __restrict__ out_masks) {
    const int obj = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int born = obj < n0 ? -1 : obj - n0;
    const uint32_t later = born + 1 >= 32 ? 0u : (0xffffffffu << (born + 1));
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t cv = cover[p];
        const bool in = obj < n0 ? masks[(int64_t)obj * HW + p] != 0 : ((cv >> born) & 1u) != 0;
        out_masks[(int64_t)obj * HW + p] = (in && !(cv & later)) ? (obj < n0 ? masks[(int64_t)obj * HW + p] : 1) : 0;
    }
}

extern "C" int dgx_copy_paste(uint8_t* image, const uint8_t* masks, const float* boxes0, int n0, int H, int W,
                              const uint8_t* src_rgba, const int32_t* src_desc, int K, uint8_t* out_masks,
                              float* out_boxes, uint8_t* out_valid, int32_t* stats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (K <= 0 || H <= 0 || W <= 0) return K < 0 ? DGX_ERR_BAD_ARG : DGX_OK;
    if (K > CP_MAX_K) return DGX_ERR_UNSUPPORTED;
    if (!image || !src_rgba || !src_desc || !out_masks || !out_boxes || !out_valid || !stats || n0 < 0 ||
        (n0 > 0 && (!masks || !boxes0)))
        return DGX_ERR_BAD_ARG;
    // cover words live at the tail of the stats workspace: (n0+K)*(K+1)*5 ints, then H*W words
    const int64_t ns = (int64_t)(n0 + K) * (K + 1) * 5;
    uint32_t* cover = reinterpret_cast<uint32_t*>(stats + ns);
    const int64_t HW = (int64_t)H * W;
    const int gp = (int)((HW + 255) / 256 < 4096 ? (HW + 255) / 256 : 4096);
    hipLaunchKernelGGL(cp_stats_init_kernel, dim3((int)((ns + 255) / 256)), dim3(256), 0, st, stats, ns);
    hipLaunchKernelGGL(cp_cover_blend_kernel, dim3(gp), dim3(256), 0, st, image, H, W, src_rgba, src_desc, K, cover);
    const int rpb = 32;
    hipLaunchKernelGGL(cp_stats_kernel, dim3(n0 + K, (H + rpb - 1) / rpb), dim3(256), 0, st, masks, cover, n0, H, W, K,
                       rpb, stats);
    hipLaunchKernelGGL(cp_resolve_kernel, dim3((n0 + K + 63) / 64), dim3(64), 0, st, stats, boxes0, n0, K, out_boxes,
                       out_valid);
    hipLaunchKernelGGL(cp_masks_kernel, dim3(gp, n0 + K), dim3(256), 0, st, masks, cover, n0, H, W, K, out_masks);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
