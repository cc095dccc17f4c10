// Instance copy-paste compositor for gfx950 ('basic' blend), all-integer, bit-exact with the
// sequential reference (custom_build_copypaste_mapper.py:488-566, :79-92; custom_cp_method.py:5-9).
//
// The reference touches the whole (n,H,W) mask stack and the image once PER PASTE.  Here the K
// pastes are resolved in one sweep: per pixel a K-bit "cover" word says which pastes have alpha>0
// there; the image takes the top-most paste; an object's pixel dies at the first later paste that
// covers it, so per (object, death step) pixel count and extents are enough to replay the
// reference's per-step occlusion filter exactly (boxes are small integers in fp32).
//   k1 cover+blend  : HBM-bound, reads image once, writes image + cover (4 B/pixel)
//   k2 stats        : per object rows, 16 pixels per lane -> runs folded in registers -> LDS histogram over death step -> global atomics
//   k3 resolve      : statistics staged in LDS, one lane per object: suffixes over the death step, then the replay of the filter
//   k4 masks        : final masks of all objects, 16 pixels per lane, the cover words read once per pixel group
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
// One workgroup = (object, block of rows); a lane owns CPX = 16 consecutive pixels of one row per trip: ONE 64-byte read of
// the cover words and ONE 16-byte read of the mask (round 2 read a byte and a word per lane: 1/16 of the bytes per
// request), and pixels that die at the same step are folded in registers before the LDS histogram sees them (a run of a
// row almost always shares its death step, so the LDS atomics drop ~16x).  A paste's footprint lies inside its rectangle:
// row blocks outside it leave at once.
constexpr int CPX = 16;
struct CpRun { int t, cnt, x0, x1; };
__device__ __forceinline__ void cp_flush(int32_t* s, const CpRun& r, int y) {
    if (r.cnt == 0) return;
    atomicAdd(&s[5 * r.t], r.cnt);
    atomicMin(&s[5 * r.t + 1], r.x0);
    atomicMax(&s[5 * r.t + 2], r.x1);
    atomicMin(&s[5 * r.t + 3], y);
    atomicMax(&s[5 * r.t + 4], y);
}
__global__ __launch_bounds__(256) void cp_stats_kernel(const uint8_t* __restrict__ masks, const uint32_t* __restrict__ cover,
                                                       const int32_t* __restrict__ desc, int n0, int H, int W, int K,
                                                       int rows_per_block, int32_t* __restrict__ stats) {
    __shared__ int32_t s[(CP_MAX_K + 1) * 5];
    const int obj = blockIdx.x;
    int y0 = blockIdx.y * rows_per_block, y1 = min(H, y0 + rows_per_block);
    const int born = obj < n0 ? -1 : obj - n0;
    int xlo = 0, xhi = W;
    if (born >= 0) {                               // the paste's rectangle: [dx, dx + w) x [dy, dy + h)
        const int h = desc[5 * born + 1], w = desc[5 * born + 2], dx = desc[5 * born + 3], dy = desc[5 * born + 4];
        y0 = max(y0, dy); y1 = min(y1, dy + h);
        xlo = max(0, dx); xhi = min(W, dx + w);
        if (y0 >= y1 || xlo >= xhi) return;
    }
    for (int i = threadIdx.x; i < (K + 1) * 5; i += blockDim.x) {
        const int f = i % 5;
        s[i] = (f == 0 || f == 2 || f == 4) ? (f == 0 ? 0 : -1) : 0x7fffffff;
    }
    __syncthreads();
    const uint32_t later = born + 1 >= 32 ? 0u : (0xffffffffu << (born + 1));
    const int c0 = xlo / CPX, c1 = (xhi + CPX - 1) / CPX, ncol = c1 - c0;      // chunk columns that can hold the object
    const int64_t HW = (int64_t)H * W;
    const bool vec = (W % CPX) == 0;               // rows (and every object's plane) start on 16-byte boundaries
    const int total = (y1 - y0) * ncol;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int y = y0 + i / ncol, x = (c0 + i % ncol) * CPX;
        const int64_t p = (int64_t)y * W + x;
        uint32_t cv[CPX];
        uint8_t mk[CPX];
        if (vec) {
#pragma unroll
            for (int q = 0; q < CPX / 4; ++q) {
                const uint4 v = reinterpret_cast<const uint4*>(cover + p)[q];
                cv[4 * q] = v.x; cv[4 * q + 1] = v.y; cv[4 * q + 2] = v.z; cv[4 * q + 3] = v.w;
            }
            if (obj < n0) {
                const uint4 m = *reinterpret_cast<const uint4*>(masks + (int64_t)obj * HW + p);
                const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int q = 0; q < CPX; ++q) mk[q] = (uint8_t)(mw[q >> 2] >> (8 * (q & 3)));
            }
        } else {
#pragma unroll
            for (int q = 0; q < CPX; ++q) {
                const bool inrow = x + q < W;
                cv[q] = inrow ? cover[p + q] : 0u;
                mk[q] = (inrow && obj < n0) ? masks[(int64_t)obj * HW + p + q] : 0;
            }
        }
        CpRun r = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < CPX; ++q) {
            const bool in = obj < n0 ? mk[q] != 0 : ((cv[q] >> born) & 1u) != 0;     // (pixels beyond W carry cover 0, mask 0)
            if (!in) continue;
            const uint32_t lb = cv[q] & later;
            const int t = lb ? __ffs((int)lb) - 1 : K;
            if (r.cnt && t != r.t) { cp_flush(s, r, y); r.cnt = 0; }
            if (r.cnt == 0) { r.t = t; r.x0 = x + q; }
            r.x1 = x + q;
            ++r.cnt;
        }
        cp_flush(s, r, y);
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
// The remaining mask after step k = the pixels that die LATER than k: suffix sums / extents over the death step.  A workgroup
// copies the statistics of its CP_RES_OBJ objects into LDS with coalesced loads, each lane turns its object's rows into
// suffixes in place (K steps) and replays the filter from them (K steps); round 2 re-summed the suffix from global memory
// at every step, one dependent load after the other (K^2 / 2 x 5 loads per lane: 156 us for 29 objects).
constexpr int CP_RES_OBJ = 64;
__global__ __launch_bounds__(CP_RES_OBJ) void cp_resolve_kernel(const int32_t* __restrict__ stats, const float* __restrict__ boxes0, int n0,
                                                               int K, float* __restrict__ out_boxes, uint8_t* __restrict__ out_valid) {
    extern __shared__ int32_t rs[];                // [CP_RES_OBJ][(K + 1) * 5 + 1] (odd row stride: lanes on distinct banks)
    const int nobj = n0 + K, per = (K + 1) * 5, rstride = per | 1;
    const int o0 = blockIdx.x * CP_RES_OBJ, cnt_obj = min(CP_RES_OBJ, nobj - o0);
    for (int i = threadIdx.x; i < cnt_obj * per; i += CP_RES_OBJ) rs[(i / per) * rstride + i % per] = stats[(int64_t)o0 * per + i];
    __syncthreads();
    const int obj = o0 + threadIdx.x;
    if (obj >= nobj) return;
    int32_t* s = rs + threadIdx.x * rstride;
    for (int t = K - 1; t >= 0; --t) {             // s[t] <- statistics of the pixels that die at step >= t
        s[5 * t] += s[5 * t + 5];
        s[5 * t + 1] = min(s[5 * t + 1], s[5 * t + 6]);
        s[5 * t + 2] = max(s[5 * t + 2], s[5 * t + 7]);
        s[5 * t + 3] = min(s[5 * t + 3], s[5 * t + 8]);
        s[5 * t + 4] = max(s[5 * t + 4], s[5 * t + 9]);
    }
    const int born = obj < n0 ? -1 : obj - n0;
    auto box_of = [&](int t, float (&b)[4]) -> int {      // tight box of the pixels that die at step >= t (zeros if none)
        const int cnt = s[5 * t];
        if (cnt > 0) { b[0] = (float)s[5 * t + 1]; b[1] = (float)s[5 * t + 3]; b[2] = (float)(s[5 * t + 2] + 1); b[3] = (float)(s[5 * t + 4] + 1); }
        else b[0] = b[1] = b[2] = b[3] = 0.0f;
        return cnt;
    };
    float prev[4];
    if (obj < n0) {
        for (int i = 0; i < 4; ++i) prev[i] = boxes0[4 * obj + i];
    } else {
        box_of(0, prev);                           // tight box of the paste's footprint at birth (get_bboxes of the translated mask)
    }
    bool valid = true;
    for (int k = born + 1; k < K && valid; ++k) {
        float cur[4];
        const int cnt = box_of(k + 1, cur);
        bool box_ok = true;
        for (int i = 0; i < 4; ++i) box_ok = box_ok && fabsf(cur[i] - prev[i]) <= 10.0f;
        valid = box_ok || cnt > 300;
        for (int i = 0; i < 4; ++i) prev[i] = cur[i];
    }
    for (int i = 0; i < 4; ++i) out_boxes[4 * obj + i] = prev[i];
    out_valid[obj] = valid ? 1 : 0;
}

// Final masks of all objects: a lane owns 16 consecutive pixels, reads their cover words ONCE and walks the objects of its
// group (grid.y): 16-byte mask reads / writes (round 2: one byte per lane per object, the cover word re-read per object).
__global__ __launch_bounds__(256) void cp_masks_kernel(const uint8_t* __restrict__ masks, const uint32_t* __restrict__ cover,
                                                       int n0, int H, int W, int K, int obj_per_group,
                                                       uint8_t* __restrict__ out_masks) {
    const int64_t HW = (int64_t)H * W;
    const int nobj = n0 + K;
    const int oa = blockIdx.y * obj_per_group, ob = min(nobj, oa + obj_per_group);
    const bool vec = (HW % CPX) == 0;
    const int64_t nchunk = (HW + CPX - 1) / CPX;
    for (int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ci < nchunk; ci += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = ci * CPX;
        uint32_t cv[CPX];
        if (vec) {
#pragma unroll
            for (int q = 0; q < CPX / 4; ++q) {
                const uint4 v = reinterpret_cast<const uint4*>(cover + p)[q];
                cv[4 * q] = v.x; cv[4 * q + 1] = v.y; cv[4 * q + 2] = v.z; cv[4 * q + 3] = v.w;
            }
            uint32_t zero = 0;                     // bytes of the chunk no paste covers: an original mask survives exactly there
#pragma unroll
            for (int q = 0; q < CPX; ++q) zero |= cv[q] ? 0u : (1u << q);
            for (int obj = oa; obj < ob; ++obj) {
                uint32_t ow[4] = {0u, 0u, 0u, 0u};
                if (obj < n0) {
                    const uint4 m = *reinterpret_cast<const uint4*>(masks + (int64_t)obj * HW + p);
                    const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                    for (int q = 0; q < CPX; ++q)
                        if ((zero >> q) & 1u) ow[q >> 2] |= mw[q >> 2] & (0xffu << (8 * (q & 3)));
                } else {
                    const int born = obj - n0;
                    const uint32_t later = born + 1 >= 32 ? 0u : (0xffffffffu << (born + 1));
#pragma unroll
                    for (int q = 0; q < CPX; ++q)
                        if (((cv[q] >> born) & 1u) && !(cv[q] & later)) ow[q >> 2] |= 1u << (8 * (q & 3));
                }
                *reinterpret_cast<uint4*>(out_masks + (int64_t)obj * HW + p) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        } else {
            for (int q = 0; q < CPX && p + q < HW; ++q) {
                const uint32_t c = cover[p + q];
                for (int obj = oa; obj < ob; ++obj) {
                    const int born = obj < n0 ? -1 : obj - n0;
                    const uint32_t later = born + 1 >= 32 ? 0u : (0xffffffffu << (born + 1));
                    const uint8_t mv = obj < n0 ? masks[(int64_t)obj * HW + p + q] : 1;
                    const bool in = obj < n0 ? mv != 0 : ((c >> born) & 1u) != 0;
                    out_masks[(int64_t)obj * HW + p + q] = (in && !(c & later)) ? mv : 0;
                }
            }
        }
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
    // cover words live at the tail of the stats workspace: (n0+K)*(K+1)*5 ints, rounded up to a 16-byte boundary, then H*W words
    const int64_t ns = (int64_t)(n0 + K) * (K + 1) * 5;
    if ((uintptr_t)stats & 15) return DGX_ERR_BAD_ARG;
    uint32_t* cover = reinterpret_cast<uint32_t*>(stats + ((ns + 3) & ~(int64_t)3));
    const int64_t HW = (int64_t)H * W;
    const int gp = (int)((HW + 255) / 256 < 4096 ? (HW + 255) / 256 : 4096);
    hipLaunchKernelGGL(cp_stats_init_kernel, dim3((int)((ns + 255) / 256)), dim3(256), 0, st, stats, ns);
    hipLaunchKernelGGL(cp_cover_blend_kernel, dim3(gp), dim3(256), 0, st, image, H, W, src_rgba, src_desc, K, cover);
    const int rpb = 32;
    hipLaunchKernelGGL(cp_stats_kernel, dim3(n0 + K, (H + rpb - 1) / rpb), dim3(256), 0, st, masks, cover, src_desc, n0, H, W, K,
                       rpb, stats);
    hipLaunchKernelGGL(cp_resolve_kernel, dim3((n0 + K + CP_RES_OBJ - 1) / CP_RES_OBJ), dim3(CP_RES_OBJ),
                       (size_t)CP_RES_OBJ * (((K + 1) * 5) | 1) * sizeof(int32_t), st, stats, boxes0, n0, K, out_boxes, out_valid);
    // masks: one lane per 16 pixels; the objects split into groups so that small images still fill the chip
    const int64_t nchunk = (HW + CPX - 1) / CPX;
    const int gx = (int)((nchunk + 255) / 256 < 2048 ? (nchunk + 255) / 256 : 2048);
    int groups = gx >= 2048 ? 1 : (2048 + gx - 1) / gx;
    if (groups > n0 + K) groups = n0 + K;
    const int opg = (n0 + K + groups - 1) / groups;
    hipLaunchKernelGGL(cp_masks_kernel, dim3(gx, (n0 + K + opg - 1) / opg), dim3(256), 0, st, masks, cover, n0, H, W, K, opg, out_masks);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
