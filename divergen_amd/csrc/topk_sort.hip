// Per-level top-k and the score sort of the CenterNet proposal decode (gfx950).
//
// Reference: CN/modeling/dense_heads/centernet.py:690-737 (predict_single_level: `per_candidate_inds` + `topk(pre_nms_top_n, sorted=False)`
// per FPN level) and :739-768 (nms_and_topK -> ml_nms on score-sorted boxes).  The build ran torch.topk -- at::native::sbtopk::gatherTopK,
// ONE workgroup per row, 115-150 us for 16 384 scores -- and a rocprim segmented sort (93 us + helpers for 2 x 9 344) on the step's
// pre-sync critical path.  Here: dgx_topk_index_rows = radix select of the k-th largest score (4 passes of 8 bits over the row in LDS)
// + order-preserving compaction: the k selected positions in ASCENDING index order (torch's set: everything above the k-th value, then
// the lowest-index elements equal to it); dgx_sort_rows_desc = rank sort of (score, position) pairs over the whole chip, equal to
// torch.sort(descending=True, stable=True) element for element (a 64-bit key: score bits, then inverted position).
#include "dgx_common.h"

namespace {
__device__ __forceinline__ uint32_t f2key(float f) {          // monotonic: larger float <-> larger key; -0 < +0 (torch treats them equal: no
    const uint32_t u = __float_as_uint(f);                    // score of this path is a negative zero)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int TK_THREADS = 1024;
constexpr int TK_WAVES = TK_THREADS / 64;
constexpr int TK_MAXN = 32768;          // keys of a row in LDS: 128 KB

// One workgroup per (level, image).  Phases, all on keys held in LDS:
//   1. radix select, 4 passes of 8 bits from the top: histogram of the digit among the keys that still match the prefix (a wave whose
//      64 keys share the digit -- the common case: most scores of a level are the below-threshold marker -- adds once), then the bin
//      holding the remaining-th largest, found by a 256-thread suffix scan;
//   2. order-preserving compaction: wave w owns the contiguous chunk w of the row; a count pass (ballots), a scan over the 16 waves,
//      a write pass.  Elements equal to the k-th value are taken lowest position first until k is reached.
__global__ __launch_bounds__(TK_THREADS) void topk_index_rows_kernel(const float* __restrict__ scores, int64_t row_stride,
                                                                     const int32_t* __restrict__ level_off, const int32_t* __restrict__ level_n,
                                                                     int k, int64_t* __restrict__ out, int64_t out_row_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);
    __shared__ int hist[256];
    __shared__ int wsum[4];
    __shared__ int w_gt[TK_WAVES], w_eq[TK_WAVES], w_pos[TK_WAVES], w_take[TK_WAVES];
    __shared__ uint32_t s_prefix;
    __shared__ int s_remaining;
    const int lev = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int off = level_off[lev], n = level_n[lev];
    const float* row = scores + (int64_t)b * row_stride + off;
    for (int i = tid; i < n; i += TK_THREADS) keys[i] = f2key(row[i]);
    if (tid == 0) { s_prefix = 0u; s_remaining = k; }
    uint32_t mask = 0u;
    const int n_up = (n + 63) & ~63;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const int rem = s_remaining;
        for (int i = tid; i < n_up; i += TK_THREADS) {
            const uint32_t key = i < n ? keys[i] : 0u;
            const bool in = i < n && (key & mask) == prefix;
            const uint32_t digit = (key >> shift) & 255u;
            const uint64_t act = __ballot(in);
            if (act == 0ull) continue;
            const uint32_t d0 = __shfl(digit, __ffsll((long long)act) - 1);
            if (__ballot(in && digit != d0) == 0ull) {
                if (lane == 0) atomicAdd(&hist[d0], __popcll(act));
            } else if (in) {
                atomicAdd(&hist[digit], 1);
            }
        }
        __syncthreads();
        if (tid < 256) {                 // thread t looks at bin 255 - t: inclusive sums from the top
            const int h = hist[255 - tid];
            int x = h;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
            if (lane == 63) wsum[w] = x;
            __threadfence_block();
        }
        __syncthreads();
        if (tid < 256) {
            const int h = hist[255 - tid];
            int x = h;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
            for (int q = 0; q < w; ++q) x += wsum[q];
            if (x >= rem && x - h < rem) {               // exactly one thread: the counts sum to >= rem by construction
                s_prefix = prefix | ((uint32_t)(255 - tid) << shift);
                s_remaining = rem - (x - h);
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const uint32_t kth = s_prefix;
    const int need_eq = s_remaining;     // how many of the elements EQUAL to the k-th value are taken (lowest positions first)
    const int chunk = ((n + TK_WAVES - 1) / TK_WAVES + 63) & ~63;
    const int i0 = w * chunk, i1 = min(n, i0 + chunk);
    int gt = 0, eq = 0;
    for (int i = i0 + lane; i < i0 + chunk; i += 64) {
        const uint32_t key = i < i1 ? keys[i] : 0u;
        gt += __popcll(__ballot(i < i1 && key > kth));
        eq += __popcll(__ballot(i < i1 && key == kth));
    }
    if (lane == 0) { w_gt[w] = gt; w_eq[w] = eq; }
    __syncthreads();
    if (tid == 0) {
        int eq_base = 0, pos = 0;
        for (int q = 0; q < TK_WAVES; ++q) {
            const int take = min(max(need_eq - eq_base, 0), w_eq[q]);
            w_take[q] = take;
            w_pos[q] = pos;
            pos += w_gt[q] + take;
            eq_base += w_eq[q];
        }
    }
    __syncthreads();
    int64_t* o = out + (int64_t)b * out_row_stride + (int64_t)lev * k;
    int pos = w_pos[w], eq_seen = 0;
    const int eq_take = w_take[w];
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int i = i0 + lane; i < i0 + chunk; i += 64) {
        const uint32_t key = i < i1 ? keys[i] : 0u;
        const bool is_eq = i < i1 && key == kth;
        const uint64_t eqm = __ballot(is_eq);
        const bool take = (i < i1 && key > kth) || (is_eq && eq_seen + __popcll(eqm & lt) < eq_take);
        const uint64_t tm = __ballot(take);
        if (take) o[pos + __popcll(tm & lt)] = (int64_t)off + i;
        pos += __popcll(tm);
        eq_seen += __popcll(eqm);
    }
}

// ---- sort: K <= SORT_MAXK (score, position) pairs per row, descending, stable -- a RANK sort: the place of an element is the number of
// elements that come before it, K^2 comparisons spread over the whole chip (2 x 9 344^2 = 1.7e8, ~15 us of VALU) instead of the ~100
// dependent passes of a sorting network inside two workgroups (measured: bitonic in LDS 140 us, torch's segmented merge sort 70 us).
// A workgroup ranks 64 elements (one per lane); its 16 waves each scan 1/16 of the row's keys, read as LDS broadcasts.
constexpr int SORT_MAXK = 16384;
constexpr int RS_I = 64;
__global__ __launch_bounds__(TK_THREADS) void rank_sort_rows_desc_kernel(const float* __restrict__ in, int K, int KP /* K rounded up to 32 */,
                                                                         float* __restrict__ out_vals, int64_t* __restrict__ out_order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* a = reinterpret_cast<uint64_t*>(smem);
    __shared__ int part[TK_WAVES][RS_I];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* row = in + (int64_t)b * K;
    for (int i = tid; i < KP; i += TK_THREADS)       // (score key, inverted position): larger = earlier; padding = 0 is below every real key
        a[i] = i < K ? ((uint64_t)f2key(row[i]) << 32) | (uint64_t)(0xffffffffu - (uint32_t)i) : 0ull;
    __syncthreads();
    const int i = blockIdx.x * RS_I + lane;
    const uint64_t ki = i < K ? a[i] : ~0ull;
    const int per = KP / TK_WAVES;                   // even
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a + w * per);
    int cnt = 0;
#pragma unroll 4
    for (int j = 0; j < per / 2; ++j) {
        const ulonglong2 kk = src[j];
        cnt += (kk.x > ki) + (kk.y > ki);
    }
    part[w][lane] = cnt;
    __syncthreads();
    if (w == 0 && i < K) {
        int r = 0;
#pragma unroll
        for (int q = 0; q < TK_WAVES; ++q) r += part[q][lane];
        out_vals[(int64_t)b * K + r] = key2f((uint32_t)(ki >> 32));
        out_order[(int64_t)b * K + r] = (int64_t)i;
    }
}
}  // namespace

extern "C" int dgx_topk_index_rows(const float* scores, int64_t row_stride, int B, const int32_t* level_off, const int32_t* level_n,
                                   const int32_t* level_n_host, int nlev, int k, int64_t* out, int64_t out_row_stride, void* stream) {
    if (B <= 0 || nlev <= 0) return DGX_OK;
    if (!scores || !level_off || !level_n || !level_n_host || !out || k <= 0) return DGX_ERR_BAD_ARG;
    int nmax = 0;
    for (int l = 0; l < nlev; ++l) {
        if (level_n_host[l] < k) return DGX_ERR_BAD_ARG;          // a level with fewer than k entries is taken whole by the caller
        nmax = level_n_host[l] > nmax ? level_n_host[l] : nmax;
    }
    if (nmax > TK_MAXN) return DGX_ERR_UNSUPPORTED;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)topk_index_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TK_MAXN * 4) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    hipLaunchKernelGGL(topk_index_rows_kernel, dim3(nlev, B), dim3(TK_THREADS), (size_t)nmax * 4, (hipStream_t)stream, scores, row_stride,
                       level_off, level_n, k, out, out_row_stride);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_sort_rows_desc(const float* in, int B, int K, float* out_vals, int64_t* out_order, void* stream) {
    if (B <= 0 || K <= 0) return DGX_OK;
    if (!in || !out_vals || !out_order) return DGX_ERR_BAD_ARG;
    if (K > SORT_MAXK) return DGX_ERR_UNSUPPORTED;
    const int KP = (K + 31) & ~31;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)rank_sort_rows_desc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_MAXK * 8) != hipSuccess)
            return DGX_ERR_UNSUPPORTED;
        once = true;
    }
    hipLaunchKernelGGL(rank_sort_rows_desc_kernel, dim3((K + RS_I - 1) / RS_I, B), dim3(TK_THREADS), (size_t)KP * 8, (hipStream_t)stream, in, K, KP,
                       out_vals, out_order);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
