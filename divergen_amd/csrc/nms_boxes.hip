// Greedy NMS (bitmask + on-device sweep) and fused pairwise-IoU + Matcher for gfx950.
// Integer outputs (keep flags, matched indices/labels) are bit-exact with the oracle: the IoU
// float sequence is the reference's (torchvision nms: inter/(a_i+a_j-inter) > thr;
// D2/structures/boxes.py:310-357: inter/(a1+a2-inter)), compiled with -ffp-contract=off.
#include "dgx_common.h"

// mask[i][cb] bit j: box (64*cb + j) has IoU > thr with box i and comes later in score order.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thr, int nb,
                                                      uint64_t* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    __shared__ float cbx[64 * 4];
    const int l = threadIdx.x;
    const int cj = 64 * cb + l;
    if (cj < n) {
        cbx[4 * l + 0] = boxes[4 * cj + 0];
        cbx[4 * l + 1] = boxes[4 * cj + 1];
        cbx[4 * l + 2] = boxes[4 * cj + 2];
        cbx[4 * l + 3] = boxes[4 * cj + 3];
    }
    __syncthreads();
    const int i = 64 * rb + l;
    if (i >= n) return;
    const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    const float iarea = (ix2 - ix1) * (iy2 - iy1);
    const int cols = min(64, n - 64 * cb);
    uint64_t bits = 0;
    for (int j = (rb == cb ? l + 1 : 0); j < cols; ++j) {
        const float jx1 = cbx[4 * j], jy1 = cbx[4 * j + 1], jx2 = cbx[4 * j + 2], jy2 = cbx[4 * j + 3];
        const float jarea = (jx2 - jx1) * (jy2 - jy1);
        const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
        const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
        const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (iarea + jarea - inter);
        if (ovr > thr) bits |= 1ull << j;
    }
    mask[(int64_t)i * nb + cb] = bits;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// One workgroup walks the 64-box blocks in order: lane-serial resolve inside a block, then all
// threads OR the rows of the kept boxes into the running "removed" vector held in LDS.
__global__ __launch_bounds__(1024) void nms_sweep_kernel(const uint64_t* __restrict__ mask, int n, int nb,
                                                         uint8_t* __restrict__ keep, int32_t* __restrict__ num_keep) {
    extern __shared__ uint64_t remv[];  // [nb] + kept-list scratch
    __shared__ int kept_rows[64];
    __shared__ int kept_n;
    __shared__ int total;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb; i += blockDim.x) remv[i] = 0;
    if (tid == 0) total = 0;
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int cnt = min(64, n - 64 * b);
        if (tid < 64) {
            // diagonal words of this block's rows
            const uint64_t mine = tid < cnt ? mask[(int64_t)(64 * b + tid) * nb + b] : 0ull;
            // the serial part is scalar and walks the KEPT boxes only: find-first-set, that row's diagonal word out of its lane, three bit
            // operations (rounds 1-5: 64 cross-lane shuffles per block whatever survived)
            const uint64_t vbits = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            uint64_t alive = uniform64(~remv[b] & vbits), keptbits = 0;
            while (alive) {
                const int j = __builtin_ctzll(alive);
                const uint64_t bit = 1ull << j;
                keptbits |= bit;
                alive &= ~(readlane64(mine, j) | bit);
            }
            const bool kept = (keptbits >> tid) & 1ull;
            if (tid < cnt) keep[64 * b + tid] = kept ? 1 : 0;
            if (kept) kept_rows[__popcll(keptbits & ((1ull << tid) - 1ull))] = 64 * b + tid;
            if (tid == 0) { const int k = __popcll(keptbits); kept_n = k; total += k; }
        }
        __syncthreads();
        const int kn = kept_n;
        // OR the kept rows into the running vector: columns across lanes (coalesced), the kept rows
        // split over G row-groups so all 1024 threads have independent loads in flight
        const int ncols = nb - b - 1;
        if (ncols > 0 && kn > 0) {
            int G = blockDim.x / ncols;                 // lanes per column: each walks every G-th kept row, four loads in flight
            G = G < 1 ? 1 : (G > 16 ? 16 : G);
            for (int idx = tid; idx < ncols * G; idx += blockDim.x) {
                const int grp = idx / ncols, cb = b + 1 + (idx - grp * ncols);
                uint64_t acc = 0;
                int k = grp;
                for (; k + 3 * G < kn; k += 4 * G) {
                    const uint64_t a0 = mask[(int64_t)kept_rows[k] * nb + cb];
                    const uint64_t a1 = mask[(int64_t)kept_rows[k + G] * nb + cb];
                    const uint64_t a2 = mask[(int64_t)kept_rows[k + 2 * G] * nb + cb];
                    const uint64_t a3 = mask[(int64_t)kept_rows[k + 3 * G] * nb + cb];
                    acc |= a0 | a1 | a2 | a3;
                }
                for (; k < kn; k += G) acc |= mask[(int64_t)kept_rows[k] * nb + cb];
                if (acc) atomicOr((unsigned long long*)&remv[cb], (unsigned long long)acc);
            }
        }
        __syncthreads();
    }
    if (tid == 0) *num_keep = total;
}

extern "C" int64_t dgx_nms_workspace_words(int n) { return n <= 0 ? 0 : (int64_t)n * ((n + 63) / 64); }

extern "C" int dgx_nms_sorted(const float* boxes, int n, float iou_thr, uint64_t* mask, uint8_t* keep,
                              int32_t* num_keep, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || !num_keep) return DGX_ERR_BAD_ARG;
    if (n == 0) {
        (void)hipMemsetAsync(num_keep, 0, sizeof(int32_t), st);
        return DGX_OK;
    }
    if (!boxes || !mask || !keep) return DGX_ERR_BAD_ARG;
    const int nb = (n + 63) / 64;
    if ((size_t)nb * 8 > 60000) return DGX_ERR_UNSUPPORTED;  // removed-vector must fit LDS (n <= ~480k)
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(64), 0, st, boxes, n, iou_thr, nb, mask);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(1024), (size_t)nb * 8, st, mask, n, nb, keep, num_keep);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// ---- batched NMS with device-resident counts --------------------------------------------------
// Same arithmetic as above for B images at once; the number of candidates per image is read from device
// memory (no host round trip), the sweep stops at max_keep (+ score ties) and emits the kept indices
// compacted in score order.
__global__ __launch_bounds__(64) void nms_mask_batched_kernel(const float* __restrict__ boxes_all, const int32_t* __restrict__ n_valid,
                                                              int K, float thr, int nb, uint64_t* __restrict__ mask_all) {
    const int rb = blockIdx.y, cb = blockIdx.x, b = blockIdx.z;
    const int n = min(n_valid[b], K);
    if (cb < rb || 64 * cb >= n) return;
    const float* boxes = boxes_all + (int64_t)b * K * 4;
    uint64_t* mask = mask_all + (int64_t)b * K * nb;
    __shared__ float cbx[64 * 4];
    const int l = threadIdx.x;
    const int cj = 64 * cb + l;
    if (cj < n) {
        cbx[4 * l + 0] = boxes[4 * cj + 0];
        cbx[4 * l + 1] = boxes[4 * cj + 1];
        cbx[4 * l + 2] = boxes[4 * cj + 2];
        cbx[4 * l + 3] = boxes[4 * cj + 3];
    }
    __syncthreads();
    const int i = 64 * rb + l;
    if (i >= n) return;
    const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    const float iarea = (ix2 - ix1) * (iy2 - iy1);
    const int cols = min(64, n - 64 * cb);
    uint64_t bits = 0;
    for (int j = (rb == cb ? l + 1 : 0); j < cols; ++j) {
        const float jx1 = cbx[4 * j], jy1 = cbx[4 * j + 1], jx2 = cbx[4 * j + 2], jy2 = cbx[4 * j + 3];
        const float jarea = (jx2 - jx1) * (jy2 - jy1);
        const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
        const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
        const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (iarea + jarea - inter);
        if (ovr > thr) bits |= 1ull << j;
    }
    mask[(int64_t)i * nb + cb] = bits;
}

// The two-barrier form with the kept rows read from global memory: any K (the pipelined form below holds two blocks of rows in LDS
// and serves K <= ~4 900).  One workgroup per image.  Inside a 64-box block the survivors are found by a SCALAR loop over the kept
// boxes only (find-first-set on the alive word, readlane of that row's diagonal word); then all threads OR
// the kept rows into the running "removed" vector in LDS.
__global__ __launch_bounds__(1024) void nms_sweep_batched_global_kernel(const uint64_t* __restrict__ mask_all, const float* __restrict__ scores_all,
                                                                 const int32_t* __restrict__ n_valid, int K, int nb, int max_keep,
                                                                 int32_t* __restrict__ keep_idx_all, int cap, int32_t* __restrict__ num_keep) {
    extern __shared__ uint64_t remv[];  // [nb]
    __shared__ int kept_rows[64];
    __shared__ int kept_n, total, done;
    __shared__ float tie_score;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_valid[b], K);
    const int nbv = (n + 63) / 64;
    const uint64_t* mask = mask_all + (int64_t)b * K * nb;
    const float* scores = scores_all ? scores_all + (int64_t)b * K : nullptr;
    int32_t* keep_idx = keep_idx_all + (int64_t)b * cap;
    for (int i = tid; i < nb; i += blockDim.x) remv[i] = 0;
    for (int i = tid; i < cap; i += blockDim.x) keep_idx[i] = -1;
    if (tid == 0) { total = 0; done = 0; tie_score = 0.f; }
    __syncthreads();
    // the diagonal word of a row does not depend on the sweep: the next block's words are requested while this block's kept
    // rows are OR-ed into the removed vector (one global-memory latency less on the serial path per 64-box block)
    uint64_t mine_next = (tid < 64 && tid < min(64, n)) ? mask[(int64_t)tid * nb] : 0ull;
    for (int blk = 0; blk < nbv; ++blk) {
        const uint64_t mine_cur = mine_next;
        if (tid < 64 && blk + 1 < nbv)
            mine_next = tid < min(64, n - 64 * (blk + 1)) ? mask[(int64_t)(64 * (blk + 1) + tid) * nb + blk + 1] : 0ull;
        if (tid < 64) {
            const int cnt = min(64, n - 64 * blk);
            const uint64_t mine = mine_cur;
            const uint64_t vbits = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            uint64_t alive = uniform64(~remv[blk] & vbits);
            const int tot = __builtin_amdgcn_readfirstlane(total);
            float tie = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tie_score)));
            int k = 0, stop = 0;
            while (alive) {
                const int j = __builtin_ctzll(alive);
                const int idx = 64 * blk + j;
                if (max_keep > 0 && tot + k >= max_keep) {
                    // quota reached: only boxes tied with the max_keep-th kept score are still eligible
                    // (the reference keeps every survivor with score >= the k-th score, centernet.py:727-731)
                    if (!scores || scores[idx] != tie) { stop = 1; break; }
                }
                if (tid == 0) {
                    kept_rows[k] = idx;
                    if (tot + k < cap) keep_idx[tot + k] = idx;
                }
                ++k;
                if (max_keep > 0 && scores && tot + k == max_keep) tie = scores[idx];
                alive &= ~(readlane64(mine, j) | (1ull << j));
            }
            if (max_keep > 0 && !scores && tot + k >= max_keep) stop = 1;
            if (tid == 0) { kept_n = k; total = tot + k; done = stop; tie_score = tie; }
        }
        __syncthreads();
        const int kn = kept_n;
        if (done) break;
        const int ncols = nbv - blk - 1;
        if (ncols > 0 && kn > 0) {
            int G = blockDim.x / ncols;                 // lanes per column: each walks every G-th kept row, four loads in flight
            G = G < 1 ? 1 : (G > 16 ? 16 : G);
            for (int idx = tid; idx < ncols * G; idx += blockDim.x) {
                const int grp = idx / ncols, cb = blk + 1 + (idx - grp * ncols);
                uint64_t acc = 0;
                int k = grp;
                for (; k + 3 * G < kn; k += 4 * G) {
                    const uint64_t a0 = mask[(int64_t)kept_rows[k] * nb + cb];
                    const uint64_t a1 = mask[(int64_t)kept_rows[k + G] * nb + cb];
                    const uint64_t a2 = mask[(int64_t)kept_rows[k + 2 * G] * nb + cb];
                    const uint64_t a3 = mask[(int64_t)kept_rows[k + 3 * G] * nb + cb];
                    acc |= a0 | a1 | a2 | a3;
                }
                for (; k < kn; k += G) acc |= mask[(int64_t)kept_rows[k] * nb + cb];
                if (acc) atomicOr((unsigned long long*)&remv[cb], (unsigned long long)acc);
            }
        }
        __syncthreads();
    }
    if (tid == 0) num_keep[b] = total < cap ? total : cap;
}

constexpr int NMS_PRE = 3;        // removed-word requests a worker lane keeps in flight per phase (covers 2 880 kept rows)
// Round 3: the sweep as a pipeline with ONE barrier per 64-box block and no memory latency on its serial path, and with the
// removed words computed only for the columns the sweep reaches (it stops at max_keep kept boxes: ~35 of the 167 blocks of a
// 10 688-candidate image; the eager form above ORs every kept row into ALL later columns -- ten times the loads).  The mask of
// such an image is 14 MB, i.e. every request is a memory-side access of a few microseconds -- longer than a phase -- so every
// request is made TWO phases before its use.  Column c gets
//   * the rows kept in blocks <= c - 4 from the 15 worker waves: requested in phase c - 3 (one word per kept row, list in LDS), folded
//     into remv[c] in phase c - 1;
//   * the rows kept in blocks c - 3, c - 2 and c - 1 from the resolver wave itself: every row's words of the next THREE columns are
//     requested next to its diagonal word two phases ahead, and the kept rows' words are OR-ed out of the lanes (c - 3, c - 2: into
//     remv[c]; c - 1: kept in a register for the next phase).
struct NmsRowWords { uint64_t d, c1, c2, c3; };
__global__ __launch_bounds__(1024) void nms_sweep_batched_kernel(const uint64_t* __restrict__ mask_all, const float* __restrict__ scores_all,
                                                                 const int32_t* __restrict__ n_valid, int K, int nb, int max_keep,
                                                                 int32_t* __restrict__ keep_idx_all, int cap, int32_t* __restrict__ num_keep) {
    extern __shared__ uint64_t remv[];  // [nb], then the kept rows in keep order, int [K]
    int* kept_list = reinterpret_cast<int*>(remv + nb);
    __shared__ int total2[2], done;
    __shared__ float tie_score;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_valid[b], K);
    const int nbv = (n + 63) / 64;
    const uint64_t* mask = mask_all + (int64_t)b * K * nb;
    const float* scores = scores_all ? scores_all + (int64_t)b * K : nullptr;
    int32_t* keep_idx = keep_idx_all + (int64_t)b * cap;
    for (int i = tid; i < nb; i += blockDim.x) remv[i] = 0;
    for (int i = tid; i < cap; i += blockDim.x) keep_idx[i] = -1;
    if (tid == 0) { total2[0] = total2[1] = 0; done = 0; tie_score = 0.f; }
    __syncthreads();
    auto row_words = [&](int blk) {                  // resolver lane: row 64 blk + tid
        NmsRowWords w = {0ull, 0ull, 0ull, 0ull};
        const int r = 64 * blk + tid;
        if (blk < nbv && r < n) {
            const uint64_t* p = mask + (int64_t)r * nb + blk;
            w.d = p[0];
            if (blk + 1 < nbv) w.c1 = p[1];
            if (blk + 2 < nbv) w.c2 = p[2];
            if (blk + 3 < nbv) w.c3 = p[3];
        }
        return w;
    };
    NmsRowWords w0 = {0ull, 0ull, 0ull, 0ull}, w1 = w0;     // this block's and the next block's words
    if (tid < 64) { w0 = row_words(0); w1 = row_words(1); }
    uint64_t add = 0;
    const int nworkers = blockDim.x - 64, wid = tid - 64;
    uint64_t preA[NMS_PRE], preB[NMS_PRE];                  // requested in the previous phase / two phases ago
#pragma unroll
    for (int q = 0; q < NMS_PRE; ++q) preA[q] = preB[q] = 0;
    int total_end = 0;
    for (int blk = 0; blk < nbv; ++blk) {
        const int cur = blk & 1;
        const int tot = total2[cur ^ 1];               // kept in the blocks before this one (written in the previous phase)
        if (tid < 64) {
            const NmsRowWords w = w0;
            w0 = w1;
            w1 = row_words(blk + 2);
            const int cnt = min(64, n - 64 * blk);
            const uint64_t vbits = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            uint64_t alive = uniform64(~(remv[blk] | add) & vbits);
            float tie = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tie_score)));
            // the serial part is scalar only: find-first-set, one diagonal word out of its lane, two bit operations per kept box
            int k = 0, stop = 0;
            uint64_t keptbits = 0;
            // A block that cannot reach the quota or fill the output even if every live box of it is kept -- all blocks but the last one or
            // two -- takes the bare loop: find-first-set, one diagonal word out of its lane, three bit operations per kept box.  (The
            // general loop below carries the quota / tie / capacity tests as ~25 scalar instructions and five taken branches per kept
            // box: ~200 cycles each, 17 kept boxes per block on the bench's boxes = the 1.75 us a phase took.)
            const int room = (max_keep > 0 && max_keep < cap ? max_keep : cap) - tot;
            const bool bare = (int)__popcll(alive) < room;       // (HIP's __popcll is unsigned: room goes negative behind the quota)
            while (bare && alive) {
                const int j = __builtin_ctzll(alive);
                const uint64_t bit = 1ull << j;
                keptbits |= bit;
                alive &= ~(readlane64(w.d, j) | bit);
            }
            if (bare) k = __popcll(keptbits);
            while (alive) {
                const int j = __builtin_ctzll(alive);
                if (tot + k >= cap) { stop = 1; break; }      // every slot of the fixed-length output is taken: num_keep = cap whatever follows
                if (max_keep > 0 && tot + k >= max_keep) {
                    // quota reached: only boxes tied with the max_keep-th kept score are still eligible
                    // (the reference keeps every survivor with score >= the k-th score, centernet.py:727-731)
                    if (!scores || scores[64 * blk + j] != tie) { stop = 1; break; }
                }
                ++k;
                keptbits |= 1ull << j;
                if (max_keep > 0 && scores && tot + k == max_keep) tie = scores[64 * blk + j];
                alive &= ~(readlane64(w.d, j) | (1ull << j));
            }
            if (max_keep > 0 && !scores && tot + k >= max_keep) stop = 1;
            const bool kept = (keptbits >> tid) & 1ull;
            if (kept) {                                                  // lane = box: its rank among the kept ones of the block
                const int pos = tot + __popcll(keptbits & ((1ull << tid) - 1ull));
                kept_list[pos] = 64 * blk + tid;
                if (pos < cap) keep_idx[pos] = 64 * blk + tid;
            }
            uint64_t a1 = kept ? w.c1 : 0ull, a2 = kept ? w.c2 : 0ull, a3 = kept ? w.c3 : 0ull;   // what the kept rows remove further on
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a1 |= ((uint64_t)__shfl_xor((uint32_t)(a1 >> 32), o) << 32) | __shfl_xor((uint32_t)a1, o);
                a2 |= ((uint64_t)__shfl_xor((uint32_t)(a2 >> 32), o) << 32) | __shfl_xor((uint32_t)a2, o);
                a3 |= ((uint64_t)__shfl_xor((uint32_t)(a3 >> 32), o) << 32) | __shfl_xor((uint32_t)a3, o);
            }
            add = a1;
            if (tid == 0) {
                if (a2 && blk + 2 < nbv) atomicOr((unsigned long long*)&remv[blk + 2], (unsigned long long)a2);
                if (a3 && blk + 3 < nbv) atomicOr((unsigned long long*)&remv[blk + 3], (unsigned long long)a3);
                total2[cur] = tot + k;
                done = stop;
                tie_score = tie;
            }
        } else {
            // (a) fold what was requested two phases ago: column blk + 1, rows kept in blocks <= blk - 3
            uint64_t acc = 0;
#pragma unroll
            for (int q = 0; q < NMS_PRE; ++q) { acc |= preB[q]; preB[q] = preA[q]; }
            if (acc) atomicOr((unsigned long long*)&remv[blk + 1], (unsigned long long)acc);
            // (b) request column blk + 3 for the rows kept in blocks <= blk - 1 (`tot` of them)
            const int col = blk + 3;
#pragma unroll
            for (int q = 0; q < NMS_PRE; ++q) {
                const int e = wid + q * nworkers;
                preA[q] = (col < nbv && e < tot) ? mask[(int64_t)kept_list[e] * nb + col] : 0ull;
            }
            if (col < nbv && tot > NMS_PRE * nworkers) {                 // more kept rows than the lanes hold in flight (no quota): the rest now
                uint64_t more = 0;
                for (int e = wid + NMS_PRE * nworkers; e < tot; e += nworkers) more |= mask[(int64_t)kept_list[e] * nb + col];
                if (more) atomicOr((unsigned long long*)&remv[col], (unsigned long long)more);
            }
        }
        // LDS-only barrier: __syncthreads() would also wait for the global requests in flight (s_waitcnt vmcnt(0)) -- they are made
        // precisely to be waited for two phases later; everything the phases exchange (removed words, kept list, counters) lives
        // in LDS, the mask is read-only and keep_idx write-only
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        total_end = total2[cur];
        if (done) break;
    }
    if (tid == 0) num_keep[b] = total_end < cap ? total_end : cap;
}

extern "C" int64_t dgx_nms_batched_workspace_words(int B, int K) {
    return (B <= 0 || K <= 0) ? 0 : (int64_t)B * K * ((K + 63) / 64);
}

extern "C" int dgx_nms_batched(const float* boxes, const float* scores, const int32_t* n_valid, int B, int K, float iou_thr,
                               int max_keep, uint64_t* mask, int32_t* keep_idx, int cap, int32_t* num_keep, void* stream) {
    if (B <= 0) return DGX_OK;
    if (K <= 0 || !boxes || !n_valid || !mask || !keep_idx || !num_keep || cap <= 0 || max_keep < 0) return DGX_ERR_BAD_ARG;
    const int nb = (K + 63) / 64;
    if ((size_t)nb * 8 > 60000 || B > 65535) return DGX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_mask_batched_kernel, dim3(nb, nb, B), dim3(64), 0, st, boxes, n_valid, K, iou_thr, nb, mask);
    const size_t sm = (size_t)nb * 8 + (size_t)K * 4 + 8;
    if (sm > 150 * 1024) {                          // K > ~36 000: the kept list does not fit LDS -- the eager form
        hipLaunchKernelGGL(nms_sweep_batched_global_kernel, dim3(B), dim3(1024), (size_t)nb * 8, st, mask, scores, n_valid, K, nb, max_keep,
                           keep_idx, cap, num_keep);
        DGX_LAUNCH_CHECK();
        return DGX_OK;
    }
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)nms_sweep_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(nms_sweep_batched_kernel, dim3(B), dim3(1024), sm, st, mask, scores, n_valid, K, nb, max_keep, keep_idx, cap, num_keep);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// ---- pairwise IoU + Matcher -----------------------------------------------------------------
__global__ __launch_bounds__(256) void iou_match_kernel(const float* __restrict__ gt, int M, const float* __restrict__ props,
                                                        int N, float thr, int64_t* __restrict__ midx,
                                                        int8_t* __restrict__ mlab, float* __restrict__ miou) {
    extern __shared__ float g[];  // [M][5]: box + area
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const float x1 = gt[4 * i], y1 = gt[4 * i + 1], x2 = gt[4 * i + 2], y2 = gt[4 * i + 3];
        g[5 * i] = x1; g[5 * i + 1] = y1; g[5 * i + 2] = x2; g[5 * i + 3] = y2;
        g[5 * i + 4] = (x2 - x1) * (y2 - y1);
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const float px1 = props[4 * j], py1 = props[4 * j + 1], px2 = props[4 * j + 2], py2 = props[4 * j + 3];
    const float parea = (px2 - px1) * (py2 - py1);
    float best = -1.0f;
    int bi = 0;
    for (int i = 0; i < M; ++i) {
        const float w = fmaxf(fminf(g[5 * i + 2], px2) - fmaxf(g[5 * i], px1), 0.0f);
        const float h = fmaxf(fminf(g[5 * i + 3], py2) - fmaxf(g[5 * i + 1], py1), 0.0f);
        const float inter = w * h;
        const float iou = inter > 0.0f ? inter / (g[5 * i + 4] + parea - inter) : 0.0f;
        if (iou > best) { best = iou; bi = i; }  // first maximum wins (torch.max(dim=0))
    }
    if (M == 0) { best = 0.0f; bi = 0; }
    midx[j] = bi;
    mlab[j] = (M > 0 && best >= thr) ? 1 : 0;
    if (miou) miou[j] = best;
}

extern "C" int dgx_iou_match(const float* gt, int M, const float* props, int N, float thr, int64_t* matched_idx,
                             int8_t* matched_label, float* max_iou, void* stream) {
    if (N <= 0) return DGX_OK;
    if (!props || !matched_idx || !matched_label || M < 0 || (M > 0 && !gt)) return DGX_ERR_BAD_ARG;
    if ((size_t)M * 20 > 60000) return DGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(iou_match_kernel, dim3((N + 255) / 256), dim3(256), (size_t)M * 20, (hipStream_t)stream, gt, M,
                       props, N, thr, matched_idx, matched_label, max_iou);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
