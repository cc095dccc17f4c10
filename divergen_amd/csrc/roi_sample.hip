// Proposal labelling and sampling of the RoI heads for the whole batch
// (DG/divergen/modeling/roi_heads/detic_roi_heads.py:273-307 `label_and_sample_proposals`: D2 `add_ground_truth_to_proposals`
//  proposal_utils.py:126-196, `pairwise_iou` boxes.py:334-357, `Matcher` matcher.py:62-104, and the index bookkeeping of
//  `subsample_labels` sampling.py:9-54 -- the random permutations themselves stay with torch's generator, on the host side of
//  the one device->host read of the step).
//
// The composed form is ~125 small launches per step (per image: concatenations, IoU + match, class gather, masks, two stable
// sorts, counts; then a dozen index-gathers per sampled field).  Here:
//   dgx_roi_label   one workgroup per image: rows = the image's fixed-length proposal list followed by its ground-truth boxes;
//                   per row the best ground-truth box / label (the float sequence of dgx_iou_match), then the rows of the
//                   foreground and of the background in index order (= the stable sorts of the two masks) and their counts;
//   dgx_roi_gather  one thread per SAMPLED row of the batch: position in the image's permutation -> source row -> box, label,
//                   matched ground-truth box / index / instance_source, objectness logit.
#include "dgx_common.h"

namespace {
constexpr int RS_MAXB = 16;
struct RoiGatherP {
    const int64_t* perm_pos[RS_MAXB];   // per image: the first num_pos entries of randperm(n_pos)
    const int64_t* perm_neg[RS_MAXB];
    int num_pos[RS_MAXB], num_neg[RS_MAXB], row0[RS_MAXB + 1];
    int B, K, Nmax;
};

__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot, int& total) {
    // exclusive scan of one int per thread over a 1024-thread block (16 waves): wave shuffles + LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < 16; ++i) {
        const int t = wave_tot[i];
        if (i < w) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}
}  // namespace

__global__ __launch_bounds__(1024) void roi_label_kernel(const float* __restrict__ prop, const uint8_t* __restrict__ valid, int K,
                                                         const float* __restrict__ gt, const int64_t* __restrict__ gt_cls,
                                                         const int32_t* __restrict__ gt_off, float thr, int num_classes, int append_gt,
                                                         int Nmax, int32_t* __restrict__ midx, int64_t* __restrict__ label,
                                                         int32_t* __restrict__ pos_idx, int32_t* __restrict__ neg_idx,
                                                         int32_t* __restrict__ counts) {
    extern __shared__ float g[];                   // [M][5]: box + area
    __shared__ int wave_tot[16];
    const int b = blockIdx.x;
    const int g0 = gt_off[b], M = gt_off[b + 1] - g0;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const float* q = gt + 4 * (int64_t)(g0 + i);
        g[5 * i] = q[0]; g[5 * i + 1] = q[1]; g[5 * i + 2] = q[2]; g[5 * i + 3] = q[3];
        g[5 * i + 4] = (q[2] - q[0]) * (q[3] - q[1]);
    }
    __syncthreads();
    const int N = K + (append_gt ? M : 0);
    int base_p = 0, base_n = 0;
    for (int j0 = 0; j0 < Nmax; j0 += 1024) {
        const int j = j0 + threadIdx.x;
        int fp = 0, fn = 0;
        if (j < N) {
            float px1, py1, px2, py2;
            bool ok = true;
            if (j < K) {
                const float* p = prop + 4 * ((int64_t)b * K + j);
                px1 = p[0]; py1 = p[1]; px2 = p[2]; py2 = p[3];
                ok = valid ? valid[(int64_t)b * K + j] != 0 : true;
            } else {
                px1 = g[5 * (j - K)]; py1 = g[5 * (j - K) + 1]; px2 = g[5 * (j - K) + 2]; py2 = g[5 * (j - K) + 3];
            }
            const float parea = (px2 - px1) * (py2 - py1);
            float best = -1.0f;
            int bi = 0;
            for (int i = 0; i < M; ++i) {          // pairwise_iou + max over the ground truth: first maximum wins
                const float w = fmaxf(fminf(g[5 * i + 2], px2) - fmaxf(g[5 * i], px1), 0.0f);
                const float h = fmaxf(fminf(g[5 * i + 3], py2) - fmaxf(g[5 * i + 1], py1), 0.0f);
                const float inter = w * h;
                const float iou = inter > 0.0f ? inter / (g[5 * i + 4] + parea - inter) : 0.0f;
                if (iou > best) { best = iou; bi = i; }
            }
            if (M == 0) bi = 0;
            const bool fg = M > 0 && best >= thr;
            const int64_t cls = !ok ? -1 : (fg ? gt_cls[g0 + bi] : (int64_t)num_classes);
            midx[(int64_t)b * Nmax + j] = bi;
            label[(int64_t)b * Nmax + j] = cls;
            fp = (cls != -1 && cls != num_classes) ? 1 : 0;
            fn = (cls == num_classes) ? 1 : 0;
        } else if (j < Nmax) {
            midx[(int64_t)b * Nmax + j] = 0;
            label[(int64_t)b * Nmax + j] = -1;
        }
        int tp, tn;
        const int rp = block_excl_scan(fp, wave_tot, tp);
        const int rn = block_excl_scan(fn, wave_tot, tn);
        if (fp) pos_idx[(int64_t)b * Nmax + base_p + rp] = j;
        if (fn) neg_idx[(int64_t)b * Nmax + base_n + rn] = j;
        base_p += tp;
        base_n += tn;
    }
    if (threadIdx.x == 0) { counts[2 * b] = base_p; counts[2 * b + 1] = base_n; }
}

__global__ __launch_bounds__(256) void roi_gather_kernel(RoiGatherP P, const float* __restrict__ prop, const float* __restrict__ logits,
                                                         const float* __restrict__ gt, const int64_t* __restrict__ gt_src,
                                                         const int32_t* __restrict__ gt_off, float gt_logit, const int32_t* __restrict__ midx,
                                                         const int64_t* __restrict__ label, const int32_t* __restrict__ pos_idx,
                                                         const int32_t* __restrict__ neg_idx, float* __restrict__ o_box,
                                                         int64_t* __restrict__ o_cls, float* __restrict__ o_gtb, int64_t* __restrict__ o_gti,
                                                         int64_t* __restrict__ o_src, float* __restrict__ o_logit) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.row0[P.B]) return;
    int b = 0;
    while (b + 1 < P.B && r >= P.row0[b + 1]) ++b;
    const int t = r - P.row0[b];
    const int j = t < P.num_pos[b] ? pos_idx[(int64_t)b * P.Nmax + (int)P.perm_pos[b][t]]
                                   : neg_idx[(int64_t)b * P.Nmax + (int)P.perm_neg[b][t - P.num_pos[b]]];
    const int g0 = gt_off[b], M = gt_off[b + 1] - g0;
    float bx[4];
    if (j < P.K) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bx[k] = prop[4 * ((int64_t)b * P.K + j) + k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) bx[k] = gt[4 * (int64_t)(g0 + j - P.K) + k];
    }
    const int mi = midx[(int64_t)b * P.Nmax + j];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o_box[4 * (int64_t)r + k] = bx[k];
        o_gtb[4 * (int64_t)r + k] = M > 0 ? gt[4 * (int64_t)(g0 + mi) + k] : bx[k];
    }
    o_cls[r] = label[(int64_t)b * P.Nmax + j];
    o_gti[r] = mi;
    if (o_src) o_src[r] = (M > 0 && gt_src) ? gt_src[g0 + mi] : 0;
    if (o_logit) o_logit[r] = j < P.K ? (logits ? logits[(int64_t)b * P.K + j] : 0.0f) : gt_logit;
}

extern "C" int dgx_roi_label(const float* prop, const uint8_t* valid, int B, int K, const float* gt_boxes, const int64_t* gt_classes,
                             const int32_t* gt_offsets, int max_gt, float iou_thr, int num_classes, int append_gt, int Nmax,
                             int32_t* matched_idx, int64_t* labels, int32_t* pos_idx, int32_t* neg_idx, int32_t* counts, void* stream) {
    if (B <= 0) return DGX_OK;
    if (!prop || !gt_offsets || !matched_idx || !labels || !pos_idx || !neg_idx || !counts || K < 0 || Nmax < K || max_gt < 0 ||
        (max_gt > 0 && (!gt_boxes || !gt_classes)))
        return DGX_ERR_BAD_ARG;
    if ((size_t)max_gt * 20 > 60000) return DGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(roi_label_kernel, dim3(B), dim3(1024), (size_t)max_gt * 20, (hipStream_t)stream, prop, valid, K, gt_boxes, gt_classes,
                       gt_offsets, iou_thr, num_classes, append_gt, Nmax, matched_idx, labels, pos_idx, neg_idx, counts);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int dgx_roi_gather(int B, int K, int Nmax, const int64_t* const* perm_pos, const int64_t* const* perm_neg, const int* num_pos,
                              const int* num_neg, const float* prop, const float* logits, const float* gt_boxes, const int64_t* gt_src,
                              const int32_t* gt_offsets, float gt_logit, const int32_t* matched_idx, const int64_t* labels,
                              const int32_t* pos_idx, const int32_t* neg_idx, float* out_boxes, int64_t* out_classes, float* out_gt_boxes,
                              int64_t* out_gt_index, int64_t* out_src, float* out_logits, void* stream) {
    if (B <= 0) return DGX_OK;
    if (B > RS_MAXB || !perm_pos || !perm_neg || !num_pos || !num_neg || !prop || !gt_offsets || !matched_idx || !labels || !pos_idx ||
        !neg_idx || !out_boxes || !out_classes || !out_gt_boxes || !out_gt_index)
        return DGX_ERR_BAD_ARG;
    RoiGatherP P;
    P.B = B; P.K = K; P.Nmax = Nmax;
    P.row0[0] = 0;
    for (int b = 0; b < B; ++b) {
        P.perm_pos[b] = perm_pos[b]; P.perm_neg[b] = perm_neg[b];
        P.num_pos[b] = num_pos[b]; P.num_neg[b] = num_neg[b];
        P.row0[b + 1] = P.row0[b] + num_pos[b] + num_neg[b];
    }
    const int R = P.row0[B];
    if (R <= 0) return DGX_OK;
    hipLaunchKernelGGL(roi_gather_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, prop, logits, gt_boxes, gt_src,
                       gt_offsets, gt_logit, matched_idx, labels, pos_idx, neg_idx, out_boxes, out_classes, out_gt_boxes, out_gt_index,
                       out_src, out_logits);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
