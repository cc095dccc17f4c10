/* divergen_hip.h -- C ABI of libdgx.so, the gfx950 (MI355X / CDNA4) kernel library behind the
 * DiverGen training hot path.
 *
 * The reference (aim-uofa/DiverGen) has no FFI of its own: its "native" calls on this path are
 * ATen / torchvision ops issued from Python.  Each entry point below replaces one such call site
 * (cited as file:line; DG = DiverGen/, D2 = BSGAL/third_party/CenterNet2/detectron2/,
 * CN = BSGAL/third_party/CenterNet2/projects/CenterNet2/centernet/).  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add at each site.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless marked (host);
 *   - tensors are dense, row-major in the order written in the comment;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - return value: 0 on success, DGX_ERR_* (<0) on bad arguments, or -(hipError_t) - 1000 for a
 *     launch failure.  No exceptions cross the boundary.
 *   - bf16 = raw uint16_t bfloat16 bits.  dtype selectors: DGX_F32 / DGX_BF16.
 */
#ifndef DIVERGEN_HIP_H
#define DIVERGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGX_OK 0
#define DGX_ERR_BAD_ARG (-1)
#define DGX_ERR_UNSUPPORTED (-2)
#define DGX_F32 0
#define DGX_BF16 1

/* library / device identification; returns the gfx arch string the kernels were built for */
const char* dgx_build_arch(void);
int dgx_abi_version(void);

/* Loader staging.  The reference's DataLoader hands every batch to a pin thread that allocates pinned memory and copies into it
 * (torch/utils/data/_utils/pin_memory.py behind D2/data/build.py:build_detection_train_loader, DG/train_net.py:164-239).  Here the
 * training process page-locks ONE shared-memory region once (dgx_host_register; bytes of any size, p from mmap / shm), the loader
 * workers write their sample blobs into slots of it, and a slot goes up with dgx_memcpy_h2d_async on the loader's stream: the
 * copy is asynchronous because the source is page-locked; the caller keeps the slot untouched until the stream has passed it. */
int dgx_host_register(void* p, size_t bytes);
int dgx_host_unregister(void* p);
int dgx_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Swin window attention core.  Replaces WindowAttention.forward between the qkv Linear and the
 * proj Linear: q*scale, q@k^T, + relative_position_bias_table[relative_position_index],
 * + shift mask, softmax (fp32), @v, head merge.   DG/divergen/modeling/backbone/swintransformer.py:133-154
 *
 *   qkv    bf16 (B_, N, 3, nH, 32)   output of the qkv Linear, N = ws*ws, head_dim fixed at 32
 *   table  f32                       relative_position_bias_table, entry (index i, head h) at
 *                                    table[h*table_stride_head + i*table_stride_index]: the parameter itself,
 *                                    ((2ws-1)^2, nH), is strides (1, nH); a per-head-contiguous copy is (T, 1)
 *   region i8   (nW, N) or NULL      region id of each token of each window position; the additive
 *                                    mask of swintransformer.py:368-387 is (region[i]!=region[j]) ? -100 : 0;
 *                                    B_ % nW == 0, window b uses row b % nW.  NULL = W-MSA (no mask)
 *   out    bf16 (B_, N, nH*32)
 *   lse    f32  (B_, nH, N)          log-sum-exp of each score row (saved for backward)
 * ws in {7, 12}.
 */
int dgx_window_attention_fwd(const void* qkv, const float* table, int64_t table_stride_head,
                             int64_t table_stride_index, const int8_t* region, void* out, float* lse, int B_,
                             int nW, int nH, int ws, float scale, void* stream);

/* Backward of the above.  dqkv bf16 (B_,N,3,nH,32) is fully overwritten; the bias-table gradient is
 * ADDED to dtable[h*dtable_stride_head + i*dtable_stride_index], so the caller can point it at the
 * parameter's own (T, nH) gradient (strides 1, nH) or at a zeroed (nH, T) scratch (strides T, 1).
 * `table` is read with the SAME two strides.  `out`/`lse` are the forward results.
 * The table gradient is summed in a fixed order (per workgroup run, then over the runs of a head by the
 * last run to finish: bit-reproducible for a given launch shape and CU reservation, no float atomics);
 * the partial sums live in a workspace the library keeps PER STREAM (allocated at the first call on a
 * stream, outside any graph capture): launches on one stream are ordered with each other, launches on
 * different streams do not share it.  DGX_ERR_UNSUPPORTED when that workspace cannot be allocated. */
int dgx_window_attention_bwd(const void* qkv, const float* table, const int8_t* region,
                             const void* out, const float* lse, const void* dout,
                             void* dqkv, float* dtable, int64_t dtable_stride_head,
                             int64_t dtable_stride_index, int B_, int nW, int nH, int ws, float scale,
                             void* stream);

/* The same pair over COMPACT window order (divergen_amd/csrc/winmap.h): the reference pads the token grid to multiples of ws
 * AFTER norm1 (swintransformer.py:216-221), so a padding token's qkv row is the qkv bias, its output is cropped (:248-251) and no
 * gradient reaches its query -- rows that need not exist.  Here qkv / out / dout hold the B*H*W REAL tokens only, in (image,
 * window, token) order with the padding tokens left out (every GEMM around the attention runs at M = B*H*W instead of
 * B*nW*ws*ws: -21 % rows in Swin stages 2 and 3 at 1024^2, -39 % in stage 3 at 896^2); the kernels take the bias for a padding
 * token's q / k / v, never store its output, and read 0 for its dO.
 *   qkv      bf16 (B*H*W, 3, nH, 32)          qkv_bias bf16 (3, nH, 32)
 *   out/dout bf16 (B*H*W, nH*32)              lse f32 (B*nW, nH, N) (per window position, as in the classic pair)
 *   dqkv     bf16 (B*nW*N, 3, nH, 32)         rows 0 .. B*H*W-1: the real tokens; the rest: the padding tokens' (0, dk, dv) in their
 *                                             own (image, window, token) order -- the qkv BIAS gradient sums over all rows
 *   region   i8 (nW, N) or NULL               as above (indexed by window position and token, padding tokens included)
 * B images of H x W tokens, cyclic shift `shift` (0 or ws/2), H >= shift and W >= shift. */
int dgx_window_attention_fwd_compact(const void* qkv, const void* qkv_bias, const float* table, int64_t table_stride_head,
                                     int64_t table_stride_index, const int8_t* region, void* out, float* lse, int B, int H,
                                     int W, int nH, int ws, int shift, float scale, void* stream);
int dgx_window_attention_bwd_compact(const void* qkv, const void* qkv_bias, const float* table, const int8_t* region,
                                     const void* out, const float* lse, const void* dout, void* dqkv, float* dtable,
                                     int64_t dtable_stride_head, int64_t dtable_stride_index, int B, int H, int W, int nH,
                                     int ws, int shift, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Window gather / scatter: zero-pad to a multiple of ws, cyclic shift by -shift, window_partition
 * (gather), and the exact inverse window_reverse + roll(+shift) + crop (scatter).
 * swintransformer.py:216-233 and :239-251.  dtype DGX_BF16 or DGX_F32.
 *   x  (B, H, W, C)      xw (B*nWh*nWw, ws*ws, C),  nWh = ceil(H/ws), nWw = ceil(W/ws)
 * gather: xw <- x (padding tokens = 0).  scatter: x <- xw (padding tokens dropped). */
int dgx_window_gather(const void* x, void* xw, int B, int H, int W, int C, int ws, int shift,
                      int dtype, void* stream);
int dgx_window_scatter(const void* xw, void* x, int B, int H, int W, int C, int ws, int shift,
                       int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ROIAlign (aligned=True/False, adaptive sampling when sampling_ratio == 0).  Replaces
 * torchvision.ops.roi_align at D2/layers/roi_align.py:58-65 (callers D2/modeling/poolers.py:185-245).
 *   feat   (N, H, W, C) channels-last, dtype f32 or bf16
 *   rois   f32 (R, 5) = (batch_index, x1, y1, x2, y2) in image coordinates
 *   out    (R, C, ph, pw) if out_nhwc == 0 else (R, ph, pw, C), same dtype as feat
 * Backward accumulates (atomic add) into grad_feat f32 (N,H,W,C), which the caller zeroes. */
int dgx_roi_align_fwd(const void* feat, const float* rois, void* out, int N, int H, int W, int C,
                      int R, float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                      int out_nhwc, int dtype, void* stream);
int dgx_roi_align_bwd(const void* grad_out, const float* rois, float* grad_feat, int N, int H,
                      int W, int C, int R, float spatial_scale, int ph, int pw, int sampling_ratio,
                      int aligned, int out_nhwc, int dtype, void* stream);

/* Multi-level ROIPooler in one launch: level assignment floor(4 + log2(sqrt(area)/224 + 1e-8))
 * clamped to [min_level, max_level] (D2/modeling/poolers.py:22-58) followed by ROIAlignV2 on the
 * chosen level.  feats/grad_feats: (host) array of num_levels device pointers; Hs/Ws (host) ints;
 * level l has spatial_scale = 2^-(min_level + l).  levels_out i32 (R) optional (may be NULL). */
int dgx_roi_pooler_fwd(const void* const* feats, const int* Hs, const int* Ws, int num_levels,
                       int min_level, const float* rois, void* out, int32_t* levels_out, int N,
                       int C, int R, int ph, int pw, int sampling_ratio, int out_nhwc, int dtype,
                       void* stream);
int dgx_roi_pooler_bwd(const void* grad_out, float* const* grad_feats, const int* Hs, const int* Ws,
                       int num_levels, int min_level, const float* rois, int N, int C, int R, int ph,
                       int pw, int sampling_ratio, int out_nhwc, int dtype, void* stream);

/* Backward of the two pooling entry points above as an output-stationary GATHER (the autograd backward of
 * torchvision.ops.roi_align reached from D2/modeling/poolers.py:240-245): every pixel of every level's gradient map is
 * produced by one workgroup that walks the RoIs in index order -- no atomics, no zero fill, deterministic.
 *   grad_out   (R, ph, pw, C) channels-last, dtype f32 or bf16
 *   grad_feats (host) array of num_levels device pointers to (N, H_l, W_l, C) maps of the SAME dtype, overwritten
 *   num_levels == 1: plain ROIAlign with spatial_scale / aligned; > 1: ROIPooler level rule, scales 2^-(min_level + l)
 * Returns DGX_ERR_UNSUPPORTED for C % 8 != 0, C < 64, 256 % (C/8) != 0, ph or pw > 16 or pointers not 16-byte aligned
 * (callers then use the scatter forms). */
int dgx_roi_pooler_bwd_gather(const void* grad_out, void* const* grad_feats, const int* Hs, const int* Ws,
                              int num_levels, int min_level, float spatial_scale, int aligned, const float* rois,
                              int N, int C, int R, int ph, int pw, int sampling_ratio, int dtype, void* stream);
/* The same with accumulate != 0: the maps already hold a gradient of the same features (autograd's sum over the consumers of
 * one FPN level -- D2/modeling/roi_heads/cascade_rcnn.py:137-160 pools them once per stage, the mask head once more) and this
 * pooling's contribution is ADDED: fp32 sum, one rounding into the map's dtype.  grad_scale multiplies this pooling's
 * contribution (D2/modeling/roi_heads/cascade_rcnn.py:20-28 _ScaleGradient between pooler and box head; 1 otherwise). */
int dgx_roi_pooler_bwd_gather_accum(const void* grad_out, void* const* grad_feats, const int* Hs, const int* Ws,
                                    int num_levels, int min_level, float spatial_scale, int aligned, const float* rois,
                                    int N, int C, int R, int ph, int pw, int sampling_ratio, int accumulate, float grad_scale,
                                    int dtype, void* stream);

/* GT-mask crop for the mask loss: ROIAlign(S x S, scale 1, ratio 0, aligned) on uint8/bool masks
 * and `>= 0.5`, without materialising the fp32 mask.  Replaces BitMasks.crop_and_resize,
 * D2/structures/masks.py:189-220.
 *   masks u8 (M, H, W);  boxes f32 (R,4);  mask_idx i32 (R) row of `masks` for each box;
 *   out u8 (R, S, S) in {0,1}. */
int dgx_mask_crop(const uint8_t* masks, const float* boxes, const int32_t* mask_idx, uint8_t* out,
                  int M, int H, int W, int R, int S, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Greedy NMS on score-sorted boxes.  Replaces torchvision.ops.nms behind batched_nms
 * (D2/layers/nms.py:9-20, CN/modeling/layers/ml_nms.py:26).
 *   boxes  f32 (n,4) ALREADY SORTED by descending score (stable);   iou_thr: suppress when IoU > thr
 *   mask   u64 workspace, n * ceil(n/64) words
 *   keep   u8 (n) out: 1 = kept;  num_keep i32 (1) out */
int dgx_nms_sorted(const float* boxes, int n, float iou_thr, uint64_t* mask, uint8_t* keep,
                   int32_t* num_keep, void* stream);
int64_t dgx_nms_workspace_words(int n);

/* The same NMS for B images in one launch with DEVICE-resident candidate counts (the reference reads the
 * counts back to the host between the score filter and nms, CN/modeling/dense_heads/centernet.py:690-737).
 *   boxes   f32 (B, K, 4) per image sorted by descending score; rows >= n_valid[b] are ignored
 *   scores  f32 (B, K) the sorted scores, or NULL; only used for the tie rule of max_keep
 *   n_valid i32 (B)
 *   max_keep: 0 = keep every survivor; else the sweep stops once max_keep boxes are kept, except that
 *            survivors tied with the max_keep-th kept score are kept too (centernet.py:727-731 keeps
 *            every survivor with score >= the k-th score)
 *   mask    u64 workspace, dgx_nms_batched_workspace_words(B, K) words
 *   keep_idx i32 (B, cap) out: indices (in the sorted order) of the kept boxes, ascending, -1 padded
 *   num_keep i32 (B) out (<= cap) */
int64_t dgx_nms_batched_workspace_words(int B, int K);
int dgx_nms_batched(const float* boxes, const float* scores, const int32_t* n_valid, int B, int K,
                    float iou_thr, int max_keep, uint64_t* mask, int32_t* keep_idx, int cap,
                    int32_t* num_keep, void* stream);

/* pairwise IoU + Matcher in one pass (no M x N matrix): for every proposal the best GT and the
 * label from the thresholds.  D2/structures/boxes.py:310-357 + D2/modeling/matcher.py:62-104
 * (single threshold, labels [0,1], no low-quality matches), callers
 * DG/divergen/modeling/roi_heads/detic_roi_heads.py:136-190,273-307.
 *   gt f32 (M,4), props f32 (N,4) -> matched_idx i64 (N), matched_label i8 (N), max_iou f32 (N) (optional)
 * M == 0: idx 0, label 0. */
int dgx_iou_match(const float* gt, int M, const float* props, int N, float thr,
                  int64_t* matched_idx, int8_t* matched_label, float* max_iou, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CenterNet proposal decoding around the top-k / sort (dgx_topk_index_rows / dgx_sort_rows_desc below; torch's in rounds 1-5) and dgx_nms_batched (centernet.py:627-737 `predict_instances`,
 * `predict_single_level`, `nms_and_topK`).  Level geometry as for dgx_centernet_targets (HOST arrays level_hw (L,2), strides (L));
 * per-level maps are channels-last (B, h, w, pixel_stride) of dtype f32 | bf16, HOST arrays of L device pointers; M = sum h*w.
 *   dgx_centernet_scores:   scores f32 (B, M) (levels concatenated per image) = sigmoid(logit[channel]) where > thr, else -1;
 *                           n_valid i32 (B) cleared.
 *   dgx_centernet_decode:   cand_idx i64 (B, Kc) locations in [0, M) -> boxes f32 (B, Kc, 4) = (gx - s r0, gy - s r1,
 *                           max(gx + s r2, x0 + .01), max(gy + s r3, y0 + .01)), (gx, gy) = (x s + s/2, y s + s/2), r = the 4
 *                           regression channels from `reg_channel`; out_scores = sqrt(score) where score > thr else -1;
 *                           n_valid[b] += number of candidates above thr.
 *   dgx_centernet_finalize: keep_idx i32 (B, cap) / num_keep i32 (B) from dgx_nms_batched over the score-sorted candidates
 *                           (B, K) -> out_boxes (B, cap, 4), out_scores (B, cap), out_valid u8 (B, cap); rows >= num_keep zeroed. */
int dgx_centernet_scores(const void* const* hm_levels, int hm_pixel_stride, int hm_channel, const int32_t* level_hw,
                         const int32_t* strides, int L, int B, float thr, float* scores, int32_t* n_valid, int dtype, void* stream);
int dgx_centernet_decode(const void* const* reg_levels, int reg_pixel_stride, int reg_channel, const int32_t* level_hw,
                         const int32_t* strides, int L, int B, const int64_t* cand_idx, int Kc, const float* scores, float thr,
                         float* boxes, float* out_scores, int32_t* n_valid, int dtype, void* stream);
/* boxes (B, K, 4) f32, order (B, K) i64 (the permutation torch.sort returns for the candidates' scores) -> out[b][k] =
 * boxes[b][order[b][k]]: the gather of centernet.py:704-712 (`boxlist = boxlist[keep]` behind the score sort). */
int dgx_gather_boxes(const float* boxes, const int64_t* order, int B, int K, float* out, void* stream);
/* The two selections of the proposal decode as own kernels (round 6; they were torch.topk = one workgroup per row, 115-150 us, and a
 * rocprim segmented sort, ~95 us, on the step's pre-sync critical path).
 *   dgx_topk_index_rows   per (image b, level l): the positions of the k largest of scores[b * row_stride + level_off[l] .. + level_n[l])
 *                         -- centernet.py:713-717 `per_box_cls.topk(per_pre_nms_top_n, sorted=False)` -- written to
 *                         out[b * out_row_stride + l * k ..] as level_off[l] + position, in ASCENDING position order; ties at the k-th
 *                         value go to the lowest positions (torch's set).  level_off / level_n: device arrays; level_n_host: the same
 *                         sizes on the host (every one >= k and <= 32 768).
 *   dgx_sort_rows_desc    out_vals / out_order (int64) = torch.sort(in (B, K), dim=1, descending=True, stable=True), K <= 16 384
 *                         (ml_nms' score order, centernet.py:739-768 -> D2/layers/nms.py). */
int dgx_topk_index_rows(const float* scores, int64_t row_stride, int B, const int32_t* level_off, const int32_t* level_n,
                        const int32_t* level_n_host, int nlev, int k, int64_t* out, int64_t out_row_stride, void* stream);
int dgx_sort_rows_desc(const float* in, int B, int K, float* out_vals, int64_t* out_order, void* stream);
int dgx_centernet_finalize(const float* sorted_boxes, const float* sorted_scores, const int32_t* keep_idx, const int32_t* num_keep,
                           int B, int K, int cap, float* out_boxes, float* out_scores, uint8_t* out_valid, void* stream);

/* The tail of CenterNetHead.forward and CenterNet's output flattening over ALL levels, one launch each way
 * (CN/modeling/dense_heads/centernet_head.py:113-131: `agn_hm(bbox_tower)`, `F.relu(self.scales[l](bbox_pred(bbox_tower)))`;
 * centernet.py:179-235: per level (B, C, h, w) -> (B h w, C), levels stacked).  Level l: x bf16 (rows_l, C) channels-last, the
 * grouped predictor output (channel 0 = heat-map logit, 1..4 = regression, C a multiple of 8), scale f32 (1) DEVICE pointer
 * (scales[l].scale).  Forward: reg f32 (M, 4) = relu(float(x[1..4]) * scale_l), hm f32 (M) = float(x[0]), M = sum rows_l.
 * Backward: dx bf16 (rows_l, C) = [g_hm, g_reg * (reg > 0) * scale_l, zeros], d_scale f32 (n); workspace:
 * dgx_centernet_head_outputs_bwd_workspace_floats(...) floats.  Bit-reproducible (fixed partial order). */
typedef struct dgx_head_level { const void* x; void* dx; const float* scale; int rows; } dgx_head_level;
int dgx_centernet_head_outputs(const dgx_head_level* levels, int n, int C, float* reg, float* hm, void* stream);
int64_t dgx_centernet_head_outputs_bwd_workspace_floats(const dgx_head_level* levels, int n, int C);
int dgx_centernet_head_outputs_bwd(const dgx_head_level* levels, int n, int C, const float* g_reg, const float* g_hm,
                                   float* d_scale, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Proposal labelling and sampling of the RoI heads for the whole batch (detic_roi_heads.py:273-307
 * `label_and_sample_proposals`; D2 proposal_utils.py:126-196, boxes.py:334-357, matcher.py:62-104, sampling.py:9-54).
 * dgx_roi_label: one workgroup per image.  Rows of image b = its K fixed-length proposals (prop f32 (B,K,4), valid u8 (B,K)
 *   or NULL) followed, when append_gt, by its n_b ground-truth boxes (gt_boxes f32 (G,4), gt_classes i64 (G), gt_offsets i32
 *   (B+1) on the device; max_gt = max n_b).  Per row: matched_idx i32 (B,Nmax) = best ground-truth box of the image (first
 *   maximum; 0 without ground truth) and labels i64 (B,Nmax) = its class when IoU >= iou_thr, num_classes (background)
 *   otherwise, -1 for invalid proposals and for the padding up to Nmax >= K + max_gt.  pos_idx / neg_idx i32 (B,Nmax): the rows
 *   with a foreground / background label in index order (what `nonzero` of the two masks returns), counts i32 (B,2) their
 *   lengths.  IoU float sequence = dgx_iou_match.
 * dgx_roi_gather: the sampled rows of the batch, image after image, foreground first: row t of image b comes from
 *   pos_idx[b][perm_pos[b][t]] (t < num_pos[b]) or neg_idx[b][perm_neg[b][t - num_pos[b]]]; perm_* (HOST arrays of B device
 *   pointers, i64) are the leading entries of the caller's random permutations (torch.randperm: the reference's generator).
 *   Outputs (R = sum num_pos + num_neg rows): boxes f32 (R,4), classes i64, matched ground-truth box f32 (R,4) (the row's own
 *   box for an image without ground truth) and index-within-image i64, instance_source i64 (0 without ground truth / gt_src
 *   NULL; out_src may be NULL), objectness logit f32 (logits (B,K) for proposals, gt_logit for ground-truth rows; may be NULL).
 *   B <= 16. */
int dgx_roi_label(const float* prop, const uint8_t* valid, int B, int K, const float* gt_boxes, const int64_t* gt_classes,
                  const int32_t* gt_offsets, int max_gt, float iou_thr, int num_classes, int append_gt, int Nmax,
                  int32_t* matched_idx, int64_t* labels, int32_t* pos_idx, int32_t* neg_idx, int32_t* counts, void* stream);
int dgx_roi_gather(int B, int K, int Nmax, const int64_t* const* perm_pos, const int64_t* const* perm_neg, const int* num_pos,
                   const int* num_neg, const float* prop, const float* logits, const float* gt_boxes, const int64_t* gt_src,
                   const int32_t* gt_offsets, float gt_logit, const int32_t* matched_idx, const int64_t* labels,
                   const int32_t* pos_idx, const int32_t* neg_idx, float* out_boxes, int64_t* out_classes, float* out_gt_boxes,
                   int64_t* out_gt_index, int64_t* out_src, float* out_logits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CenterNet dense target assignment for one batch, no M x N temporaries.
 * CN/modeling/dense_heads/centernet.py:338-436 (+ :505-530, :551-562, :576-592).
 *   gt_boxes f32 (sum n_i, 4); gt_offsets i32 (B+1) prefix offsets per image
 *   level_hw i32 (L,2) (h,w) per level; strides i32 (L); soi f32 (L,2) size ranges -- these three
 *   are small HOST arrays (configuration, not data)
 *   reg_targets f32 (M*B, 4) and heatmap f32 (M*B) in the reference's level-major layout
 *   (level, image, y, x); reg already divided by stride, -1e8/stride where ignored. */
int dgx_centernet_targets(const float* gt_boxes, const int32_t* gt_offsets, int B,
                          const int32_t* level_hw, const int32_t* strides, const float* soi, int L,
                          float delta, float min_radius, float* reg_targets, float* heatmap,
                          void* stream);
/* The positive-location list of the CenterNet losses (centernet.py:439-483 `_get_label_inds`), fixed length: for every box n
 * (of `total` = sum n_i, images concatenated) and level l:  ind[n * L + l] = flat index of the location holding the box centre in
 * the level-major layout above, cared[n * L + l] = 1 when the box's half diagonal lies in the level's size range (the
 * reference's list is ind[cared]; the loss kernel takes both, so no shape depends on the data). */
int dgx_centernet_label_inds(const float* gt_boxes, const int32_t* gt_offsets, int B, int total, const int32_t* level_hw,
                             const int32_t* strides, const float* soi, int L, int64_t* ind, uint8_t* cared, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Instance copy-paste compositor ('basic' blend).  Replaces the per-paste numpy passes of
 * InstPool._cat_a_new_image / _copy_paste / get_bboxes / blend_image:
 * DG/divergen/data/custom_build_copypaste_mapper.py:488-566, :79-92;
 * DG/divergen/data/transforms/custom_cp_method.py:5-9.
 *   image   u8 (3,H,W) in/out;  masks u8 (n0,H,W) original instance masks
 *   boxes0  f32 (n0,4) boxes of the original instances as the mapper holds them BEFORE pasting
 *   K pastes, applied in order 0..K-1: src_rgba u8 concatenated (h_k, w_k, 4) patches,
 *   src_desc i32 (K,5) = (byte offset into src_rgba, h, w, x0, y0) (x0,y0 may be negative)
 * outputs (object i < n0 = original i, object n0+k = paste k):
 *   out_masks u8 (n0+K, H, W) final mask of every object (occluded pixels cleared)
 *   out_boxes f32 (n0+K, 4), out_valid u8 (n0+K): 1 if the object survives the reference's
 *   occlusion filter (|box delta| <= 10 in all coords, or area > 300) at every step and, for
 *   pastes, has a non-empty footprint.  The host compacts by out_valid (order preserved).
 *   stats i32 workspace of (n0+K)*(K+1)*5 + 3 + H*W words, 16-byte aligned (per-object histograms, then -- on the next
 *   16-byte boundary -- the per-pixel cover words).
 *   K <= 31. */
int dgx_copy_paste(uint8_t* image, const uint8_t* masks, const float* boxes0, int n0, int H, int W,
                   const uint8_t* src_rgba, const int32_t* src_desc, int K, uint8_t* out_masks,
                   float* out_boxes, uint8_t* out_valid, int32_t* stats, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused parameter update over a flat arena: per-element gradient value clip, AdamW, EMA lerp of
 * the PRE-step weights (the reference updates the EMA before optimizer.step: DG/train_net.py:262-284),
 * optional bf16 shadow copy of the new weights.  Replaces the ~400 per-tensor param groups of
 * DG/divergen/custom_solver.py:19-77 + D2/solver/build.py:24-75 + DG/divergen/ema.py:49-58.
 *   p,g,m,v,ema f32 (n); lr_scale f32 (n_seg) per-segment lr multiplier, seg_end i64 (n_seg)
 *   exclusive end offsets (or both NULL);  grad_scale multiplies g first (1/loss_scale, 1/world);
 *   found_inf i32 (1) optional: if *found_inf != 0 the step is skipped (GradScaler semantics);
 *   p_bf16 optional. */
int dgx_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                       int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       float clip_value, float grad_scale, int step, float ema_decay,
                       const float* lr_scale, const int64_t* seg_end, int n_seg,
                       const int32_t* found_inf, void* stream);
/* The same step with the gradient additionally multiplied by a DEVICE scalar: the full-model norm-clip coefficient of
 * dgx_clip_coef_f32 (DG/divergen/custom_solver.py:46-60 wraps AdamW as well as SGD in FullModelGradientClippingOptimizer);
 * grad_scale_dev may be NULL (= dgx_adamw_ema_step). */
int dgx_adamw_ema_step_scaled(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float clip_value, float grad_scale, const float* grad_scale_dev, int step, float ema_decay,
                              const float* lr_scale, const int64_t* seg_end, int n_seg,
                              const int32_t* found_inf, void* stream);

/* The 'SGD' branch of build_custom_optimizer (DG/divergen/custom_solver.py:64-68 = torch.optim.SGD with momentum / nesterov and one
 * weight decay for every group) over the same arena, with the EMA lerp, bf16 shadow, per-segment lr multipliers and found_inf of
 * the AdamW step:  d = clip(g * grad_scale [* *grad_scale_dev]) + wd * p;  buf = (step == 1) ? d : momentum * buf + d;
 * d = nesterov ? d + momentum * buf : buf;  p -= lr * d.   buf f32 (n), may be NULL when momentum == 0. */
int dgx_sgd_ema_step(float* p, const float* g, float* buf, float* ema, void* p_bf16, int64_t n, float lr, float momentum,
                     int nesterov, float weight_decay, float clip_value, float grad_scale, const float* grad_scale_dev,
                     int step, float ema_decay, const float* lr_scale, const int64_t* seg_end, int n_seg,
                     const int32_t* found_inf, void* stream);
/* FullModelGradientClippingOptimizer (custom_solver.py:46-60: torch.nn.utils.clip_grad_norm_ over all parameters) without a host
 * read: out2[0] = min(1, max_norm / (||g * grad_scale||_2 + 1e-6)), out2[1] = the norm; out2 is what dgx_sgd_ema_step takes as
 * grad_scale_dev.  workspace: dgx_clip_coef_workspace_floats() floats.  Deterministic (fixed partial order, double fold). */
/* g[first .. first + count) = 0 for n ranges, ranges i64 (n, 2) DEVICE, first / count multiples of 4, count <= 65536: the gradient
 * arena minus the segments their first writer overwrites (`optimizer.zero_grad()` of train_net.py:248-304 without touching what the
 * weight-gradient launches rewrite anyway). */
int dgx_zero_ranges_f32(float* g, const int64_t* ranges, int64_t n, void* stream);
int64_t dgx_clip_coef_workspace_floats(void);
int dgx_clip_coef_f32(const float* g, int64_t n, float grad_scale, float max_norm, float* workspace, float* out2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 / pad 1 / stride 1|2 convolution as im2col + library GEMM on channels-last tensors.
 * Replaces the cuDNN (here: MIOpen) calls behind nn.Conv2d at D2/modeling/backbone/fpn.py:126-154,
 * CN/modeling/backbone/fpn_p5.py:30-33, CN/modeling/dense_heads/centernet_head.py:141-162 and
 * D2/modeling/roi_heads/mask_head.py:209-284.
 *   x   (N,H,W,C)                 col  (N*Ho*Wo, 9*C), tap-major (ky,kx,c)
 *   col2im is the adjoint (gather form, fp32 accumulation): dx (N,H,W,C) <- dcol (N*Ho*Wo, 9*C). */
int dgx_im2col3x3(const void* x, void* col, int N, int H, int W, int C, int stride, int dtype,
                  void* stream);
int dgx_col2im3x3(const void* dcol, void* dx, int N, int H, int W, int C, int stride, int dtype,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient of a Linear / im2col convolution:  gw[Nn][Kk] = beta*gw + dy^T x  (fp32 out),
 * dy bf16 (M,Nn), x bf16 (M,Kk), both row-major; M-split MFMA kernel + slab reduction.  Replaces the
 * `grad_output.t().mm(input)` of nn.Linear's backward (call sites as dgx_window_attention_*: qkv /
 * proj / fc1 / fc2 of swintransformer.py:118-120,35-37, box_head.py fc1/fc2) where the GEMM library
 * runs the long-K transposed shape at a fraction of its rate.  Nn, Kk multiples of 8.
 * workspace: dgx_wgrad_workspace_bytes(M,Nn,Kk) bytes of device scratch. */
int64_t dgx_wgrad_workspace_bytes(int M, int Nn, int Kk);
int dgx_linear_wgrad(const void* dy, const void* x, float* gw, int M, int Nn, int Kk, float beta,
                     void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused LayerNorm (+ bf16 cast + optional window gather).  Replaces norm1 + F.pad + torch.roll +
 * window_partition (swintransformer.py:213-233) and norm2 (:255) with one pass each way.
 *   x f32 or bf16 (x_dtype) (T,C) with T = B*H*W;  ws > 0: y bf16 is written in window order (B*nW, ws*ws, C), zero rows
 *   for padding tokens;  ws == 0: y bf16 (T,C).  mean/rstd f32 (T) are saved for backward.
 * Backward: dy bf16 in the same order as y -> dx (x_dtype) (T,C) written; dgamma/dbeta f32 (C) ACCUMULATED;
 *   part: f32 scratch of dgx_layernorm_bwd_blocks(T)*2*C.  C % 4 == 0, C <= 1536 for backward.
 *   dres (backward, nullable, dtype of x, may alias dx): gradient arriving on the residual branch that
 *   bypasses the norm (x + f(LN(x)), swintransformer.py:254-255); dx = dres + LN-gradient in one pass.
 *   ws < 0 (here, in dgx_layernorm_bwd_emit's ews and in DgxGemmEpilogue.ws): window size -ws with the rows in COMPACT window order
 *   (csrc/winmap.h; see dgx_window_attention_fwd_compact): rows 0 .. T-1 are the real tokens in (image, window, token) order.  The
 *   forward still writes B*nW*ws*ws rows -- the padding tokens' zero rows follow the real ones (the qkv weight gradient's operand);
 *   backward reads / emits the T real rows only.  Needs H >= shift and W >= shift. */
int dgx_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                      float* rstd, int64_t T, int C, float eps, int B, int H, int W, int ws, int shift,
                      int x_dtype, void* stream);
int dgx_layernorm_bwd_blocks(int64_t T);
int dgx_layernorm_bwd(const void* dy_bf16, const void* x, const float* mean, const float* rstd,
                      const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta, float* part,
                      int64_t T, int C, int B, int H, int W, int ws, int shift, int x_dtype, void* stream);
/* dgx_layernorm_bwd that ALSO writes the operand of the next GEMM of the backward pass (what a separate dgx_residual_bwd pass over
 * dx produces: the reference's roll / window_partition / DropPath backward of swintransformer.py:216-255):
 *   emit_bf16[row(tok)] = bf16(emit_scale[b] * dx[tok]),  b = tok / (eH * eW),  emit_scale f32 (eB) or NULL,
 * rows in token order (ews == 0) or in window order with zero rows for the padding tokens (ews > 0, eshift: the block's shift).
 * eB * eH * eW == T.  The value is formed from dx as stored, so it equals the two-kernel result bit for bit. */
int dgx_layernorm_bwd_emit(const void* dy_bf16, const void* x, const float* mean, const float* rstd, const float* gamma,
                           const void* dres, void* dx, float* dgamma, float* dbeta, float* part, int64_t T, int C, int B, int H,
                           int W, int ws, int shift, int x_dtype, void* emit_bf16, const float* emit_scale, int eB, int eH,
                           int eW, int ews, int eshift, void* stream);
/* dgx_layernorm_bwd with dgamma = dbeta = NULL leaves its per-block partial sums in `part`; this folds the partial rows of TWO
 * such calls (same T and C: norm2 and norm1 of one Swin block) into their parameter gradients in one launch, in the summation
 * order of the single-norm second stage (bit-identical results). */
int dgx_layernorm_param_reduce2(const float* part_a, float* dgamma_a, float* dbeta_a, const float* part_b, float* dgamma_b,
                                float* dbeta_b, int64_t T, int C, void* stream);
/* ... and of up to 16 norms at once (the two norms of up to eight consecutive Swin blocks of one stage: same T and C): n pointer triples,
 * per norm the summation order of the single-norm kernel. */
int dgx_layernorm_param_reduce_n(const float* const* parts, float* const* dgammas, float* const* dbetas, int n, int64_t T, int C,
                                 void* stream);
/* LayerNorm with an fp32 result (PatchEmbed.norm, swintransformer.py:440-442; under autocast nn.LayerNorm returns fp32
 * and that tensor is the stage-0 residual stream).  x f32|bf16 (T,C) -> y f32 (T,C); backward: dy f32, dx in x's dtype,
 * ADDS into dgamma / dbeta; part as for dgx_layernorm_bwd.  C % 4 == 0, C <= 768 for backward. */
int dgx_layernorm_f32out_fwd(const void* x, const float* gamma, const float* beta, float* y, float* mean,
                             float* rstd, int64_t T, int C, float eps, int x_dtype, void* stream);
int dgx_layernorm_f32out_bwd(const float* dy, const void* x, const float* mean, const float* rstd,
                             const float* gamma, void* dx, float* dgamma, float* dbeta, float* part, int64_t T,
                             int C, int x_dtype, void* stream);
/* PatchMerging front half (swintransformer.py:272-298: pad to even H/W, x0|x1|x2|x3 concatenation of the 2x2
 * neighbourhood, LayerNorm(4*C0)) in one pass.  x (B,H,W,C0) f32|bf16 -> y bf16 (B*H2*W2, 4*C0), H2 = ceil(H/2);
 * mean, rstd f32 (B*H2*W2).  Backward writes dx (B,H,W,C0) in x's dtype, every element once, and ADDS into
 * dgamma / dbeta (4*C0);  part: f32 scratch of dgx_layernorm_bwd_blocks(B*H2*W2)*2*4*C0.  C0 % 4 == 0, 4*C0 <= 3072. */
int dgx_patch_merge_ln_fwd(const void* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                           float* rstd, int B, int H, int W, int C0, float eps, int x_dtype, void* stream);
int dgx_patch_merge_ln_bwd(const void* dy_bf16, const void* x, const float* mean, const float* rstd,
                           const float* gamma, void* dx, float* dgamma, float* dbeta, float* part, int B, int H,
                           int W, int C0, int x_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Residual + DropPath epilogue of a Swin block: out = x + scale[b] * y, y bf16 in token order (ws == 0)
 * or in window order (ws > 0: window_reverse + roll(+shift) + crop folded into the read).  Replaces
 * swintransformer.py:239-255 (window_reverse, roll, crop, drop_path, add).  x/out dtype f32|bf16;
 * scale f32 (B) or NULL (= 1).  Backward: dy (bf16, same order as y incl. zero padding rows) =
 * scale[b] * g;  the gradient w.r.t. x is g itself (no kernel).  C % 8 == 0. */
int dgx_residual_fwd(const void* x, const void* y_bf16, const float* scale, void* out, int B, int H, int W,
                     int C, int ws, int shift, int x_dtype, void* stream);
int dgx_residual_bwd(const void* g, const float* scale, void* dy_bf16, int B, int H, int W, int C, int ws,
                     int shift, int g_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FPN top-down step (D2/modeling/backbone/fpn.py:139-145: F.interpolate(prev, scale_factor=2, mode="nearest") followed by
 * `lateral + top_down`) on NHWC maps: out (N,H,W,C) = lat (N,H,W,C) + top (N,H/2,W/2,C)[y>>1][x>>1], fp32 sum, one rounding.
 * Backward: the lateral's gradient is g itself; gtop (N,H/2,W/2,C) = the sum of g over each 2x2 block (fp32, one rounding) -- what
 * autograd computes through upsample_nearest2d_backward.  dtype DGX_BF16 / DGX_F32; C % 8 == 0, H and W even, 16-byte aligned pointers
 * (DGX_ERR_UNSUPPORTED otherwise). */
int dgx_upsample2x_add_fwd(const void* lat, const void* top, void* out, int N, int H, int W, int C, int dtype, void* stream);
int dgx_upsample2x_add_bwd(const void* g, void* gtop, int N, int H, int W, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Grouped form of dgx_linear_wgrad: the weight gradients of several Linear layers (the four of a Swin
 * block: qkv/proj/fc1/fc2, swintransformer.py:101-108,36-46) in ONE launch, so that large output tiles
 * fill the GPU with a small M-split.  n <= 12 problems -- or up to 32 when dgx_wgrad_grouped_form(problems, n) == 1: groups
 * with enough 256x192 output tiles to fill the chip for whole rounds (the Linears of ~7 Swin-L stage-2 blocks) run on the
 * persistent loader-wave kernel (wgrad_lw.hip: every tile contracts its whole M, no split, no workspace).  For each problem
 * gw (Nn,Kk) = beta*gw + dy^T x  and, when gb != NULL, the
 * layer's BIAS gradient gb (Nn) = beta*gb + column sums of dy from the same pass (the sum over rows autograd performs for the
 * bias; computed as dy^T 1 on fragments the kernel holds anyway -- ABI version 3 added the field).
 * Nn % 8 == 0, Kk % 8 == 0.  workspace: dgx_wgrad_grouped_workspace_bytes(problems, n) bytes. */
typedef struct dgx_wgrad_problem {
    const void* dy;   /* bf16 (M, Nn) row-major */
    const void* x;    /* bf16 (M, Kk) row-major */
    float* gw;        /* f32 (Nn, Kk) */
    int M, Nn, Kk;
    float* gb;        /* f32 (Nn) or NULL */
} dgx_wgrad_problem;
int64_t dgx_wgrad_grouped_workspace_bytes(const dgx_wgrad_problem* problems, int n);
int dgx_linear_wgrad_grouped(const dgx_wgrad_problem* problems, int n, float beta, void* workspace,
                             void* stream);
/* 1 when dgx_linear_wgrad_grouped would run this group on the loader-wave kernel (n <= 32 allowed), 0 when on the split-M
 * kernel (n <= 12): the caller sizes its launches with it. */
int dgx_wgrad_grouped_form(const dgx_wgrad_problem* problems, int n);

/* ---------------------------------------------------------------------------------------------
 * Bias gradient of nn.Linear (the sum over rows autograd performs for the bias of qkv/proj/fc1/fc2,
 * swintransformer.py:36-46,101-108): out[n] = beta*out[n] + sum_m dy[m][n]; dy bf16 (M, N) row-major,
 * out fp32.  N % 8 == 0.  workspace: dgx_colsum_workspace_bytes(M, N) bytes. */
int64_t dgx_colsum_workspace_bytes(int M, int N);
int dgx_colsum_bf16(const void* dy_bf16, float* out, int M, int N, float beta, void* workspace, void* stream);
/* grouped form: n <= 8 (dy, out, M, N) problems -- the bias gradients of a whole block -- in two launches */
typedef struct dgx_colsum_problem {
    const void* dy;   /* bf16 (M, N) row-major */
    float* out;       /* f32 (N) */
    int M, N;
} dgx_colsum_problem;
int64_t dgx_colsum_grouped_workspace_bytes(const dgx_colsum_problem* problems, int n);
int dgx_colsum_grouped(const dgx_colsum_problem* problems, int n, float beta, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ ReLU) over channels-last bf16 activations with 8 channels per group: the Conv -> GroupNorm(32)
 * -> ReLU unit of the CenterNet tower (CN/modeling/dense_heads/centernet_head.py:52-75; torch.nn.GroupNorm
 * semantics: biased variance over (HW, C/G), eps inside the sqrt).  x, y, dy, dx bf16 (N, HW, C); C == 8*G.
 * mean / rstd f32 (N*G) saved by forward.  Backward ACCUMULATES into dgamma / dbeta (f32 C);
 * scratch / part: f32, dgx_groupnorm_scratch_floats(N, HW, G) elements (both directions). */
int64_t dgx_groupnorm_scratch_floats(int N, int HW, int G);
int dgx_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      float* scratch, int N, int HW, int C, int G, float eps, int relu, void* stream);
int dgx_groupnorm_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, void* dx, float* dgamma, float* dbeta, float* part, int N, int HW,
                      int C, int G, int relu, void* stream);
/* The same GroupNorm (+ ReLU) over SEVERAL tensors at once -- the FPN levels of one CenterNet tower layer share its weights
 * (centernet_head.py:141-162 runs the tower level by level): n <= 8 items, one launch per pass for all of them, the bias / weight
 * gradients accumulated over the items in list order.  `out` = y (forward) | dx (backward); `dy` unused in the forward;
 * `scratch` = dgx_groupnorm_scratch_floats(N, HW, G) floats per item, the forward's for the forward, a fresh one for the backward. */
typedef struct dgx_gn_item {
    const void* x;      /* (N, HW, C) bf16 */
    const void* dy;     /* (N, HW, C) bf16, backward only */
    void* out;          /* (N, HW, C) bf16 */
    float* mean;        /* (N * G) */
    float* rstd;        /* (N * G) */
    float* scratch;
    int N, HW;
} dgx_gn_item;
int dgx_groupnorm_fwd_multi(const dgx_gn_item* items, int n, const float* gamma, const float* beta, int C, int G, float eps, int relu,
                            void* stream);
int dgx_groupnorm_bwd_multi(const dgx_gn_item* items, int n, const float* gamma, const float* beta, float* dgamma, float* dbeta, int C,
                            int G, int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Detic box-head losses of one cascade stage in one pass: sigmoid cross-entropy with per-class weights
 * (federated loss / zero-category mask, detic_fast_rcnn.py:271-304), class-agnostic L1 box regression on the
 * Box2Box deltas of the foreground rows [with instance_source == 0] (:160-235, box_regression.py:43-118) and the
 * classification statistics of D2 fast_rcnn.py:88-114 -- forward values AND gradients.
 *   logits (R, C+1), deltas (R, 4): f32 or bf16 (dtype);  gt_classes i64 (R) in [0, C] (C = background, <0 ignored
 *   for the box term);  class_w f32 (C) or NULL (= 1);  prop, gtb f32 (R,4);  src i64 (R) or NULL
 *   rows with gt_classes < 0 are "ignore" rows: no loss, no gradient, not counted in the normalisers
 *   dlogits (R, C+1) same dtype as logits  = rows * d loss_cls / d logits   (multiply by out16[14] = 1/rows)
 *   dsign   f32 (R, 4)                     = sign(deltas - target) on selected rows, else 0
 *   out16   f32 [16]: [8] loss_cls, [9] loss_box_reg, [10] 1/max(4*rows,1) (scale of dsign in backward),
 *                     [11] cls_accuracy, [12] fg_cls_accuracy, [13] false_negative, [14] 1/rows, [0..7] the raw sums
 *   part    f32 scratch R*8 */
int dgx_detic_losses(const void* logits, const void* deltas, const int64_t* gt_classes, const float* class_w,
                     const float* prop, const float* gtb, const int64_t* src, int R, int C, float wx, float wy,
                     float ww, float wh, void* dlogits, float* dsign, float* out16, float* part, int dtype,
                     void* stream);
/* The same with row strides (elements), for the JOINT output of the predictor pair cls_score | bbox_pred
 * (detic_fast_rcnn.py:437-466 runs two Linears on the same features; here their rows are one GEMM operand):
 *   logits = y, ld_logits = ld;  deltas = y + (C+1), ld_deltas = ld   with y (R, ld), ld >= C + 5
 *   grad_cols > C + 1: row r of dlogits (ld_dlogits) additionally gets dsign[r] in columns [C+1, C+5) and zeros in
 *   [C+5, grad_cols) -- the K-padded operand of the pair's input- and weight-gradient GEMMs, still unscaled;
 *   dgx_detic_grad_scale then multiplies columns [0, C+1) by g_cls * out16[14] and [C+1, C+5) by g_box * out16[10]
 *   (g_cls, g_box: device scalars = the gradients of the two loss values), in place. */
int dgx_detic_losses_strided(const void* logits, int64_t ld_logits, const void* deltas, int64_t ld_deltas,
                             const int64_t* gt_classes, const float* class_w, const float* prop, const float* gtb,
                             const int64_t* src, int R, int C, float wx, float wy, float ww, float wh, void* dlogits,
                             int64_t ld_dlogits, int grad_cols, float* dsign, float* out16, float* part, int dtype,
                             void* stream);
int dgx_detic_grad_scale(void* dy, int64_t ld, int rows, int C, const float* out16, const float* g_cls, const float* g_box,
                         int dtype, void* stream);

/* Federated-loss class set of one cascade stage (DG/divergen/modeling/utils.py:16-28 get_fed_loss_inds, used by
 * detic_fast_rcnn.py:271-304) as a 0/1 mask u8 (C+1): the classes present in gt_classes i64 (R), plus -- when fewer than K are
 * present -- the K - #present classes with the largest prob[c] / expo[c] among the others (prob f32 (C+1) >= 0, the sampling
 * weights with 0 for the background; expo f32 (C+1) = Exponential(1) draws made by the caller's generator: multinomial without
 * replacement IS this top-k, so the set is the reference's for the same generator state).  One workgroup; C + 1 <= 8192. */
int dgx_fed_class_mask(const int64_t* gt_classes, int R, const float* prob, const float* expo, int C, int K, uint8_t* mask,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hand-over between two cascade stages for all images of the batch in one launch
 * (detic_roi_heads.py:192-244 `_forward_box`, `_create_proposals_from_boxes`, :136-190 `_match_and_label_boxes`):
 * boxes = clip(apply_deltas(deltas, prop)) (box_regression.py:76-118, class-agnostic), valid_out = valid_in & nonempty,
 * then pairwise IoU + Matcher (threshold iou_thr, labels {0,1}) against the image's own ground truth and the gather
 * of the match: out_cls = gt class | num_classes (background) | -1 (row dropped by the reference: empty box),
 * out_gtb = matched gt box (zeros without ground truth), out_src = instance_source of a foreground match else 0.
 *   prop f32 (R,4); deltas (R,4) f32|bf16; valid_in u8 (R) or NULL; HOST arrays row0/gt0 (B+1 prefix offsets of the
 *   RoI rows / GT rows of each image), img_h/img_w (B);  gt_* concatenated over images; gt_src may be NULL (then
 *   out_src may be NULL);  num_fg i32 (1) out = foreground matches over the batch.  B <= 32. */
int dgx_cascade_refine(const float* prop, const void* deltas, const uint8_t* valid_in, int B, const int* row0,
                       const int* gt0, const float* img_h, const float* img_w, const float* gt_boxes,
                       const int64_t* gt_classes, const int64_t* gt_src, float iou_thr, int num_classes, float wx,
                       float wy, float ww, float wh, float scale_clamp, float* boxes, uint8_t* valid_out,
                       int64_t* out_cls, float* out_gtb, int64_t* out_src, int32_t* num_fg, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Exact (erf) GELU of the Swin MLP on bf16 activations (swintransformer.py:40-46, nn.GELU()).
 *   dgx_gelu_fwd: y = gelu(x), n elements (n % 8 == 0).
 *   dgx_gelu_bwd_colsum: dx (M, N) = dy * gelu'(x) and, when bias_grad != NULL, bias_grad (N) = beta*bias_grad +
 *   column sums of the bf16 dx (the fc1 bias gradient) from the same pass.  N % 8 == 0;
 *   workspace dgx_gelu_bwd_workspace_bytes(M, N) bytes. */
int dgx_gelu_fwd(const void* x, void* y, int64_t n, void* stream);
int64_t dgx_gelu_bwd_workspace_bytes(int M, int N);
int dgx_gelu_bwd_colsum(const void* dy, const void* x, void* dx, float* bias_grad, int M, int N, float beta,
                        void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CenterNet proposal losses (centernet/modeling/dense_heads/centernet.py:237-314 `losses`; layers/iou_loss.py:10-63 'giou';
 * layers/heatmap_focal_loss.py:51-85).  reg_pred, reg_targets f32 (M,4); hms f32 (M,C) (peak = max over C);
 * logit f32 (M) agnostic heat-map logits; pos_idx i64 (P) locations of the positives, pos_cared u8 (P) or NULL (all).
 * Returns RAW sums in out8 = {sum of regression weights, weighted GIoU sum, neg loss, pos loss, #positives, -, -, -}
 * (pos/neg already multiplied by pos_mul / neg_mul = alpha / 1-alpha) -- the caller all-reduces the normalisers
 * (:243-262) -- and UNSCALED gradients: g_reg (M,4) = d(weighted GIoU sum)/d reg_pred, g_neg (M) = d neg / d logit,
 * g_pos (M) = d pos / d logit (scatter-added; cleared by the call).  part: f32 scratch 3*dgx_centernet_losses_blocks(M). */
int dgx_centernet_losses_blocks(int M);
int dgx_centernet_losses(const float* reg_pred, const float* reg_targets, const float* hms, const float* logit,
                         const int64_t* pos_idx, const uint8_t* pos_cared, int M, int C, int P, int not_norm_reg,
                         float beta, float gamma, float sigmoid_clamp, float ignore_high_fp, float pos_mul,
                         float neg_mul, float* g_reg, float* g_neg, float* g_pos, float* out8, float* part,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluation post-processing.  paste_masks_in_image (D2/layers/mask_ops.py:17-150; called from
 * custom_rcnn.py:265-332 detector_postprocess): the SxS mask probabilities of detection n are sampled bilinearly
 * (F.grid_sample, align_corners=False, zero padding) at every image pixel through its box and thresholded.
 *   masks f32 (N,S,S) in [0,1];  boxes f32 (N,4) in output-image pixels;  out u8 (N,H,W) in {0,1}.
 * dgx_paste_rle evaluates the same bits on the fly and emits, per detection, the COCO run-length encoding of the
 * (H,W) mask (column-major order, first run counts zeros; what pycocotools.mask.encode produces before string
 * compression, D2/evaluation/lvis_evaluation.py:80-97 / coco_evaluation instances_to_coco_json) without the N*H*W tensor:
 *   counts i32 (N, cap) run lengths;  nruns i32 (N): number of runs, or -(needed) if it did not fit cap. */
int dgx_paste_masks(const float* masks, const float* boxes, uint8_t* out, int N, int S, int H, int W,
                    float threshold, void* stream);
int dgx_paste_rle(const float* masks, const float* boxes, int32_t* counts, int32_t* nruns, int N, int S, int H,
                  int W, float threshold, int cap, void* stream);
/* Run lengths of existing bitmasks (bits u8 (N,H,W), non-zero = set), same output contract as dgx_paste_rle: the device
 * form of mask_util.encode(np.array(mask[:, :, None], order="F")) at coco_evaluation.py:400-410. */
int dgx_rle_encode(const uint8_t* bits, int32_t* counts, int32_t* nruns, int N, int H, int W, int cap, void* stream);
/* HOST function: COCO compressed-RLE string of one run-length list (pycocotools maskApi.c rleToString, reached from
 * coco_evaluation.py:406 mask_util.encode).  Returns the length written, or -(needed) if cap is too small. */
int64_t dgx_rle_to_string(const int32_t* counts, int64_t n, char* out, int64_t cap);

/* ---------------------------------------------------------------------------------------------
 * BSGAL gain scoring on flat gradient arenas (BS/bsgal/modeling/meta_arch/custom_rcnn.py).
 * dgx_grad_bank_update replaces update_grad_bank (:1046-1062): bank = bank * a + grad * b in that fp32 order
 *   ("AVERAGE": a = it/(it+1), b = 1/(it+1);  "MOMENTUMm": a = m, b = 1 - m).  bank, grad: f32 (n), 16-byte aligned.
 * dgx_grad_sim replaces compute_grad_sim (:1074-1086) and the two .norm() passes: one pass over both vectors,
 *   out3 f64 = {g1.g2, |g1|^2, |g2|^2},  out4 f32 = {g1.g2, |g1|, |g2|, g1.g2 / (|g1||g2| + 1e-8)}.
 *   workspace: dgx_grad_sim_workspace_bytes(n) bytes.  Deterministic (fixed-order two-stage fp64 reduction). */
int dgx_grad_bank_update(float* bank, const float* grad, int64_t n, float a, float b, void* stream);
int64_t dgx_grad_sim_workspace_bytes(int64_t n);
int dgx_grad_sim(const float* g1, const float* g2, int64_t n, double* out3, float* out4, void* workspace,
                 void* stream);


/* ---------------------------------------------------------------------------------------------
 * bf16 GEMM with fused epilogues: the forward and input-gradient matrix products of every nn.Linear / 1x1 convolution
 * on the path (the reference issues them as F.linear -> cuBLAS):  DG/divergen/modeling/backbone/swintransformer.py:133
 * (qkv), :155 (proj), :40-46 (Mlp fc1 / GELU / fc2), :296 (PatchMerging.reduction); D2/modeling/backbone/fpn.py:126-154
 * (lateral 1x1); D2/modeling/roi_heads/box_head.py:26-98 and DG/divergen/modeling/roi_heads/detic_fast_rcnn.py:437-466 (FCs).
 *
 *   acc[m][n] = sum_k A[m][k] * B[n][k]      A bf16 (M, K) row stride lda, B bf16 (N, K) row stride ldb (elements),
 *                                             fp32 accumulation on MFMA;  K % 8 == 0, N % 8 == 0, lda/ldb % 8 == 0.
 *   y = bf16(acc + bias[n])  (bias bf16 (N) or NULL), then by `mode`:
 *     DGX_EPI_NONE / DGX_EPI_BIAS   c[m][n] = y                                   (ldc)
 *     DGX_EPI_BIAS_GELU             c[m][n] = y;  c2[m][n] = bf16(GELU_erf(y))    (Mlp.fc1 + act, swintransformer.py:41-42)
 *     DGX_EPI_GELU_GRAD             c[m][n] = bf16(y * GELU'(aux[m][n]))          (backward of the above; aux = saved c)
 *     DGX_EPI_RELU_GRAD             c[m][n] = aux[m][n] > 0 ? y : 0               (input gradient carried through the ReLU that
 *                                   follows a Linear / convolution: box_head.py:26-98 fc_relu, mask_head.py:209-284; aux = the
 *                                   saved activation)
 *     DGX_EPI_BIAS_RESIDUAL         out[tok(m)][n] = residual[tok(m)][n] + scale[b(m)] * y
 *         rows m are in WINDOW order when ws > 0 (window_reverse + roll(+shift) + crop folded into tok(m), padding rows
 *         dropped; swintransformer.py:239-255) or token order when ws == 0; residual / out are (B, H*W, N) of
 *         residual_dtype (DGX_F32 | DGX_BF16); scale f32 (B) = per-sample DropPath factor or NULL.
 *         M must equal B*nW*ws*ws (ws > 0) or B*H*W.
 * The input gradient dx = dy W of a Linear is this same call on the TRANSPOSED weight image (K and N swapped).
 */
#define DGX_EPI_NONE 0
#define DGX_EPI_BIAS 1
#define DGX_EPI_BIAS_GELU 2
#define DGX_EPI_BIAS_RESIDUAL 3
#define DGX_EPI_GELU_GRAD 4
#define DGX_EPI_RELU_GRAD 5
typedef struct dgx_gemm_epilogue {
    int mode;
    void* c;              /* bf16 (M, ldc) */
    int64_t ldc;
    const void* bias;     /* bf16 (N) or NULL */
    void* c2;             /* bf16 (M, ldc): DGX_EPI_BIAS_GELU */
    const void* aux;      /* bf16 (M, ldaux): DGX_EPI_GELU_GRAD, DGX_EPI_RELU_GRAD */
    int64_t ldaux;
    const void* residual; /* DGX_EPI_BIAS_RESIDUAL */
    void* out;
    const float* scale;
    int residual_dtype;
    int B, H, W, ws, shift;
    void* workspace;      /* optional fp32 scratch for split-K (few output tiles, long K): S slabs of M*N floats are used when */
    int64_t workspace_bytes; /* they fit, folded by a second launch with the same epilogue; NULL / 0 = never split */
    int relu;             /* DGX_EPI_NONE / DGX_EPI_BIAS: c = max(y, 0) (Conv2d / ConvTranspose2d followed by ReLU, mask_head.py:209-284) */
} dgx_gemm_epilogue;
int dgx_gemm_bf16_nt(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb,
                     const dgx_gemm_epilogue* epilogue, void* stream);
/* Compute units that the library's persistent kernels (one workgroup per CU walking a tile list: the loader-wave GEMM) leave free
 * for work that runs beside them -- the RCCL channels of the overlapped gradient all-reduce (DG/train_net.py:357-362 wraps the model
 * in DistributedDataParallel: same overlap).  Rounded up to a multiple of 8 (one per XCD); 0 = use the whole chip (default). */
void dgx_set_reserved_cus(int n);
int dgx_get_reserved_cus(void);
/* Test / measurement hooks of the GEMM dispatch.  The product path never calls them and the library reads NO environment variable:
 * which kernel form trains the model cannot depend on a user's shell.
 *   dgx_dev_set(key, value)   "gemm_lw" (-1 plan | 0 gemm_nt everywhere | 1 gemm_lw everywhere), "gemm_2wg" (0 = never the
 *                             two-workgroup form), "gemm_tile" (bm * 1000 + bn, 0 = plan), "gemm_splitk" (forced slab count, 0 = plan),
 *                             "wgrad_lw" (1 plan | 0 never the loader-wave weight-gradient form | 2 always), "gemm_k192" (0 = never
 *                             the resident-panel kernel of the K = 192 problems), "reset" (all of them back to the plan).
 *                             DGX_ERR_BAD_ARG for an unknown key.
 *   dgx_gemm_last_form        which kernel the most recent dgx_gemm_bf16_nt / dgx_conv3x3_gemm call launched: returns 0 = gemm_nt,
 *                             1 = gemm_lw (persistent loader-wave form), 2 = gemm_nt's two-workgroups-per-CU form, 3 = gemm_k192
 *                             (K = 192, M >= 32 768: weight panel resident in LDS, one wave per row tile), -1 = none yet;
 *                             tile and split-K slab count through the pointers (NULL = not wanted).  The parity tests assert with
 *                             it that the form the bench times is the form they checked.
 *   dgx_dev_gemm_log(path)    one line "M N K mode bm bn" per GEMM launch into path (NULL / "" closes it); joined with a kernel
 *                             trace by tools/gemm_insitu.py. */
int dgx_dev_set(const char* key, int value);
int dgx_gemm_last_form(int* bm, int* bn, int* splits);
int dgx_dev_gemm_log(const char* path);


/* Transposed twins of the matrix parameters inside a flat bf16 arena (the operand layout dgx_gemm_bf16_nt needs for the
 * input gradient dx = dy W of F.linear; autograd's mm(dy, W) behind swintransformer.py:40-46,133,155,296): for every job,
 * dst[off + c*rows + r] = src[off + r*cols + c].  jobs = DEVICE array of njobs records {int64 off; int32 rows; int32 cols;
 * int64 tile0; int64 cin} sorted by tile0 = number of 64x64 tiles of all earlier jobs; total_tiles = their total.  cin > 0 marks a
 * 3x3 convolution weight stored (Cout, 3, 3, Cin) (rows = Cout, cols = 9 cin, cin % 64 == 0): its twin is the tap-flipped
 * transpose (Cin, 3, 3, Cout) that dgx_conv3x3_gemm takes for the input gradient.  One launch. */
int dgx_transpose_bf16_grouped(const void* src, void* dst, const void* jobs, int njobs, int64_t total_tiles, void* stream);


/* ---------------------------------------------------------------------------------------------
 * 3x3 convolution (pad 1, stride 1) as an IMPLICIT GEMM over channels-last bf16 images -- the F.conv2d calls of
 * D2/modeling/backbone/fpn.py:126-154 (output convs), CN/modeling/dense_heads/centernet_head.py:58-77,141-162 (towers and
 * predictors) and D2/modeling/roi_heads/mask_head.py:209-284 (mask head), forward AND both gradients, with no column matrix.
 *   dgx_conv3x3_pad    xpad = zero-bordered copy of x (N,H,W,C): rows = (W+3) zero rows, the (N,H+2,W+2) grid, (W+3) zero rows;
 *                      dgx_conv3x3_pad_rows(N,H,W) rows of C elements.  C % 8 == 0.
 *   dgx_conv3x3_gemm   y (N,H,W,Cout) = conv(x, w) (+ bias) (ReLU when relu != 0):  w bf16 (Cout, 3, 3, Cin) = (Cout, 9 Cin),
 *                      Cin % 64 == 0, Cout % 8 == 0.  The MFMA pipeline of dgx_gemm_bf16_nt; K-tile kt of the A operand is tap
 *                      kt / (Cin/64): the same rows of xpad shifted by a constant, so every tap is a plain strided read.
 *                      The INPUT gradient is the same call: dx = dgx_conv3x3_gemm(dypad, wflip, NULL, ...) with
 *                      wflip (Cin, 3, 3, Cout)[ci][ey][ex][co] = w[co][2-ey][2-ex][ci]  (and Cin / Cout exchanged).
 *                      workspace (optional, f32): maps with few output tiles cut the 9 Cin contraction into slabs summed by a
 *                      second launch (split-K) when S * N (H+2) (W+2) * Cout * 4 bytes fit; NULL = never split.
 *   dgx_conv3x3_wgrad  gw f32 (Cout,3,3,Cin) = beta*gw + sum over positions of dypad (x) xpad shifted per tap: nine problems of
 *                      the grouped weight-gradient kernel over the two padded images.  workspace: ..._workspace_bytes().
 */
int64_t dgx_conv3x3_pad_rows(int N, int H, int W);
int dgx_conv3x3_pad(const void* x, void* xpad, int N, int H, int W, int C, void* stream);
/* dgx_conv3x3_pad over n <= 8 images of the same channel count in one launch (the FPN levels a shared tower layer reads). */
typedef struct dgx_pad_item { const void* x; void* xpad; int N, H, W; } dgx_pad_item;
int dgx_conv3x3_pad_multi(const dgx_pad_item* items, int n, int C, void* stream);
/* dgx_conv3x3_gemm over n <= 6 zero-bordered images that share the weights (and bias / ReLU) in ONE launch: the FPN levels under a
 * CenterNet tower layer (centernet_head.py:141-162).  Cout <= 256.  The input gradient is the same call on the tap-flipped twin. */
/* ConvTranspose2d(kernel 2, stride 2) as ONE GEMM over the weight as stored (Cin, Cout*2*2) + a pixel shuffle (mask_head.py:209-284
 * `deconv`): GEMM row m = (n, h, w), columns (co, dy, dx), bf16.
 *   dgx_deconv2x2_shuffle:             out (N, 2H, 2W, Cout) channels-last: out[n][2h+dy][2w+dx][co] = y2[m][4 co + 2 dy + dx].
 *   dgx_deconv2x2_unshuffle_relu_grad: g2 (N*H*W, 4 Cout): the inverse on the output gradient gy (N, 2H, 2W, Cout), zeroed where the
 *                                      layer's own ReLU-ed output yout (same layout as gy) is not positive; yout NULL = no ReLU. */
int dgx_deconv2x2_shuffle(const void* y2, void* out, int N, int H, int W, int Cout, void* stream);
int dgx_deconv2x2_unshuffle_relu_grad(const void* gy, const void* yout, void* g2, int N, int H, int W, int Cout, void* stream);
typedef struct dgx_conv_item { const void* xpad; void* y; int N, H, W; } dgx_conv_item;
int dgx_conv3x3_gemm_multi(const dgx_conv_item* items, int n, const void* w, const void* bias, int Cin, int Cout, int relu, void* stream);
/* The weight (+ bias) gradient of ONE convolution accumulated over n <= 6 zero-bordered image pairs (output gradient, input) -- the FPN
 * levels under a shared tower layer -- in one partial + one reduce launch: gw (Cout, 3, 3, Cin) f32 = beta*gw + sum_i dW_i, gb likewise.
 * workspace: dgx_conv3x3_wgrad_bias_multi_workspace_bytes(...) bytes. */
typedef struct dgx_conv_wgrad_item { const void* dypad; const void* xpad; int N, H, W; } dgx_conv_wgrad_item;
int64_t dgx_conv3x3_wgrad_bias_multi_workspace_bytes(const dgx_conv_wgrad_item* items, int n, int Cin, int Cout);
int dgx_conv3x3_wgrad_bias_multi(const dgx_conv_wgrad_item* items, int n, float* gw, float* gb, int Cin, int Cout, float beta,
                                 void* workspace, void* stream);
/* gpad = the zero-bordered copy of the output gradient g (N,H,W,C bf16) masked by ReLU': element kept where the saved activation
 * y (same shape, bf16: the convolution's ReLU-ed output) is > 0.  Replaces `g * (y > 0)` in front of dgx_conv3x3_pad in the backward
 * of a convolution with a fused ReLU (D2/layers/wrappers.py Conv2d with activation = relu: mask_head.py:209-284, fpn.py). */
int dgx_conv3x3_pad_relu_grad(const void* g, const void* y, void* gpad, int N, int H, int W, int C, void* stream);
int dgx_conv3x3_gemm(const void* xpad, const void* w, const void* bias, void* y, int N, int H, int W, int Cin, int Cout,
                     int relu, void* workspace, int64_t workspace_bytes, void* stream);
int64_t dgx_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int dgx_conv3x3_wgrad(const void* dypad, const void* xpad, float* gw, int N, int H, int W, int Cin, int Cout, float beta,
                      void* workspace, void* stream);
/* the same, also producing the convolution's bias gradient gb f32 (Cout) = beta*gb + sum over pixels of dy (NULL = none) from the
 * same pass; workspace: dgx_conv3x3_wgrad_bias_workspace_bytes */
int64_t dgx_conv3x3_wgrad_bias_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int dgx_conv3x3_wgrad_bias(const void* dypad, const void* xpad, float* gw, float* gb, int N, int H, int W, int Cin, int Cout,
                           float beta, void* workspace, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Image normalisation + zero padding + 4x4 patch gather in one pass: the A operand of the PatchEmbed projection straight
 * from the uint8 image.  Replaces `(x - pixel_mean) / pixel_std` (D2/modeling/meta_arch/rcnn.py:220-227),
 * ImageList.from_tensors' padded batch (D2/structures/image_list.py:59-110) and the unfold of the stride-4 convolution
 * (DG/divergen/modeling/backbone/swintransformer.py:317-338).
 *   img   u8 (3, h, w) one image;  mean, stdv f32 (3) device pointers
 *   rows  bf16 (Hp*Wp, 48) the rows of THIS image: rows[py*Wp + px][c*16 + dy*4 + dx], zero where 4py+dy >= h or 4px+dx >= w
 *   patch must be 4 (DGX_ERR_UNSUPPORTED otherwise); h <= 4 Hp, w <= 4 Wp. */
int dgx_preprocess_patches(const uint8_t* img, int h, int w, const float* mean, const float* stdv, void* rows, int Hp, int Wp,
                           int patch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ResNet-50 glue of the R50 configurations (DG/divergen/modeling/backbone/timm.py:27-151 over timm 0.4.9's ResNet /
 * Bottleneck; DG/configs/Base-C2_L_R5021k_640b64_4x.yaml): everything between the GEMMs, channels-last bf16.
 *   dgx_stem_im2col7x7   conv1 (7x7, stride 2, pad 3) as GEMM rows: x f32 (N,3,H,W) -> rows bf16 (N*Ho*Wo, 152),
 *                        rows[r][c*49 + ky*7 + kx] (the order of conv1.weight.view(64, 147)), columns 147..151 zero;
 *                        Ho = (H-1)/2 + 1.  The contraction runs on dgx_gemm_bf16_nt.
 *   dgx_affine_act_fwd   FrozenBatchNorm2d (D2/layers/batch_norm.py: y = x*scale + shift with scale = weight*rsqrt(var+eps),
 *                        shift = bias - mean*scale) + optional residual add + optional ReLU in one pass: x, residual, y bf16
 *                        (rows, C), scale / shift f32 (C), C % 8 == 0.
 *   dgx_affine_act_bwd   g = dy*[y > 0] (when relu), dx = g*scale, dres = g (NULL: not wanted; may alias dy).
 *   dgx_maxpool3x3s2_*   nn.MaxPool2d(3, 2, 1) over (N,H,W,C) bf16; idx u8 (N,Ho,Wo,C) = winning tap (first maximum in scan
 *                        order, NaN wins: ATen's rule); backward = gather, one writer per input pixel. */
int dgx_stem_im2col7x7(const float* x_nchw, void* rows_bf16, int N, int H, int W, void* stream);
int dgx_affine_act_fwd(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows, int C,
                       int relu, void* stream);
int dgx_affine_act_bwd(const void* dy, const void* y, const float* scale, void* dx, void* dres, int64_t rows, int C, int relu,
                       void* stream);
int dgx_maxpool3x3s2_fwd(const void* x, void* y, void* idx_u8, int N, int H, int W, int C, void* stream);
int dgx_maxpool3x3s2_bwd(const void* dy, const void* idx_u8, void* dx, int N, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mask loss of mask_rcnn_loss (D2/modeling/roi_heads/mask_head.py:35-110): F.binary_cross_entropy_with_logits(pred, gt,
 * reduction="mean"), its gradient and the accuracy / false-positive / false-negative counts in one pass.
 *   logits  (R, inner) f32 or bf16, row r at logits + r*row_stride (the class-specific branch hands a strided gather view)
 *   gt      u8 (n = R*inner) in {0,1} (dgx_mask_crop's output)
 *   grad    same dtype as logits, (n) contiguous, = (sigmoid(x) - t) / n; may be NULL
 *   out     f32 (5): mean loss, #incorrect, #false positive, #false negative, #positive
 *   workspace f32 (dgx_mask_bce_workspace_floats(n)); sums are two-stage in a fixed order (reproducible). */
int64_t dgx_mask_bce_workspace_floats(int64_t n);
int dgx_mask_bce(const void* logits, int64_t row_stride, int64_t inner, const uint8_t* gt, int64_t n, void* grad, float* out,
                 float* workspace, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Launch accounting for measurement (bench.py's `roofline` objects; no reference counterpart -- the reference has no
 * device-side instrumentation).  After dgx_prof_enable(1) the entry points of a family bracket each call with HIP events
 * on the stream they launch on and add the call's algorithmic FLOP and bytes (operands read once, results written once)
 * to the family's tally.  Calls made while the stream is captured into a hipGraph are tallied in the captured_* fields
 * instead (no events inside a graph): the caller multiplies them by its replay count.
 * dgx_prof_read synchronises the events recorded so far; dgx_prof_enable (either value) resets all tallies;
 * dgx_prof_pause(1) / (0) stops / resumes the accounting without touching them (sampling a subset of steps: two event
 * records per launch cost the host ~10 us). */
#define DGX_PROF_GEMM_NT 0   /* dgx_gemm_bf16_nt, dgx_conv3x3_gemm */
#define DGX_PROF_WGRAD 1     /* dgx_linear_wgrad_grouped, dgx_conv3x3_wgrad, dgx_linear_wgrad */
#define DGX_PROF_ATTN_FWD 2  /* dgx_window_attention_fwd */
#define DGX_PROF_ATTN_BWD 3  /* dgx_window_attention_bwd */
#define DGX_PROF_FAMILIES 4
typedef struct dgx_prof_stats {
    double ms;                  /* sum of event-bracketed durations */
    int64_t launches;           /* calls timed */
    double flops, bytes;        /* algorithmic work of the timed calls */
    int64_t captured_launches;  /* calls recorded into hipGraphs (once per capture) */
    double captured_flops, captured_bytes;
} dgx_prof_stats;
int dgx_prof_enable(int on);
int dgx_prof_pause(int paused);
int dgx_prof_read(int family, dgx_prof_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* DIVERGEN_HIP_H */
