// Between two cascade stages (DG/divergen/modeling/roi_heads/detic_roi_heads.py:192-244 `_forward_box`,
// :136-190 `_match_and_label_boxes`, `_create_proposals_from_boxes`): decode the previous stage's box deltas
// (D2/modeling/box_regression.py:76-118), clip to the image, flag empty boxes, match every box to its image's ground
// truth (D2/structures/boxes.py:310-357 + D2/modeling/matcher.py:62-104, single threshold, no low-quality matches) and
// gather class / box / instance_source of the match.  ~50 tiny launches per stage in the eager form; here one
// thread per RoI does all of it for every image of the batch in one launch.  The float sequences are those of
// apply_deltas / pairwise_iou (compiled with -ffp-contract=off), so matched indices agree with dgx_iou_match.
#include "dgx_common.h"

#define REFINE_MAX_IMAGES 32
struct RefineImages {
    int B;
    int row0[REFINE_MAX_IMAGES + 1];   // RoI rows of image b: [row0[b], row0[b+1])
    int gt0[REFINE_MAX_IMAGES + 1];    // its ground-truth rows in the concatenated GT arrays
    float H[REFINE_MAX_IMAGES], W[REFINE_MAX_IMAGES];
};

namespace {
template <typename T> __device__ __forceinline__ float ldr(const T* p);
template <> __device__ __forceinline__ float ldr<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldr<uint16_t>(const uint16_t* p) { return bf2f(*p); }
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void cascade_refine_kernel(const float* __restrict__ prop, const T* __restrict__ deltas,
                                                             const uint8_t* __restrict__ valid_in, RefineImages P,
                                                             const float* __restrict__ gt_boxes, const int64_t* __restrict__ gt_classes,
                                                             const int64_t* __restrict__ gt_src, float thr, int num_classes, float wx,
                                                             float wy, float ww, float wh, float scale_clamp, float* __restrict__ boxes,
                                                             uint8_t* __restrict__ valid_out, int64_t* __restrict__ out_cls,
                                                             float* __restrict__ out_gtb, int64_t* __restrict__ out_src,
                                                             int32_t* __restrict__ num_fg) {
    const int R = P.row0[P.B];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= R) return;
    int b = 0;
#pragma unroll 1
    for (int i = 1; i < P.B; ++i)
        if (j >= P.row0[i]) b = i;
    // ---- Box2BoxTransform.apply_deltas (class-agnostic) + Boxes.clip
    const float bx1 = prop[4 * j], by1 = prop[4 * j + 1], bx2 = prop[4 * j + 2], by2 = prop[4 * j + 3];
    const float w = bx2 - bx1, h = by2 - by1;
    const float cx = bx1 + 0.5f * w, cy = by1 + 0.5f * h;
    const float dx = ldr<T>(deltas + 4 * (int64_t)j) / wx, dy = ldr<T>(deltas + 4 * (int64_t)j + 1) / wy;
    const float dw = fminf(ldr<T>(deltas + 4 * (int64_t)j + 2) / ww, scale_clamp);
    const float dh = fminf(ldr<T>(deltas + 4 * (int64_t)j + 3) / wh, scale_clamp);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    const float Wd = P.W[b], Hd = P.H[b];
    const float px1 = fminf(fmaxf(pcx - 0.5f * pw, 0.0f), Wd), py1 = fminf(fmaxf(pcy - 0.5f * ph, 0.0f), Hd);
    const float px2 = fminf(fmaxf(pcx + 0.5f * pw, 0.0f), Wd), py2 = fminf(fmaxf(pcy + 0.5f * ph, 0.0f), Hd);
    boxes[4 * j] = px1; boxes[4 * j + 1] = py1; boxes[4 * j + 2] = px2; boxes[4 * j + 3] = py2;
    const bool ok = (px2 - px1 > 0.0f) && (py2 - py1 > 0.0f) && (!valid_in || valid_in[j]);   // Boxes.nonempty()
    valid_out[j] = ok ? 1 : 0;
    // ---- pairwise_iou + Matcher against this image's ground truth
    const int g0 = P.gt0[b], M = P.gt0[b + 1] - g0;
    const float parea = (px2 - px1) * (py2 - py1);
    float best = -1.0f;
    int bi = 0;
    for (int i = 0; i < M; ++i) {
        const float* gb = gt_boxes + 4 * (int64_t)(g0 + i);
        const float gx1 = gb[0], gy1 = gb[1], gx2 = gb[2], gy2 = gb[3];
        const float garea = (gx2 - gx1) * (gy2 - gy1);
        const float iw = fmaxf(fminf(gx2, px2) - fmaxf(gx1, px1), 0.0f);
        const float ih = fmaxf(fminf(gy2, py2) - fmaxf(gy1, py1), 0.0f);
        const float inter = iw * ih;
        const float iou = inter > 0.0f ? inter / (garea + parea - inter) : 0.0f;
        if (iou > best) { best = iou; bi = i; }   // first maximum wins (torch.max(dim=0))
    }
    const bool fg = M > 0 && best >= thr;
    int64_t cls = num_classes, src = 0;
    float gb4[4] = {0.f, 0.f, 0.f, 0.f};
    if (M > 0) {
        const float* gb = gt_boxes + 4 * (int64_t)(g0 + bi);
        gb4[0] = gb[0]; gb4[1] = gb[1]; gb4[2] = gb[2]; gb4[3] = gb[3];
        if (fg) {
            cls = gt_classes[g0 + bi];
            if (gt_src) src = gt_src[g0 + bi];
        }
    }
    out_cls[j] = ok ? cls : -1;          // an empty box is dropped by the reference; here it is an "ignore" row
    out_gtb[4 * j] = gb4[0]; out_gtb[4 * j + 1] = gb4[1]; out_gtb[4 * j + 2] = gb4[2]; out_gtb[4 * j + 3] = gb4[3];
    if (out_src) out_src[j] = src;
    if (fg && ok) atomicAdd(num_fg, 1);
}

extern "C" int dgx_cascade_refine(const float* prop, const void* deltas, const uint8_t* valid_in, int B, const int* row0,
                                  const int* gt0, const float* img_h, const float* img_w, const float* gt_boxes,
                                  const int64_t* gt_classes, const int64_t* gt_src, float iou_thr, int num_classes, float wx,
                                  float wy, float ww, float wh, float scale_clamp, float* boxes, uint8_t* valid_out,
                                  int64_t* out_cls, float* out_gtb, int64_t* out_src, int32_t* num_fg, int dtype, void* stream) {
    if (B <= 0) return DGX_OK;
    if (B > REFINE_MAX_IMAGES || !row0 || !gt0 || !img_h || !img_w) return DGX_ERR_BAD_ARG;
    RefineImages P;
    P.B = B;
    for (int i = 0; i <= B; ++i) { P.row0[i] = row0[i]; P.gt0[i] = gt0[i]; }
    for (int i = 0; i < B; ++i) { P.H[i] = img_h[i]; P.W[i] = img_w[i]; }
    const int R = row0[B];
    hipStream_t st = (hipStream_t)stream;
    if (num_fg) (void)hipMemsetAsync(num_fg, 0, sizeof(int32_t), st);
    if (R <= 0) return DGX_OK;
    if (!prop || !deltas || !boxes || !valid_out || !out_cls || !out_gtb || !num_fg || (gt0[B] > 0 && (!gt_boxes || !gt_classes)))
        return DGX_ERR_BAD_ARG;
    const int grid = (R + 255) / 256;
    if (dtype == DGX_BF16)
        hipLaunchKernelGGL(cascade_refine_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, prop, (const uint16_t*)deltas, valid_in, P,
                           gt_boxes, gt_classes, gt_src, iou_thr, num_classes, wx, wy, ww, wh, scale_clamp, boxes, valid_out, out_cls,
                           out_gtb, out_src, num_fg);
    else
        hipLaunchKernelGGL(cascade_refine_kernel<float>, dim3(grid), dim3(256), 0, st, prop, (const float*)deltas, valid_in, P, gt_boxes,
                           gt_classes, gt_src, iou_thr, num_classes, wx, wy, ww, wh, scale_clamp, boxes, valid_out, out_cls, out_gtb,
                           out_src, num_fg);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
