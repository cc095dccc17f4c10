/* Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): plain-C restatement of the two
 * torchvision ops the reference calls on this path.  torchvision is NOT vendored in
 * /root/reference and its version is unpinned (D2/layers/roi_align.py:43-47 asserts only
 * torchvision >= 0.7); the algorithm below is the published one:
 *
 *   roi_align  -- call sites D2/layers/roi_align.py:58-65, D2/modeling/poolers.py:142-159,
 *                 D2/structures/masks.py:214-218.  Pinned by the reference's own known-answer
 *                 test D2T/layers/test_roi_align.py:14-47 (tests/test_oracle_roi.py) and
 *                 cross-checked against oracle/_ref (ROIAlignRotated_cpu.cpp, angle 0).
 *   nms        -- call site D2/layers/nms.py:20 (batched_nms), used by
 *                 CN/modeling/layers/ml_nms.py:4-31.  PARITY UNPINNED by reference vectors.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/build.py); no FMA contraction so
 * that float results are the plain IEEE sequence the GPU kernels reproduce bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int pos[4]; float w[4]; } tap_t;

static int bilinear_taps(int H, int W, float y, float x, tap_t *t) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - y_low, lx = x - x_low, hy = 1.0f - ly, hx = 1.0f - lx;
    t->pos[0] = y_low * W + x_low;  t->w[0] = hy * hx;
    t->pos[1] = y_low * W + x_high; t->w[1] = hy * lx;
    t->pos[2] = y_high * W + x_low; t->w[2] = ly * hx;
    t->pos[3] = y_high * W + x_high; t->w[3] = ly * lx;
    return 1;
}

/* input NCHW float, rois (R,5) = (batch, x1,y1,x2,y2), out (R,C,ph,pw).  backward != 0:
 * `io` is grad_out (read) and `input` is grad_in (accumulated, must be zeroed by caller). */
static void roi_align_impl(float *input, int N, int C, int H, int W, const float *rois, int R,
                           float scale, int ph, int pw, int sampling_ratio, int aligned,
                           float *io, int backward) {
    (void)N;
    for (int r = 0; r < R; ++r) {
        const float *roi = rois + 5 * r;
        int b = (int)roi[0];
        float off = aligned ? 0.5f : 0.0f;
        float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
        float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
        float rw = ew - sw, rh = eh - sh;
        if (!aligned) { rw = rw > 1.0f ? rw : 1.0f; rh = rh > 1.0f ? rh : 1.0f; }
        float bin_h = rh / (float)ph, bin_w = rw / (float)pw;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            float *plane = input + ((size_t)b * C + c) * H * W;
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    float *o = io + (((size_t)r * C + c) * ph + i) * pw + j;
                    float acc = 0.0f;
                    float g = backward ? (*o / count) : 0.0f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + i * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + j * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                            tap_t t;
                            if (!bilinear_taps(H, W, y, x, &t)) continue;
                            if (backward) {
                                for (int k = 0; k < 4; ++k) plane[t.pos[k]] += g * t.w[k];
                            } else {
                                acc += t.w[0] * plane[t.pos[0]] + t.w[1] * plane[t.pos[1]] +
                                       t.w[2] * plane[t.pos[2]] + t.w[3] * plane[t.pos[3]];
                            }
                        }
                    }
                    if (!backward) *o = acc / count;
                }
        }
    }
}

void oracle_roi_align_forward(const float *input, int N, int C, int H, int W, const float *rois,
                              int R, float scale, int ph, int pw, int sampling_ratio,
                              int aligned, float *out) {
    roi_align_impl((float *)input, N, C, H, W, rois, R, scale, ph, pw, sampling_ratio, aligned, out, 0);
}

void oracle_roi_align_backward(const float *grad_out, int N, int C, int H, int W,
                               const float *rois, int R, float scale, int ph, int pw,
                               int sampling_ratio, int aligned, float *grad_in) {
    memset(grad_in, 0, sizeof(float) * (size_t)N * C * H * W);
    roi_align_impl(grad_in, N, C, H, W, rois, R, scale, ph, pw, sampling_ratio, aligned,
                   (float *)grad_out, 1);
}

/* Greedy NMS.  `order` = candidate indices sorted by descending score (stable), computed by
 * the caller.  keep[] receives kept indices in that order; returns their count.
 * Suppress when IoU > thr (strict), IoU = inter / (a_i + a_j - inter), no +1 offsets. */
int64_t oracle_nms(const float *boxes, const int64_t *order, int64_t n, float thr, int64_t *keep) {
    unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    float *area = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i)
        area[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    int64_t nk = 0;
    for (int64_t a = 0; a < n; ++a) {
        int64_t i = order[a];
        if (dead[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        for (int64_t bq = a + 1; bq < n; ++bq) {
            int64_t j = order[bq];
            if (dead[j]) continue;
            float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
            float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            float inter = w * h;
            float ovr = inter / (area[i] + area[j] - inter);
            if (ovr > thr) dead[j] = 1;
        }
    }
    free(dead);
    free(area);
    return nk;
}
