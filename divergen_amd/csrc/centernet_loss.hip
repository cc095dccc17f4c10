// CenterNet proposal losses in one pass (CN/modeling/dense_heads/centernet.py:237-314 `losses`, with
// CN/modeling/layers/iou_loss.py:10-63 'giou' and heatmap_focal_loss.py:51-85 binary_heatmap_focal_loss):
// per location the GIoU regression term (weighted by the heat-map peak) and the negative focal term, per positive entry the
// positive focal term -- values AND the gradients w.r.t. reg_pred / the agnostic heat-map logit, which the torch formulation
// spreads over ~100 elementwise / reduction launches on 43 k-element tensors.  The kernel returns RAW sums (the
// normalisers num_pos / reg_norm are all-reduced across ranks by the caller, centernet.py:243-262) and UNSCALED gradients.
#include "dgx_common.h"

namespace {
constexpr int CL_T = 256;
struct CnlCfg {
    int M, C, P, not_norm_reg;
    float beta, gamma, clampv, ignore_high_fp, pos_mul, neg_mul;
};
__device__ __forceinline__ float powg(float x, float g) { return g == 2.0f ? x * x : powf(x, g); }
// derivative weight of min(a, b) w.r.t. a (torch.minimum: ties split evenly); max(a, b) gets 1 - this
__device__ __forceinline__ float dmin_w(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); }

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < CL_T / 64; ++i) s += sm[i];
    return s;
}

// sigmoid + clamp and d pred / d logit (clamp passes the gradient on the closed interval, like torch.clamp)
__device__ __forceinline__ void clamped_sigmoid(float x, float c, float& p, float& dp) {
    const float sg = 1.0f / (1.0f + expf(-x));
    p = fminf(fmaxf(sg, c), 1.0f - c);
    dp = (sg >= c && sg <= 1.0f - c) ? sg * (1.0f - sg) : 0.0f;
}

__global__ __launch_bounds__(CL_T) void cn_loss_rows_kernel(const float* __restrict__ reg_pred, const float* __restrict__ reg_tgt,
                                                            const float* __restrict__ hms, const float* __restrict__ logit,
                                                            float* __restrict__ g_reg, float* __restrict__ g_neg,
                                                            float* __restrict__ part, CnlCfg cf) {
    __shared__ float sm[CL_T / 64];
    const int i = blockIdx.x * CL_T + threadIdx.x;
    float s_w = 0.f, s_loc = 0.f, s_neg = 0.f;
    if (i < cf.M) {
        float w = hms[(int64_t)i * cf.C];
        for (int c = 1; c < cf.C; ++c) w = fmaxf(w, hms[(int64_t)i * cf.C + c]);
        const float4 t = reinterpret_cast<const float4*>(reg_tgt)[i];
        const float4 p = reinterpret_cast<const float4*>(reg_pred)[i];
        float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)) >= 0.f) {          // reg_mask
            const float wt = cf.not_norm_reg ? 1.0f : w;
            const float pl = p.x, pt = p.y, pr = p.z, pb = p.w, tl = t.x, tt = t.y, tr = t.z, tb = t.w;
            const float A = pl + pr, Bh = pt + pb;
            const float ta = (tl + tr) * (tt + tb), pa = A * Bh;
            const float wi = fminf(pl, tl) + fminf(pr, tr), hi = fminf(pb, tb) + fminf(pt, tt);
            const float gw = fmaxf(pl, tl) + fmaxf(pr, tr), gh = fmaxf(pb, tb) + fmaxf(pt, tt);
            const float ac = gw * gh, ai = wi * hi, au = ta + pa - ai;
            const float ious = (ai + 1.0f) / (au + 1.0f);
            const float loss = 1.0f - (ious - (ac - au) / ac);
            s_w = wt;
            s_loc = loss * wt;
            // d loss = -dI + dG,  I = (ai+1)/(au+1),  G = (ac-au)/ac
            const float iu = 1.0f / (au + 1.0f), iac = 1.0f / ac;
            auto dloss = [&](float dai, float dpa, float dac) {
                const float dau = dpa - dai;
                const float dI = (dai * (au + 1.0f) - (ai + 1.0f) * dau) * iu * iu;
                const float dG = -(dau * ac - au * dac) * iac * iac;
                return (-dI + dG) * wt;
            };
            const float ml = dmin_w(pl, tl), mr = dmin_w(pr, tr), mt = dmin_w(pt, tt), mb = dmin_w(pb, tb);
            gr.x = dloss(ml * hi, Bh, (1.0f - ml) * gh);
            gr.y = dloss(wi * mt, A, gw * (1.0f - mt));
            gr.z = dloss(mr * hi, Bh, (1.0f - mr) * gh);
            gr.w = dloss(wi * mb, A, gw * (1.0f - mb));
        }
        reinterpret_cast<float4*>(g_reg)[i] = gr;
        // negative focal term on every location
        float pr_, dp;
        clamped_sigmoid(logit[i], cf.clampv, pr_, dp);
        const float nw = powf(1.0f - w, cf.beta);
        const float l1 = logf(1.0f - pr_), pg = powg(pr_, cf.gamma);
        float n = l1 * pg * nw;
        float dn = (-pg / (1.0f - pr_) + cf.gamma * powg(pr_, cf.gamma - 1.0f) * l1) * nw;
        if (cf.gamma == 2.0f) dn = (-pg / (1.0f - pr_) + 2.0f * pr_ * l1) * nw;
        if (cf.ignore_high_fp > 0.f && !(pr_ < cf.ignore_high_fp)) { n = 0.f; dn = 0.f; }
        s_neg = -n * cf.neg_mul;
        g_neg[i] = -cf.neg_mul * dn * dp;
    }
    const float a = block_sum(s_w, sm), b = block_sum(s_loc, sm), c = block_sum(s_neg, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = a;
        part[blockIdx.x * 3 + 1] = b;
        part[blockIdx.x * 3 + 2] = c;
    }
}

// one workgroup: positive entries (gather at pos_idx, scatter-add of the gradient) + fold of the row partials
// out: [0] sum of regression weights, [1] weighted GIoU sum, [2] neg loss, [3] pos loss, [4] number of cared positives
__global__ __launch_bounds__(CL_T) void cn_loss_tail_kernel(const float* __restrict__ logit, const int64_t* __restrict__ pos_idx,
                                                            const uint8_t* __restrict__ cared, float* __restrict__ g_pos,
                                                            const float* __restrict__ part, int blocks, float* __restrict__ out,
                                                            CnlCfg cf) {
    __shared__ float sm[CL_T / 64];
    float s_pos = 0.f, n_pos = 0.f;
    for (int j = threadIdx.x; j < cf.P; j += CL_T) {
        if (cared && !cared[j]) continue;
        const int64_t ix = pos_idx[j];
        float q, dp;
        clamped_sigmoid(logit[ix], cf.clampv, q, dp);
        const float lq = logf(q), og = powg(1.0f - q, cf.gamma);
        float dt = og / q - cf.gamma * powg(1.0f - q, cf.gamma - 1.0f) * lq;
        if (cf.gamma == 2.0f) dt = og / q - 2.0f * (1.0f - q) * lq;
        s_pos += -lq * og * cf.pos_mul;
        n_pos += 1.0f;
        atomicAdd(g_pos + ix, -cf.pos_mul * dt * dp);
    }
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < blocks; b += CL_T)
        for (int k = 0; k < 3; ++k) acc[k] += (double)part[b * 3 + k];
    const float r0 = block_sum((float)acc[0], sm), r1 = block_sum((float)acc[1], sm), r2 = block_sum((float)acc[2], sm);
    const float r3 = block_sum(s_pos, sm), r4 = block_sum(n_pos, sm);
    if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; out[4] = r4; }
}
}  // namespace

extern "C" int dgx_centernet_losses_blocks(int M) { return M > 0 ? (M + CL_T - 1) / CL_T : 1; }

extern "C" int dgx_centernet_losses(const float* reg_pred, const float* reg_targets, const float* hms, const float* logit,
                                    const int64_t* pos_idx, const uint8_t* pos_cared, int M, int C, int P, int not_norm_reg,
                                    float beta, float gamma, float sigmoid_clamp, float ignore_high_fp, float pos_mul,
                                    float neg_mul, float* g_reg, float* g_neg, float* g_pos, float* out8, float* part,
                                    void* stream) {
    if (M <= 0 || C <= 0 || P < 0 || !reg_pred || !reg_targets || !hms || !logit || !g_reg || !g_neg || !g_pos || !out8 ||
        !part || (P > 0 && !pos_idx))
        return DGX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const CnlCfg cf = {M, C, P, not_norm_reg, beta, gamma, sigmoid_clamp, ignore_high_fp, pos_mul, neg_mul};
    const hipError_t me = hipMemsetAsync(g_pos, 0, (size_t)M * sizeof(float), st);
    if (me != hipSuccess) return -(int)me - 1000;
    const int blocks = dgx_centernet_losses_blocks(M);
    hipLaunchKernelGGL(cn_loss_rows_kernel, dim3(blocks), dim3(CL_T), 0, st, reg_pred, reg_targets, hms, logit, g_reg, g_neg, part, cf);
    hipLaunchKernelGGL(cn_loss_tail_kernel, dim3(1), dim3(CL_T), 0, st, logit, pos_idx, pos_cared, g_pos, part, blocks, out8, cf);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
