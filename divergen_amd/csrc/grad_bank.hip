// BSGAL gain scoring on flat gradient arenas.  The reference flattens every parameter gradient with
// torch.cat (BS/bsgal/modeling/meta_arch/custom_rcnn.py:973-1001 get_loss_grad), keeps a running "gradient bank" of the
// held-out batch (:1046-1072 update_grad_bank) and scores a training batch by the dot / cosine of its gradient with the
// bank (:1074-1086 compute_grad_sim): three full passes plus two temporaries over ~78 M..200 M floats.  With the gradients
// already living in one arena these are two streaming kernels:
//   dgx_grad_bank_update : bank = bank * a + grad * b            (AVERAGE: a = it/(it+1), b = 1/(it+1); MOMENTUMm: a = m, b = 1-m)
//                          same fp32 operation order as `mul_` then `+= grad * b` (no contraction)
//   dgx_grad_sim         : (g1.g2, |g1|^2, |g2|^2) in ONE pass, fp64 accumulation, deterministic two-stage reduce
#include "dgx_common.h"

namespace {
constexpr int GB_T = 256;

__global__ __launch_bounds__(GB_T) void grad_bank_update_kernel(f32x4* __restrict__ bank, const f32x4* __restrict__ grad,
                                                                int64_t n4, float a, float b, float* __restrict__ tail_bank,
                                                                const float* __restrict__ tail_grad, int tail) {
    const int64_t stride = (int64_t)gridDim.x * GB_T;
    for (int64_t i = (int64_t)blockIdx.x * GB_T + threadIdx.x; i < n4; i += stride) {
        const f32x4 g = __builtin_nontemporal_load(grad + i);
        f32x4 k = bank[i];
        k = k * a;
        k = k + g * b;
        bank[i] = k;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) {
        float k = tail_bank[threadIdx.x] * a;
        tail_bank[threadIdx.x] = k + tail_grad[threadIdx.x] * b;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ __launch_bounds__(GB_T) void grad_sim_partial_kernel(const f32x4* __restrict__ g1, const f32x4* __restrict__ g2,
                                                                int64_t n4, const float* __restrict__ t1,
                                                                const float* __restrict__ t2, int tail,
                                                                double* __restrict__ part) {
    double dot = 0.0, n1 = 0.0, n2 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * GB_T;
    for (int64_t i = (int64_t)blockIdx.x * GB_T + threadIdx.x; i < n4; i += stride) {
        const f32x4 a = __builtin_nontemporal_load(g1 + i), b = __builtin_nontemporal_load(g2 + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = a[k], y = b[k];
            dot += x * y;
            n1 += x * x;
            n2 += y * y;
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) {
        const double x = t1[threadIdx.x], y = t2[threadIdx.x];
        dot += x * y;
        n1 += x * x;
        n2 += y * y;
    }
    __shared__ double sm[3][GB_T / 64];
    dot = wave_sum(dot);
    n1 = wave_sum(n1);
    n2 = wave_sum(n2);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = dot; sm[1][wv] = n1; sm[2][wv] = n2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int w = 0; w < GB_T / 64; ++w) s += sm[threadIdx.x][w];
        part[(int64_t)blockIdx.x * 3 + threadIdx.x] = s;
    }
}

// one workgroup: fixed-order sum of the per-block partials -> out[0..2] (fp64) and out_f[0..3] fp32 (dot, |g1|, |g2|, cosine)
__global__ __launch_bounds__(GB_T) void grad_sim_final_kernel(const double* __restrict__ part, int blocks, double* __restrict__ out,
                                                              float* __restrict__ out_f) {
    __shared__ double sm[3][GB_T];
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < blocks; i += GB_T)
        for (int k = 0; k < 3; ++k) s[k] += part[(int64_t)i * 3 + k];
    for (int k = 0; k < 3; ++k) sm[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = GB_T / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = sm[0][0];
        out[1] = sm[1][0];
        out[2] = sm[2][0];
        const float dot = (float)sm[0][0], na = (float)sqrt(sm[1][0]), nb = (float)sqrt(sm[2][0]);
        out_f[0] = dot;
        out_f[1] = na;
        out_f[2] = nb;
        out_f[3] = dot / (na * nb + 1e-8f);           // compute_grad_sim with active_grad_norm (custom_rcnn.py:1084)
    }
}

int sim_blocks(int64_t n4) {
    int64_t b = (n4 + GB_T - 1) / GB_T;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace

extern "C" int dgx_grad_bank_update(float* bank, const float* grad, int64_t n, float a, float b, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!bank || !grad || ((uintptr_t)bank & 15) || ((uintptr_t)grad & 15)) return DGX_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    const int tail = (int)(n - n4 * 4);
    hipLaunchKernelGGL(grad_bank_update_kernel, dim3(sim_blocks(n4)), dim3(GB_T), 0, (hipStream_t)stream,
                       reinterpret_cast<f32x4*>(bank), reinterpret_cast<const f32x4*>(grad), n4, a, b, bank + n4 * 4, grad + n4 * 4,
                       tail);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

extern "C" int64_t dgx_grad_sim_workspace_bytes(int64_t n) { return (int64_t)sim_blocks(n / 4) * 3 * sizeof(double); }

extern "C" int dgx_grad_sim(const float* g1, const float* g2, int64_t n, double* out3, float* out4, void* workspace,
                            void* stream) {
    if (!out3 || !out4 || !workspace || n < 0) return DGX_ERR_BAD_ARG;
    if (n > 0 && (!g1 || !g2 || ((uintptr_t)g1 & 15) || ((uintptr_t)g2 & 15))) return DGX_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    const int tail = (int)(n - n4 * 4), blocks = sim_blocks(n4);
    hipLaunchKernelGGL(grad_sim_partial_kernel, dim3(blocks), dim3(GB_T), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(g1), reinterpret_cast<const f32x4*>(g2), n4, g1 + n4 * 4, g2 + n4 * 4, tail,
                       (double*)workspace);
    DGX_LAUNCH_CHECK();
    hipLaunchKernelGGL(grad_sim_final_kernel, dim3(1), dim3(GB_T), 0, (hipStream_t)stream, (const double*)workspace, blocks, out3,
                       out4);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
