// Issue rate of v_mfma_f32_16x16x32_bf16 on gfx950 by accumulator-reuse distance, waves per SIMD and duration.
//   mfma_rate_probe            prints cycles per MFMA (shader clock, s_memtime) and the shader clock (against the 100 MHz counter)
// Patterns: D accumulators used round-robin (the same accumulator comes back every D MFMAs), operands from VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int D>
__global__ __launch_bounds__(768) void k(unsigned long long* out, int iters, int active_waves) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (w >= active_waves) return;
    f32x4 acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[6], b[2];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = bf16x8{(short)l, (short)i, 1, 2, 3, 4, 5, 6};
    b[0] = bf16x8{(short)(l + 1), 1, 1, 2, 3, 4, 5, 6}; b[1] = bf16x8{(short)(l + 2), 2, 1, 2, 3, 4, 5, 6};
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u) acc[u % D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % 6], b[(u / 3) & 1], acc[u % D], 0, 0, 0);
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) s += acc[i][0] + acc[i][3];
    if (l == 0 && blockIdx.x == 0) { out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }
    if (s == 12345.f) out[63] = 1;
}
template <int D> void run(unsigned long long* o, int iters, int waves) {
    hipMemset(o, 0, 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<D><<<256, 768>>>(o, iters / 10, waves);
    hipEventRecord(e0);
    k<D><<<256, 768>>>(o, iters, waves);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64]; hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
    const double n = 48.0 * iters;
    printf("distance %2d  waves/WG %2d  %8.0f MFMAs/wave  %6.2f cycles/MFMA/wave  shader clock %4.0f MHz  kernel %7.3f ms  %7.1f TF/s\n", D, waves, n,
           (double)h[0] / n, (double)h[0] / ((double)h[1] / 100.0), ms, 256.0 * waves * n * 16384.0 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned long long* o; hipMalloc(&o, 512);
    for (int waves : {4, 8, 12})
        for (int iters : {200, 20000}) {
            run<3>(o, iters, waves); run<6>(o, iters, waves); run<12>(o, iters, waves); run<24>(o, iters, waves);
        }
    return 0;
}
