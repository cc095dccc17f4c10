// How fast can ONE workgroup per CU pull operand tiles out of L2?  Compares, for the access pattern of gemm_nt_kernel's loader
// (a wave-instruction = 8 rows x 128 B, rows `ld` bytes apart), per CU and per clock:
//   mode 0  buffer_load_dwordx4 ... lds   (global -> LDS direct, what the GEMM uses)
//   mode 1  global_load_dwordx4 -> VGPR   (register path; the result is xor-folded so that nothing is optimised away)
//   mode 2  both, alternating (half the bytes each)
// `waves` waves per workgroup issue `depth` loads back to back, wait for all of them, repeat.  The buffer (rows x ld bytes) is
// small enough to stay in L2 after the first pass.  Prints bytes / clock / CU and GB/s over the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k(const unsigned char* buf, unsigned long long* out, int iters, int ld, int rows_per_wg, uint32_t bytes, int shared, int seg) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t a = (uint64_t)buf;
    const u32x4 rsrc = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + 1024u * 8 * w);
    const uint32_t row0 = shared ? (uint32_t)(blockIdx.x & 7) * rows_per_wg : (uint32_t)blockIdx.x * rows_per_wg;   // shared: one panel per XCD (L2 hits)
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // instruction q covers rows 8q .. 8q+7 of this workgroup's panel, 128 B per row at K offset (it % kt) * 128
            const uint32_t q = (uint32_t)((it * DEPTH + d) * nw + w);
            // seg = bytes per row an instruction covers (128: 8 whole cache lines of 8 rows; 64: half lines of 16 rows)
            const uint32_t lpr = (uint32_t)seg / 16u, rpi = 64u / lpr;
            const uint32_t row = row0 + (rpi * q + (uint32_t)l / lpr) % rows_per_wg;
            const uint32_t koff = ((uint32_t)it % (uint32_t)(ld / seg)) * (uint32_t)seg;
            const uint32_t voff = row * (uint32_t)ld + koff + ((uint32_t)l % lpr) * 16u;
            const bool lds_path = MODE == 0 || (MODE == 2 && (d & 1) == 0);
            if (lds_path) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(ldsw + 1024u * (d & 7)) : "memory");
            } else {
                v[d] = *reinterpret_cast<const u32x4*>(buf + voff);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (!(MODE == 0 || (MODE == 2 && (d & 1) == 0))) acc ^= v[d];
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678u && acc[1] == 7u) out[4095] = acc[2];
}
template <int MODE, int DEPTH>
void run(const unsigned char* d, unsigned long long* o, int waves, int ld, int rows_per_wg, size_t bytes, int iters, int shared, int seg) {
    hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(64 * waves), 1024 * 8 * 8, 0, d, o, iters, ld, rows_per_wg, (uint32_t)bytes, shared, seg);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < 256; ++i) cyc += (double)h[i]; cyc /= 256;
    const double per_wg = (double)iters * DEPTH * waves * 1024.0;
    printf("%s seg %3d mode %d depth %2d waves %d ld %5d: %6.1f B/clk/CU  %7.0f GB/s chip  (%.0f cycles, %.1f us)\n", shared ? "L2 " : "HBM", seg, MODE, DEPTH, waves, ld,
           per_wg / cyc, per_wg * 256 / (ms * 1e-3) / 1e9, cyc, ms * 1e3);
}
int main(int argc, char** argv) {
    const int ld = argc > 1 ? atoi(argv[1]) : 1536;          // bytes per row (K = 768 bf16)
    const int rows_per_wg = 448;                             // a 256 + 192 row operand pair
    const size_t bytes = (size_t)256 * rows_per_wg * ld;
    unsigned char* d; unsigned long long* o;
    hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&o, 4096 * 8);
    const int iters = 400;
    for (int shared : {1, 0})
        for (int seg : {128, 64}) {
            run<0, 7>(d, o, 8, ld, rows_per_wg, bytes, iters, shared, seg);
            run<0, 14>(d, o, 8, ld, rows_per_wg, bytes, iters, shared, seg);
            run<0, 28>(d, o, 8, ld, rows_per_wg, bytes, iters, shared, seg);
        }
    return 0;
}
