// Throughput of ds_read_b64_tr_b16 for a given per-lane address pattern (table of 64 byte offsets from the host),
// 4 waves (one per SIMD) hammering the LDS concurrently.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const uint32_t* offs, unsigned long long* out, int iters, int stride_bytes) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i;
    __syncthreads();
    const int l = threadIdx.x & 63;
    auto* base = (__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t o = offs[l];
    s16x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            auto* p = (__attribute__((address_space(3))) s16x4*)(base + ((o + u * stride_bytes + (it & 3) * 8192) & 65535));
            acc += __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
        }
    }
    const unsigned long long t1 = clock64();
    if (l == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (acc[0] == 12345) out[7] = acc[1];
}
int main() {
    uint32_t* d; unsigned long long* o; hipMalloc(&d, 256); hipMalloc(&o, 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    struct Pat { const char* name; int rs; int sw; int hi; };   // row stride bytes, swizzle kind, second-group shift
    // lane (c16 = l&15, g = l>>4): row = 8g + (c16>>2), 8-byte piece (c16&3) of a 16-column block
    Pat pats[] = {{"512B rows, no swizzle", 512, 0, 0}, {"512B rows, xor (row&3)<<1 [current]", 512, 1, 0},
                  {"512B rows, xor (row&3)<<1 | ((row>>3)&1)<<3", 512, 2, 0}, {"528B rows (pad 16)", 528, 0, 0},
                  {"544B rows (pad 32)", 544, 0, 0}, {"576B rows (pad 64)", 576, 0, 0}, {"640B rows (pad 128)", 640, 0, 0},
                  {"512B rows, xor (row&7)<<1", 512, 3, 0}, {"512B rows, xor ((row&3)<<1) ^ ((row>>2)&1)", 512, 4, 0},
                  {"80B rows (attention images)", 80, 0, 0}, {"336B rows (attention dS image)", 336, 0, 0}};
    for (auto& p : pats) {
        std::vector<uint32_t> h(64);
        for (int l = 0; l < 64; ++l) {
            const int c16 = l & 15, g = l >> 4, row = 8 * g + (c16 >> 2);
            int chunk16 = (c16 & 3) >> 1, half = c16 & 1;       // which 16-byte chunk / 8-byte half of the 32-byte block
            int sw = 0;
            if (p.sw == 1) sw = (row & 3) << 1;
            if (p.sw == 2) sw = ((row & 3) << 1) | (((row >> 3) & 1) << 3);
            if (p.sw == 3) sw = (row & 7) << 1;
            if (p.sw == 4) sw = ((row & 3) << 1) ^ ((row >> 2) & 1);
            h[l] = row * p.rs + ((chunk16 ^ sw) * 16) + half * 8;
        }
        hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, d, o, 2000, 32);   // successive reads: next fragment (32 B further)
        unsigned long long r[8]; hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
        printf("%-52s %6.1f cycles per tr read (4 waves concurrently => x4 reads per that time)\n", p.name, (double)r[0] / (2000.0 * 8));
    }
    return 0;
}
