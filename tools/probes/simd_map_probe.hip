// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8] on gfx9-family parts.)
//   hipcc --offload-arch=gfx950 -O2 simd_map_probe.hip -o simd_map_probe && ./simd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 16 * 4);
    for (int waves : {9, 12, 8, 16}) {
        hipMemset(d, 0, 4 * 16 * 4);
        hipLaunchKernelGGL(probe, dim3(2), dim3(waves * 64), 0, 0, d);
        unsigned h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%2d waves: simd of wave 0.. :", waves);
        for (int w = 0; w < waves; ++w) printf(" %u", (h[w] >> 4) & 3);
        printf("   (cu %u)\n", (h[0] >> 8) & 15);
    }
    return 0;
}
