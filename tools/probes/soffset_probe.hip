// Is the SGPR offset of a raw buffer load part of the range check on gfx950?  A 256-byte descriptor over a 4 KB array of 0x1234: loads at
// voffset 0 / soffset 1024 and voffset 1024 / soffset 0 (the second is certainly out of range -> 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const uint32_t* src, uint32_t* out) {
    const uint64_t a = (uint64_t)src;
    const u32x4 rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffff)), 256u, 0x00020000u};
    uint32_t v0, v1, v2;
    uint32_t z = threadIdx.x * 4, o = 1024 + threadIdx.x * 4;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v0) : "v"(z), "s"(rs), "s"(1024u) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v1) : "v"(o), "s"(rs) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v2) : "v"(z), "s"(rs), "s"(128u) : "memory");
    out[threadIdx.x] = v0; out[64 + threadIdx.x] = v1; out[128 + threadIdx.x] = v2;
}
int main() {
    uint32_t *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 4096);
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x1234; hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(s, o); hipMemcpy(h, o, 768, hipMemcpyDeviceToHost);
    printf("voffset 0..252 + soffset 1024 (beyond num_records 256): lane 0 -> 0x%x (0 = soffset is range-checked)\n", h[0]);
    printf("voffset 1024.. + soffset 0: lane 0 -> 0x%x\n", h[64]);
    printf("voffset 0..252 + soffset 128: lanes 0 / 31 / 32 / 63 -> 0x%x 0x%x 0x%x 0x%x (lanes >= 32 are beyond 256 only with the soffset)\n", h[128], h[128 + 31], h[128 + 32], h[128 + 63]);
    return 0;
}
