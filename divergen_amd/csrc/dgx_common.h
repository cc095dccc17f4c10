// Shared device helpers for libdgx (gfx950 only: wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/divergen_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 in 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) float f32x4;   // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// D(16x16) += A(16x32) * B(32x16); lane l: a = A[l&15][slot (l>>4, 0..7)], b = B[slot][l&15],
// d[r] = D[(l>>4)*4 + r][l&15].  A and B use the same (lane-group, i) -> k slot map.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LDS transpose read (gfx950 ds_read_b64_tr_b16).  Measured semantics (tools/probes/tr_probe.hip): within
// each 16-lane group, lane p supplies the address of an 8-byte chunk (4 bf16); result lane i, element j is
// element (i & 3) of the chunk supplied by lane 4*j + (i >> 2).  With lane p pointing at row (p >> 2),
// columns 4*(p & 3).. of a row-major [4][16] block, lane i receives column i of that block (rows 0..3):
// exactly the k-contiguous MFMA operand layout, from an image stored the way global memory has it.
//   lds_byte_addr: byte address (LDS address space) of THIS lane's chunk.
__device__ __forceinline__ uint2 ds_read_tr16_b64(uint32_t lds_byte_addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_byte_addr) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// 8 k-slots (two 4-row blocks at rows r0 and r0+4... given explicitly) of column block c0..c0+15 of a
// row-major bf16 LDS image with row stride `rsb` bytes; p = lane & 15.
__device__ __forceinline__ bf16x8 ld_frag_tr(uint32_t img, int rsb, int rowA, int rowB, int c0, int p) {
    const uint32_t off = (uint32_t)((p >> 2) * rsb + (c0 + 4 * (p & 3)) * 2);
    const uint2 a = ds_read_tr16_b64(img + rowA * rsb + off);
    const uint2 b = ds_read_tr16_b64(img + rowB * rsb + off);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 v = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(bf16x8, v);
}

#define DGX_LAUNCH_CHECK()                                 \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return -(int)e__ - 1000;    \
    } while (0)

static inline int dgx_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
