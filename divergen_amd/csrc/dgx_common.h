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

#define DGX_LAUNCH_CHECK()                                 \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return -(int)e__ - 1000;    \
    } while (0)

static inline int dgx_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
