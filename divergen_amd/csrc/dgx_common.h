// Shared device helpers for libdgx (gfx950 only: wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/divergen_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 in 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) float f32x4;   // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even (NaN stays a quiet NaN): gfx950 has it in hardware (v_cvt_pk_bf16_f32)
typedef __bf16 dgx_bf2_t __attribute__((ext_vector_type(2)));
typedef float dgx_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const dgx_f2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dgx_bf2_t));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)pack_bf2(f, 0.f); }

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
// The compiler builtin is used (not inline asm) so that constant offsets fold into the instruction's
// offset field and the waits are placed by the scheduler.
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define DGX_LDS __attribute__((address_space(3)))
// LDS-qualified copy of a shared-memory pointer, opaque to the optimiser: everything derived from it is
// "base register + constant", which the backend folds into the DS instruction's 16-bit offset field instead
// of materialising (and spilling) one address register per unrolled access.
template <typename T>
__device__ __forceinline__ DGX_LDS T* lds_opaque(T* p) {
    DGX_LDS T* q = (DGX_LDS T*)p;
    asm volatile("" : "+v"(q));
    return q;
}
__device__ __forceinline__ s16x4 ds_read_tr16(DGX_LDS const uint16_t* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((DGX_LDS s16x4*)lds_ptr);
}
// This lane's chunk pointer inside a row-major bf16 LDS image (row stride `rs` elements) for a transpose
// read of the 4-row block starting at `row0`, columns c0 .. c0+15; p = lane & 15.
__device__ __forceinline__ DGX_LDS const uint16_t* tr_lane_ptr(const uint16_t* img, int rs, int row0, int c0, int p) {
    return lds_opaque(img + (row0 + (p >> 2)) * rs + c0 + 4 * (p & 3));
}
// MFMA operand (8 k-slots) = two 4-row blocks at element offsets offA / offB from the lane pointer.
__device__ __forceinline__ bf16x8 tr_frag(DGX_LDS const uint16_t* lane_ptr, int offA, int offB) {
    const s16x4 a = ds_read_tr16(lane_ptr + offA), b = ds_read_tr16(lane_ptr + offB);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

#define DGX_LAUNCH_CHECK()                                 \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return -(int)e__ - 1000;    \
    } while (0)

static inline int dgx_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Launch accounting (prof.hip): no-op unless dgx_prof_enable(1) was called.
struct DgxProfScope {
    DgxProfScope(int family, void* stream, double flops, double bytes);
    ~DgxProfScope();
    void *a, *b, *st;
    int fam;
};
