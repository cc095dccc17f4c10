// Image normalisation fused with the 4x4 patch gather of PatchEmbed (gfx950).
// The reference normalises the uint8 image ((x - mean) / std, rcnn.py:220-227), zero-pads the batch to the backbone's
// divisibility (ImageList.from_tensors) and hands it to a stride-4 4x4 convolution (swintransformer.py:317-338), i.e. a
// Linear over the 48 values (c, dy, dx) of every patch.  Here one pass reads the bytes and writes exactly that GEMM operand:
// rows[(b*Hp + py)*Wp + px][c*16 + dy*4 + dx] = bf16((img[c][4py+dy][4px+dx] - mean[c]) / std[c]), 0 outside the image --
// the fp32 batch tensor (48 MB at 2 x 1024^2) is never written.  HBM-bound: 3 B in, 6 B out per pixel.
#include "dgx_common.h"

__global__ __launch_bounds__(256) void preprocess_patches_kernel(const uint8_t* __restrict__ img, int h, int w,
                                                                 const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                 uint16_t* __restrict__ rows, int Hp, int Wp) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)Hp * Wp * 3) return;
    const int c = (int)(t % 3);
    const int64_t patch = t / 3;
    const int py = (int)(patch / Wp), px = (int)(patch - (int64_t)py * Wp);
    const float m = mean[c], s = stdv[c];
    const uint8_t* src = img + (int64_t)c * h * w;
    float v[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        const int y = 4 * py + dy;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int x = 4 * px + dx;
            v[4 * dy + dx] = (y < h && x < w) ? ((float)src[(int64_t)y * w + x] - m) / s : 0.0f;
        }
    }
    u32x4* out = reinterpret_cast<u32x4*>(rows + patch * 48 + c * 16);
    out[0] = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    out[1] = u32x4{pack_bf2(v[8], v[9]), pack_bf2(v[10], v[11]), pack_bf2(v[12], v[13]), pack_bf2(v[14], v[15])};
}

extern "C" int dgx_preprocess_patches(const uint8_t* img, int h, int w, const float* mean, const float* stdv, void* rows,
                                      int Hp, int Wp, int patch, void* stream) {
    if (Hp <= 0 || Wp <= 0) return DGX_OK;
    if (!img || !mean || !stdv || !rows || h <= 0 || w <= 0 || h > 4 * Hp || w > 4 * Wp || ((uintptr_t)rows & 15)) return DGX_ERR_BAD_ARG;
    if (patch != 4) return DGX_ERR_UNSUPPORTED;
    const int64_t n = (int64_t)Hp * Wp * 3;
    hipLaunchKernelGGL(preprocess_patches_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, h, w, mean,
                       stdv, (uint16_t*)rows, Hp, Wp);
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
