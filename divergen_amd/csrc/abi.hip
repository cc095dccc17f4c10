#include "dgx_common.h"
extern "C" const char* dgx_build_arch(void) { return "gfx950"; }
extern "C" int dgx_abi_version(void) { return 3; }

// Loader staging (round 6): the training process page-locks ONE shared-memory region once (the loader workers write their sample blobs
// into slots of it) and uploads a slot with a plain asynchronous copy -- no per-batch pinned allocation, no pin thread.
extern "C" int dgx_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return DGX_ERR_BAD_ARG;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    return e == hipSuccess ? DGX_OK : -(int)e - 1000;
}
extern "C" int dgx_host_unregister(void* p) {
    if (!p) return DGX_ERR_BAD_ARG;
    const hipError_t e = hipHostUnregister(p);
    return e == hipSuccess ? DGX_OK : -(int)e - 1000;
}
extern "C" int dgx_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream) {
    if (!bytes) return DGX_OK;
    if (!dst || !src) return DGX_ERR_BAD_ARG;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    return e == hipSuccess ? DGX_OK : -(int)e - 1000;
}
