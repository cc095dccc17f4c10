// Standalone timing harness for the window-attention kernels (no torch): hipcc -DDIAG_CLOCK -DDIAG_WAVE=0 ...
#include <cstdlib>
#include "../../divergen_amd/csrc/window_attention.hip"
#include <cstdio>
extern "C" int dgx_get_reserved_cus(void) { return getenv("RESERVED_CUS") ? atoi(getenv("RESERVED_CUS")) : 0; }   // (libdgx: gemm_lw.hip)
#include <vector>
int main(int argc, char** argv) {
    const int B_ = argc > 1 ? atoi(argv[1]) : 968, nH = argc > 2 ? atoi(argv[2]) : 6, N = 144, C = nH * 32;
    const int nWm = argc > 3 ? atoi(argv[3]) : 0;      // > 0: shifted-window launch with that many windows per image (region ids 0..3)
    size_t nq = (size_t)B_ * N * 3 * C, no = (size_t)B_ * N * C;
    std::vector<uint16_t> h(nq);
    for (size_t i = 0; i < nq; ++i) h[i] = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);  // ~ +-[0.0078..0.0156]... small bf16
    uint16_t *qkv, *out, *dout, *dqkv; float *table, *lse, *dtable;
    hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&out, no * 2); hipMalloc(&dout, no * 2);
    hipMalloc(&table, 529 * nH * 4); hipMalloc(&dtable, 529 * nH * 4); hipMalloc(&lse, (size_t)B_ * nH * N * 4);
    hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice);
    hipMemcpy(dout, h.data(), no * 2, hipMemcpyHostToDevice);
    hipMemset(table, 0, 529 * nH * 4); hipMemset(dtable, 0, 529 * nH * 4);
    int8_t* region = nullptr;
    if (nWm > 0) {
        std::vector<int8_t> hr((size_t)nWm * N);
        for (size_t i = 0; i < hr.size(); ++i) hr[i] = (int8_t)(getenv("REGION_ZERO") || (i / N) % 3 == 0 ? 0 : rand() & 3);
        hipMalloc(&region, hr.size()); hipMemcpy(region, hr.data(), hr.size(), hipMemcpyHostToDevice);
    }
    const int nWv = nWm > 0 ? nWm : 1;
    const bool strided = getenv("TABLE_STRIDED") != nullptr;      // the model's layout: table (529, nH), head stride 1, index stride nH
    const int64_t t_sh = strided ? 1 : 529, t_si = strided ? nH : 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        dgx_window_attention_fwd(qkv, table, t_sh, t_si, region, out, lse, B_, nWv, nH, 12, 0.17677f, nullptr);
        dgx_window_attention_bwd(qkv, table, region, out, lse, dout, dqkv, dtable, t_sh, t_si, B_, nWv, nH, 12, 0.17677f, nullptr);
    }
    hipDeviceSynchronize();
#ifdef DIAG_CLOCK
    unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(dgx_clk), z, sizeof z);
#endif
    const int iters = 10; float msf = 0, msb = 0, ms;
    for (int it = 0; it < iters; ++it) {
        hipEventRecord(e0); dgx_window_attention_fwd(qkv, table, t_sh, t_si, region, out, lse, B_, nWv, nH, 12, 0.17677f, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); msf += ms;
        hipEventRecord(e0); dgx_window_attention_bwd(qkv, table, region, out, lse, dout, dqkv, dtable, t_sh, t_si, B_, nWv, nH, 12, 0.17677f, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); msb += ms;
    }
    printf("B_=%d nH=%d%s fwd %.1f us bwd %.1f us\n", B_, nH, nWm > 0 ? " shifted" : "", msf / iters * 1e3, msb / iters * 1e3);
#ifdef DIAG_CLOCK
    hipMemcpyFromSymbol(z, HIP_SYMBOL(dgx_clk), sizeof z);
    const int nchunks = 256 / nH; const int chunk = (B_ + nchunks - 1) / nchunks;
    const char* nm[] = {"load+stage", "sync1", "phase1", "dKdV store", "sync2", "phase2", "dQ store"};
    for (int i = 0; i < 7; ++i) printf("  %-12s %8.0f cycles/window\n", nm[i], (double)z[i] / iters / chunk);
#endif
    return 0;
}
