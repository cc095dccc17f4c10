// Standalone timing harness for the window-attention FORWARD (no torch): the four Swin-L stage shapes of the 1024^2 bench, W-MSA and
// SW-MSA launches, event-timed; with -DDIAG_CLOCK -DDIAG_WAVE=w the per-workgroup phase clocks of wave w summed over ALL workgroups.
#include <cstdlib>
#include "../../divergen_amd/csrc/window_attention.hip"
#include <cstdio>
#include <cstring>
extern "C" int dgx_get_reserved_cus(void) { return 0; }
#include <vector>
int main(int argc, char** argv) {
    struct Shape { const char* name; int B_, nH, nW; } shapes[] = {{"stage0", 968, 6, 484}, {"stage1", 242, 12, 121}, {"stage2", 72, 24, 36}, {"stage3", 18, 48, 9}};
    const int N = 144, iters = 20;
    for (const Shape& sh : shapes) {
        const int B_ = sh.B_, nH = sh.nH, C = nH * 32;
        size_t nq = (size_t)B_ * N * 3 * C, no = (size_t)B_ * N * C;
        std::vector<uint16_t> h(nq);
        for (size_t i = 0; i < nq; ++i) h[i] = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
        uint16_t *qkv, *out; float *table, *lse;
        hipMalloc(&qkv, nq * 2); hipMalloc(&out, no * 2);
        hipMalloc(&table, 529 * nH * 4); hipMalloc(&lse, (size_t)B_ * nH * N * 4);
        hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice);
        hipMemset(table, 0, 529 * nH * 4);
        std::vector<int8_t> hr((size_t)sh.nW * N);
        for (size_t i = 0; i < hr.size(); ++i) hr[i] = (int8_t)((i / N) % 3 == 0 ? 0 : rand() & 3);
        int8_t* region; hipMalloc(&region, hr.size()); hipMemcpy(region, hr.data(), hr.size(), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int masked = 0; masked < 2; ++masked) {
            const int8_t* rg = masked ? region : nullptr;
            const int nWv = masked ? sh.nW : 1;
            for (int it = 0; it < 3; ++it) dgx_window_attention_fwd(qkv, table, 1, nH, rg, out, lse, B_, nWv, nH, 12, 0.17677f, nullptr);
            hipDeviceSynchronize();
#ifdef DIAG_CLOCK
#endif
            float ms = 0, t;
            for (int it = 0; it < iters; ++it) {
                hipEventRecord(e0); dgx_window_attention_fwd(qkv, table, 1, nH, rg, out, lse, B_, nWv, nH, 12, 0.17677f, nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1); ms += t;
            }
            const double us = ms / iters * 1e3, bytes = (double)(nq + no) * 2;
            std::vector<uint16_t> ho(no); std::vector<float> hl((size_t)B_ * nH * N);
            hipMemcpy(ho.data(), out, no * 2, hipMemcpyDeviceToHost); hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long ck = 0; for (size_t i = 0; i < no; ++i) ck = ck * 1315423911ull + ho[i];
            unsigned long long cl = 0; for (size_t i = 0; i < hl.size(); ++i) { unsigned int u; memcpy(&u, &hl[i], 4); cl = cl * 1315423911ull + u; }
            printf("%s B_=%d nH=%d %s: fwd %.1f us  %.2f TB/s of q,k,v,out bytes (%.1f MB)  out %016llx lse %016llx\n", sh.name, B_, nH, masked ? "SW-MSA" : "W-MSA ", us, bytes / us / 1e6, bytes / 1e6, ck, cl);
#ifdef DIAG_CLOCK
            {   // the LAST launch's stamps: per-phase mean over the workgroups, the mean lifetime, and the launch's span first start -> last end
                const int nwg = B_ * nH < 8192 ? B_ * nH : 8192;
                std::vector<unsigned long long> f((size_t)nwg * 8);
                hipMemcpyFromSymbol(f.data(), HIP_SYMBOL(dgx_fclk), f.size() * 8);
                const char* nm[] = {"address + issue", "load latency + park", "barrier", "scores + softmax", "P V + store issue"};
                double ph[5] = {0, 0, 0, 0, 0};
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int w = 0; w < nwg; ++w) {
                    for (int i = 0; i < 5; ++i) ph[i] += (double)(f[w * 8 + i + 1] - f[w * 8 + i]);
                    t0 = f[w * 8] < t0 ? f[w * 8] : t0;
                    t1 = f[w * 8 + 5] > t1 ? f[w * 8 + 5] : t1;
                }
                double tot = 0;
                for (int i = 0; i < 5; ++i) { printf("    %-20s %8.0f cycles / (window, head)\n", nm[i], ph[i] / nwg); tot += ph[i] / nwg; }
                printf("    %-20s %8.0f cycles; launch span %llu cycles (clock64: shader clock)\n", "lifetime (to store issue)", tot, t1 - t0);
            }
#endif
        }
        hipFree(qkv); hipFree(out); hipFree(table); hipFree(lse); hipFree(region);
    }
    return 0;
}
