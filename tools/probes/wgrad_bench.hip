// Standalone harness: grouped 256x256 weight-gradient kernel vs the 128x128 split-M kernel (timing + agreement).
#include "../../divergen_amd/csrc/wgrad256.hip"
#include "../../divergen_amd/csrc/wgrad_gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
struct Shape { int M, Nn, Kk; };
int main(int argc, char** argv) {
    // the four weight gradients of a Swin-L stage-2 block at 1024^2, B=2 (window-padded tokens for qkv/proj)
    std::vector<Shape> shapes = {{8192, 768, 3072}, {8192, 3072, 768}, {10368, 768, 768}, {10368, 2304, 768}};
    if (argc > 1 && atoi(argv[1]) == 0) shapes = {{131072, 192, 768}, {131072, 768, 192}, {139392, 192, 192}, {139392, 576, 192}};
    if (argc > 1 && atoi(argv[1]) == 9) shapes = {{100, 192, 264}, {8192, 40, 768}, {4000, 768, 776}};
    if (argc > 1 && atoi(argv[1]) == 3) shapes = {{2048, 1536, 6144}, {2048, 6144, 1536}, {2592, 1536, 1536}, {2592, 4608, 1536}};
    const int n = (int)shapes.size();
    dgx_wgrad_problem pr[8];
    std::vector<float*> ref(n);
    double flops = 0;
    for (int i = 0; i < n; ++i) {
        const Shape s = shapes[i];
        size_t na = (size_t)s.M * s.Nn, nb = (size_t)s.M * s.Kk;
        std::vector<uint16_t> ha(na), hb(nb);
        for (auto& v : ha) v = 0x3c00 + (rand() & 0x1ff) - ((rand() & 1) << 15);
        for (auto& v : hb) v = 0x3c00 + (rand() & 0x1ff) - ((rand() & 1) << 15);
        void *a, *b; float* c;
        hipMalloc(&a, na * 2); hipMalloc(&b, nb * 2); hipMalloc(&c, (size_t)s.Nn * s.Kk * 4); hipMalloc(&ref[i], (size_t)s.Nn * s.Kk * 4);
        hipMemcpy(a, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), nb * 2, hipMemcpyHostToDevice);
        pr[i] = {a, b, c, s.M, s.Nn, s.Kk};
        flops += 2.0 * s.M * s.Nn * s.Kk;
    }
    int64_t wsb = dgx_wgrad_grouped_workspace_bytes(pr, n);
    void* ws; hipMalloc(&ws, wsb);
    int64_t wsb1 = 0;
    for (int i = 0; i < n; ++i) { int64_t b = dgx_wgrad_workspace_bytes(pr[i].M, pr[i].Nn, pr[i].Kk); if (b > wsb1) wsb1 = b; }
    void* ws1; hipMalloc(&ws1, wsb1);
    printf("grouped workspace %.1f MB, single-kernel workspace %.1f MB\n", wsb / 1e6, wsb1 / 1e6);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) dgx_linear_wgrad_grouped(pr, n, 0.f, ws, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
#ifdef DIAG_CLOCK
    { unsigned long long z[8]; hipMemcpyFromSymbol(z, HIP_SYMBOL(w256_clk), sizeof z);
      const char* nm[] = {"wait vmcnt", "wait lgkm", "barrier", "issue BL", "issue TR", "MFMA"}; double tot = 0; for (int i = 0; i < 6; ++i) tot += z[i];
      for (int i = 0; i < 6; ++i) printf("   %-12s %5.1f %%  (%.0f cycles/stage)\n", nm[i], 100.0 * z[i] / tot, (double)z[i] / 20 / 128); }
#endif
    printf("grouped 256x256 : %.1f us per group  (%.0f TF/s)\n", ms * 100, flops / (ms * 1e-4) / 1e12 / 1e0 * 1e-0);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it)
            for (int i = 0; i < n; ++i) dgx_linear_wgrad(pr[i].dy, pr[i].x, ref[i], pr[i].M, pr[i].Nn, pr[i].Kk, 0.f, ws1, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("4 x split-M 128 : %.1f us per group  (%.0f TF/s)\n", ms * 100, flops / (ms * 1e-4) / 1e12);
    for (int i = 0; i < n; ++i) {
        size_t ne = (size_t)pr[i].Nn * pr[i].Kk;
        std::vector<float> x(ne), y(ne);
        hipMemcpy(x.data(), pr[i].gw, ne * 4, hipMemcpyDeviceToHost); hipMemcpy(y.data(), ref[i], ne * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t k = 0; k < ne; ++k) { md = fmax(md, fabs((double)x[k] - y[k])); mx = fmax(mx, fabs((double)y[k])); }
        printf("  problem %d (%d x %d x %d): max |diff| %.3g of max |ref| %.3g\n", i, pr[i].M, pr[i].Nn, pr[i].Kk, md, mx);
    }
    return 0;
}
