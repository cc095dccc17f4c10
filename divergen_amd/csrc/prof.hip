// Launch accounting for the roofline objects of bench.py (gfx950).
// When switched on, the heavy entry points (own GEMM, grouped weight-gradient GEMM, window attention) bracket their launches
// with HIP events ON THE STREAM THEY LAUNCH ON and add the algorithmic FLOP / bytes of the call to a per-family tally.
// Launches issued while the stream is being captured into a hipGraph cannot carry events: they are tallied separately
// ("captured"), and the host side multiplies them by the number of replays (divergen_amd/utils/graphs.py).
#include "dgx_common.h"

#include <mutex>
#include <utility>
#include <vector>

namespace {
struct Family {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double flops = 0, bytes = 0, cflops = 0, cbytes = 0;
    int64_t launches = 0, claunches = 0;
};
Family g_fam[DGX_PROF_FAMILIES];
std::vector<hipEvent_t> g_pool;
std::mutex g_mu;
bool g_on = false;

hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

DgxProfScope::DgxProfScope(int family, void* stream, double flops, double bytes) : a(nullptr), b(nullptr), st(stream), fam(family) {
    if (!g_on || family < 0 || family >= DGX_PROF_FAMILIES) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family& F = g_fam[family];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        F.cflops += flops; F.cbytes += bytes; F.claunches += 1;
        return;
    }
    F.flops += flops; F.bytes += bytes; F.launches += 1;
    a = take_event();
    b = take_event();
    if (a && b) (void)hipEventRecord((hipEvent_t)a, (hipStream_t)stream);
}

DgxProfScope::~DgxProfScope() {
    if (!a || !b) return;
    (void)hipEventRecord((hipEvent_t)b, (hipStream_t)st);
    std::lock_guard<std::mutex> lk(g_mu);
    g_fam[fam].ev.emplace_back((hipEvent_t)a, (hipEvent_t)b);
}

extern "C" int dgx_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Family& F : g_fam) {
        for (auto& p : F.ev) { g_pool.push_back(p.first); g_pool.push_back(p.second); }
        F = Family();
    }
    g_on = on != 0;
    return DGX_OK;
}

extern "C" int dgx_prof_pause(int paused) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = paused == 0;
    return DGX_OK;
}

extern "C" int dgx_prof_read(int family, dgx_prof_stats* out) {
    if (!out || family < 0 || family >= DGX_PROF_FAMILIES) return DGX_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    Family& F = g_fam[family];
    double ms = 0.0;
    for (auto& p : F.ev) {
        if (hipEventSynchronize(p.second) != hipSuccess) return DGX_ERR_BAD_ARG;
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
    }
    out->ms = ms; out->launches = F.launches; out->flops = F.flops; out->bytes = F.bytes;
    out->captured_launches = F.claunches; out->captured_flops = F.cflops; out->captured_bytes = F.cbytes;
    return DGX_OK;
}
