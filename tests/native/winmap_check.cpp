// CPU check of divergen_amd/csrc/winmap.h (the header is shared by the HIP kernels): the compact window order against a brute-force
// enumeration of the reference's pad -> roll -> partition (swintransformer.py:216-233).  Prints "ok <cases>" or the first mismatch.
#define __host__
#define __device__
#define __forceinline__ inline
#include "../../divergen_amd/csrc/winmap.h"
#include <cstdio>
#include <vector>

static int classic_src(int B, int H, int W, int ws, int shift, int orow) {
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws, Hp = nWh * ws, Wp = nWw * ws, N = ws * ws;
    const int n = orow % N;
    int t = orow / N;
    const int wc = t % nWw; t /= nWw;
    const int wr = t % nWh;
    const int b = t / nWh;
    int hh = wr * ws + n / ws + shift, ww = wc * ws + n % ws + shift;
    if (hh >= Hp) hh -= Hp;
    if (ww >= Wp) ww -= Wp;
    (void)B;
    return (hh >= H || ww >= W) ? -1 : (b * H + hh) * W + ww;
}

int main() {
    const int cases[][3] = {{64, 64, 12}, {32, 32, 12}, {56, 56, 12}, {28, 28, 12}, {30, 26, 12}, {13, 25, 12}, {24, 36, 12}, {12, 12, 12},
                            {7, 7, 7}, {10, 9, 7}, {20, 15, 7}, {256, 256, 12}, {6, 18, 12}, {9, 40, 12}, {224, 224, 12}, {112, 96, 12}};
    int ncases = 0;
    for (auto& c : cases)
        for (int sh = 0; sh < 2; ++sh) {
            const int H = c[0], W = c[1], ws = c[2], shift = sh ? ws / 2 : 0, B = 2;
            if (!wm_compact_ok(H, W, ws, shift)) continue;
            const WmGeom g = wm_geom(H, W, ws, shift);
            const int N = ws * ws, Tw = B * g.nWh * g.nWw * N, T = B * H * W;
            int r = 0, p = 0;
            for (int orow = 0; orow < Tw; ++orow) {
                const int tok = classic_src(B, H, W, ws, shift, orow);
                const int n = orow % N, wi = (orow / N) % (g.nWh * g.nWw), b = orow / N / (g.nWh * g.nWw);
                const int wr = wi / g.nWw, wc = wi % g.nWw;
                const WmWindow w = wm_window(g, wr, wc);
                int before;
                const bool real = wm_token_real(g, w, wr, wc, n / ws, n % ws, before);
                if (real != (tok >= 0)) { printf("real? H%d W%d ws%d s%d orow %d\n", H, W, ws, shift, orow); return 1; }
                if (real) {
                    int bb;
                    if (b * H * W + w.base + before != r) { printf("window rank H%d W%d ws%d s%d orow %d: %d vs %d\n", H, W, ws, shift, orow, b * H * W + w.base + before, r); return 1; }
                    if (wm_row_of_token(g, tok / (H * W), (tok / W) % H, tok % W) != r) { printf("row_of_token H%d W%d ws%d s%d tok %d\n", H, W, ws, shift, tok); return 1; }
                    if (wm_token_of_row(g, r, bb) != tok || bb != tok / (H * W)) { printf("token_of_row H%d W%d ws%d s%d row %d\n", H, W, ws, shift, r); return 1; }
                    ++r;
                } else {
                    // padding rows: T + (image, window, token) order among the padding tokens
                    const int P = g.nWh * g.nWw * N - H * W;
                    const int prow = T + b * P + (wi * N - w.base) + (n - before);
                    if (prow != T + p) { printf("pad row H%d W%d ws%d s%d orow %d: %d vs %d\n", H, W, ws, shift, orow, prow, T + p); return 1; }
                    ++p;
                }
            }
            if (r != T || r + p != Tw) { printf("counts\n"); return 1; }
            ++ncases;
        }
    printf("ok %d\n", ncases);
    return 0;
}
