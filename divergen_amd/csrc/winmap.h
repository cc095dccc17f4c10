// Compact window order (round 6): the rows between a Swin block's LayerNorm-1 and its proj GEMM without the padding tokens.
//
// The reference zero-pads the (H, W) token grid to multiples of the window size BEFORE the cyclic shift and the partition
// (DG/divergen/modeling/backbone/swintransformer.py:216-233), runs qkv / attention / proj over all Hp x Wp tokens and crops afterwards
// (:248-251).  A padding token's LayerNorm-ed value is 0, so its qkv row is the qkv BIAS, its attention output and proj row are
// cropped away, no gradient flows into its query, and it contributes nothing to a weight gradient -- only its keys / values take part
// (and their gradients reach the qkv bias).  At 1024^2 the padded grid holds 26.6 % more rows than the real one in stages 2 and 3
// (72^2 for 64^2, 36^2 for 32^2), 6.3 % in stages 0 and 1; at 896^2 stage 3 holds 65 % more.
//
// CLASSIC window order: row = ((b * nWh + wr) * nWw + wc) * ws^2 + n, all Hp * Wp tokens of an image.
// COMPACT window order: the REAL tokens in the same (b, wr, wc, n) order with the padding tokens left out -- B*H*W rows, so every GEMM
// over them runs at M = T instead of Tw -- followed, where a buffer keeps them at all (xw, dqkv: the qkv weight / bias gradient), by
// the padding tokens in their own (b, wr, wc, n) order at rows T .. Tw-1.
//
// In ROLLED coordinates (after the shift) the padding rows are one band [H - shift, Hp - shift) (empty when Hp == H), the padding
// columns one band [W - shift, Wp - shift): a window's real tokens are (its rows outside the row band) x (its columns outside the
// column band), so every count below is a clamp:  rb(x) = number of real coordinates in [0, x).
#pragma once
#include <stdint.h>

struct WmAxis { int lo, hi; };   // the padding band of one axis in rolled coordinates

__host__ __device__ __forceinline__ int wm_rb(int x, WmAxis a) {
    int d = x - a.lo;
    d = d < 0 ? 0 : d;
    const int w = a.hi - a.lo;
    return x - (d > w ? w : d);
}
// the r-th real coordinate of the axis
__host__ __device__ __forceinline__ int wm_unrb(int r, WmAxis a) { return r < a.lo ? r : r + (a.hi - a.lo); }

struct WmGeom {
    int H, W, ws, shift, nWh, nWw;
    WmAxis ah, aw;
};
__host__ __device__ __forceinline__ WmGeom wm_geom(int H, int W, int ws, int shift) {
    WmGeom g;
    g.H = H; g.W = W; g.ws = ws; g.shift = shift;
    g.nWh = (H + ws - 1) / ws; g.nWw = (W + ws - 1) / ws;
    g.ah.lo = H - shift; g.ah.hi = g.nWh * ws - shift;
    g.aw.lo = W - shift; g.aw.hi = g.nWw * ws - shift;
    return g;
}
// usable when the bands do not wrap (H >= shift, W >= shift: every real configuration; tiny test grids fall back to the classic order)
__host__ __device__ __forceinline__ bool wm_compact_ok(int H, int W, int ws, int shift) { return ws > 0 && H >= shift && W >= shift; }

// per window (wr, wc): first compact row of the window inside its image, its real rows / columns, first rolled row / column counts
struct WmWindow { int base, rh, rw, r0, c0; };
__host__ __device__ __forceinline__ WmWindow wm_window(const WmGeom& g, int wr, int wc) {
    WmWindow w;
    w.r0 = wm_rb(wr * g.ws, g.ah);
    w.c0 = wm_rb(wc * g.ws, g.aw);
    w.rh = wm_rb(wr * g.ws + g.ws, g.ah) - w.r0;
    w.rw = wm_rb(wc * g.ws + g.ws, g.aw) - w.c0;
    w.base = g.W * w.r0 + w.rh * w.c0;
    return w;
}
// token n = i * ws + j of window (wr, wc): is it real, and how many real tokens of the window come before it
__host__ __device__ __forceinline__ bool wm_token_real(const WmGeom& g, const WmWindow& w, int wr, int wc, int i, int j, int& before) {
    const int hs = wr * g.ws + i, wsx = wc * g.ws + j;
    const int a = wm_rb(hs, g.ah) - w.r0, c = wm_rb(wsx, g.aw) - w.c0;
    const bool rreal = hs < g.ah.lo || hs >= g.ah.hi, creal = wsx < g.aw.lo || wsx >= g.aw.hi;
    before = a * w.rw + (rreal ? c : 0);
    return rreal && creal;
}

// source token (b, hh0, ww0) -> compact row
__host__ __device__ __forceinline__ int wm_row_of_token(const WmGeom& g, int b, int hh0, int ww0) {
    const int Hp = g.nWh * g.ws, Wp = g.nWw * g.ws;
    int hs = hh0 - g.shift, wsx = ww0 - g.shift;
    if (hs < 0) hs += Hp;
    if (wsx < 0) wsx += Wp;
    const int wr = hs / g.ws, wc = wsx / g.ws;
    const WmWindow w = wm_window(g, wr, wc);
    return b * g.H * g.W + w.base + (wm_rb(hs, g.ah) - w.r0) * w.rw + (wm_rb(wsx, g.aw) - w.c0);
}
// compact row (< B*H*W) -> source token index, image in b
__host__ __device__ __forceinline__ int wm_token_of_row(const WmGeom& g, int row, int& b) {
    const int HW = g.H * g.W;
    b = row / HW;
    const int r = row - b * HW;
    const int wr = wm_unrb(r / g.W, g.ah) / g.ws;
    const int r0 = wm_rb(wr * g.ws, g.ah), rh = wm_rb(wr * g.ws + g.ws, g.ah) - r0;
    const int q = r - g.W * r0;
    const int wc = wm_unrb(q / rh, g.aw) / g.ws;
    const int c0 = wm_rb(wc * g.ws, g.aw), rw = wm_rb(wc * g.ws + g.ws, g.aw) - c0;
    const int q2 = q - rh * c0;
    const int a = q2 / rw, c = q2 - a * rw;
    const int Hp = g.nWh * g.ws, Wp = g.nWw * g.ws;
    int hh = wm_unrb(r0 + a, g.ah) + g.shift, ww = wm_unrb(c0 + c, g.aw) + g.shift;
    if (hh >= Hp) hh -= Hp;
    if (ww >= Wp) ww -= Wp;
    return (b * g.H + hh) * g.W + ww;
}
