// Grouped weight-gradient GEMM for gfx950:  for each problem p:  C_p[Nn][Kk] = beta*C_p + sum_m A_p[m][Nn] * B_p[m][Kk]
// (A = dY, B = X, row-major bf16; C fp32 in the gradient arena).  One launch covers the (up to 8) weight
// gradients of a transformer block, so that 256x256 output tiles -- the size at which a CU is MFMA-bound
// rather than LDS-bound: 64 MFMAs per 32 transpose reads -- still fill the 256 CUs with a SMALL M-split
// (2-3 slabs instead of 7-28 for one matrix alone); the split's fp32 partial tiles are what a
// small-output / long-contraction GEMM otherwise drowns in.
//   workgroup = 4 waves, wave = 128x128 quadrant = 8x8 MFMA 16x16x32 bf16 tiles (256 accumulator
//   registers), M-depth 64 per stage; both operands sit in LDS exactly as in memory ([m][n], 16-byte
//   stores, 32-byte row padding = conflict-free ds_read_b64_tr_b16 within a 16-lane group);
//   double-buffered LDS (2 x 68 KiB), next stage prefetched into registers under the MFMAs.
//   Partials go to a workspace; a second (also grouped) kernel folds them into the arena.
#include "dgx_common.h"
#include <type_traits>

namespace {
constexpr int T256 = 256;               // tile edge
constexpr int BM256 = 32;               // m-depth per stage (one K=32 MFMA step)
constexpr int NB256 = 4;                // LDS stage buffers: the stage being computed + 3 in flight
constexpr int OPB256 = BM256 * T256 * 2;       // bytes of one operand image per stage (16 KiB)
constexpr int STB256 = 2 * OPB256;             // bytes per stage (A then B)
constexpr int MAXP256 = 12;             // >= 9 (the taps of a 3x3 convolution, dgx_conv3x3_wgrad); 12: two Swin blocks and a part of a third (252 of 256 tiles)

struct Prob256 {
    const uint16_t* A;
    const uint16_t* B;
    float* C;
    float* gb;          // bias gradient (Nn) = beta*gb + column sums of A, or null
    float* wsb;         // split problems: partial column sums [S][Nn]
    float* ws;          // partial tiles [S][Nn][Kk]
    int M, Nn, Kk, tiles_k, tiles, S, slab, wg0;
    int ldc;            // row stride of C (elements): Kk for a Linear, 9 Cin for one tap block of a convolution weight
    int64_t red0;       // first float4 of this problem in the reduce kernel's index space
};
struct Params256 {
    Prob256 p[MAXP256];
    int n, total, per_xcd;
    int interleave;     // identical problems that share one operand (the 9 taps of a convolution): consecutive workgroups take the
                        // SAME (slab, tile) of consecutive problems, so the shared dY slab is read from HBM once per XCD, not 9 x
    float beta;
    // multi-image convolution (dgx_conv3x3_wgrad_bias_multi): the M dimension of the nine tap problems runs over nimg zero-bordered
    // image pairs; slab s belongs to the last image with slab0 <= s and carries ITS pointers and tap shift
    int nimg;
    struct Img { const uint16_t* dy; const uint16_t* x; int wp, M, slab0; } img[6];
};

// 16 bytes per lane from a raw buffer straight into LDS (no staging registers): lane i of the wave lands at
// lds_wave_base + 16*i.  Out-of-range lanes (voff >= num_records) do not touch memory.  The caller orders the
// data with s_waitcnt vmcnt + barrier; the compiler does not track these loads.  soff: scalar byte offset, part of the range check on
// gfx950 (tools/probes/soffset_probe.hip) -- the row offset of a stage travels there, no VALU add per load in a loop whose lone wave per
// SIMD pays for every instruction next to its MFMAs.
__device__ __forceinline__ void buffer_load_lds16(uint32_t voff, u32x4 rsrc, uint32_t lds_wave_base, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_wave_base), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
// MFMA with the accumulator pinned to the AGPR half of the register file: 256 accumulator registers + two
// fragment sets do not fit the 256 architectural VGPRs, and left to itself the allocator shuttles accumulator
// tiles between the two halves around every MFMA.  Each accumulator is touched once per 64 MFMAs, far beyond
// the MFMA->SrcC hazard window, so no software wait states are needed inside the loop.
__device__ __forceinline__ void mfma16_agpr(f32x4& c, bf16x8 a, bf16x8 b) {
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// ... and with the accumulator pinned to the VGPR half: the bias sums' four tiles.  Left to the allocator they were given AGPRs -- all 256
// of which hold the weight-gradient tile -- and every bias MFMA swapped a tile out and back (8 v_accvgpr moves + wait states per MFMA: the
// workgroups of a problem's first tile column, i.e. ALL of them where Kk <= 256, ran 160 such moves per 128 MFMAs; found in round 4).
__device__ __forceinline__ void mfma16_vgpr(f32x4& c, bf16x8 a, bf16x8 b) {
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace

#ifdef DIAG_CLOCK
__device__ unsigned long long w256_clk[8];
#define WCLK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t__ = clock64(); w256_clk[i] += t__ - tprev; tprev = t__; } } while (0)
#else
#define WCLK(i)
#endif
// LDS image of one operand stage: [32 rows (m)][256 columns] bf16, 512-byte rows, NO padding (a wave-wide
// LDS-direct load writes 1 KiB = two whole rows).  Bank conflicts of the transpose reads (4 consecutive rows,
// same columns) are removed by an XOR swizzle of the 16-byte chunk index:  physical = logical ^ ((row & 3) << 1).
// (Additionally moving the rows of the second lane group of an LDS cycle, (row >> 3) & 1, to the other 32 banks
// was measured and is NOT faster on hardware HERE (LDS cycles are half the MFMA cycles even with the 2-way conflict), so the simpler
// swizzle stays; wgrad_lw.hip, whose eight smaller wave tiles read 0.9 LDS cycles per MFMA cycle under the conflict, has it.)
__global__ __launch_bounds__(256) void wgrad256_partial_kernel(Params256 P) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];   // [NB256][A|B][32][512 B]
    // XCD-aware order: workgroups that share an XCD (blockIdx % 8, one L2 each) take CONSECUTIVE logical ids =
    // neighbouring tiles of one (problem, slab), i.e. they stream the same dY / X panels through that L2
    const int L = (blockIdx.x & 7) * P.per_xcd + (blockIdx.x >> 3);
    if (L >= P.total) return;
    int pi = 0;
    if (P.interleave) {
        pi = L % P.n;
    } else {
#pragma unroll
        for (int i = 1; i < MAXP256; ++i)
            if (i < P.n && L >= P.p[i].wg0) pi = i;
    }
    Prob256 q = P.p[pi];
    const int local = P.interleave ? L / P.n : L - q.wg0;
    const int s = local / q.tiles, tile = local - s * q.tiles;
    int s_in = s;                                  // slab index inside its image
    if (P.nimg > 0) {                              // multi-image convolution: this slab's image (constant indices: see gemm_nt.hip)
        const int su = __builtin_amdgcn_readfirstlane(s), tap = __builtin_amdgcn_readfirstlane(pi);
        Params256::Img im = P.img[0];
#pragma unroll
        for (int k = 1; k < 6; ++k)
            if (k < P.nimg && su >= P.img[k].slab0) im = P.img[k];
        q.A = im.dy + (int64_t)(im.wp + 1) * q.Nn;                           // grid position 0 (behind the slack)
        q.B = im.x + (int64_t)((tap / 3) * im.wp + tap % 3) * q.Kk;          // position 0 shifted by tap - (wp + 1)
        q.M = im.M;
        s_in = su - im.slab0;
    }
#ifdef DIAG_SAMEPANEL   // every workgroup streams the same two panels: isolates the CU-side limit from L2 / fabric
    const int n0 = 0, k0 = 0;
    const int m_begin = 0, m_end = min(q.M, q.slab);
#else
    const int n0 = (tile / q.tiles_k) * T256, k0 = (tile % q.tiles_k) * T256;
    const int m_begin = s_in * q.slab, m_end = min(q.M, m_begin + q.slab);
#endif
    const int nst = (m_end - m_begin + BM256 - 1) / BM256;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    const int wn = (w >> 1) * 128, wk = (w & 1) * 128;    // this wave's quadrant

    // ---- loader role: load j (0..3) of a stage brings rows 2*(4j + w), +1 of each operand; this lane's slot is
    // row-in-pair l >> 5, physical chunk l & 31, i.e. logical chunk (l & 31) ^ ((row & 3) << 1), row & 3 = 2*(w&1) + (l>>5)
    const int rip = l >> 5;
    const int lc = (l & 31) ^ (((2 * (w & 1) + rip) & 3) << 1);
    const uint32_t OOB = 0x80000000u;
    const uint32_t vA = (n0 + 8 * lc < q.Nn) ? (uint32_t)(((2 * w + rip) * q.Nn + n0 + 8 * lc) * 2) : OOB;
    const uint32_t vB = (k0 + 8 * lc < q.Kk) ? (uint32_t)(((2 * w + rip) * q.Kk + k0 + 8 * lc) * 2) : OOB;
    const u32x4 rA = make_rsrc(q.A, (uint32_t)((int64_t)m_end * q.Nn * 2));   // rows >= m_end are out of range
    const u32x4 rB = make_rsrc(q.B, (uint32_t)((int64_t)m_end * q.Kk * 2));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(DGX_LDS unsigned char*)lds_raw;
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + 1024u * w);
    // one LDS-direct load (16 B per lane, 1 KiB per wave) of stage st: h = 0..3 -> A rows 8h.., h = 4..7 -> B rows 8(h-4)..
    auto issue1 = [&](int st, int h) {
        const uint32_t row0 = (uint32_t)(m_begin + st * BM256);
        const uint32_t dst = ldsw + (uint32_t)(st & (NB256 - 1)) * STB256;
        const int j = h & 3;
        if (h < 4) buffer_load_lds16(vA, rA, dst + 4096u * j, (row0 + 8 * j) * (uint32_t)(q.Nn * 2));
        else buffer_load_lds16(vB, rB, dst + OPB256 + 4096u * j, (row0 + 8 * j) * (uint32_t)(q.Kk * 2));
    };
    auto issue = [&](int st) {
#pragma unroll
        for (int h = 0; h < 8; ++h) issue1(st, h);
    };

    // ---- MFMA role: fragment i = 4a + b of an operand = columns base + 16 i .. +15, k-slots = rows 8g .. 8g+7.
    // Lane p (= c16) reads row 8g + (p >> 2) (+4), 8-byte piece (p & 3) of the 16 columns; under the swizzle
    // that is physical byte  row*512 + colbase*2 + 32*(i ^ x) + 16*hi + 8*lo,  x = p >> 2, hi = (p>>1)&1, lo = p&1;
    // i ^ x = 4a + (b ^ x): four lane-dependent bases per operand, everything else is an immediate.
    const int x = c16 >> 2, hi = (c16 >> 1) & 1, lo = c16 & 1;
    // Round 4: the lane pointers are formed ONCE (two per base: the DS offset field reaches 64 KB of the 128 KB ring); the stage slot is an
    // immediate because the loop is unrolled over the ring -- a v_add per fragment in a loop whose lone wave per SIMD pays ~6 cycles of
    // MFMA issue for every instruction it interleaves was 16 of them per 128 MFMAs.
    DGX_LDS const uint16_t* pfa[2][4];
    DGX_LDS const uint16_t* pfb[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t o = (uint32_t)((8 * g + x) * 512 + 32 * (b ^ x) + 16 * hi + 8 * lo);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            pfa[h2][b] = lds_opaque(reinterpret_cast<const uint16_t*>(lds_raw + o + wn * 2 + h2 * 2 * STB256));
            pfb[h2][b] = lds_opaque(reinterpret_cast<const uint16_t*>(lds_raw + OPB256 + o + wk * 2 + h2 * 2 * STB256));
        }
    }

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Bias gradient = column sums of dY = the product of the A fragments with a vector of ones: the workgroups of the first tile
    // column of a problem run 4 extra MFMAs per wave and stage (+6 %) on fragments they hold anyway -- no second pass over dY,
    // no extra launch.  The two waves that share an n-half split its 8 fragments (w & 1 selects fragments 4 (w & 1) ..).
    const bool do_bias = q.gb != nullptr && (tile % q.tiles_k) == 0;
    f32x4 bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};
    asm volatile("" : "+v"(ones));                // kept in registers: rematerialised in front of the inline-asm MFMA it would be a VALU write the
                                                  // hazard recogniser cannot pair with its reader
    auto bias_mfmas = [&](const bf16x8 (&af)[8]) {
        if (do_bias) {
            if (w & 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) mfma16_vgpr(bacc[i], af[4 + i], ones);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) mfma16_vgpr(bacc[i], af[i], ones);
            }
        }
    };

    // Software pipeline: while the MFMAs of stage st run out of registers, the transpose reads of stage st+1
    // fill the other fragment set and the LDS-direct loads of stages st+2 .. st+4 are in flight.
    auto read_frag = [&](auto SLOT, int f, bf16x8 (&af)[8], bf16x8 (&bfr)[8]) {   // f = 0..7: A fragment f, 8..15: B fragment f-8; SLOT: ring slot
        constexpr int slot = decltype(SLOT)::value;
        const int i = f & 7;
        const int off = (slot & 1) * (STB256 / 2) + 64 * (i >> 2);        // elements
        if (f < 8) af[i] = tr_frag(pfa[slot >> 1][i & 3], off, off + 4 * 256);
        else bfr[i] = tr_frag(pfb[slot >> 1][i & 3], off, off + 4 * 256);
    };
    auto read_frags = [&](auto SLOT, bf16x8 (&af)[8], bf16x8 (&bfr)[8]) {
#pragma unroll
        for (int f = 0; f < 16; ++f) read_frag(SLOT, f, af, bfr);
    };
    auto mfmas = [&](const bf16x8 (&af)[8], const bf16x8 (&bfr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) mfma16_agpr(acc[i][j], af[i], bfr[j]);
        bias_mfmas(af);
    };
    // One pipeline step.  Stage st+1 must have landed for everyone; then, in 16 groups pinned by scheduling
    // barriers, the wave issues {a global->LDS load of stage st+4 (every other group), the two transpose reads of
    // one fragment of stage st+1, four MFMAs of stage st}.  The three pipes (TA 64 B/clk, LDS, MFMA) are fed at
    // their own rates; issued in bulk, the loads and reads fill their queues and stall the wave's MFMA issue.
    auto step = [&](int st, auto NSLOT, const bf16x8 (&caf)[8], const bf16x8 (&cbf)[8], bf16x8 (&naf)[8], bf16x8 (&nbf)[8]) {   // NSLOT = (st + 1) & 3
        wait_vmcnt<16>();                                      // own loads of stage st+1 done (st+2, st+3 may fly)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // own LDS reads of stage st done before its buffer is recycled
        __syncthreads();
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) {
            if ((gq & 1) == 0) issue1(st + 4, gq >> 1);        // into the buffer stage st used
            read_frag(NSLOT, gq, naf, nbf);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * gq + u;
                mfma16_agpr(acc[e >> 3][e & 7], caf[e >> 3], cbf[e & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        bias_mfmas(caf);
    };
    bf16x8 af0[8], bf0[8], af1[8], bf1[8];
    issue(0);
    issue(1);
    issue(2);
    issue(3);
    wait_vmcnt<24>();
    __syncthreads();
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2; typedef std::integral_constant<int, 3> S3;
    read_frags(S0{}, af0, bf0);
    int st = 0;
    for (; st + 4 <= nst; st += 4) {         // st = 0 (mod 4) here: the slots of the stages read ahead are static
        step(st, S1{}, af0, bf0, af1, bf1);
        step(st + 1, S2{}, af1, bf1, af0, bf0);
        step(st + 2, S3{}, af0, bf0, af1, bf1);
        step(st + 3, S0{}, af1, bf1, af0, bf0);
    }
    if (st + 2 <= nst) {
        step(st, S1{}, af0, bf0, af1, bf1);
        step(st + 1, S2{}, af1, bf1, af0, bf0);
        st += 2;
    }
    if (st < nst) mfmas(af0, bf0);   // odd stage count: the last stage's fragments are already in registers
    wait_vmcnt<0>();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA -> accumulator read-out hazard
    // unsplit problems (S == 1) fold straight into the gradient; split ones leave a partial tile for the reduce kernel.
    // The accumulator layout (lane = one column of a 16-wide tile) would store 64-byte runs; each wave passes its
    // 16 x 128 strips through a private LDS staging area (the pipeline buffers are idle now) and stores / read-modify-
    // writes 512-byte row segments instead.
    const bool direct = q.S == 1;
    if (do_bias && c16 == 0) {
        // D[n][.] layout: lane group g holds rows 4g .. 4g + 3 of the 16-row fragment (replicated over the 16 columns)
        float* bo = direct ? q.gb : q.wsb + (int64_t)s * q.Nn;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn + 16 * (4 * (w & 1) + i) + 4 * g + r;
                if (n < q.Nn) bo[n] = (direct && P.beta != 0.f) ? P.beta * bo[n] + bacc[i][r] : bacc[i][r];
            }
    }
    float* out = direct ? q.C : q.ws + (int64_t)s * q.Nn * q.Kk;
    const float beta = direct ? P.beta : 0.f;
    __syncthreads();                                  // every wave is done reading the operand images
    constexpr int SR = 132;                           // staging row stride (floats): 4 rows apart = 16 banks apart
    DGX_LDS float* stg = reinterpret_cast<DGX_LDS float*>((DGX_LDS unsigned char*)lds_raw) + w * (16 * SR);
    DGX_LDS float* stg_w = stg + (4 * g) * SR + c16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg_w[r * SR + 16 * j] = acc[i][j][r];
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) {
            const int idx = qd * 64 + l, row = idx >> 5, c4 = idx & 31;
            const f32x4 v = *reinterpret_cast<DGX_LDS const f32x4*>(stg + row * SR + 4 * c4);
            const int n = n0 + wn + 16 * i + row, k = k0 + wk + 4 * c4;
            if (n < q.Nn && k < q.Kk) {                // Kk % 8 == 0: a quad never straddles the edge
                f32x4* o = reinterpret_cast<f32x4*>(out + (int64_t)n * (direct ? q.ldc : q.Kk) + k);
                *o = beta != 0.f ? beta * *o + v : v;
            }
        }
    }
}

// LPQ = lanes per output quad: 1 for large outputs; 16 when the output is small and the split deep (a 192 x 48 patch-embed
// weight is cut into 256 slabs: one lane per quad would walk them serially, 2 300 lanes for the whole launch).
// The last `bias_blocks` workgroups of the launch fold the bias-gradient slabs of the split problems instead
// (gb[n] = beta * gb[n] + sum_s wsb[s][n], one lane per column, fixed order; block = (problem, 256 columns)): round 2 ran a
// third launch for them (41 launches and 0.4 ms per step of one-wave-deep latency).
template <int LPQ>
__global__ __launch_bounds__(256) void wgrad256_reduce_kernel(Params256 P, int64_t total4, int bias_blocks, int bias_cb) {
    const int main_blocks = (int)gridDim.x - bias_blocks;
    if ((int)blockIdx.x >= main_blocks) {
        const int j = (int)blockIdx.x - main_blocks, pi = j / bias_cb;
        const Prob256& q = P.p[pi];
        if (pi >= P.n || q.S <= 1 || !q.gb) return;
        const int n = (j - pi * bias_cb) * 256 + threadIdx.x;
        if (n >= q.Nn) return;
        const float* w = q.wsb + n;
        float a = 0.f;
        int s = 0;
        for (; s + 4 <= q.S; s += 4) {             // four slabs in flight; the additions keep the slab order
            const float v0 = w[(int64_t)s * q.Nn], v1 = w[(int64_t)(s + 1) * q.Nn], v2 = w[(int64_t)(s + 2) * q.Nn], v3 = w[(int64_t)(s + 3) * q.Nn];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; s < q.S; ++s) a += w[(int64_t)s * q.Nn];
        q.gb[n] = P.beta != 0.f ? P.beta * q.gb[n] + a : a;
        return;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4 * LPQ; t += (int64_t)main_blocks * blockDim.x) {
        const int64_t i = t / LPQ;
        const int sub = (int)(t - i * LPQ);
        int pi = 0;
#pragma unroll
        for (int k = 1; k < MAXP256; ++k)
            if (k < P.n && P.p[k].S > 1 && i >= P.p[k].red0) pi = k;
        const Prob256& q = P.p[pi];
        const int64_t e = i - q.red0, n4 = (int64_t)q.Nn * q.Kk / 4;
        const float4* ws = reinterpret_cast<const float4*>(q.ws);
        float4 a = {0.f, 0.f, 0.f, 0.f};
        int s = sub;
        for (; s + 3 * LPQ < q.S; s += 4 * LPQ) {      // four slabs in flight; the additions keep the slab order
            const float4 v0 = ws[(int64_t)s * n4 + e], v1 = ws[(int64_t)(s + LPQ) * n4 + e], v2 = ws[(int64_t)(s + 2 * LPQ) * n4 + e],
                         v3 = ws[(int64_t)(s + 3 * LPQ) * n4 + e];
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
            a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
            a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
        }
        for (; s < q.S; s += LPQ) {
            const float4 v = ws[(int64_t)s * n4 + e];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (LPQ > 1) {
#pragma unroll
            for (int o = LPQ / 2; o > 0; o >>= 1) {
                a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
            }
            if (sub != 0) continue;
        }
        const int64_t k4 = q.Kk / 4, rowc = e / k4;
        float4* C = reinterpret_cast<float4*>(q.C + rowc * q.ldc) + (e - rowc * k4);
        if (P.beta != 0.f) {
            const float4 c = *C;
            a.x += P.beta * c.x; a.y += P.beta * c.y; a.z += P.beta * c.z; a.w += P.beta * c.w;
        }
        *C = a;
    }
}

// rows per workgroup R (multiple of BM256) such that sum_p tiles_p * ceil(M_p / R) fits one round of 256 workgroups
static void plan256(const dgx_wgrad_problem* pr, int n, int* S, int* slab) {
    int tiles[MAXP256], maxM = 0;
    for (int i = 0; i < n; ++i) {
        tiles[i] = ((pr[i].Nn + T256 - 1) / T256) * ((pr[i].Kk + T256 - 1) / T256);
        if (pr[i].M > maxM) maxM = pr[i].M;
    }
    int R = BM256;
    for (;; R += BM256) {
        int64_t wgs = 0;
        for (int i = 0; i < n; ++i) wgs += (int64_t)tiles[i] * ((pr[i].M + R - 1) / R);
        if (wgs <= 256 || R >= maxM) break;
    }
    for (int i = 0; i < n; ++i) {
        S[i] = (pr[i].M + R - 1) / R;
        if (S[i] < 1) S[i] = 1;
        int sl = (pr[i].M + S[i] - 1) / S[i];
        slab[i] = (sl + BM256 - 1) / BM256 * BM256;
    }
}

static int64_t ws_floats256(const dgx_wgrad_problem* pr, int n, const int* S, int64_t* off, int64_t* offb = nullptr) {
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        if (off) off[i] = tot;
        if (S[i] > 1) tot += (int64_t)S[i] * pr[i].Nn * pr[i].Kk;
    }
    for (int i = 0; i < n; ++i) {                  // partial column sums of the split problems that carry a bias gradient
        if (offb) offb[i] = tot;
        if (S[i] > 1 && pr[i].gb) tot += ((int64_t)S[i] * pr[i].Nn + 3) / 4 * 4;
    }
    return tot;
}

// wgrad_lw.hip: the persistent loader-wave form for groups of many tiles (no M-split, no workspace)
bool wgrad_lw_wants(const dgx_wgrad_problem* pr, int n);
int wgrad_lw_launch(const dgx_wgrad_problem* pr, int n, float beta, hipStream_t st);
constexpr int MAXP_GROUP = 32;           // problems per dgx_linear_wgrad_grouped call: <= 12 on the split-M form, <= 32 on the loader-wave form

extern "C" int dgx_wgrad_grouped_form(const dgx_wgrad_problem* problems, int n) {
    return problems && n > 0 && n <= MAXP_GROUP && wgrad_lw_wants(problems, n) ? 1 : 0;
}

// workspace of the split-M form for a group (what wgrad_grouped_impl writes): the ONLY place its size is computed
static int64_t split_workspace_bytes(const dgx_wgrad_problem* problems, int n) {
    if (!problems || n <= 0 || n > MAXP256) return 0;
    int S[MAXP256], slab[MAXP256];
    plan256(problems, n, S, slab);
    return ws_floats256(problems, n, S, nullptr) * 4;
}
extern "C" int64_t dgx_wgrad_grouped_workspace_bytes(const dgx_wgrad_problem* problems, int n) {
    if (!problems || n <= 0 || n > MAXP_GROUP) return 0;
    // a group the loader-wave form takes needs none -- unless its launch checks can still send it to the split-M form (n <= 12):
    // then the split form's size is reported, so the fall-back in dgx_linear_wgrad_grouped never writes past the caller's buffer
    if (wgrad_lw_wants(problems, n) && n > MAXP256) return 0;
    return split_workspace_bytes(problems, n);
}

static int wgrad_grouped_impl(const dgx_wgrad_problem* problems, const int* ldc, int n, float beta, void* workspace, void* stream);

extern "C" int dgx_linear_wgrad_grouped(const dgx_wgrad_problem* problems, int n, float beta, void* workspace,
                                        void* stream) {
    double fl = 0.0, by = 0.0;
    if (!problems || n <= 0 || n > MAXP_GROUP) return DGX_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {      // dY, X read once (bf16); fp32 gradient read + written (bias: free)
        fl += 2.0 * problems[i].M * problems[i].Nn * problems[i].Kk;
        by += 2.0 * problems[i].M * ((double)problems[i].Nn + problems[i].Kk) + 8.0 * problems[i].Nn * problems[i].Kk;
    }
    DgxProfScope prof(DGX_PROF_WGRAD, stream, fl, by);
    if (wgrad_lw_wants(problems, n)) {
        const int rc = wgrad_lw_launch(problems, n, beta, (hipStream_t)stream);
        if (rc == DGX_OK) { DGX_LAUNCH_CHECK(); return DGX_OK; }
        if (rc != DGX_ERR_UNSUPPORTED || n > MAXP256) return rc;      // an operand over 2^31 bytes, ...: the split-M form takes <= 12 problems
    }
    return wgrad_grouped_impl(problems, nullptr, n, beta, workspace, stream);
}

// Weight gradient of a 3x3 convolution (pad 1, stride 1) without a column matrix: dW[co][tap][ci] = sum over the padded grid of
// dypad[m][co] * xpad[m + shift(tap)][ci] -- nine Linear-type problems over the SAME two zero-bordered images (dgx_conv3x3_pad:
// border rows of dypad are zero, so border positions add nothing), the tap being a pointer offset; gw f32 (Cout, 3, 3, Cin).
extern "C" int64_t dgx_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    dgx_wgrad_problem pr[9];
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    for (int t = 0; t < 9; ++t) { pr[t].dy = pr[t].x = nullptr; pr[t].gw = nullptr; pr[t].gb = nullptr; pr[t].M = (int)Mp; pr[t].Nn = Cout; pr[t].Kk = Cin; }
    return split_workspace_bytes(pr, 9);       // the tap problems always run on the split-M form (ldc = 9 Cin), whatever wgrad_lw_wants says
}
extern "C" int dgx_conv3x3_wgrad_bias(const void* dypad, const void* xpad, float* gw, float* gb, int N, int H, int W, int Cin, int Cout,
                                      float beta, void* workspace, void* stream);
extern "C" int dgx_conv3x3_wgrad(const void* dypad, const void* xpad, float* gw, int N, int H, int W, int Cin, int Cout, float beta,
                                 void* workspace, void* stream) {
    return dgx_conv3x3_wgrad_bias(dypad, xpad, gw, nullptr, N, H, W, Cin, Cout, beta, workspace, stream);
}
extern "C" int64_t dgx_conv3x3_wgrad_bias_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    dgx_wgrad_problem pr[9];
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    float dummy;
    for (int t = 0; t < 9; ++t) { pr[t].dy = pr[t].x = nullptr; pr[t].gw = nullptr; pr[t].gb = t == 0 ? &dummy : nullptr; pr[t].M = (int)Mp; pr[t].Nn = Cout; pr[t].Kk = Cin; }
    return split_workspace_bytes(pr, 9);
}
extern "C" int dgx_conv3x3_wgrad_bias(const void* dypad, const void* xpad, float* gw, float* gb, int N, int H, int W, int Cin, int Cout,
                                      float beta, void* workspace, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return DGX_OK;
    if (!dypad || !xpad || !gw || (Cin & 7) || (Cout & 7)) return DGX_ERR_BAD_ARG;
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    const int wp = W + 2;
    dgx_wgrad_problem pr[9];
    int ldc[9];
    for (int t = 0; t < 9; ++t) {
        pr[t].dy = (const uint16_t*)dypad + (int64_t)(wp + 1) * Cout;                          // grid position 0 (behind the slack)
        pr[t].x = (const uint16_t*)xpad + (int64_t)((t / 3) * wp + t % 3) * Cin;               // position 0 shifted by tap - (wp + 1)
        pr[t].gw = gw + (int64_t)t * Cin;
        pr[t].gb = t == 0 ? gb : nullptr;        // the bias gradient = column sums of dypad (its border rows are zero): once
        pr[t].M = (int)Mp; pr[t].Nn = Cout; pr[t].Kk = Cin;
        ldc[t] = 9 * Cin;
    }
    // the nine taps share the two padded images; useful work = the H x W interior
    DgxProfScope prof(DGX_PROF_WGRAD, stream, 2.0 * N * H * W * 9.0 * Cin * Cout, 2.0 * Mp * ((double)Cin + Cout) + 8.0 * 9.0 * Cin * Cout);
    return wgrad_grouped_impl(pr, ldc, 9, beta, workspace, stream);
}

static int wgrad_grouped_impl(const dgx_wgrad_problem* problems, const int* ldc, int n, float beta, void* workspace, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!problems || n > MAXP256) return DGX_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
        const dgx_wgrad_problem& p = problems[i];
        if (!p.dy || !p.x || !p.gw || p.M <= 0 || p.Nn <= 0 || p.Kk <= 0 || (p.Nn & 7) || (p.Kk & 7)) return DGX_ERR_BAD_ARG;
        if ((int64_t)p.M * p.Nn * 2 >= (1ll << 31) || (int64_t)p.M * p.Kk * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    }
    int S[MAXP256], slab[MAXP256];
    int64_t off[MAXP256], offb[MAXP256];
    plan256(problems, n, S, slab);
    if (ws_floats256(problems, n, S, off, offb) > 0 && !workspace) return DGX_ERR_BAD_ARG;      // a split plan without a buffer to fold through
    Params256 P;
    P.n = n;
    P.beta = beta;
    P.nimg = 0;
    int wg = 0;
    int64_t red = 0;
    for (int i = 0; i < n; ++i) {
        const dgx_wgrad_problem& p = problems[i];
        Prob256& q = P.p[i];
        q.A = (const uint16_t*)p.dy;
        q.B = (const uint16_t*)p.x;
        q.C = p.gw;
        q.gb = p.gb;
        q.wsb = (float*)workspace + offb[i];
        q.ws = (float*)workspace + off[i];
        q.M = p.M; q.Nn = p.Nn; q.Kk = p.Kk;
        q.ldc = ldc ? ldc[i] : p.Kk;
        q.tiles_k = (p.Kk + T256 - 1) / T256;
        q.S = S[i];
        q.slab = slab[i];
        q.tiles = ((p.Nn + T256 - 1) / T256) * q.tiles_k;
        q.wg0 = wg;
        q.red0 = red;
        wg += q.tiles * S[i];
        if (S[i] > 1) red += (int64_t)p.Nn * p.Kk / 4;   // unsplit problems are finished by the GEMM kernel itself
    }
    for (int i = n; i < MAXP256; ++i) P.p[i] = P.p[0];
    hipStream_t st = (hipStream_t)stream;
    const size_t sm = (size_t)NB256 * STB256;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)wgrad256_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        once = true;
    }
    P.total = wg;
    P.interleave = 0;
    if (ldc && n > 1) {                            // dgx_conv3x3_wgrad: same M / Nn / Kk for every tap
        P.interleave = 1;
        for (int i = 1; i < n; ++i)
            if (P.p[i].tiles != P.p[0].tiles || P.p[i].S != P.p[0].S || P.p[i].M != P.p[0].M) P.interleave = 0;
    }
    P.per_xcd = (wg + 7) / 8;
    hipLaunchKernelGGL(wgrad256_partial_kernel, dim3(8 * P.per_xcd), dim3(256), sm, st, P);
    int bias_cb = 1, bias_blocks = 0;              // bias-gradient folds of the split problems ride in the reduce launch
    {
        int maxNn = 0;
        for (int i = 0; i < n; ++i)
            if (S[i] > 1 && problems[i].gb) maxNn = problems[i].Nn > maxNn ? problems[i].Nn : maxNn;
        if (maxNn > 0) { bias_cb = (maxNn + 255) / 256; bias_blocks = bias_cb * n; }
    }
    if (red > 0 || bias_blocks > 0) {
        int maxS = 1;
        for (int i = 0; i < n; ++i) maxS = S[i] > maxS ? S[i] : maxS;
        if (red > 0 && red <= 32768 && maxS >= 16) {   // small output, deep split: 16 lanes share the walk over the slabs
            // (total4 * 16 is a multiple of 64, so the xor-shuffles never mix lanes of different quads with idle ones)
            hipLaunchKernelGGL(wgrad256_reduce_kernel<16>, dim3((int)((red * 16 + 255) / 256) + bias_blocks), dim3(256), 0, st, P, red,
                               bias_blocks, bias_cb);
        } else {
            const int grid = (int)((red + 255) / 256 < 8192 ? (red + 255) / 256 : 8192);
            hipLaunchKernelGGL(wgrad256_reduce_kernel<1>, dim3(grid + bias_blocks), dim3(256), 0, st, P, red, bias_blocks, bias_cb);
        }
    }
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}

// dW += sum over SEVERAL zero-bordered image pairs (dypad_i, xpad_i) of the same convolution -- the FPN levels under one CenterNet tower
// layer, whose weights are shared -- in ONE partial launch + ONE reduce launch: the nine tap problems run over all images' rows, cut
// into slabs of R rows that never straddle an image (R = the smallest multiple of 32 for which 9 x sum_i ceil(M_i / R) workgroups fit
// one round of 256).  Round 2 / early round 3: one partial + one reduce launch PER image (the P5 - P7 levels: ~30 us of launches for
// microseconds of work each).
static int conv_multi_plan(const dgx_conv_wgrad_item* items, int n, int tiles, int* R_out, int* nsl, int* M) {
    int maxM = 0;
    for (int i = 0; i < n; ++i) {
        M[i] = items[i].N * (items[i].H + 2) * (items[i].W + 2);
        maxM = M[i] > maxM ? M[i] : maxM;
    }
    int R = BM256;
    for (;; R += BM256) {
        int64_t wgs = 0;
        for (int i = 0; i < n; ++i) wgs += (int64_t)9 * tiles * ((M[i] + R - 1) / R);
        if (wgs <= 256 || R >= maxM) break;
    }
    int S = 0;
    for (int i = 0; i < n; ++i) { nsl[i] = (M[i] + R - 1) / R; S += nsl[i]; }
    *R_out = R;
    return S;
}

extern "C" int64_t dgx_conv3x3_wgrad_bias_multi_workspace_bytes(const dgx_conv_wgrad_item* items, int n, int Cin, int Cout) {
    if (!items || n <= 0 || n > 6) return 0;
    if (n == 1) return dgx_conv3x3_wgrad_bias_workspace_bytes(items[0].N, items[0].H, items[0].W, Cin, Cout);
    int R, nsl[6], M[6];
    const int tiles = ((Cout + T256 - 1) / T256) * ((Cin + T256 - 1) / T256);
    const int S = conv_multi_plan(items, n, tiles, &R, nsl, M);
    return ((int64_t)9 * S * Cout * Cin + ((int64_t)S * Cout + 3) / 4 * 4) * 4;
}

extern "C" int dgx_conv3x3_wgrad_bias_multi(const dgx_conv_wgrad_item* items, int n, float* gw, float* gb, int Cin, int Cout, float beta,
                                            void* workspace, void* stream) {
    if (n <= 0) return DGX_OK;
    if (!items || !gw || !workspace || (Cin & 7) || (Cout & 7) || Cin <= 0 || Cout <= 0) return DGX_ERR_BAD_ARG;
    if (n > 6) return DGX_ERR_UNSUPPORTED;
    if (n == 1)                                    // (a single slab would be finished by the partial kernel itself: the per-image plan)
        return dgx_conv3x3_wgrad_bias(items[0].dypad, items[0].xpad, gw, gb, items[0].N, items[0].H, items[0].W, Cin, Cout, beta, workspace,
                                      stream);
    int R, nsl[6], M[6];
    const int tiles_k = (Cin + T256 - 1) / T256, tiles = ((Cout + T256 - 1) / T256) * tiles_k;
    for (int i = 0; i < n; ++i) {
        if (!items[i].dypad || !items[i].xpad || items[i].N <= 0 || items[i].H <= 0 || items[i].W <= 0) return DGX_ERR_BAD_ARG;
        const int64_t rows = (int64_t)items[i].N * (items[i].H + 2) * (items[i].W + 2) + 2 * (int64_t)(items[i].W + 3);
        if (rows * Cout * 2 >= (1ll << 31) || rows * Cin * 2 >= (1ll << 31)) return DGX_ERR_UNSUPPORTED;
    }
    const int S = conv_multi_plan(items, n, tiles, &R, nsl, M);
    Params256 P;
    P.n = 9;
    P.beta = beta;
    P.nimg = n;
    int s0 = 0;
    double fl = 0.0, by = 8.0 * 9.0 * Cin * Cout;
    for (int i = 0; i < n; ++i) {
        P.img[i].dy = (const uint16_t*)items[i].dypad; P.img[i].x = (const uint16_t*)items[i].xpad;
        P.img[i].wp = items[i].W + 2; P.img[i].M = M[i]; P.img[i].slab0 = s0;
        s0 += nsl[i];
        fl += 2.0 * items[i].N * items[i].H * items[i].W * 9.0 * Cin * Cout;
        by += 2.0 * M[i] * ((double)Cin + Cout);
    }
    for (int i = n; i < 6; ++i) P.img[i] = P.img[0];
    float* ws = (float*)workspace;
    const int64_t per = (int64_t)S * Cout * Cin;
    int wg = 0;
    int64_t red = 0;
    for (int t = 0; t < 9; ++t) {
        Prob256& q = P.p[t];
        q.A = P.img[0].dy; q.B = P.img[0].x;       // patched per slab in the kernel
        q.C = gw + (int64_t)t * Cin;
        q.gb = t == 0 ? gb : nullptr;              // the bias gradient = column sums of dypad (its border rows are zero): once
        q.wsb = ws + 9 * per;
        q.ws = ws + (int64_t)t * per;
        q.M = M[0]; q.Nn = Cout; q.Kk = Cin;
        q.ldc = 9 * Cin;
        q.tiles_k = tiles_k;
        q.S = S;
        q.slab = R;
        q.tiles = tiles;
        q.wg0 = wg;
        q.red0 = red;
        wg += tiles * S;
        red += (int64_t)Cout * Cin / 4;
    }
    for (int i = 9; i < MAXP256; ++i) P.p[i] = P.p[0];
    hipStream_t st = (hipStream_t)stream;
    const size_t sm = (size_t)NB256 * STB256;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)wgrad256_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        once = true;
    }
    P.total = wg;
    P.interleave = 1;
    P.per_xcd = (wg + 7) / 8;
    DgxProfScope prof(DGX_PROF_WGRAD, stream, fl, by);
    hipLaunchKernelGGL(wgrad256_partial_kernel, dim3(8 * P.per_xcd), dim3(256), sm, st, P);
    const int bias_cb = (Cout + 255) / 256, bias_blocks = gb ? bias_cb * 9 : 0;
    if (red <= 32768 && S >= 16) {
        hipLaunchKernelGGL(wgrad256_reduce_kernel<16>, dim3((int)((red * 16 + 255) / 256) + bias_blocks), dim3(256), 0, st, P, red, bias_blocks,
                           bias_cb);
    } else {
        const int grid = (int)((red + 255) / 256 < 8192 ? (red + 255) / 256 : 8192);
        hipLaunchKernelGGL(wgrad256_reduce_kernel<1>, dim3(grid + bias_blocks), dim3(256), 0, st, P, red, bias_blocks, bias_cb);
    }
    DGX_LAUNCH_CHECK();
    return DGX_OK;
}
