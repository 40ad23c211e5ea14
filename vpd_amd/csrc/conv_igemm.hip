// Implicit-GEMM convolution on bf16 MFMA (gfx950).  One kernel serves the
// forward convolutions (7x7 s2 stem as 7 row-taps of 64 contiguous values,
// 3x3 s1/s2, 1x1 s2) and the data-gradient convolutions (3x3 s1 as a flipped
// conv; stride-2 convs as one launch per output parity class).
//
// GEMM view: rows = output channels (Co), cols = output pixels (M), K = taps x Kc.
// The MFMA "A" operand is the weight tile, the "B" operand the gathered pixel
// tile, so each lane ends up with 4 consecutive channels of one pixel and the
// NHWC store is 8 bytes per lane.
//
// Pipeline per 64-deep K-step: global->register prefetch of step s+1 is issued
// before the MFMAs of step s, written to the other LDS buffer after them; one
// barrier per step.  LDS tiles are [rows][64] bf16 (128-B rows) with the 16-B
// chunk index XOR-ed by (row & 7): conflict-free for ds_read_b128 fragments.
#include <stdlib.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

template <int BM, int BN, int WM, int WN, int EPM>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p0) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM;          // pixels per wave
    constexpr int WTN = BN / WN;          // channels per wave
    constexpr int MI = WTM / 16;
    constexpr int NI = WTN / 16;
    constexpr int PR = BM / 32;           // pixel rows staged per thread
    constexpr int WR = BN / 32;           // weight rows staged per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sP = reinterpret_cast<bf16_t*>(smem);                 // [2][BM*64]
    bf16_t* sW = sP + 2 * BM * 64;                                // [2][BN*64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int mtile = blockIdx.x;
    // second convolution of the launch (ConvParams::alt_*): these blocks see it as THE convolution
    ConvParams p = p0;
    int by = blockIdx.y;
    if (p0.alt_w && by >= p0.alt_y0) {
        by -= p0.alt_y0;
        p.w = p0.alt_w; p.y = p0.alt_y; p.stats = p0.alt_stats; p.taps = p0.alt_taps;
        p.ep_scale = p0.alt_ep_scale; p.ep_shift = p0.alt_ep_shift; p.ep_relu = p0.alt_ep_relu; p.res = nullptr;
    }
    const int n0 = by * BN;
    const int m0 = mtile * BM;
    // parity class of a merged stride-2 data gradient (blockIdx.z); class 0 lives in the top-level fields
    ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    TapSet taps = p.taps;
    // heaviest class first: the classes come in ascending tap count (1, 2, 2, 4 taps for a 3x3 kernel) and blocks are
    // dispatched z-slowest, so walking them backwards keeps the 4-tap blocks out of the tail of the launch
    const int cls = (int)gridDim.z - 1 - (int)blockIdx.z;
    switch (cls) {
        case 1: geo = p.cls[0].geo; taps = p.cls[0].taps; break;
        case 2: geo = p.cls[1].geo; taps = p.cls[1].taps; break;
        case 3: geo = p.cls[2].geo; taps = p.cls[2].taps; break;
        default: break;
    }
    if (m0 >= geo.M) return;

    const int piece = tid & 7;
    const int row0 = tid >> 3;

    // per-thread gather bases (element offsets into x) for its PR pixel rows
    int pixbase[PR];
    const int HW = geo.Hs * geo.Ws;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
        int m = m0 + row0 + 32 * i;
        m = m < geo.M ? m : geo.M - 1;
        const int b = m / HW;
        const int r = m - b * HW;
        const int yy = r / geo.Ws;
        const int xx = r - yy * geo.Ws;
        pixbase[i] = ((b * p.xHp + yy * p.istr) * p.xWp + xx * p.istr) * p.xC + piece * 8;
    }
    const int kchunks = p.Kc >> 6;
    const int nsteps1 = taps.nr * taps.nc * kchunks;
    // class 0 only: extra K-steps from the second input tensor (same pixels, first tap's offset, its own weights)
    const int nsteps = nsteps1 + ((p.x2 && cls == 0) ? (p.Kc2 >> 6) : 0);

    u32x4 rp[PR], rw[WR];
    auto load_step = [&](int s) __attribute__((always_inline)) {
        const bf16_t* xs = p.x;
        const bf16_t* wb;
        int toff, wstride;
        if (s < nsteps1) {
            const int tap = s / kchunks;
            const int cc = s - tap * kchunks;
            const int ir = tap / taps.nc;
            const int ic = tap - ir * taps.nc;
            toff = ((taps.dy0 + ir * taps.dys) * p.xWp + (taps.dx0 + ic * taps.dxs)) * p.xC + cc * 64;
            const int wsl = taps.w0 + ir * taps.wrs + ic * taps.wcs;
            wb = p.w + ((size_t)wsl * p.Co + n0) * p.Kc + cc * 64 + piece * 8;
            wstride = p.Kc;
        } else {
            const int cc = s - nsteps1;
            xs = p.x2;
            toff = (taps.dy0 * p.xWp + taps.dx0) * p.xC + cc * 64;
            wb = p.w2 + (size_t)n0 * p.Kc2 + cc * 64 + piece * 8;
            wstride = p.Kc2;
        }
#pragma unroll
        for (int i = 0; i < PR; ++i)
            rp[i] = *reinterpret_cast<const u32x4*>(xs + pixbase[i] + toff);
#pragma unroll
        for (int i = 0; i < WR; ++i)
            rw[i] = *reinterpret_cast<const u32x4*>(wb + (size_t)(row0 + 32 * i) * wstride);
    };
    auto store_step = [&](int buf) __attribute__((always_inline)) {
        bf16_t* dP = sP + buf * BM * 64;
        bf16_t* dW = sW + buf * BN * 64;
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            const int r = row0 + 32 * i;
            *reinterpret_cast<u32x4*>(dP + r * 64 + ((piece ^ (r & 7)) << 3)) = rp[i];
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int r = row0 + 32 * i;
            *reinterpret_cast<u32x4*>(dW + r * 64 + ((piece ^ (r & 7)) << 3)) = rw[i];
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_step(0);
    store_step(0);
    __syncthreads();

    const int fr = lane & 15;
    const int fq = lane >> 4;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_step(s + 1);
        const bf16_t* cP = sP + buf * BM * 64;
        const bf16_t* cW = sW + buf * BN * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[NI], bfm[MI];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int r = wn * WTN + a * 16 + fr;
                af[a] = *reinterpret_cast<const bf16x8*>(cW + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int r = wm * WTM + b * 16 + fr;
                bfm[b] = *reinterpret_cast<const bf16x8*>(cP + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
        }
        if (s + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }
    float st1[BN / WN / 16][4], st2[BN / WN / 16][4];
#pragma unroll
    for (int a = 0; a < BN / WN / 16; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo);
    if (p.stats) conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, mtile, n0, smem);
}

// ---------------------------------------------------------------------------
// 3x3 stride-1 convolution with an LDS-resident NHWC halo tile.
//
// The output tile is BM consecutive pixels of whole image rows (BM % W == 0).  In the zero-bordered
// NHWC layout the input pixels all nine taps need form ONE contiguous run of padded rows (also across
// images, whose padded planes are adjacent), so the halo [NHP pixels][64 channels] is brought to LDS
// once per 64-channel chunk by LDS-DMA (global_load_lds, 16 B per lane, swizzle applied on the source
// address) and every tap reads its pixel fragments from it at a shifted row.  Only the per-tap weight
// tile [BN][64] streams (double-buffered LDS-DMA, prefetched one tap ahead).  No register staging, one
// barrier per 64-deep K-step.
// ---------------------------------------------------------------------------
#include "conv_pws.h"

template <int BM, int BN, int HROWS, bool HALO2>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvParams p, const HaloGeom g) {
    constexpr int WM = 2, WN = 2;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int HB = HALO2 ? 2 : 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sH = reinterpret_cast<bf16_t*>(smem);                 // [HB][HROWS*64]
    bf16_t* sW = sH + HB * HROWS * 64;                            // [2][BN*64]

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const int mtile = blockIdx.x;
    const int n0 = blockIdx.y * BN;
    const int W = p.Ws, H = p.Hs, Wp = W + 2;
    const int Ci = p.Kc;

    // first padded pixel of the halo in the input tensor
    const int gr0 = mtile * g.TR;                                 // global output row = b*H + y
    int prow0;
    if (g.multi) prow0 = (gr0 / H) * (H + 2);
    else { const int b = gr0 / H; prow0 = b * (H + 2) + (gr0 - b * H); }
    const int gp0 = prow0 * Wp;

    // per-lane halo row of each pixel fragment (tap offset added per step)
    int hbase[MI];
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = wm * WTM + b * 16 + fr;
        const int lr = m / W;
        const int xx = m - lr * W;
        const int hrow = g.multi ? (lr / H) * (H + 2) + (lr % H) : lr;
        hbase[b] = hrow * Wp + xx;
    }

    const int piece = tid & 7;
    const int srow = tid >> 3;                                    // 0..31: row within a 32-row pass
    auto issue_halo = [&](int cc, int hb) __attribute__((always_inline)) {
        bf16_t* dst = sH + hb * HROWS * 64;
#pragma unroll
        for (int ps = 0; ps < HROWS / 32; ++ps) {
            const int hp = ps * 32 + srow;
            int gp = gp0 + hp;
            gp = gp < g.total_pix ? gp : g.total_pix - 1;
            const bf16_t* src = p.x + (size_t)gp * Ci + cc * 64 + ((piece ^ (hp & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (ps * 32 + wave * 8) * 64), 16, 0, 0);
        }
    };
    auto issue_w = [&](int step, int wb) __attribute__((always_inline)) {
        const int cc = step / 9;
        const int tap = step - cc * 9;
        const int ir = tap / 3, ic = tap - ir * 3;
        const int wsl = p.taps.w0 + ir * p.taps.wrs + ic * p.taps.wcs;
        bf16_t* dst = sW + wb * BN * 64;
        const bf16_t* wbp = p.w + ((size_t)wsl * p.Co + n0) * Ci + cc * 64;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int n = i * 32 + srow;
            const bf16_t* src = wbp + (size_t)n * Ci + ((piece ^ (n & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (i * 32 + wave * 8) * 64), 16, 0, 0);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = Ci >> 6;
    const int nsteps = nchunks * 9;
    issue_halo(0, 0);
    issue_w(0, 0);

    int step = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        const bf16_t* cH = sH + (HALO2 ? (cc & 1) : 0) * HROWS * 64;
        for (int tap = 0; tap < 9; ++tap, ++step) {
            // everything issued so far (this step's weights, this chunk's halo) has landed; all waves
            // are past the previous step's LDS reads
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (step + 1 < nsteps && !VPD_ABL(p, 1)) issue_w(step + 1, (step + 1) & 1);
            if (HALO2 && tap == 0 && cc + 1 < nchunks && !VPD_ABL(p, 4)) issue_halo(cc + 1, (cc + 1) & 1);
            if (VPD_ABL(p, 2)) continue;

            const int ir = tap / 3, ic = tap - ir * 3;
            const int toff = (p.taps.dy0 + ir * p.taps.dys) * Wp + (p.taps.dx0 + ic * p.taps.dxs);
            const bf16_t* cW = sW + (step & 1) * BN * 64;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[NI], bfm[MI];
                const int chunk = kk * 4 + fq;
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int r = wn * WTN + a * 16 + fr;
                    af[a] = *reinterpret_cast<const bf16x8*>(cW + r * 64 + ((chunk ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const int r = hbase[b] + toff;
                    bfm[b] = *reinterpret_cast<const bf16x8*>(cH + r * 64 + ((chunk ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b)
                        acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
            }
        }
        if (!HALO2 && cc + 1 < nchunks) {
            __builtin_amdgcn_s_barrier();          // every wave is done reading this chunk's halo
            issue_halo(cc + 1, 0);
        }
    }
    __syncthreads();                               // LDS is reused by the statistics reduction
    float st1[BN / WN / 16][4], st2[BN / WN / 16][4];
#pragma unroll
    for (int a = 0; a < BN / WN / 16; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    conv_epilogue<BM, BN, WM, WN>(p, acc, mtile, n0, st1, st2, geo);
    if (p.stats) conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, mtile, n0, smem);
}

// ---------------------------------------------------------------------------
// Warp-specialised 3x3 stride-1 convolution (forward and data-gradient): 8 waves per block.
//   waves 4..7 (loaders): all LDS-DMA.  Per K-step s they issue the weight tile of step s+2 into a
//       3-stage ring and one eighth of the NEXT 64-channel chunk's halo into the other halo buffer, a
//       fixed number of instructions per step so that one counted s_waitcnt vmcnt(PER_STEP) per step
//       retires exactly "everything except the bundle issued one step ago".
//   waves 0..3 (MFMA): only ds_read_b128 + MFMA; wave tile 64 channels x BM/WM pixels.
// One barrier per step: READY_s = weights of step s (and at chunk boundaries the halo) have landed;
// since the MFMA waves only arrive after finishing step s-1 it also frees ring stage (s+2)%3.
// 256-pixel tiles halve the weight bytes streamed per FLOP with respect to the 128-pixel kernels.
// ---------------------------------------------------------------------------
// HB: halo buffers (1 when the conv has a single 64-channel chunk: the smaller footprint lets two blocks share a
// CU and overlap each other's prologue / epilogue); WPS: launch-bounds waves per SIMD (4 = two blocks per CU).
// NMW: MFMA waves (4 = one per SIMD; 8 = two per SIMD, which hide each other's LDS round trips: 12 waves per block, three
// per SIMD, at most 168 registers each)
template <int BM, int BN, int HROWS, int HB, int WPS, int EPM, int NS, int NMW = 4>
__global__ __launch_bounds__((NMW + 4) * 64, WPS) void conv3x3_ws_kernel(const ConvParams p, const HaloGeom g) {
    constexpr int WN = BN / 64;
    constexpr int WM = NMW / WN;
    constexpr int WTM = BM / WM;
    constexpr int MI = WTM / 16, NI = 4;
    constexpr int AHEAD = NS - 1;                    // weight tiles in flight beyond the one being consumed
    constexpr int WSTAGE = BN * 64;
    constexpr int HBUF = HROWS * 64;
    constexpr int W_PER = BN / 32;                   // weight-tile DMA instructions per loader wave per step
    constexpr int HPASS = HROWS / 32;                // halo DMA instructions per loader wave per chunk
    constexpr int H_PER = (HPASS + 7) / 8;           // ... spread over steps 0..7 of the previous chunk
    constexpr int PER_STEP = W_PER + H_PER;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sH = reinterpret_cast<bf16_t*>(smem);                 // [HB][HBUF]
    bf16_t* sW = sH + HB * HBUF;                                  // [NS][WSTAGE]

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = blockIdx.x;
    const int n0 = blockIdx.y * BN;
    const int W = p.Ws, H = p.Hs, Wp = W + 2;
    const int Ci = p.Kc;
    const int nchunks = Ci >> 6;
    const int nsteps = nchunks * 9;

    if (wave >= NMW) {
        // ------------------------------ loader waves ------------------------------
        const int lw = wave - NMW;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        const int gr0 = mtile * g.TR;
        int prow0;
        if (g.multi) prow0 = (gr0 / H) * (H + 2);
        else { const int b = gr0 / H; prow0 = b * (H + 2) + (gr0 - b * H); }
        const int gp0 = prow0 * Wp;
        // chunk rotation (HaloGeom::rot, as conv3x3_pws_kernel: bit-identical results): logical chunk cc is chunk (cc + tile) % nchunks
        const int crot = g.rot ? mtile % nchunks : 0;
        auto halo_instr = [&](int cc, int k) __attribute__((always_inline)) {      // pass k of chunk cc
            const int hp = (lw + 4 * k) * 8 + lrow;
            int gp = gp0 + hp;
            gp = gp < g.total_pix ? gp : g.total_pix - 1;
            const int cce = cc + crot >= nchunks ? cc + crot - nchunks : cc + crot;
            const bf16_t* src = p.x + (size_t)gp * Ci + cce * 64 + ((piece ^ (hp & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sH + (cc & (HB - 1)) * HBUF + (lw + 4 * k) * 8 * 64), 16, 0, 0);
        };
        auto issue_w = [&](int step, int stage) __attribute__((always_inline)) {
            const int cc = step / 9;
            const int tap = step - cc * 9;
            const int ir = tap / 3, ic = tap - ir * 3;
            const int wsl = p.taps.w0 + ir * p.taps.wrs + ic * p.taps.wcs;
            const int cce = cc + crot >= nchunks ? cc + crot - nchunks : cc + crot;
            const bf16_t* wbp = p.w + ((size_t)wsl * p.Co + n0) * Ci + cce * 64;
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const int n = (lw + 4 * i) * 8 + lrow;
                const bf16_t* src = wbp + (size_t)n * Ci + ((piece ^ (n & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + stage * WSTAGE + (lw + 4 * i) * 8 * 64), 16, 0, 0);
            }
        };
#pragma unroll
        for (int k = 0; k < HPASS; ++k) halo_instr(0, k);
        // prologue: the halo of chunk 0 and the first AHEAD weight tiles; only the halo and tile 0 are waited for
        issue_w(0, 0);
#pragma unroll
        for (int k = 1; k < AHEAD; ++k) issue_w(k < nsteps ? k : nsteps - 1, k);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * W_PER) : "memory");
        for (int s = 0; s < nsteps; ++s) {
            // bundles s+1 .. s+AHEAD-1 may still be in flight; bundle s (and everything older) has landed
            // (the first tiles were issued in the prologue without halo slices: exact counts of what is younger than tile s)
            static_assert(AHEAD <= 4 && (AHEAD - 1) * PER_STEP < 64, "vmcnt range / peeled steps");
            if (s == 1 && AHEAD > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD > 2 ? AHEAD - 2 : 0) * W_PER + PER_STEP) : "memory");
            else if (s == 2 && AHEAD > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD > 3 ? AHEAD - 3 : 0) * W_PER + 2 * PER_STEP) : "memory");
            else if (s > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PER_STEP) : "memory");
            __builtin_amdgcn_s_barrier();                         // READY_s
            const int sw = s + AHEAD < nsteps ? s + AHEAD : nsteps - 1;   // tail: harmless reload into a free stage
            if (VPD_ABL(p, 1)) continue;
            issue_w(sw, (s + AHEAD) % NS);
            const int cc = s / 9;
            const int tap = s - cc * 9;
            // next chunk's halo, slice `tap`; on the last chunk (and in slice 8) re-load identical bytes
            const int hc = cc + 1 < nchunks ? cc + 1 : cc;
#pragma unroll
            for (int u = 0; u < H_PER; ++u) {
                int k = tap * H_PER + u;
                k = k < HPASS ? k : HPASS - 1;
                halo_instr(hc, k);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing may land after the LDS is re-purposed
        __builtin_amdgcn_s_barrier();                             // END: every MFMA wave is done with the tiles
        if (p.stats) __builtin_amdgcn_s_barrier();                // matches the barrier inside conv_stats_flush
        if (EPM == 8) {                                           // second flush: scratch re-use barrier + its own
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    int hbase[MI];
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = wm * WTM + b * 16 + fr;
        const int lr = m / W;
        const int xx = m - lr * W;
        const int hrow = g.multi ? (lr / H) * (H + 2) + (lr % H) : lr;
        hbase[b] = hrow * Wp + xx;
    }
    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    int s = 0;
    auto tap_body = [&](int tap, const bf16_t* cH) __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();                             // READY_s
        if (VPD_ABL(p, 2)) return;
        const int ir = tap / 3, ic = tap - ir * 3;
        const int toff = (p.taps.dy0 + ir * p.taps.dys) * Wp + (p.taps.dx0 + ic * p.taps.dxs);
        const bf16_t* cW = sW + (s % NS) * WSTAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[NI], bfm[MI];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int r = wn * 64 + a * 16 + fr;
                af[a] = *reinterpret_cast<const bf16x8*>(cW + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int r = hbase[b] + toff;
                bfm[b] = *reinterpret_cast<const bf16x8*>(cH + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
        }
    };
    for (int cc = 0; cc < nchunks; ++cc) {
        const bf16_t* cH = sH + (cc & (HB - 1)) * HBUF;
        if constexpr (NMW == 8) {
            // 168 registers per wave here: keep the tap loop a loop (unrolled, hipcc hoists the fragment addresses of all nine
            // taps out of the chunk loop: 212 bytes of scratch per lane)
#pragma nounroll
            for (int tap = 0; tap < 9; ++tap, ++s) tap_body(tap, cH);
        } else {
            for (int tap = 0; tap < 9; ++tap, ++s) tap_body(tap, cH);
        }
    }
    // EPM 6 / 7: the epilogue's z fragments and mask bits (first 4 pixel groups) are requested in front of the END barrier
    // (requesting them a chunk earlier costs the main loop 34 registers it does not have: 28-104 bytes of scratch per lane)
    constexpr bool BST = EPM == 6 || EPM == 7 || EPM == 8;
    BstFrag<NI, VPD_BST_MB(MI)> bst;
    BstPair<NI, VPD_BST_MB(MI)> pr;
    if (BST) conv_bst_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, bst);
    if (EPM == 8) {
        conv_bst2_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, pr);
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) pr.s3[a][j] = 0.f;
    }
    __builtin_amdgcn_s_barrier();                                 // END
    float st1[BN / WN / 16][4], st2[BN / WN / 16][4];
#pragma unroll
    for (int a = 0; a < BN / WN / 16; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    if constexpr (EPM == 8) conv_epilogue_pre2<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst, pr);
    else if constexpr (BST) conv_epilogue_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst);
    else conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo);
    if (p.stats) conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, mtile, n0, smem);
    if constexpr (EPM == 8) {      // the second BatchNorm's rows: (sum g, sum g * z2)
        __builtin_amdgcn_s_barrier();                             // the first flush has read the scratch
        conv_stats_flush<BM, BN, WM, WN>(p, st1, pr.s3, mtile, n0, smem, p.stats2);
    }
}

// ---------------------------------------------------------------------------
// Persistent 64 -> 64 channel 3x3 stride-1 convolution (ResNet layer1, forward and data-gradient).
// K is only 576, so a per-tile block would spend most of its life re-streaming the 73.7 KB of weights and
// in prologue / epilogue latency.  Here one block per CU keeps ALL NINE weight taps resident in LDS and
// walks over pixel tiles: the four loader waves fetch the next tile's halo into the other buffer while the
// four MFMA waves run the 9 taps x 2 K-halves of the current tile out of LDS without any barrier, then store.
// One barrier per 128-pixel tile.
// ---------------------------------------------------------------------------
template <int HROWS, int EPM>
__global__ __launch_bounds__(512) void conv3x3_c64_persistent_kernel(const ConvParams p, const HaloGeom g, int ntiles, int per) {
    constexpr int BM = 128, BN = 64, WM = 4, WN = 1;
    constexpr int WTM = BM / WM;
    constexpr int MI = WTM / 16, NI = 4;
    constexpr int HBUF = HROWS * 64;
    constexpr int HPASS = HROWS / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);                 // [9][64*64] resident weights
    bf16_t* sH = sW + 9 * BN * 64;                                // [2][HBUF]
    unsigned char* red = reinterpret_cast<unsigned char*>(sH + 2 * HBUF);      // statistics scratch (2 KiB)

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.Ws, H = p.Hs, Wp = W + 2;
    // Tile walk of a block.  per > 0: `per` CONSECUTIVE tiles (round 4) -- vertically adjacent 128-pixel tiles share two of their
    // six halo rows, and walked by one block the second read of those rows hits the XCD's L2 a few microseconds after the first;
    // strided over the grid (per == 0: tiles b, b + G, ...) the neighbours run at the same time on other XCDs and both
    // reads go out to memory (1.6 x the activation bytes instead of ~1.1 x).
    const int G = per > 0 ? 1 : gridDim.x;
    const int tbeg = per > 0 ? blockIdx.x * per : blockIdx.x;
    const int tend = per > 0 ? (tbeg + per < ntiles ? tbeg + per : ntiles) : ntiles;

    if (wave >= 4) {
        const int lw = wave - 4;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        const float rWp = g.rWp;
        auto issue_halo = [&](int mtile, int buf) __attribute__((always_inline)) {
            const int gr0 = mtile * g.TR;
            int prow0;
            if (g.multi) prow0 = (gr0 / H) * (H + 2);
            else { const int b = gr0 / H; prow0 = b * (H + 2) + (gr0 - b * H); }
            const int gp0 = prow0 * Wp;
#pragma unroll
            for (int k = 0; k < HPASS; ++k) {
                const int hp = (lw + 4 * k) * 8 + lrow;
                // halo pixel hp = (row hr, column xp) of the padded tile keeps its 16-byte pieces XOR-ed with the COLUMN's low bits
                // (HaloGeom::kmask; 16-pixel fragments lie inside one image row here): the MFMA waves' fragment addresses then
                // split into a per-lane constant per tap column and a wave-uniform term (see `lo` below)
                const int hr = vpd_fdiv(hp, rWp);
                const int key = (hp - hr * Wp) & 7;
                int gp = gp0 + hp;
                gp = gp < g.total_pix ? gp : g.total_pix - 1;
                const bf16_t* src = p.x + (size_t)gp * 64 + ((piece ^ key) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sH + buf * HBUF + (lw + 4 * k) * 8 * 64), 16, 0, 0);
            }
        };
        // resident weights: tap t (in loop order) -> slice w0 + ir*wrs + ic*wcs, rows n, 64 channels
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int wsl = p.taps.w0 + (t / 3) * p.taps.wrs + (t % 3) * p.taps.wcs;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = (lw + 4 * i) * 8 + lrow;
                const bf16_t* src = p.w + ((size_t)wsl * 64 + n) * 64 + ((piece ^ (n & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + t * BN * 64 + (lw + 4 * i) * 8 * 64), 16, 0, 0);
            }
        }
        if (tbeg < tend) issue_halo(tbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // B_0
        int i = 0;
        for (int t = tbeg; t < tend; t += G, ++i) {
            if (t + G < tend && !VPD_ABL(p, 4)) issue_halo(t + G, (i + 1) & 1);   // buffer last read in tile i-1, finished before B_i
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // B_{i+1}
        }
        if (p.stats) __builtin_amdgcn_s_barrier();                // matches the barrier inside conv_stats_flush
        return;
    }

    const int wm = wave;                                          // WN == 1
    const int fr = lane & 15;
    const int fq = lane >> 4;
    // pixel-fragment LDS byte offsets inside a halo buffer for tap column ic (K-half 0; K-half 1 = ^ 64): with the column-keyed
    // swizzle the lane-dependent part is tap-ROW independent, so a tap's address is lo[ic][b] + (wave-uniform row term) -- one
    // VALU add instead of five per fragment (this kernel's LDS pipe and issue slots, not its MFMAs, bound it)
    unsigned lo[3][MI];
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = wm * WTM + b * 16 + fr;
        const int lr = m / W;
        const int xx = m - lr * W;
        const int hrow = g.multi ? (lr / H) * (H + 2) + (lr % H) : lr;
#pragma unroll
        for (int ic = 0; ic < 3; ++ic)
            lo[ic][b] = (unsigned)((hrow * Wp + xx) * 128) + ((((unsigned)(xx + p.taps.dx0 + ic * p.taps.dxs) & 7u) ^ (unsigned)fq) << 4);
    }
    // weight-fragment byte offset of this lane inside a tap's [64][64] tile (row fr, K-half 0; rows a * 16 + fr: + a * 2048)
    const unsigned wl0 = (unsigned)(fr * 128 + ((fq ^ (fr & 7)) << 4));
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    typedef const bf16x8 __attribute__((address_space(3)))* frag_t;
    // statistics stay in registers across this block's tiles: one reduction + 128 atomics per BLOCK, not per tile
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    __builtin_amdgcn_s_barrier();                                 // B_0
    int i = 0;
    for (int t = tbeg; t < tend; t += G, ++i) {
        const bf16_t* cH = sH + (i & 1) * HBUF;
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        BstFrag<NI, VPD_BST_MB(MI)> bst;      // EPM 6 / 7: the epilogue's z fragments and mask bits, in flight behind the MFMA loop
        AccFrag<NI, MI> accf;                 // EPM 2 / 7: the old values of y and their mask bits, likewise
        if (EPM == 6 || EPM == 7) conv_bst_prefetch<BM, BN, WM, WN>(p, t, 0, geo, bst);
        if (EPM == 2 || EPM == 7) conv_acc_prefetch<BM, BN, WM, WN>(p, t, 0, geo, accf);
        // ONE MFMA wave per SIMD and no barrier inside a tile: nothing hides an LDS round trip except this wave's own MFMAs.
        // Two fragment sets; step s+1's six ds_read_b128 are issued between the eight MFMAs of step s (hipcc left to itself
        // re-uses one set and waits for each pair of reads right before the MFMA that needs it: ~150 exposed cycles per
        // 128 cycles of matrix work).
        bf16x8 af[2][NI], bfm[2][MI];
        // fragment i of step st: 0 -> af[0], 1 -> bfm[0], 2 -> bfm[1], 3.. -> af[1..] (the order the MFMAs below first need them)
        // (row term of tap row ir: halo buffer + (dy * Wp + dx0) * 128, wave-uniform)
        const unsigned hbb = lds0 + (unsigned)(9 * BN * 64 * 2) + (unsigned)(i & 1) * (HBUF * 2u);
        const unsigned rowS[3] = {hbb + (unsigned)(((p.taps.dy0) * Wp + p.taps.dx0) * 128),
                                  hbb + (unsigned)(((p.taps.dy0 + p.taps.dys) * Wp + p.taps.dx0) * 128),
                                  hbb + (unsigned)(((p.taps.dy0 + 2 * p.taps.dys) * Wp + p.taps.dx0) * 128)};
        const int dxs128 = p.taps.dxs * 128;
        auto ldone = [&](int st, int buf, int i) __attribute__((always_inline)) {
            const int tap = st >> 1, kk = st & 1;
            if (i == 1 || i == 2) {
                const unsigned ad = (lo[tap % 3][i - 1] + rowS[tap / 3] + (unsigned)((tap % 3) * dxs128)) ^ (unsigned)(kk * 64);
                bfm[buf][i - 1] = *(frag_t)(size_t)ad;
            } else {
                const int a = i == 0 ? 0 : i - 2;
                af[buf][a] = *(frag_t)(size_t)(lds0 + (unsigned)(tap * BN * 64 * 2 + a * 2048) + (wl0 ^ (unsigned)(kk * 64)));
            }
        };
        static_assert(NI == 4 && MI == 2, "fragment schedule below");
        if (!VPD_ABL(p, 2)) {
#pragma unroll
            for (int i = 0; i < NI + MI; ++i) ldone(0, 0, i);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int cur = st & 1, nxt = cur ^ 1;
                const bool more = st + 1 < 18;
#define C64_MFMA(a, b) acc[a][b] = VPD_MFMA16(af[cur][a], bfm[cur][b], acc[a][b])
                // source order pinned by scheduling barriers: the next step's reads leave in pairs behind the first three MFMAs
                C64_MFMA(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { ldone(st + 1, nxt, 0); ldone(st + 1, nxt, 1); }
                __builtin_amdgcn_sched_barrier(0);
                C64_MFMA(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { ldone(st + 1, nxt, 2); ldone(st + 1, nxt, 3); }
                __builtin_amdgcn_sched_barrier(0);
                C64_MFMA(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { ldone(st + 1, nxt, 4); ldone(st + 1, nxt, 5); }
                __builtin_amdgcn_sched_barrier(0);
                C64_MFMA(1, 1); C64_MFMA(2, 0); C64_MFMA(2, 1); C64_MFMA(3, 0); C64_MFMA(3, 1);
                __builtin_amdgcn_sched_barrier(0);
#undef C64_MFMA
            }
        }
        if (!VPD_ABL(p, 8)) {
            if constexpr (EPM == 2 || EPM == 7) conv_epilogue_acc_pre<BM, BN, WM, WN, EPM>(p, acc, t, 0, st1, st2, geo, bst, accf);
            else if constexpr (EPM == 6) conv_epilogue_pre<BM, BN, WM, WN, EPM>(p, acc, t, 0, st1, st2, geo, bst);
            else conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, t, 0, st1, st2, geo);
        }
        __builtin_amdgcn_s_barrier();                             // B_{i+1}
    }
    if (p.stats) conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, blockIdx.x, 0, red);
}

// ---------------------------------------------------------------------------
// Two MFMA wave groups per block for the 64 -> 64 channel convolutions of the INFERENCE forward (eval epilogue).
// conv3x3_c64_persistent_kernel keeps ONE MFMA wave per SIMD: its 8 MFMAs per half K-step (128 cycles) sit behind ~150
// cycles of exposed ds_read latency, and the eval epilogue (scale / shift, residual loads, ReLU, padded stores) runs with
// the matrix cores idle.  Here a tile is 256 pixels: MFMA waves 0..3 take its first 128 pixels, waves 4..7 the second 128 --
// two MFMA waves per SIMD, one computing while the other waits for LDS or stores -- on ONE shared halo (344 pixels instead of
// 2 x 204).  LDS: 9 weight taps (72 KiB) + two halo buffers of 43 KiB = 158 KiB.  12 waves.
// Train-mode launches stay on the single-group kernel: there this kernel made its own class 6 % faster and every OTHER class
// of the step 3-5 % slower (the chip held a lower clock: profiles/r02_negative_results.txt).
// ---------------------------------------------------------------------------
#define C64X2_HPIX 344                           // halo pixels staged per tile (multiple of 8: one LDS-DMA instruction each)
template <int EPM>
__global__ __launch_bounds__(768) void conv3x3_c64x2_persistent_kernel(const ConvParams p, const HaloGeom g, int ntiles, int per) {
    constexpr int BN = 64;
    constexpr int MI = 2, NI = 4;                                 // per wave: 32 pixels x 64 channels
    constexpr int HBUF = C64X2_HPIX * 64;
    constexpr int NINSTR = C64X2_HPIX / 8;                        // 43 LDS-DMA instructions per halo
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);                 // [9][64*64] resident weights
    bf16_t* sH = sW + 9 * BN * 64;                                // [2][HBUF]

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.Ws, H = p.Hs, Wp = W + 2;
    // (tile walk as in conv3x3_c64_persistent_kernel: per > 0 = `per` consecutive tiles per block)
    const int G = per > 0 ? 1 : gridDim.x;
    const int tbeg = per > 0 ? blockIdx.x * per : blockIdx.x;
    const int tend = per > 0 ? (tbeg + per < ntiles ? tbeg + per : ntiles) : ntiles;

    if (wave >= 8) {
        const int lw = wave - 8;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        const float rWp = 1.0f / (float)Wp;
        auto issue_halo = [&](int mtile, int buf) __attribute__((always_inline)) {
            const int gr0 = mtile * g.TR;
            int prow0;
            if (g.multi) prow0 = (gr0 / H) * (H + 2);
            else { const int b = gr0 / H; prow0 = b * (H + 2) + (gr0 - b * H); }
            const int gp0 = prow0 * Wp;
            for (int k = lw; k < NINSTR; k += 4) {
                const int hp = k * 8 + lrow;
                const int hr = vpd_fdiv(hp, rWp);                // column-keyed swizzle, as conv3x3_c64_persistent_kernel
                const int key = (hp - hr * Wp) & 7;
                int gp = gp0 + hp;
                gp = gp < g.total_pix ? gp : g.total_pix - 1;
                const bf16_t* src = p.x + (size_t)gp * 64 + ((piece ^ key) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sH + buf * HBUF + k * 8 * 64), 16, 0, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int wsl = p.taps.w0 + (t / 3) * p.taps.wrs + (t % 3) * p.taps.wcs;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = (lw + 4 * i) * 8 + lrow;
                const bf16_t* src = p.w + ((size_t)wsl * 64 + n) * 64 + ((piece ^ (n & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + t * BN * 64 + (lw + 4 * i) * 8 * 64), 16, 0, 0);
            }
        }
        if (tbeg < tend) issue_halo(tbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // B_0
        int i = 0;
        for (int t = tbeg; t < tend; t += G, ++i) {
            if (t + G < tend) issue_halo(t + G, (i + 1) & 1);   // buffer last read in tile i-1, finished before B_i
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // B_{i+1}
        }
        if (EPM == 1) {
            __builtin_amdgcn_s_barrier();                         // halo buffers are free for the statistics scratch
            __builtin_amdgcn_s_barrier();                         // matches the barrier inside conv_stats_flush
        }
        return;
    }

    const int grp = wave >> 2;                                    // pixel half of the tile
    const int wm = wave & 3;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    unsigned lo[3][MI];                                           // as in conv3x3_c64_persistent_kernel
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = grp * 128 + wm * 32 + b * 16 + fr;
        const int lr = m / W;
        const int xx = m - lr * W;
        const int hrow = g.multi ? (lr / H) * (H + 2) + (lr % H) : lr;
#pragma unroll
        for (int ic = 0; ic < 3; ++ic)
            lo[ic][b] = (unsigned)((hrow * Wp + xx) * 128) + ((((unsigned)(xx + p.taps.dx0 + ic * p.taps.dxs) & 7u) ^ (unsigned)fq) << 4);
    }
    const unsigned wl0 = (unsigned)(fr * 128 + ((fq ^ (fr & 7)) << 4));
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    typedef const bf16x8 __attribute__((address_space(3)))* frag_t;
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    __builtin_amdgcn_s_barrier();                                 // B_0
    int i = 0;
    for (int t = tbeg; t < tend; t += G, ++i) {
        const bf16_t* cH = sH + (i & 1) * HBUF;
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned hbb = lds0 + (unsigned)(9 * BN * 64 * 2) + (unsigned)(i & 1) * (HBUF * 2u);
        // eval with a residual: its fragments are requested before the tile's nine K-steps instead of inside the epilogue
        ResFrag<NI, MI> resf;
        const bool res_pre = EPM == 3 && p.res != nullptr;
        if (res_pre) conv_res_prefetch<128, BN, 4, 1>(p, 2 * t + grp, 0, geo, resf, grp * 4);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const unsigned S = hbb + (unsigned)(((p.taps.dy0 + (tap / 3) * p.taps.dys) * Wp + p.taps.dx0 + (tap % 3) * p.taps.dxs) * 128);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[NI], bfm[MI];
#pragma unroll
                for (int a = 0; a < NI; ++a)
                    af[a] = *(frag_t)(size_t)(lds0 + (unsigned)(tap * BN * 64 * 2 + a * 2048) + (wl0 ^ (unsigned)(kk * 64)));
#pragma unroll
                for (int b = 0; b < MI; ++b) bfm[b] = *(frag_t)(size_t)((lo[tap % 3][b] + S) ^ (unsigned)(kk * 64));
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b)
                        acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
            }
        }
        // this group's 128 pixels are output tile 2t + grp of the 128-pixel tiling
        if (res_pre) conv_epilogue_res_pre<128, BN, 4, 1, EPM>(p, acc, 2 * t + grp, 0, st1, st2, geo, resf, grp * 4);
        else conv_epilogue<128, BN, 4, 1, EPM>(p, acc, 2 * t + grp, 0, st1, st2, geo, grp * 4);
        __builtin_amdgcn_s_barrier();                             // B_{i+1}
    }
    if (EPM == 1) {
        __builtin_amdgcn_s_barrier();                             // (loaders too) nobody reads the halo buffers any more
        conv_stats_flush<256, BN, 8, 1>(p, st1, st2, blockIdx.x, 0, reinterpret_cast<unsigned char*>(sH));
    }
}

// 256-pixel tiles of whole image rows whose halo fits C64X2_HPIX pixels; eval epilogue only (see the kernel's comment)
static bool c64x2_geom(const ConvParams& p, HaloGeom* g) {
    static const int off = getenv("VPD_C64X2") ? !atoi(getenv("VPD_C64X2")) : 0;
    const int W = p.Ws, H = p.Hs;
    if (off || conv_ep_mode(p) != 3 || W <= 0 || 256 % W != 0) return false;      // inference only (in training it measured slower)
    const int TR = 256 / W;
    if (TR <= H) { if (H % TR != 0) return false; g->multi = 0; g->HR = TR + 2; }
    else { if (TR % H != 0) return false; g->multi = 1; g->HR = (TR / H) * (H + 2); }
    g->TR = TR;
    g->NHP = g->HR * (W + 2);
    g->total_pix = p.N * (H + 2) * (W + 2);
    g->rot = 0; g->rnch = 1.0f;
    return g->NHP <= C64X2_HPIX && p.M >= 256 * 256;      // at least one 256-pixel tile per CU (256 crops of 32 x 32: 1024 tiles)
}
static hipError_t launch_c64x2(const ConvParams& p, const HaloGeom& g, hipStream_t stream) {
    const int ntiles = (p.M + 255) / 256;
    const int ncu = vpd_cu_budget();
    int grid = ntiles < ncu ? ntiles : ncu;
    static const int contig = getenv("VPD_C64_CONTIG") ? atoi(getenv("VPD_C64_CONTIG")) : 1;
    const int per = contig ? (ntiles + grid - 1) / grid : 0;
    if (per > 0) grid = (ntiles + per - 1) / per;
    const size_t lds = ((size_t)9 * 64 + 2 * C64X2_HPIX) * 64 * sizeof(bf16_t);
    ConvParams q = p;
    switch (conv_ep_mode(q)) {
        case 0: VPD_LAUNCH((conv3x3_c64x2_persistent_kernel<0>), dim3(grid), dim3(768), lds, stream, q, g, ntiles, per); break;
        case 1: VPD_LAUNCH((conv3x3_c64x2_persistent_kernel<1>), dim3(grid), dim3(768), lds, stream, q, g, ntiles, per); break;
        case 2: VPD_LAUNCH((conv3x3_c64x2_persistent_kernel<2>), dim3(grid), dim3(768), lds, stream, q, g, ntiles, per); break;
        default: VPD_LAUNCH((conv3x3_c64x2_persistent_kernel<3>), dim3(grid), dim3(768), lds, stream, q, g, ntiles, per); break;
    }
    return hipGetLastError();
}

template <int HROWS>
static hipError_t launch_c64(const ConvParams& p, const HaloGeom& g, hipStream_t stream) {
    const int ntiles = (p.M + 127) / 128;
    const int ncu = vpd_cu_budget();
    int grid = ntiles < ncu ? ntiles : ncu;
    // consecutive tiles per block (VPD_C64_CONTIG=0: strided); the grid shrinks to the blocks that get tiles
    static const int contig = getenv("VPD_C64_CONTIG") ? atoi(getenv("VPD_C64_CONTIG")) : 1;
    const int per = contig ? (ntiles + grid - 1) / grid : 0;
    if (per > 0) grid = (ntiles + per - 1) / per;
    const size_t lds = ((size_t)9 * 64 + 2 * HROWS) * 64 * sizeof(bf16_t) + 2048;
    ConvParams q = p;
    // (the accumulate modes fetch the old values of y by dense pixel index ahead of the MFMA loop: conv_acc_prefetch)
    if (q.accumulate && (q.ypad != 0 || q.osub != 1 || q.yC != q.Co)) return hipErrorInvalidValue;
    switch (conv_ep_mode(q)) {
        case 0: VPD_LAUNCH((conv3x3_c64_persistent_kernel<HROWS, 0>), dim3(grid), dim3(512), lds, stream, q, g, ntiles, per); break;
        case 1: VPD_LAUNCH((conv3x3_c64_persistent_kernel<HROWS, 1>), dim3(grid), dim3(512), lds, stream, q, g, ntiles, per); break;
        case 2: VPD_LAUNCH((conv3x3_c64_persistent_kernel<HROWS, 2>), dim3(grid), dim3(512), lds, stream, q, g, ntiles, per); break;
        case 6: VPD_LAUNCH((conv3x3_c64_persistent_kernel<HROWS, 6>), dim3(grid), dim3(512), lds, stream, q, g, ntiles, per); break;
        case 7: VPD_LAUNCH((conv3x3_c64_persistent_kernel<HROWS, 7>), dim3(grid), dim3(512), lds, stream, q, g, ntiles, per); break;
        default: VPD_LAUNCH((conv3x3_c64_persistent_kernel<HROWS, 3>), dim3(grid), dim3(512), lds, stream, q, g, ntiles, per); break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Persistent stem convolution (7x7 stride 2 on the 8-channel, border-3 NHWC input; 64 output channels).
// A tile is 128 output pixels of whole output rows; the 2*TR+5 raw input rows it needs are one contiguous
// range, staged ONCE per tile by LDS-DMA (plain copy: the 32-byte pixel pitch of consecutive output
// pixels is already bank-conflict free for ds_read_b128).  Each of the 7 kernel rows is one 64-deep K-step
// whose pixel fragments are read straight out of the raw rows (8 column taps x 8 channels contiguous),
// instead of the gather kernel's 7 x 128 B per output pixel (14x read amplification).  The 7 weight
// taps (56 KiB) stay resident; structure as conv3x3_c64_persistent_kernel.
// ---------------------------------------------------------------------------
template <int HROWS, int EPM>
__global__ __launch_bounds__(512) void conv_stem_persistent_kernel(const ConvParams p, int TR, int ntiles, long xelems, int lds_store) {
    constexpr int BM = 128, BN = 64, WM = 4, WN = 1;
    constexpr int WTM = BM / WM;
    constexpr int MI = WTM / 16, NI = 4;
    constexpr int HBUF = HROWS * 64;
    constexpr int HPASS = HROWS / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);                 // [7][64*64] resident weights
    bf16_t* sH = sW + 7 * BN * 64;                                // [2][HBUF] raw input rows
    unsigned char* red = reinterpret_cast<unsigned char*>(sH + 2 * HBUF);

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W0 = p.Ws, H0 = p.Hs, Wp = p.xWp, Hp = p.xHp;
    const int G = gridDim.x;
    // Tile walk.  Plain modes: tiles blockIdx.x, + G, ...  Pooled mode (EPM 9): a block takes WHOLE IMAGES (blockIdx.x, + G, ...)
    // top to bottom, because the pooled rows of a tile need the last conv row of the tile above it (kept in LDS).
    constexpr bool POOL = EPM == 9;
    const int TPI = H0 / TR;                                      // tiles per image
    const int cnt = POOL ? ((p.N - (int)blockIdx.x + G - 1) / G) * TPI : (ntiles - (int)blockIdx.x + G - 1) / G;
    auto tile_at = [&](int i) __attribute__((always_inline)) {
        if (!POOL) return (int)blockIdx.x + i * G;
        const int im = i / TPI;
        return ((int)blockIdx.x + im * G) * TPI + (i - im * TPI);
    };

    if (wave >= 4) {
        const int lw = wave - 4;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        auto issue_rows = [&](int mtile, int buf) __attribute__((always_inline)) {
            const int gr0 = mtile * TR;                           // global output row = b*H0 + y0
            const int b = gr0 / H0, y0 = gr0 - b * H0;
            const long e0 = ((long)(b * Hp + 2 * y0) * Wp) * 8;   // first element of input row 2*y0 (padded coords)
#pragma unroll
            for (int k = 0; k < HPASS; ++k) {
                const int row = (lw + 4 * k) * 8 + lrow;
                long e = e0 + (long)row * 64 + piece * 8;
                e = e < xelems - 8 ? e : xelems - 8;
                __builtin_amdgcn_global_load_lds((gptr_t)(p.x + e), (lptr_t)(sH + buf * HBUF + (lw + 4 * k) * 8 * 64), 16, 0, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int wsl = p.taps.w0 + t * p.taps.wrs;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = (lw + 4 * i) * 8 + lrow;
                const bf16_t* src = p.w + ((size_t)wsl * 64 + n) * 64 + ((piece ^ (n & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + t * BN * 64 + (lw + 4 * i) * 8 * 64), 16, 0, 0);
            }
        }
        issue_rows(tile_at(0), 0);
        if constexpr (POOL) {
            // Pooled eval mode: the LOADER waves also run the max-pool.  The MFMA waves park a tile's activations in LDS and go
            // on to the next tile's MFMA loop; behind the tile's barrier the 256 loader threads (idle but for five DMA
            // instructions per tile) reduce the parked tile to its 32 pooled pixels and store them.  Input rows are issued two
            // tiles ahead (tile i + 2 into the buffer tile i has just been read from), so there is still ONE barrier per tile.
            bf16_t* sT = reinterpret_cast<bf16_t*>(red) + 1024;   // [2][128 pixels][72]: parked activations (144-byte pitch)
            bf16_t* sC = sT + 2 * BM * 72;                        // [2][32][64]: last conv row of a tile, max over each pooled column's window
            const int ltid = tid - 256;
            const int q = ltid >> 3, c8 = (ltid & 7) << 3;        // pooled pixel of the tile (32 of them), channel slice
            const int Wo = W0 >> 1, Ho = H0 >> 1;
            const int j = q / Wo, xo = q - j * Wo;                // pooled row inside the tile, pooled column
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            // activations are >= +0: bf16 bit patterns order like unsigned integers (v_pk_max_u16, two channels per instruction)
            auto max4 = [](uint4 a, const uint4& b) __attribute__((always_inline)) {
                us2 r;
                r = __builtin_elementwise_max(__builtin_bit_cast(us2, a.x), __builtin_bit_cast(us2, b.x)); a.x = __builtin_bit_cast(unsigned, r);
                r = __builtin_elementwise_max(__builtin_bit_cast(us2, a.y), __builtin_bit_cast(us2, b.y)); a.y = __builtin_bit_cast(unsigned, r);
                r = __builtin_elementwise_max(__builtin_bit_cast(us2, a.z), __builtin_bit_cast(us2, b.z)); a.z = __builtin_bit_cast(unsigned, r);
                r = __builtin_elementwise_max(__builtin_bit_cast(us2, a.w), __builtin_bit_cast(us2, b.w)); a.w = __builtin_bit_cast(unsigned, r);
                return a;
            };
            if (cnt > 1) issue_rows(tile_at(1), 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // B_0
            for (int i = 0; i < cnt; ++i) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // rows of tile i + 1 have landed; the carried row is written
                __builtin_amdgcn_s_barrier();                     // B_{i+1}: tile i is parked, its input buffer is free
                if (i + 2 < cnt) issue_rows(tile_at(i + 2), i & 1);
                const int t = tile_at(i);
                const int img = t / TPI, ti = t - img * TPI;      // image, tile inside the image
                const bf16_t* cur = sT + (i & 1) * (BM * 72);
                uint4 best = uint4{0u, 0u, 0u, 0u};
                uint4 last = best;                                // window maximum over the tile's last conv row (threads of the last pooled row)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int row = 2 * j - 1 + r;                // conv row inside the tile; -1 = last row of the tile above
                    if (row < 0) {
                        if (ti != 0) best = max4(best, *reinterpret_cast<const uint4*>(sC + ((i & 1) ^ 1) * (32 * 64) + xo * 64 + c8));
                        continue;
                    }
                    uint4 m3 = uint4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int cx = 0; cx < 3; ++cx) {
                        const int x = 2 * xo - 1 + cx;
                        if (x < 0) continue;                      // (x <= W0 - 1 always: W0 is even)
                        m3 = max4(m3, *reinterpret_cast<const uint4*>(cur + (row * W0 + x) * 72 + c8));
                    }
                    best = max4(best, m3);
                    if (r == 2) last = m3;
                }
                if (2 * j + 1 == TR - 1) *reinterpret_cast<uint4*>(sC + (i & 1) * (32 * 64) + xo * 64 + c8) = last;
                const int oy = ti * (TR >> 1) + j;
                *reinterpret_cast<uint4*>(p.pool_y + ((size_t)(img * (Ho + 2) + oy + 1) * (Wo + 2) + xo + 1) * 64 + c8) = best;
            }
            return;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // B_0
        for (int i = 0; i < cnt; ++i) {
            if (i + 1 < cnt) issue_rows(tile_at(i + 1), (i + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // B_{i+1}
        }
        if (EPM == 1) __builtin_amdgcn_s_barrier();               // matches the barrier inside conv_stats_flush
        return;
    }

    const int wm = wave;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    int pbase[MI];                                                // element offset of (input row 2*lr, pixel 2*xx) in the tile
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = wm * WTM + b * 16 + fr;
        const int lr = m / W0;
        const int xx = m - lr * W0;
        pbase[b] = (2 * lr * Wp + 2 * xx) * 8;
    }
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    // pooled mode: this lane's scale / shift of the folded BatchNorm (channels a*16 + 4*fq .. + 3)
    float4 esc[NI], esh[NI];
    if (POOL) {
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            esc[a] = *reinterpret_cast<const float4*>(p.ep_scale + a * 16 + 4 * fq);
            esh[a] = *reinterpret_cast<const float4*>(p.ep_shift + a * 16 + 4 * fq);
        }
    }
    __builtin_amdgcn_s_barrier();                                 // B_0
    for (int i = 0; i < cnt; ++i) {
        const int t = tile_at(i);
        const bf16_t* cH = sH + (i & 1) * HBUF;
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // as in conv3x3_c64_persistent_kernel: one MFMA wave per SIMD, no barrier inside a tile -- two fragment sets, the next
        // half-step's six reads pinned behind the first three MFMAs of the current one
        static_assert(NI == 4 && MI == 2, "fragment schedule below");
        bf16x8 af[2][NI], bfm[2][MI];
        // fragment i of half-step st = 2 * (kernel row r) + kk: 0 -> af[0], 1 -> bfm[0], 2 -> bfm[1], 3.. -> af[1..]
        auto ldone = [&](int st, int buf, int i) __attribute__((always_inline)) {
            const int r = st >> 1, kk = st & 1;
            const int chunk = kk * 4 + fq;                        // column tap t = chunk (8 channels each)
            if (i == 1 || i == 2) {
                bfm[buf][i - 1] = *reinterpret_cast<const bf16x8*>(cH + pbase[i - 1] + (r * Wp + chunk) * 8);
            } else {
                const int a = i == 0 ? 0 : i - 2;
                const int rr = a * 16 + fr;
                af[buf][a] = *reinterpret_cast<const bf16x8*>(sW + r * BN * 64 + rr * 64 + ((chunk ^ (rr & 7)) << 3));
            }
        };
#pragma unroll
        for (int i2 = 0; i2 < NI + MI; ++i2) ldone(0, 0, i2);
#pragma unroll
        for (int st = 0; st < 14; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            const bool more = st + 1 < 14;
#define STEM_MFMA(a, b) acc[a][b] = VPD_MFMA16(af[cur][a], bfm[cur][b], acc[a][b])
            STEM_MFMA(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) { ldone(st + 1, nxt, 0); ldone(st + 1, nxt, 1); }
            __builtin_amdgcn_sched_barrier(0);
            STEM_MFMA(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) { ldone(st + 1, nxt, 2); ldone(st + 1, nxt, 3); }
            __builtin_amdgcn_sched_barrier(0);
            STEM_MFMA(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) { ldone(st + 1, nxt, 4); ldone(st + 1, nxt, 5); }
            __builtin_amdgcn_sched_barrier(0);
            STEM_MFMA(1, 1); STEM_MFMA(2, 0); STEM_MFMA(2, 1); STEM_MFMA(3, 0); STEM_MFMA(3, 1);
            __builtin_amdgcn_sched_barrier(0);
#undef STEM_MFMA
        }
        if constexpr (POOL) {
            // Eval with the max-pool behind the conv.  The tile's activations a = bf16(relu(bf16(acc) * scale + shift)) -- the
            // rounding points of the two-launch path (conv stores bf16 z, stem_pool_pair_kernel rounds the activation before
            // the maximum), so the results are bit-identical -- are parked pixel-major in one of two LDS buffers ([128 pixels]
            // [72]: 144-byte pitch, 16-byte aligned for the pooling threads' b128 reads); the loader waves pool them (above)
            // while this wave is already in the next tile's MFMA loop.  z is never written: 523 MB of stores and 523 MB of
            // loads per 1,000 crops go away.
            bf16_t* cur = reinterpret_cast<bf16_t*>(red) + 1024 + (i & 1) * (BM * 72);
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int ml = wm * WTM + b * 16 + fr;
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    uint2 zv;
                    zv.x = pack2bf(acc[a][b][0], acc[a][b][1]);
                    zv.y = pack2bf(acc[a][b][2], acc[a][b][3]);
                    float v0 = bf2f((unsigned short)(zv.x & 0xffff)) * esc[a].x + esh[a].x;
                    float v1 = bf2f((unsigned short)(zv.x >> 16)) * esc[a].y + esh[a].y;
                    float v2 = bf2f((unsigned short)(zv.y & 0xffff)) * esc[a].z + esh[a].z;
                    float v3 = bf2f((unsigned short)(zv.y >> 16)) * esc[a].w + esh[a].w;
                    v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
                    uint2 ov;
                    ov.x = pack2bf(v0, v1);
                    ov.y = pack2bf(v2, v3);
                    *reinterpret_cast<uint2*>(cur + ml * 72 + a * 16 + 4 * fq) = ov;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (!lds_store) {
            conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, t, 0, st1, st2, geo);
        } else {
            // Coalesced store path: the accumulator layout gives a lane 4 channels of 16 different pixels (32-byte pieces of
            // 16 rows per store instruction).  The wave parks its 32 pixels x 64 channels in LDS (136-byte pixel pitch) and
            // reads them back pixel-major: 16 bytes per lane, 8 consecutive dense NHWC pixels = 1 KiB contiguous per
            // instruction.  Only the wave's own pixels are involved: no block barrier.
            bf16_t* sT = reinterpret_cast<bf16_t*>(red) + 1024 + wm * 32 * 68;
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int ml = b * 16 + fr;
                const bool valid = t * BM + wm * WTM + ml < geo.M;
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    uint2 ov;
                    ov.x = pack2bf(acc[a][b][0], acc[a][b][1]);
                    ov.y = pack2bf(acc[a][b][2], acc[a][b][3]);
                    *reinterpret_cast<uint2*>(sT + ml * 68 + a * 16 + 4 * fq) = ov;
                    if (EPM == 1 && valid) {
                        const float q0 = bf2f((unsigned short)(ov.x & 0xffff)), q1 = bf2f((unsigned short)(ov.x >> 16));
                        const float q2 = bf2f((unsigned short)(ov.y & 0xffff)), q3 = bf2f((unsigned short)(ov.y >> 16));
                        st1[a][0] += q0; st2[a][0] += q0 * q0;
                        st1[a][1] += q1; st2[a][1] += q1 * q1;
                        st1[a][2] += q2; st2[a][2] += q2 * q2;
                        st1[a][3] += q3; st2[a][3] += q3 * q3;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int pl = lane >> 3, pc = (lane & 7) << 3;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int ml = it * 8 + pl;
                const int m = t * BM + wm * WTM + ml;
                const uint2 lo = *reinterpret_cast<const uint2*>(sT + ml * 68 + pc);
                const uint2 hi = *reinterpret_cast<const uint2*>(sT + ml * 68 + pc + 4);
                if (m < geo.M) *reinterpret_cast<uint4*>(p.y + (size_t)m * 64 + pc) = uint4{lo.x, lo.y, hi.x, hi.y};
            }
        }
        __builtin_amdgcn_s_barrier();                             // B_{i+1}
    }
    if (EPM == 1) conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, blockIdx.x, 0, red);
}

// stem shape test + launch; returns false when the generic gather kernel has to take it
static bool stem_eligible(const ConvParams& p, int* TR) {
    if (!(p.xC == 8 && p.Kc == 64 && p.Co == 64 && p.istr == 2 && p.osub == 1 && p.taps.nr == 7 && p.taps.nc == 1 &&
          p.taps.dy0 == 0 && p.taps.dys == 1 && p.taps.dx0 == 0 && p.ypad == 0 && !p.accumulate && (!p.ep_scale || p.pool_y)))
        return false;
    if (p.Ws <= 0 || 128 % p.Ws != 0) return false;
    *TR = 128 / p.Ws;
    if (*TR > p.Hs || p.Hs % *TR != 0) return false;
    // pooled eval epilogue: whole pooled rows per tile (even TR, even W), folded BatchNorm + ReLU, no statistics
    if (p.pool_y && ((*TR & 1) || (p.Ws & 1) || !p.ep_scale || !p.ep_shift || !p.ep_relu || p.stats || p.res)) return false;
    const int rows = ((2 * *TR + 5) * p.xWp + 7) / 8;            // 128-byte LDS rows of the raw input range
    return rows <= 160 && p.M % 128 == 0;
}

static hipError_t launch_stem(const ConvParams& p, int TR, hipStream_t stream) {
    const int ntiles = p.M / 128;
    const int ncu = vpd_cu_budget();
    const int grid = p.pool_y ? (p.N < ncu ? p.N : ncu) : (ntiles < ncu ? ntiles : ncu);      // pooled: whole images per block
    // statistics scratch (2 KiB) + the coalesced-store staging of the four MFMA waves (4 x 32 pixels x 136 B; pooled mode:
    // two parked tiles of 128 pixels x 144 B + two carried rows of 32 x 128 B)
    const size_t lds = ((size_t)7 * 64 + 2 * 160) * 64 * sizeof(bf16_t) + 2048 +
                       (p.pool_y ? (size_t)(2 * 128 * 72 + 2 * 32 * 64) : (size_t)4 * 32 * 68) * sizeof(bf16_t);
    const long xelems = (long)p.N * p.xHp * p.xWp * 8 + 64;      // the plan allocates 256 elements of slack behind xin
    ConvParams q = p;
    // dense 64-channel output (the stem's only use): stores through LDS, 1 KiB contiguous per instruction (VPD_STEM_LDS_STORE=0: direct)
    static const int lds_store_on = getenv("VPD_STEM_LDS_STORE") ? atoi(getenv("VPD_STEM_LDS_STORE")) : 1;
    const int lds_store = lds_store_on && p.ypad == 0 && p.yC == 64 && p.osub == 1 && p.yWp == p.Ws && p.yHp == p.Hs;
    if (p.pool_y) VPD_LAUNCH((conv_stem_persistent_kernel<160, 9>), dim3(grid), dim3(512), lds, stream, q, TR, ntiles, xelems, 0);
    else if (p.stats) VPD_LAUNCH((conv_stem_persistent_kernel<160, 1>), dim3(grid), dim3(512), lds, stream, q, TR, ntiles, xelems, lds_store);
    else VPD_LAUNCH((conv_stem_persistent_kernel<160, 0>), dim3(grid), dim3(512), lds, stream, q, TR, ntiles, xelems, lds_store);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// conv3x3_pws_kernel (conv_pws.h): persistent blocks, one barrier per K-step.  VPD_PWS=0 restores conv3x3_ws_kernel.
// ---------------------------------------------------------------------------
// ring depths per tile class (NS = A + 2: two readable steps + A weight bundles in flight); -D overrides for same-box A/B builds
// (ring depths PWS_NS_C*: conv_pws.h)
int pws_cu_count() {
    // VPD_PWS_BLOCKS: pretend the device has this many CUs (tests: many tiles per block on small problems)
    static const int forced = getenv("VPD_PWS_BLOCKS") && atoi(getenv("VPD_PWS_BLOCKS")) > 0 ? atoi(getenv("VPD_PWS_BLOCKS")) : 0;
    return forced ? forced : vpd_cu_budget();
}
static bool pws_enabled(const ConvParams& p) {
    static const int on = getenv("VPD_PWS") ? atoi(getenv("VPD_PWS")) : 1;
    const int mode = conv_ep_mode(p);
    // (global output rows below 2^21: the kernel's float-reciprocal divisions, vpd_fdiv)
    return on && mode != 4 && mode != 5 && (long)p.N * p.Hs < VPD_FDIV_MAX;
}
template <int BM, int BN, int HROWS, int NS, int NMW, bool PIPE>
static hipError_t launch_pws(const ConvParams& p, const HaloGeom& g, hipStream_t stream) {
    constexpr int WN = BN / 64, WM = NMW / WN;
    constexpr size_t lds = (size_t)2 * HROWS * 128 + (size_t)NS * BN * 128 + 1024 + (size_t)3 * WM * BN * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    PwsGrid sg;
    sg.MT = (p.M + BM - 1) / BM;
    sg.NT = p.Co / BN;
    int lanes = pws_cu_count() / sg.NT;
    if (lanes < 1) lanes = 1;
    if (lanes > sg.MT) lanes = sg.MT;
    if (lanes >= 8) lanes &= ~7;
    sg.lanes = lanes;
    sg.xcd = lanes % 8 == 0;
    sg.rNT = 1.0f / (float)sg.NT; sg.rlanes = 1.0f / (float)lanes;
    const dim3 grid(lanes * sg.NT), block((NMW + 4) * 64);
    ConvParams q = p;
#ifdef PWS_STAMPS
    // diagnostic build: stamps of launch 30 of each kernel shape, as differences from the block's entry, median over blocks
    static unsigned long long* dstamps = nullptr;
    static int nlaunch = 0;
    if (!dstamps) (void)hipMalloc(&dstamps, 4096 * 16 * 8);
    (void)hipMemsetAsync(dstamps, 0, 4096 * 16 * 8, stream);
    q.err = reinterpret_cast<unsigned*>(dstamps);
    struct Dump { static void run(int nb, const char* tag) {
        static unsigned long long h[4096 * 16];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, dstamps, (size_t)nb * 16 * 8, hipMemcpyDeviceToHost);
        const char* names[16] = {"entry", "setup done", "first READY", "K loop done", "epilogue issued", "tiles done", "stats flushed",
                                 "stores drained", "L entry", "L halo issued", "L first landed", "L done", "L origin known", "(realtime)", "all epilogues", "(realtime)"};
        fprintf(stderr, "[pws stamps %s, %d blocks] cycles from the consumer's entry (median / max over blocks)\n", tag, nb);
        for (int k = 1; k < 15; ++k) {
            if (k == 13) continue;
            std::vector<long long> d;
            for (int b = 0; b < nb; ++b) if (h[b * 16 + k] && h[b * 16]) d.push_back((long long)(h[b * 16 + k] - h[b * 16]));
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            fprintf(stderr, "  %-16s %8lld %8lld\n", names[k], d[d.size() / 2], d.back());
        }
        {      // the clock the shader held between "halo issued" (slot 9 ~ slot 13's time) and "loader done" (slots 11 / 15)
            std::vector<double> ghz;
            for (int b = 0; b < nb; ++b)
                if (h[b * 16 + 15] > h[b * 16 + 13] && h[b * 16 + 11] > h[b * 16 + 9])
                    ghz.push_back((double)(h[b * 16 + 11] - h[b * 16 + 9]) / (double)(h[b * 16 + 15] - h[b * 16 + 13]) * 0.1);
            if (!ghz.empty()) { std::sort(ghz.begin(), ghz.end()); fprintf(stderr, "  s_memtime ticks per ns over the K loops (median over blocks): %.3f  (min %.3f max %.3f)\n", ghz[ghz.size() / 2], ghz.front(), ghz.back()); }
        }
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < nb; ++b) { if (h[b * 16] && h[b * 16] < lo) lo = h[b * 16]; if (h[b * 16 + 7] > hi) hi = h[b * 16 + 7]; }
        fprintf(stderr, "  first entry -> last drained: %llu cycles; entry spread: ", hi - lo);
        unsigned long long emax = 0; for (int b = 0; b < nb; ++b) if (h[b * 16] > emax) emax = h[b * 16];
        fprintf(stderr, "%llu cycles\n", emax - lo);
    } };
#endif
    bool geo_launched = false;
    {
        static const int geo_on = getenv("VPD_PWS_GEO") ? atoi(getenv("VPD_PWS_GEO")) : 1;
        geo_launched = geo_on && vpd_launch_pws_geo(BM, BN, HROWS, NS, NMW, q, g, sg, grid, block, lds, stream);
    }
    if (!geo_launched) switch (conv_ep_mode(q)) {
        case 0: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 0, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        case 1: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 1, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        case 2: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 2, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        case 3: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 3, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        case 6: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 6, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        case 7: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 7, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        case 8: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 8, NMW, PIPE>), grid, block, lds, stream, q, g, sg); break;
        default: return hipErrorInvalidValue;
    }
#ifdef PWS_STAMPS
    if (++nlaunch == 30) { char tag[64]; snprintf(tag, sizeof tag, "<%d,%d> NS %d mode %d", BM, BN, NS, conv_ep_mode(q)); Dump::run((int)grid.x, tag); }
#endif
    return hipGetLastError();
}

template <int BM, int BN, int HROWS, int HB, int WPS, int NS>
static hipError_t launch_ws_ns(const ConvParams& p, const HaloGeom& g, hipStream_t stream) {
    dim3 grid((p.M + BM - 1) / BM, p.Co / BN);
    const size_t lds = ((size_t)HB * HROWS + NS * BN) * 64 * sizeof(bf16_t);
    static_assert(((size_t)HB * HROWS + NS * BN) * 64 * sizeof(bf16_t) <= 160 * 1024, "LDS");
    ConvParams q = p;
    // layer2's 256 x 128 tile with EIGHT MFMA waves (64 x 64 wave tiles, two waves per SIMD) + the four loader waves
    // (the 256 x 64 tile with eight MFMA waves -- 32 x 64 wave tiles, six fragment reads per eight MFMAs -- is LDS-read-bound:
    //  class 525 -> 590 us)
    if constexpr (BM == 256 && BN == 128 && NS == 4) {
        static const int mw8 = getenv("VPD_WS_MW8") ? atoi(getenv("VPD_WS_MW8")) : 1;
        if (mw8) {
            switch (conv_ep_mode(q)) {
                case 0: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 0, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                case 1: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 1, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                case 2: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 2, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                case 3: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 3, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                case 6: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 6, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                case 7: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 7, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                case 8: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, 3, 8, NS, 8>), grid, dim3(768), lds, stream, q, g); return hipGetLastError();
                default: break;
            }
        }
    }
    switch (conv_ep_mode(q)) {
        case 0: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 0, NS>), grid, dim3(512), lds, stream, q, g); break;
        case 1: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 1, NS>), grid, dim3(512), lds, stream, q, g); break;
        case 2: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 2, NS>), grid, dim3(512), lds, stream, q, g); break;
        case 6: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 6, NS>), grid, dim3(512), lds, stream, q, g); break;
        case 7: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 7, NS>), grid, dim3(512), lds, stream, q, g); break;
        case 8: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 8, NS>), grid, dim3(512), lds, stream, q, g); break;
        default: VPD_LAUNCH((conv3x3_ws_kernel<BM, BN, HROWS, HB, WPS, 3, NS>), grid, dim3(512), lds, stream, q, g); break;
    }
    return hipGetLastError();
}
// NS: weight-ring depth (four stages: profiles/r02_ring_depth.txt)
template <int BM, int BN, int HROWS, int HB, int WPS, int NS>
static hipError_t launch_ws(const ConvParams& p, const HaloGeom& g, hipStream_t stream) {
    return launch_ws_ns<BM, BN, HROWS, HB, WPS, NS>(p, g, stream);
}

// ---------------------------------------------------------------------------
// 1x1 stride-1 convolution (the Bottleneck students' conv1 / conv3, forward and data gradient) as a warp-specialised ring
// GEMM: out[M][Co] = X[M][Kc] W[Co][Kc]^T with X the interior pixels of a padded NHWC activation.
//   waves 4..7 (loaders): per 64-channel K-step the [BM pixels][64] slice of X (each lane's pixel rows resolved to padded
//     addresses ONCE, in the prologue) and the [BN][64] weight slice go by LDS-DMA into stage s % NS of an NS-deep ring, AHEAD =
//     NS - 1 steps in front of the MFMA waves; counted vmcnt waits, one workgroup barrier per K-step.
//   waves 0..3 (MFMA): 64-channel x (BM / WM)-pixel wave tiles out of LDS, the shared epilogue (conv_epilogue.h).
// The gather kernel (conv_igemm_kernel: four waves, register staging, two LDS buffers, two blocks per CU) runs these
// 8.6 GFLOP launches at 180-250 TFLOP/s; its loads are one K-step ahead at best.
// ---------------------------------------------------------------------------
template <int BM, int BN, int NS, int EPM>
__global__ __launch_bounds__(512) void conv1x1_ws_kernel(const ConvParams p0) {
    // second convolution of the launch (ConvParams::alt_*: a BasicBlock's 1x1 down-sampling branch beside its stride-2 3x3):
    // the blocks with blockIdx.y >= alt_y0 see it as THE convolution
    ConvParams p = p0;
    int by = blockIdx.y;
    if (p0.alt_w && by >= p0.alt_y0) {
        by -= p0.alt_y0;
        p.w = p0.alt_w; p.y = p0.alt_y; p.stats = p0.alt_stats; p.taps = p0.alt_taps;
        p.ep_scale = p0.alt_ep_scale; p.ep_shift = p0.alt_ep_shift; p.ep_relu = p0.alt_ep_relu; p.res = nullptr;
    }
    constexpr int WN = BN / 64;
    constexpr int WM = 4 / WN;
    constexpr int WTM = BM / WM;
    constexpr int MI = WTM / 16, NI = 4;
    constexpr int ASTAGE = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int A_PER = BM / 32, W_PER = BN / 32, PER_STEP = A_PER + W_PER;      // LDS-DMA instructions per loader wave and step
    constexpr int AHEAD = NS - 1;
    static_assert(AHEAD >= 1 && AHEAD <= 3 && 3 * PER_STEP < 64, "vmcnt immediates");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);               // [NS][STAGE]: X slice, then W slice

    // parity class of a merged stride-2 data gradient (blockIdx.z, heaviest class first as in conv_igemm_kernel)
    ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    TapSet taps = p.taps;
    const int cls = (int)gridDim.z - 1 - (int)blockIdx.z;
    switch (cls) {
        case 1: geo = p.cls[0].geo; taps = p.cls[0].taps; break;
        case 2: geo = p.cls[1].geo; taps = p.cls[1].taps; break;
        case 3: geo = p.cls[2].geo; taps = p.cls[2].taps; break;
        default: break;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = blockIdx.x;
    const int n0 = by * BN;
    const int m0 = mtile * BM;
    if (m0 >= geo.M) return;                                      // (a smaller class: whole block, before any barrier)
    const int kchunks = p.Kc >> 6;
    const int nsteps1 = taps.nr * taps.nc * kchunks;              // K-step s = (tap s / kchunks, 64-channel chunk s % kchunks)
    // class 0 only: extra K-steps from the second input tensor (same pixels, first tap's offset, its own weights)
    const int nsteps = nsteps1 + ((p.x2 && cls == 0) ? (p.Kc2 >> 6) : 0);

    if (wave >= 4) {
        const int lw = wave - 4;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        const int HW = geo.Hs * geo.Ws;
        int abase[A_PER], wbase[W_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int r = (lw + 4 * i) * 8 + lrow;
            int m = m0 + r;
            m = m < geo.M ? m : geo.M - 1;
            const int b = m / HW;
            const int rr = m - b * HW;
            const int yy = rr / geo.Ws;
            const int xx = rr - yy * geo.Ws;
            abase[i] = ((b * p.xHp + yy * p.istr) * p.xWp + xx * p.istr) * p.xC + ((piece ^ (r & 7)) << 3);
        }
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int n = (lw + 4 * i) * 8 + lrow;
            wbase[i] = (n0 + n) * p.Kc + ((piece ^ (n & 7)) << 3);      // (Kc2 == Kc: checked by the launcher)
        }
        auto issue = [&](int s) __attribute__((always_inline)) {
            bf16_t* st = ring + (s % NS) * STAGE;
            const bf16_t* xs = p.x;
            const bf16_t* wsrc = p.w;
            int aoff, woff;
            if (s < nsteps1) {
                const int tap = s / kchunks;
                const int cc = s - tap * kchunks;
                const int ir = tap / taps.nc;
                const int ic = tap - ir * taps.nc;
                aoff = ((taps.dy0 + ir * taps.dys) * p.xWp + (taps.dx0 + ic * taps.dxs)) * p.xC + cc * 64;
                woff = (taps.w0 + ir * taps.wrs + ic * taps.wcs) * p.Co * p.Kc + cc * 64;
            } else {
                const int cc = s - nsteps1;
                xs = p.x2; wsrc = p.w2;
                aoff = (taps.dy0 * p.xWp + taps.dx0) * p.xC + cc * 64;
                woff = cc * 64;
            }
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(xs + abase[i] + aoff), (lptr_t)(st + (lw + 4 * i) * 8 * 64), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < W_PER; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + wbase[i] + woff), (lptr_t)(st + ASTAGE + (lw + 4 * i) * 8 * 64), 16, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            if (s < nsteps) issue(s);
        for (int s = 0; s < nsteps; ++s) {
            // step s must have landed; the steps issued behind it (at most AHEAD - 1, fewer at the tail) may still be in flight
            int behind = nsteps - 1 - s;
            behind = behind < AHEAD - 1 ? behind : AHEAD - 1;
            if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_STEP) : "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // READY_s (stage (s - 1) % NS is free again)
            if (s + AHEAD < nsteps) issue(s + AHEAD);
        }
        __builtin_amdgcn_s_barrier();                             // END
        if (p.stats) __builtin_amdgcn_s_barrier();                // matches the barrier inside conv_stats_flush
        return;
    }

    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsteps; ++s) {
        __builtin_amdgcn_s_barrier();                             // READY_s
        const bf16_t* cA = ring + (s % NS) * STAGE;
        const bf16_t* cW = cA + ASTAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[NI], bfm[MI];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int r = wn * 64 + a * 16 + fr;
                af[a] = *reinterpret_cast<const bf16x8*>(cW + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int r = wm * WTM + b * 16 + fr;
                bfm[b] = *reinterpret_cast<const bf16x8*>(cA + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
        }
    }
    __builtin_amdgcn_s_barrier();                                 // END
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo);
    if (p.stats) conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, mtile, n0, smem);
}

// 1x1, stride 1, one tap, one class, no second convolution / input, plain epilogue modes, K deep enough for the ring to matter
static bool conv1x1_ws_eligible(const ConvParams& p) {
    static const int on = getenv("VPD_CONV1X1_WS") ? atoi(getenv("VPD_CONV1X1_WS")) : 1;
    // K >= 512 (8+ steps), or 4+ steps when the launch is at most a few rounds of blocks: layer1's 256 -> 64 convs (2,048
    // blocks of four steps) stream their 168 MB faster through the gather kernel's two small blocks per CU
    const int kmin = 256, mmax = 32768;
    const bool common = on && !p.bst_z2 && p.xC == p.Kc && p.Kc % 64 == 0 && p.Co % 64 == 0 &&
                        (long)p.N * p.xHp * p.xWp * p.xC < (1l << 31) && (long)9 * p.Co * p.Kc < (1l << 31);
    if (!common) return false;
    // the merged parity classes of a stride-2 3x3 data gradient (+ the 1x1 branch's as extra K-steps of class 0) at the layer3
    // and layer4 boundaries (layer2's has 2,048 blocks of 2-8 K-steps: gather kernel)
    static const int dgon = getenv("VPD_CONV_S2_DGRAD_WS") ? atoi(getenv("VPD_CONV_S2_DGRAD_WS")) : 1;
    const int dgk = 256;      // (layer2's boundary, K = 128, on the ring GEMM: measured slower, DESIGN_HISTORY.md round 4)
    if (p.ncls > 1 || p.x2 || p.osub != 1)
        return dgon && p.ncls == 4 && p.osub == 2 && p.istr == 1 && !p.alt_w && !p.accumulate && !p.ep_scale && p.Kc >= dgk &&
               p.Co % 128 == 0 && (!p.x2 || p.Kc2 == p.Kc) && (!p.bst_z || (p.yC == p.Co && p.ypad == 0));
    if (p.oph != 0 || p.opw != 0) return false;
    // BatchNorm sums in the epilogue (mode 6): a deep stride-1 1x1 data gradient writing a dense tensor (Bottleneck students, layer3 / 4)
    if (p.bst_z && !(p.istr == 1 && p.taps.nr == 1 && p.taps.nc == 1 && !p.alt_w && !p.bst_z2 && !p.accumulate && p.yC == p.Co && p.ypad == 0))
        return false;
    // the stride-2 convs at the ResNet stage boundaries (3x3 forward, with the BasicBlock's 1x1 branch as second convolution):
    // 9-36 K-steps, one round of blocks with the tile choice below
    static const int s2on = getenv("VPD_CONV_S2_WS") ? atoi(getenv("VPD_CONV_S2_WS")) : 1;
    if (p.istr == 2) return s2on && p.taps.nr == 3 && p.taps.nc == 3 && (!p.alt_w || (p.alt_taps.nr == 1 && p.alt_taps.nc == 1));
    if (p.Kc < 512 && !(p.Kc >= kmin && p.M <= mmax)) return false;
    return p.taps.nr == 1 && p.taps.nc == 1 && p.istr == 1 && !p.alt_w;
}
template <int BM, int BN, int NS>
static hipError_t launch_1x1_ws(const ConvParams& p, hipStream_t stream) {
    int maxM = p.M;
    for (int k = 1; k < p.ncls; ++k) maxM = p.cls[k - 1].geo.M > maxM ? p.cls[k - 1].geo.M : maxM;
    dim3 grid((maxM + BM - 1) / BM, (p.Co / BN) * (p.alt_w ? 2 : 1), p.ncls > 1 ? p.ncls : 1);
    const size_t lds = (size_t)NS * (BM + BN) * 64 * sizeof(bf16_t);
    static_assert((size_t)NS * (BM + BN) * 64 * sizeof(bf16_t) <= 160 * 1024, "LDS");
    ConvParams q = p;
    if (q.alt_w) q.alt_y0 = p.Co / BN;      // the second convolution's blocks follow the first's
    switch (conv_ep_mode(q)) {
        case 0: VPD_LAUNCH((conv1x1_ws_kernel<BM, BN, NS, 0>), grid, dim3(512), lds, stream, q); break;
        case 1: VPD_LAUNCH((conv1x1_ws_kernel<BM, BN, NS, 1>), grid, dim3(512), lds, stream, q); break;
        case 2: VPD_LAUNCH((conv1x1_ws_kernel<BM, BN, NS, 2>), grid, dim3(512), lds, stream, q); break;
        case 3: VPD_LAUNCH((conv1x1_ws_kernel<BM, BN, NS, 3>), grid, dim3(512), lds, stream, q); break;
        case 6: VPD_LAUNCH((conv1x1_ws_kernel<BM, BN, NS, 6>), grid, dim3(512), lds, stream, q); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// tile choice: the largest tile that still gives every CU a block
static hipError_t launch_1x1(const ConvParams& p, hipStream_t stream) {
    long t256 = (long)((p.M + 255) / 256), t128 = (long)((p.M + 127) / 128);
    for (int k = 1; k < p.ncls; ++k) { t256 += (p.cls[k - 1].geo.M + 255) / 256; t128 += (p.cls[k - 1].geo.M + 127) / 128; }
    const int nconv = p.alt_w ? 2 : 1;
    if (p.Co % 128 == 0 && t256 * (p.Co / 128) * nconv >= 200) return launch_1x1_ws<256, 128, 3>(p, stream);
    if (p.Co % 128 == 0 && t128 * (p.Co / 128) * nconv >= 200) return launch_1x1_ws<128, 128, 4>(p, stream);
    return launch_1x1_ws<128, 64, 4>(p, stream);
}

template <int BM, int BN, int HROWS, bool HALO2>
static hipError_t launch_halo(const ConvParams& p, const HaloGeom& g, hipStream_t stream) {
    dim3 grid((p.M + BM - 1) / BM, p.Co / BN);
    size_t lds = ((size_t)(HALO2 ? 2 : 1) * HROWS + 2 * BN) * 64 * sizeof(bf16_t);
    const size_t red = (size_t)2 * 2 * BN * sizeof(float);
    if (lds < red) lds = red;
    ConvParams q = p;
    VPD_LAUNCH((conv3x3_halo_kernel<BM, BN, HROWS, HALO2>), grid, dim3(256), lds, stream, q, g);
    return hipGetLastError();
}

// Geometry of the halo tiling for a BM-pixel tile, or false when the shape does not fit it.
static bool halo_geom(const ConvParams& p, int BM, int hrows_max, HaloGeom* g) {
    const int W = p.Ws, H = p.Hs;
    if (BM % W != 0) return false;
    const int TR = BM / W;
    if (TR <= H) { if (H % TR != 0) return false; g->multi = 0; g->HR = TR + 2; }
    else { if (TR % H != 0) return false; g->multi = 1; g->HR = (TR / H) * (H + 2); }
    g->TR = TR;
    g->NHP = g->HR * (W + 2);
    g->total_pix = p.N * (H + 2) * (W + 2);
    // a 16-pixel MFMA fragment spans 16 / W image rows.  W >= 8: the column index alone separates its pixels (two rows of 8 use
    // opposite halves of the eight 16-byte pieces of a bank row); W = 4: two column bits + the row parity
    if (W >= 8) { g->kmask = 7; g->kshift = 0; g->rowmask = 0; }
    else { g->kmask = 3; g->kshift = 2; g->rowmask = 1; }
    g->rH = 1.0f / (float)H; g->rW = 1.0f / (float)W; g->rWp = 1.0f / (float)(W + 2);
    static const int rot = getenv("VPD_PWS_ROT") ? atoi(getenv("VPD_PWS_ROT")) : 1;
    // (not in the inference epilogue: there a crop's embedding must not depend on where in the batch it sits -- a batch, its split, its
    //  ragged tail and its hipGraph launch agree bit for bit, tests/test_fullsize_gpu.py -- and the rotation is a function of the tile index)
    g->rot = rot && p.Kc > 64 && conv_ep_mode(p) != 3;
    g->rnch = 1.0f / (float)(p.Kc / 64);
    return g->NHP <= hrows_max;
}

// 3x3, stride 1, border-1 input, dense sub-grid: eligible for the halo kernel
static bool halo_eligible(const ConvParams& p) {
    return p.taps.nr == 3 && p.taps.nc == 3 && p.istr == 1 && p.osub == 1 && p.oph == 0 && p.opw == 0 &&
           p.xC == p.Kc && p.xHp == p.Hs + 2 && p.xWp == p.Ws + 2 && p.taps.dy0 + 2 * p.taps.dys >= 0 &&
           p.taps.dy0 >= 0 && p.taps.dy0 <= 2 && p.taps.dx0 >= 0 && p.taps.dx0 <= 2 &&
           p.taps.dy0 + 2 * p.taps.dys <= 2 && p.taps.dx0 + 2 * p.taps.dxs >= 0 && p.taps.dx0 + 2 * p.taps.dxs <= 2;
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    int maxM = p.M;
    for (int k = 1; k < p.ncls; ++k) maxM = p.cls[k - 1].geo.M > maxM ? p.cls[k - 1].geo.M : maxM;
    ConvParams q = p;
    if (q.alt_w) q.alt_y0 = p.Co / BN;      // the second convolution's blocks follow the first's
    dim3 grid((maxM + BM - 1) / BM, (p.Co / BN) * (q.alt_w ? 2 : 1), p.ncls > 1 ? p.ncls : 1);
    const size_t lds = (size_t)2 * (BM + BN) * 64 * sizeof(bf16_t);
    switch (conv_ep_mode(q)) {
        case 0: VPD_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 0>), grid, dim3(256), lds, stream, q); break;
        case 1: VPD_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), lds, stream, q); break;
        case 2: VPD_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), lds, stream, q); break;
        case 6: VPD_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 6>), grid, dim3(256), lds, stream, q); break;
        case 7: VPD_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 7>), grid, dim3(256), lds, stream, q); break;
        default: VPD_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), lds, stream, q); break;
    }
    return hipGetLastError();
}

// BM used for a problem: the stats partial buffer has ceil(M/BM) rows.
extern "C" int vpd_conv_bm(int M, int Co) {
    const int bn = (Co % 128 == 0) ? 128 : 64;
    if (bn == 128) {
        if ((long)((M + 127) / 128) * (Co / 128) >= 384) return 128;
        return 64;
    }
    return 128;
}

// Kernel selection = timing class of vpd_plan_read_timing (one class per kernel function):
//   0 conv3x3_c64_persistent_kernel<224>   1 conv3x3_ws_kernel<256,128,352>   2 conv3x3_ws_kernel<128,128,288>
//   3 conv3x3_ws_kernel<128,64,288>        4 conv_igemm_kernel (gather; also the legacy conv3x3_halo fallback)
//   5 conv_stem_persistent_kernel<160>
int vpd_conv_kernel_class(const ConvParams& p, HaloGeom* g) {
    static const int no_ws = getenv("VPD_NO_WS") ? atoi(getenv("VPD_NO_WS")) : 0;
    int tr_stem;
    if (p.alt_w || p.x2) return 4;                             // two convolutions / two inputs in one launch: gather kernel only
    if (!no_ws && stem_eligible(p, &tr_stem)) return 5;        // conv_stem_persistent_kernel
    if (halo_eligible(p) && !no_ws) {
        if (p.Co % 128 == 0) {
            const long t256 = (long)((p.M + 255) / 256) * (p.Co / 128);
            const long t128 = (long)((p.M + 127) / 128) * (p.Co / 128);
            // 256 x 64 tiles on the pipelined persistent kernel where 256 x 128 tiles would give every block exactly one tile: twice
            // the tiles, so a block overlaps one tile's epilogue with the next one's loads (layer2 at 256 crops: 23.2 vs 25.2 us;
            // with several 256 x 128 tiles per block the eight-wave kernel below is faster: 83.7 vs 90.6 us at 1000 crops)
            if ((p.M + 255) / 256 <= pws_cu_count() / (p.Co / 128) && pws_enabled(p) && t256 >= 200 &&
                halo_geom(p, 256, 416, g))
                return 6;
            if (t256 >= 200 && halo_geom(p, 256, 352, g)) return 1;
            // 256 pixels x 64 channels: the FLOPs of a 128 x 128 tile for 30 % fewer staged bytes per K-step (8 KB of weights +
            // 1/9 of a 52 KB halo instead of 16 KB + 1/9 of 36 KB) where 256-pixel tiles alone would leave half the chip idle
            // (layer3 at 256 crops: 64 pixel tiles x 4 channel tiles)
            static const int w64 = getenv("VPD_WS_256x64") ? atoi(getenv("VPD_WS_256x64")) : 1;
            if (w64 && t128 >= 200 && (long)((p.M + 255) / 256) * (p.Co / 64) >= 200 && halo_geom(p, 256, 416, g)) return 6;
            if (halo_geom(p, 128, 288, g)) return t128 >= 200 ? 2 : 3;    // few pixel tiles: 64-channel tiles fill the chip
        } else if (p.Kc == 64 && p.Co == 64) {
            // 64 -> 64 channels (layer1): persistent blocks with resident weights
            if (halo_geom(p, 128, 224, g)) return 0;
        }
    }
    return 4;
}
int vpd_conv_kernel_class(const ConvParams& p) { HaloGeom g; return vpd_conv_kernel_class(p, &g); }

// Pixels per tile when `p` runs on conv3x3_pws_kernel with its XCD-affine tile order (launch_pws: lanes a multiple of 8, so that
// pixel tile t -- and every later tile of its block -- sits on XCD t % 8), else 0.  The fused BatchNorm launches on either side of
// such a convolution order their blocks to match (bn.hip, vpd_bn_virtual_block).
int vpd_conv_xcd_tile_px(const ConvParams& p) {
    HaloGeom g;
    if (!pws_enabled(p)) return 0;
    int bm = 0, bn = 0;
    switch (vpd_conv_kernel_class(p, &g)) {
        case 1: if ((p.M + 255) / 256 > pws_cu_count() / (p.Co / 128)) { bm = 256; bn = 128; } break;
        case 2: bm = 128; bn = 128; break;
        case 3: bm = 128; bn = 64; break;
        case 6: bm = 256; bn = 64; break;
        default: break;
    }
    if (!bm) return 0;
    const int MT = (p.M + bm - 1) / bm, NT = p.Co / bn;
    int lanes = pws_cu_count() / NT;
    if (lanes > MT) lanes = MT;
    return lanes >= 8 ? bm : 0;
}

// true when this launch's kernel can take the sums of the consuming BatchNorm's backward in its epilogue (bst_z / bst_mask)
bool vpd_conv_takes_bn_sums(const ConvParams& p) {
    if (p.ep_scale || p.alt_w || p.yC != p.Co || p.ypad != 0) return false;
    HaloGeom g;
    const int kc = vpd_conv_kernel_class(p, &g);
    if (kc == 4) {      // gather kernel: the merged parity classes of a stride-2 data gradient (plain store only)
        if (p.bst_z2) return false;
        static const int s2 = getenv("VPD_DGRAD_SUMS_S2") ? atoi(getenv("VPD_DGRAD_SUMS_S2")) : 1;
        static const int g1 = getenv("VPD_DGRAD_SUMS_1X1") ? atoi(getenv("VPD_DGRAD_SUMS_1X1")) : 1;
        // ... or a dense stride-1 launch of it (the Bottleneck students' 1x1 convs), plain or accumulating
        if (p.osub == 1 && p.ncls <= 1 && !p.x2) return g1 != 0;
        return s2 && !p.accumulate && p.osub == 2;
    }
    if (p.x2 || p.osub != 1) return false;
    if (p.yWp != p.Ws || p.yHp != p.Hs) return false;      // (the 3x3 kernels index z / the bit map by dense pixel number)
    if (p.bst_z2) return p.accumulate && p.stats2 && (kc == 1 || kc == 2 || kc == 3 || kc == 6);      // mode 8: conv3x3_ws_kernel only
    if (kc == 0) {      // layer1's persistent kernel (not its two-group variant)
        static const int l1 = getenv("VPD_DGRAD_SUMS_L1") ? atoi(getenv("VPD_DGRAD_SUMS_L1")) : 1;
        HaloGeom g2;
        return l1 && !c64x2_geom(p, &g2);
    }
    return kc == 1 || kc == 2 || kc == 3 || kc == 6;
}

hipError_t vpd_launch_conv(const ConvParams& p0, hipStream_t stream) {
    if (p0.Kc % 64 != 0 || p0.Co % 64 != 0 || p0.M <= 0) return hipErrorInvalidValue;
    if (p0.bst_z && !vpd_conv_takes_bn_sums(p0)) return hipErrorInvalidValue;      // (the caller asks first)
    // epilogue modes 6 / 7 / 8 are selected by bst_z alone, and their statistics flush (with its workgroup barriers, which the
    // loader waves of the warp-specialised kernels match one for one) runs only with rows to add to
    if (p0.bst_z && (!p0.stats || !p0.bst_mask)) return hipErrorInvalidValue;
    if (p0.bst_z2 && !p0.stats2) return hipErrorInvalidValue;
    ConvParams p = p0;
#ifdef VPD_ENABLE_ABLATE
    static const int ablate = getenv("VPD_ABLATE") ? atoi(getenv("VPD_ABLATE")) : 0;
#else
    constexpr int ablate = 0;
#endif
    p.ablate = ablate;
    HaloGeom g;
    switch (vpd_conv_kernel_class(p, &g)) {
        case 0: {
            HaloGeom g2;
            if (c64x2_geom(p, &g2)) return launch_c64x2(p, g2, stream);      // inference: two MFMA wave groups on 256-pixel tiles
            return launch_c64<224>(p, g, stream);
        }
        case 1:
            // (eight MFMA waves, no fragment pipeline: two waves per SIMD cover each other's LDS round trips.  With ONE tile per
            //  block conv3x3_ws_kernel, whose loaders run three bundles ahead instead of two, is 3 % faster: 25.5 vs 26.3 us)
            if (pws_enabled(p) && (p.M + 255) / 256 > pws_cu_count() / (p.Co / 128)) {
                return launch_pws<256, 128, 352, PWS_NS_C1, 8, false>(p, g, stream);
            }
            return launch_ws<256, 128, 352, 2, 2, 4>(p, g, stream);      // 88 + 64 KiB
        // four ring stages (the loaders three weight tiles ahead): same-box A/B against 3 / 5 stages in
        // profiles/r02_ring_depth.txt (4 is +0.5 % on the step, 5 is slower than 3)
        case 2:
            if (pws_enabled(p)) {
                return launch_pws<128, 128, 288, PWS_NS_C2, 4, true>(p, g, stream);
            }
            return launch_ws<128, 128, 288, 2, 2, 4>(p, g, stream);      // 72 + 64 KiB
        case 3:
            if (pws_enabled(p)) {
                return launch_pws<128, 64, 288, PWS_NS_C3, 4, true>(p, g, stream);
            }
            return launch_ws<128, 64, 288, 2, 2, 4>(p, g, stream);       // 72 + 32 KiB
        case 6:
            if (pws_enabled(p)) {
                // (round 4: layer2's 18 x 18 halo needs 328 rows, not 416, and the 22 KB that frees were given to a deeper weight ring,
                //  NS 7 and 9 -- same-box 72.02 / 72.02 / 71.86 k crops/s for NS 7 / 9 / 5, profiles/r04_ab_layer2_ring_depth.txt: the K
                //  loop is not bound by the loaders' bytes in flight; instantiations removed)
                return launch_pws<256, 64, 416, PWS_NS_C6, 4, true>(p, g, stream);
            }
            return launch_ws<256, 64, 416, 2, 2, 4>(p, g, stream);       // 104 + 32 KiB
        case 5: { int tr; stem_eligible(p, &tr); return launch_stem(p, tr, stream); }
        default: break;
    }
    if (vpd_conv1x1_stream_eligible(p)) return vpd_launch_conv1x1_stream(p, stream);
    if (conv1x1_ws_eligible(p)) return launch_1x1(p, stream);
    // the statistics accumulator rows only depend on the block index, so the tile choice is free
    const int bm = vpd_conv_bm(p.M, p.Co);
    if (halo_eligible(p) && !p.alt_w && !p.x2) {
        if (p.Co % 128 == 0) {
            if (bm == 128 && halo_geom(p, 128, 224, &g))
                return p.Kc > 128 ? launch_halo<128, 128, 224, true>(p, g, stream)
                                  : launch_halo<128, 128, 224, false>(p, g, stream);
            if (halo_geom(p, 64, 160, &g)) return launch_halo<64, 128, 160, true>(p, g, stream);
        } else if (halo_geom(p, 128, 224, &g)) {
            return launch_halo<128, 64, 224, false>(p, g, stream);
        }
    }
    if (p.Co % 128 == 0) {
        if (bm == 128) return launch_cfg<128, 128, 2, 2>(p, stream);
        return launch_cfg<64, 64, 2, 2>(p, stream);
    }
    return launch_cfg<128, 64, 2, 2>(p, stream);
}
