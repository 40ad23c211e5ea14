// 1x1 convolutions with few input channels on many pixels (the Bottleneck students' layer1 / layer2: 64 <-> 256, 128 <-> 512
// channels on 262,144 / 65,536 pixels; forward, data gradient, eval) as a persistent STREAMING kernel.
//
// These launches are 8.6 GFLOP over 100-170 MB: memory-bound by a factor of ten.  What they need is bytes in flight and nothing
// in the way of the stores.  conv_igemm_kernel (the gather kernel they ran on) stages through registers one K-step ahead and
// re-reads its weight tile per pixel tile: 3.3-4 TB/s.  Here
//   * the whole [BN][Kc] weight tile is brought to LDS ONCE per block (Kc <= 256: 8-64 KB),
//   * a block walks the pixel tiles t = blockIdx.x, + gridDim.x, ... ; a tile's [BM][Kc] input slice is ONE K extent, brought by
//     LDS-DMA into an NSA-deep ring by four loader waves that run NSA - 1 tiles ahead (counted vmcnt waits: the loaders issue no
//     stores, so nothing but their own tiles stands in their queue),
//   * four MFMA waves multiply a tile out of LDS and leave through the shared epilogue (conv_epilogue.h: statistics, accumulate
//     with the ReLU bit map, eval scale / shift / residual / ReLU) -- residual and old-value fragments requested BEFORE the tile's
//     barrier, so that no round trip is exposed between the MFMAs and the stores,
//   * one workgroup barrier per tile.
// Pixel tiles are whole image rows (BM % Ws == 0, Hs * Ws % BM == 0): a lane's share of a tile sits at the same offsets in every
// tile, and the tile's base address is one scalar division away.  Anything else stays on the other kernels.
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"
#include "conv_pws.h"      // gptr_t / lptr_t
#include <string.h>

namespace {

template <int N>
static __device__ __forceinline__ void stream_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `behind` newer tiles of PER instructions each may still be in flight
template <int PER, int MAXB>
static __device__ __forceinline__ void stream_wait_tiles(int behind) {
    static_assert(PER * MAXB < 64, "vmcnt immediate");
    if constexpr (MAXB >= 7) { if (behind >= 7) { stream_wait_vm<7 * PER>(); return; } }
    if constexpr (MAXB >= 6) { if (behind == 6) { stream_wait_vm<6 * PER>(); return; } }
    if constexpr (MAXB >= 5) { if (behind == 5) { stream_wait_vm<5 * PER>(); return; } }
    if constexpr (MAXB >= 4) { if (behind == 4) { stream_wait_vm<4 * PER>(); return; } }
    if constexpr (MAXB >= 3) { if (behind == 3) { stream_wait_vm<3 * PER>(); return; } }
    if constexpr (MAXB >= 2) { if (behind == 2) { stream_wait_vm<2 * PER>(); return; } }
    if constexpr (MAXB >= 1) { if (behind == 1) { stream_wait_vm<1 * PER>(); return; } }
    stream_wait_vm<0>();
}

struct StreamGeo {
    int mtiles;             // pixel tiles of the launch
    int tiles_per_img;      // Hs * Ws / BM
    int rows_per_tile;      // BM / Ws
};

// The loader waves' part of a block (the four waves behind the MFMA waves, lw = 0..3): the resident weight tile, then the block's pixel tiles through
// the ring.  PRE: extra workgroup barriers the MFMA waves run before their first tile (a coefficient prologue), matched here
// once the first AHEAD tiles are on their way.  Returns in front of the END barrier.
// DUAL: two convolutions of 64 input channels each on the same pixels (same padded geometry): chunk 0 = (p.x, p.w), chunk 1 =
// (p.x2, p.w2) -- a down-sampling Bottleneck's closing 1x1 and its 1x1 branch (conv1x1_bn2_stream_kernel)
template <int KC, int BM, int BN, int NSA, int PRE = 0, bool DUAL = false>
static __device__ __forceinline__ void stream_loader(const ConvParams& p, const StreamGeo& sg, bf16_t* sW, bf16_t* ring, int n0,
                                                     int first, int lanes, int ntile, int lw, int lane) {
    constexpr int KCH = KC / 64;
    constexpr int ASTAGE = KCH * BM * 64;
    constexpr int A_PER = BM / 32;
    constexpr int PER_TILE = A_PER * KCH;
    constexpr int W_PER = BN / 32;
    constexpr int AHEAD = NSA - 1;
    const int piece = lane & 7;
    const int lrow = lane >> 3;
    // weights: [BN][Kc] rows n0 .. of the one tap, chunk by chunk
    const bf16_t* const wsrc = p.w + (size_t)p.taps.w0 * p.Co * p.Kc;
    static_assert(!DUAL || KC == 128, "DUAL: two chunks of 64 channels");
#pragma unroll
    for (int cc = 0; cc < KCH; ++cc)
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int n = (lw + 4 * i) * 8 + lrow;
            const bf16_t* const ws = DUAL ? (cc ? p.w2 : p.w) + (size_t)(n0 + n) * 64 : wsrc + (size_t)(n0 + n) * p.Kc + cc * 64;
            __builtin_amdgcn_global_load_lds((gptr_t)(ws + ((piece ^ (n & 7)) << 3)),
                                             (lptr_t)(sW + (cc * BN + (lw + 4 * i) * 8) * 64), 16, 0, 0);
        }
    // this lane's pixel rows of a tile: offsets from the tile's first pixel (same in every tile)
    int aoff[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int r = (lw + 4 * i) * 8 + lrow;
        const int dy = r / p.Ws;
        const int xx = r - dy * p.Ws;
        aoff[i] = (dy * p.istr * p.xWp + xx * p.istr) * p.xC + ((piece ^ (r & 7)) << 3);
    }
    const int tap_off = (p.taps.dy0 * p.xWp + p.taps.dx0) * p.xC;
    auto issue = [&](int it) __attribute__((always_inline)) {
        const int t = first + it * lanes;
        const int b = t / sg.tiles_per_img;
        const int r0 = (t - b * sg.tiles_per_img) * sg.rows_per_tile;
        const size_t toff = (size_t)((b * p.xHp + r0 * p.istr) * p.xWp) * p.xC + tap_off;
        const bf16_t* const src = p.x + toff;
        bf16_t* const st = ring + (it % NSA) * ASTAGE;
#pragma unroll
        for (int cc = 0; cc < KCH; ++cc)
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(DUAL ? (cc ? p.x2 : p.x) + toff + aoff[i] : src + aoff[i] + cc * 64),
                                                 (lptr_t)(st + (cc * BM + (lw + 4 * i) * 8) * 64), 16, 0, 0);
    };
#pragma unroll
    for (int it = 0; it < AHEAD; ++it)
        if (it < ntile) issue(it);
#pragma unroll
    for (int k = 0; k < PRE; ++k) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < ntile; ++it) {
        // tile `it` (and the weights, issued before everything) must have landed; the tiles issued behind it may be in flight
        int behind = ntile - 1 - it;
        behind = behind < AHEAD - 1 ? behind : AHEAD - 1;
        stream_wait_tiles<PER_TILE, AHEAD - 1>(behind);
        __builtin_amdgcn_s_barrier();                         // READY_it (stage (it - 1) % NSA is free again)
        if (it + AHEAD < ntile) issue(it + AHEAD);
    }
    if (ntile == 0) stream_wait_vm<0>();
}

// One pixel tile out of LDS: acc[a][b] += W[channels wn*WTN + 16a ..][K] . X[pixels wm*WTM + 16b ..][K]
template <int KC, int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void stream_mma(const bf16_t* sW, const bf16_t* cA, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                  int wm, int wn, int fr, int fq) {
    constexpr int KCH = KC / 64, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
#pragma unroll
    for (int cc = 0; cc < KCH; ++cc)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[NI], bfm[MI];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int r = wn * WTN + a * 16 + fr;
                af[a] = *reinterpret_cast<const bf16x8*>(sW + (cc * BN + r) * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int r = wm * WTM + b * 16 + fr;
                bfm[b] = *reinterpret_cast<const bf16x8*>(cA + (cc * BM + r) * 64 + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
        }
}

// per-lane partial sums -> 16 pixel lanes (DPP) -> WM pixel waves (LDS) -> ONE fp64 atomic per channel and block into accumulator
// row blockIdx.x % rows.  MFMA waves only; contains ONE block barrier, which the loader waves match.
template <int BN, int WM, int WN>      // WM * WN MFMA waves
static __device__ __forceinline__ void stream_stats_flush(double* rows, int stat_rows, int Co, int n0, float (&st1)[BN / WN / 16][4],
                                                          float (&st2)[BN / WN / 16][4], unsigned char* smem) {
    constexpr int WTN = BN / WN, NI = WTN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM, fr = lane & 15, fq = lane >> 4;
    float* red = reinterpret_cast<float*>(smem);              // [WM][2][BN] (the ring is idle now)
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = row16_sum(st1[a][j]), v = row16_sum(st2[a][j]);
            if (fr == 0) {
                const int c = wn * WTN + a * 16 + 4 * fq + j;
                red[(wm * 2 + 0) * BN + c] = u;
                red[(wm * 2 + 1) * BN + c] = v;
            }
        }
    __syncthreads();
    if (rows) {
        const int rmask = (stat_rows ? stat_rows : VPD_STAT_ROWS) - 1;
        for (int i = tid; i < 2 * BN; i += WM * WN * 64) {
            const int which = i / BN;
            const int c = i - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + c];
            atomicAdd(&rows[((size_t)((int)blockIdx.x & rmask) * 2 + which) * Co + n0 + c], (double)t);
        }
    }
}

template <int KC, int BM, int BN, int NSA, int EPM, int NMW = 4>      // NMW MFMA waves (4 or 8) + 4 loader waves
__global__ __launch_bounds__((NMW + 4) * 64) void conv1x1_stream_kernel(const ConvParams p, const StreamGeo sg) {
    constexpr int KCH = KC / 64;
    constexpr int WN = BN >= 256 ? 4 : BN / 64;
    constexpr int WM = NMW / WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int WELEMS = KCH * BN * 64;              // resident weights [KCH][BN][64]
    constexpr int ASTAGE = KCH * BM * 64;              // one ring stage   [KCH][BM][64]
    constexpr int A_PER = BM / 32;                     // LDS-DMA instructions per loader wave, chunk and tile
    constexpr int PER_TILE = A_PER * KCH;
    constexpr int W_PER = BN / 32;
    constexpr int AHEAD = NSA - 1;
    static_assert(NSA >= 2 && AHEAD * PER_TILE < 64, "ring depth");
    static_assert((size_t)(WELEMS + NSA * ASTAGE) * 2 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const sW = reinterpret_cast<bf16_t*>(smem);
    bf16_t* const ring = sW + WELEMS;

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * BN;
    const int lanes = gridDim.x;
    const int first = blockIdx.x;
    const int ntile = first < sg.mtiles ? (sg.mtiles - first + lanes - 1) / lanes : 0;      // tiles of this block

    if (wave >= NMW) {
        stream_loader<KC, BM, BN, NSA>(p, sg, sW, ring, n0, first, lanes, ntile, wave - NMW, lane);
        __builtin_amdgcn_s_barrier();                             // END
        if (EPM == 1 || EPM == 4) __builtin_amdgcn_s_barrier();   // matches the barrier inside the statistics flush
        return;
    }

    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    for (int it = 0; it < ntile; ++it) {
        const int mtile = first + it * lanes;
        // fragments of the other tensors the epilogue reads: requested before the tile's barrier
        ResFrag<NI, MI> resf;
        AccFrag<NI, MI> accf;
        if (EPM == 3 && p.res) conv_res_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, resf);
        if (EPM == 2) conv_acc_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, accf);
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_barrier();                             // READY_it
        stream_mma<KC, BM, BN, WM, WN>(sW, ring + (it % NSA) * ASTAGE, acc, wm, wn, fr, fq);
        if (EPM == 3 && p.res) {
            conv_epilogue_res_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, resf);
        } else if (EPM == 2) {
            BstFrag<NI, VPD_BST_MB(MI)> none;
            conv_epilogue_acc_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, none, accf);
        } else {
            conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo);
        }
    }
    __builtin_amdgcn_s_barrier();                                 // END
    if (EPM == 1 || EPM == 4) stream_stats_flush<BN, WM, WN>(p.stats, p.stat_rows, p.Co, n0, st1, st2, smem);
}

// ---------------------------------------------------------------------------
// A Bottleneck's closing 1x1 convolution TOGETHER with its BatchNorm (train mode), the convolution computed twice instead of
// written and read back.  z3 = conv3(a2) has 4x the channels of a2 and K is 64 / 128: recomputing a tile is 128 MFMAs per wave,
// writing and re-reading it is 2 x 32 KB through a memory system that is the bound of every launch around it.
//   forward:  [statistics]  conv1x1_stream_kernel<.., 4>: z3 tile -> bf16 -> sum / sum of squares -> the BatchNorm's rows; no store
//             [MODE 1]      z3 tile again -> relu(scale z3 + shift + identity) -> block output + ReLU bit map.  Every block
//                           finalizes the statistics of its own 256 channels in its prologue (bn_finalize_channel, as
//                           bn_fwd_fused_kernel does); the blocks of the first pixel lane store mean / rstd / scale / shift and
//                           update the running statistics.
//   backward: [MODE 2]      z3 tile again; g = d(out) * mask; sum g, sum g z3 -> rows
//             [MODE 3]      z3 tile again; dz3 = A g + B z3 + D (bn_bwd_apply_coef in the prologue, dgamma / dbeta from the first
//                           lane) -> padded dz3 for the weight- and data-gradient launches.
// z3 is rounded to bf16 in registers exactly where the unfused path stores it: the forward is bit-identical to conv + bn_fwd_fused,
// the backward has bn_bwd_apply_fused_kernel's arithmetic.  64 x 256 tiles (eight MFMA waves x 32 pixels x 64 channels; the statistics
// pass keeps the storing launch's four), Kc = 64 / 128.
// ---------------------------------------------------------------------------
struct StreamBn {
    const double* rows; float count;                       // MODE 1 / 3: the sums to finalize ([VPD_FUSED_ROWS][2][Co])
    const float* gamma; const float* beta;
    float* mean; float* rstd; float* scale; float* shift;  // MODE 1: written (first lane); MODE 3: mean / rstd read
    float* rm; float* rv; float momentum, eps;             // MODE 1: running statistics (may be null)
    unsigned char* mask_out;                               // MODE 1: ReLU bit map [M][Co / 8] (may be null)
    double* rows_out;                                      // MODE 2: rows to add sum g / sum g z to
    float* dgamma; float* dbeta;                           // MODE 3 (first lane)
    bf16_t* dz; int dzHp, dzWp, dzpad;                     // MODE 3: output (p.y = d(out), dense, with p.acc_mask = the bit map)
};

static __device__ __forceinline__ unsigned stream_relu_bits(const uint4& ov) {
    // [half != 0] of eight stored non-negative bf16 values as one byte (bn_fwd_fused_kernel's form)
    const unsigned w[4] = {ov.x, ov.y, ov.z, ov.w};
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned f1;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(f1) : "v"(w[j]), "v"(0x00010001u));
        acc |= f1 << (2 * j);
    }
    return (acc | (acc >> 15)) & 0xffu;
}

template <int KC, int NSA, int MODE>
__global__ __launch_bounds__(768) void conv1x1_bn_stream_kernel(const ConvParams p, const StreamGeo sg, const StreamBn bn) {
    // eight MFMA waves (two per SIMD: one's fragment reads and epilogue arithmetic under the other's MFMAs -- with four, the
    // statistics passes ran at 1.8-3 TB/s on phases that do not overlap inside one wave) x 32 pixels x 64 channels
    constexpr int BM = 64, BN = 256, WM = 2, WN = 4, NMW = 8;
    constexpr int KCH = KC / 64;
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
    constexpr int WELEMS = KCH * BN * 64, ASTAGE = KCH * BM * 64;
    static_assert((size_t)(WELEMS + NSA * ASTAGE) * 2 + 3 * BN * 4 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const sW = reinterpret_cast<bf16_t*>(smem);
    bf16_t* const ring = sW + WELEMS;
    float* const coef = reinterpret_cast<float*>(ring + NSA * ASTAGE);      // [3][BN]
    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * BN;
    const int lanes = gridDim.x;
    const int first = blockIdx.x;
    const int ntile = first < sg.mtiles ? (sg.mtiles - first + lanes - 1) / lanes : 0;
    constexpr int PRE = (MODE == 1 || MODE == 3) ? 1 : 0;
    if (wave >= NMW) {
        stream_loader<KC, BM, BN, NSA, PRE>(p, sg, sW, ring, n0, first, lanes, ntile, wave - NMW, lane);
        __builtin_amdgcn_s_barrier();                             // END
        if (MODE == 2) __builtin_amdgcn_s_barrier();              // statistics flush
        return;
    }
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const int nw = n0 + wn * WTN;
    // ---- coefficient prologue: one channel per thread (the first BN of the MFMA waves' threads) ----
    if (MODE == 1 && tid < BN) {
        const int ch = n0 + tid;
        float mu, r, sc, sh; double var;
        bn_finalize_channel(bn.rows, p.Co, ch, bn.count, bn.eps, bn.gamma[ch], bn.beta[ch], &mu, &r, &sc, &sh, &var);
        coef[tid] = sc; coef[BN + tid] = sh;
        if (blockIdx.x == 0) {
            bn.mean[ch] = mu; bn.rstd[ch] = r; bn.scale[ch] = sc; bn.shift[ch] = sh;
            if (bn.rm) {
                const double unb = bn.count > 1.f ? var * (double)bn.count / ((double)bn.count - 1.0) : var;
                bn.rm[ch] = (1.f - bn.momentum) * bn.rm[ch] + bn.momentum * mu;
                bn.rv[ch] = (1.f - bn.momentum) * bn.rv[ch] + bn.momentum * (float)unb;
            }
        }
    } else if (MODE == 3 && tid < BN) {
        const int ch = n0 + tid;
        bn_bwd_apply_coef(bn.rows, p.Co, ch, bn.count, bn.gamma[ch], bn.mean[ch], bn.rstd[ch], coef, coef + BN, coef + 2 * BN,
                          bn.dgamma, bn.dbeta, blockIdx.x == 0, tid);
    }
    float4 k0[NI] = {}, k1[NI] = {}, k2[NI] = {};      // this lane's channels' coefficients: (scale, shift) or (A, B, D)
    if (PRE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            k0[a] = *reinterpret_cast<const float4*>(coef + wn * WTN + a * 16 + 4 * fq);
            k1[a] = *reinterpret_cast<const float4*>(coef + BN + wn * WTN + a * 16 + 4 * fq);
            if (MODE == 3) k2[a] = *reinterpret_cast<const float4*>(coef + 2 * BN + wn * WTN + a * 16 + 4 * fq);
        }
    }
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    const PixSplit ps = pix_split_init(p, geo);
    for (int it = 0; it < ntile; ++it) {
        const int mtile = first + it * lanes;
        ResFrag<NI, MI> resf;      // MODE 1: the identity path (padded activation)
        AccFrag<NI, MI> accf;      // MODE 2 / 3: d(out) (dense) and its ReLU bits
        if (MODE == 1) conv_res_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, resf);
        else conv_acc_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, accf);
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_barrier();                             // READY_it
        stream_mma<KC, BM, BN, WM, WN>(sW, ring + (it % NSA) * ASTAGE, acc, wm, wn, fr, fq);
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = mtile * BM + wm * WTM + b * 16 + fr;    // (M % BM == 0: every row is a pixel)
            int bi, yy, xx;
            pix_split(ps, m, bi, yy, xx);
            const FragRow<NI>& other = MODE == 1 ? resf.r[b] : accf.old[b];
            const unsigned long long mbits = MODE == 1 ? 0ull : accf.bits[b];
#pragma unroll
            for (int a0 = 0; a0 < NI; a0 += 2) {
                uint2 ov2[2], ovs[2];
                ov2[0] = uint2{other.q[a0 / 2].x, other.q[a0 / 2].y};
                ov2[1] = uint2{other.q[a0 / 2].z, other.q[a0 / 2].w};
                frag_pair_swap(ov2[0], ov2[1]);                   // back to the MFMA layout
#pragma unroll
                for (int ai = 0; ai < 2; ++ai) {
                    const int a = a0 + ai;
                    // z3 as the unfused path stores it
                    const unsigned z01 = pack2bf(acc[a][b][0], acc[a][b][1]), z23 = pack2bf(acc[a][b][2], acc[a][b][3]);
                    const float z[4] = {bf2f((unsigned short)(z01 & 0xffff)), bf2f((unsigned short)(z01 >> 16)),
                                        bf2f((unsigned short)(z23 & 0xffff)), bf2f((unsigned short)(z23 >> 16))};
                    const float o[4] = {bf2f((unsigned short)(ov2[ai].x & 0xffff)), bf2f((unsigned short)(ov2[ai].x >> 16)),
                                        bf2f((unsigned short)(ov2[ai].y & 0xffff)), bf2f((unsigned short)(ov2[ai].y >> 16))};
                    const float c0[4] = {k0[a].x, k0[a].y, k0[a].z, k0[a].w};
                    const float c1[4] = {k1[a].x, k1[a].y, k1[a].z, k1[a].w};
                    const float c2[4] = {k2[a].x, k2[a].y, k2[a].z, k2[a].w};
                    float v[4];
                    if (MODE == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = __builtin_fmaf(z[j], c0[j], c1[j]) + o[j];
                            v[j] = v[j] > 0.f ? v[j] : 0.f;
                        }
                    } else {
                        const unsigned bits = (unsigned)(mbits >> (a * 16 + 4 * fq));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float g = ((bits >> j) & 1u) ? o[j] : 0.f;
                            if (MODE == 2) { st1[a][j] += g; st2[a][j] += g * z[j]; }
                            else v[j] = __builtin_fmaf(c0[j], g, __builtin_fmaf(c1[j], z[j], c2[j]));
                        }
                    }
                    if (MODE != 2) ovs[ai] = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                }
                if (MODE == 2) continue;
                frag_pair_swap(ovs[0], ovs[1]);
                const uint4 ov = uint4{ovs[0].x, ovs[0].y, ovs[1].x, ovs[1].y};
                const int chan = frag_pair_chan(a0, fq);
                if (MODE == 1) {
                    bf16_t* const dpix = p.y + ((size_t)(bi * p.yHp + yy + p.ypad) * p.yWp + (xx + p.ypad)) * p.yC + nw;
                    vpd_store16<VPD_CP_EPI>(dpix + chan, ov);
                    if (bn.mask_out) bn.mask_out[(size_t)m * (p.Co >> 3) + ((nw + chan) >> 3)] = (unsigned char)stream_relu_bits(ov);
                } else {
                    bf16_t* const dpix = bn.dz + ((size_t)(bi * bn.dzHp + yy + bn.dzpad) * bn.dzWp + (xx + bn.dzpad)) * p.Co + nw;
                    vpd_store16<VPD_CP_EPI>(dpix + chan, ov);
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();                                 // END
    if (MODE == 2) stream_stats_flush<BN, WM, WN>(bn.rows_out, VPD_FUSED_ROWS, p.Co, n0, st1, st2, smem);
}

// ---------------------------------------------------------------------------
// The same for a DOWN-SAMPLING Bottleneck whose two 1x1 convolutions have 64 input channels each and stride 1 (layer1's first
// block): out = relu(BatchNorm3(conv3(a2)) + BatchNormD(convD(x))).  Both weight tiles are resident, a ring stage holds the
// a2 tile and the x tile of the same pixels, two accumulator sets per wave.  Neither z3 nor zd is ever stored.
//   MODE 1: both BatchNorms finalized in the prologue; out + ReLU bit map
//   MODE 2: g = d(out) * mask; sum g, sum g z3 -> BatchNorm3's rows; sum g, sum g zd -> BatchNormD's rows
//   MODE 3: dz3 = A3 g + B3 z3 + D3, dzd = Ad g + Bd zd + Dd -> two padded tensors; dgamma / dbeta of both
// (the statistics passes are two launches of conv1x1_stream_kernel<.., 4>, one per convolution)
// ---------------------------------------------------------------------------
struct StreamBnB {                                          // the branch's BatchNorm (fields as in StreamBn)
    const double* rows; const float* gamma; const float* beta;
    float* mean; float* rstd; float* scale; float* shift; float* rm; float* rv;
    double* rows_out; float* dgamma; float* dbeta; bf16_t* dz;
};

template <int NSA, int MODE>
__global__ __launch_bounds__(768) void conv1x1_bn2_stream_kernel(const ConvParams p, const StreamGeo sg, const StreamBn bn,
                                                                 const StreamBnB bb) {
    constexpr int BM = 64, BN = 256, WM = 2, WN = 4, NMW = 8;
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
    constexpr int WELEMS = 2 * BN * 64, ASTAGE = 2 * BM * 64;
    static_assert((size_t)(WELEMS + NSA * ASTAGE) * 2 + 6 * BN * 4 <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const sW = reinterpret_cast<bf16_t*>(smem);
    bf16_t* const ring = sW + WELEMS;
    float* const coef = reinterpret_cast<float*>(ring + NSA * ASTAGE);      // [6][BN]
    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * BN;
    const int lanes = gridDim.x;
    const int first = blockIdx.x;
    const int ntile = first < sg.mtiles ? (sg.mtiles - first + lanes - 1) / lanes : 0;
    constexpr int PRE = (MODE == 1 || MODE == 3) ? 1 : 0;
    if (wave >= NMW) {
        stream_loader<128, BM, BN, NSA, PRE, true>(p, sg, sW, ring, n0, first, lanes, ntile, wave - NMW, lane);
        __builtin_amdgcn_s_barrier();                             // END
        if (MODE == 2) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
        return;
    }
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const int nw = n0 + wn * WTN;
    if (MODE == 1 && tid < BN) {
        const int ch = n0 + tid;
        float mu, r, sc, sh; double var;
        bn_finalize_channel(bn.rows, p.Co, ch, bn.count, bn.eps, bn.gamma[ch], bn.beta[ch], &mu, &r, &sc, &sh, &var);
        coef[tid] = sc; coef[BN + tid] = sh;
        if (blockIdx.x == 0) {
            bn.mean[ch] = mu; bn.rstd[ch] = r; bn.scale[ch] = sc; bn.shift[ch] = sh;
            if (bn.rm) {
                const double unb = bn.count > 1.f ? var * (double)bn.count / ((double)bn.count - 1.0) : var;
                bn.rm[ch] = (1.f - bn.momentum) * bn.rm[ch] + bn.momentum * mu;
                bn.rv[ch] = (1.f - bn.momentum) * bn.rv[ch] + bn.momentum * (float)unb;
            }
        }
        bn_finalize_channel(bb.rows, p.Co, ch, bn.count, bn.eps, bb.gamma[ch], bb.beta[ch], &mu, &r, &sc, &sh, &var);
        coef[2 * BN + tid] = sc; coef[3 * BN + tid] = sh;
        if (blockIdx.x == 0) {
            bb.mean[ch] = mu; bb.rstd[ch] = r; bb.scale[ch] = sc; bb.shift[ch] = sh;
            if (bb.rm) {
                const double unb = bn.count > 1.f ? var * (double)bn.count / ((double)bn.count - 1.0) : var;
                bb.rm[ch] = (1.f - bn.momentum) * bb.rm[ch] + bn.momentum * mu;
                bb.rv[ch] = (1.f - bn.momentum) * bb.rv[ch] + bn.momentum * (float)unb;
            }
        }
    } else if (MODE == 3 && tid < BN) {
        const int ch = n0 + tid;
        bn_bwd_apply_coef(bn.rows, p.Co, ch, bn.count, bn.gamma[ch], bn.mean[ch], bn.rstd[ch], coef, coef + BN, coef + 2 * BN,
                          bn.dgamma, bn.dbeta, blockIdx.x == 0, tid);
        bn_bwd_apply_coef(bb.rows, p.Co, ch, bn.count, bb.gamma[ch], bb.mean[ch], bb.rstd[ch], coef + 3 * BN, coef + 4 * BN,
                          coef + 5 * BN, bb.dgamma, bb.dbeta, blockIdx.x == 0, tid);
    }
    if (PRE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float st1[NI][4], st2[NI][4], st3[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; st3[a][j] = 0.f; }
    const PixSplit ps = pix_split_init(p, geo);
    auto rounded = [](const f32x4& v, float (&z)[4]) __attribute__((always_inline)) {      // as the unfused path stores it
        const unsigned z01 = pack2bf(v[0], v[1]), z23 = pack2bf(v[2], v[3]);
        z[0] = bf2f((unsigned short)(z01 & 0xffff)); z[1] = bf2f((unsigned short)(z01 >> 16));
        z[2] = bf2f((unsigned short)(z23 & 0xffff)); z[3] = bf2f((unsigned short)(z23 >> 16));
    };
    auto cf = [&](int k, int a) __attribute__((always_inline)) {      // coefficient row k, this lane's four channels of group a
        return *reinterpret_cast<const float4*>(coef + k * BN + wn * WTN + a * 16 + 4 * fq);
    };
    for (int it = 0; it < ntile; ++it) {
        const int mtile = first + it * lanes;
        AccFrag<NI, MI> accf;      // MODE 2 / 3: d(out) (dense) and its ReLU bits
        if (MODE != 1) conv_acc_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, accf);
        f32x4 acc3[NI][MI], accd[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) { acc3[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; accd[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        __builtin_amdgcn_s_barrier();                             // READY_it
        const bf16_t* const stg = ring + (it % NSA) * ASTAGE;
        if (MODE == 2) {
            // one accumulator set at a time (three sum sets of 16 registers beside it): conv3's sums, then the branch's
            auto sums = [&](f32x4 (&acc)[NI][MI], bool with_g, float (&sz)[NI][4]) __attribute__((always_inline)) {
#pragma unroll
                for (int b = 0; b < MI; ++b)
#pragma unroll
                    for (int a0 = 0; a0 < NI; a0 += 2) {
                        uint2 dv[2];
                        dv[0] = uint2{accf.old[b].q[a0 / 2].x, accf.old[b].q[a0 / 2].y};
                        dv[1] = uint2{accf.old[b].q[a0 / 2].z, accf.old[b].q[a0 / 2].w};
                        frag_pair_swap(dv[0], dv[1]);
#pragma unroll
                        for (int ai = 0; ai < 2; ++ai) {
                            const int a = a0 + ai;
                            float z[4];
                            rounded(acc[a][b], z);
                            const float d[4] = {bf2f((unsigned short)(dv[ai].x & 0xffff)), bf2f((unsigned short)(dv[ai].x >> 16)),
                                                bf2f((unsigned short)(dv[ai].y & 0xffff)), bf2f((unsigned short)(dv[ai].y >> 16))};
                            const unsigned bits = (unsigned)(accf.bits[b] >> (a * 16 + 4 * fq));
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float g = ((bits >> j) & 1u) ? d[j] : 0.f;
                                if (with_g) st1[a][j] += g;
                                sz[a][j] += g * z[j];
                            }
                        }
                    }
            };
            stream_mma<64, BM, BN, WM, WN>(sW, stg, acc3, wm, wn, fr, fq);
            sums(acc3, true, st2);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b) acc3[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            stream_mma<64, BM, BN, WM, WN>(sW + BN * 64, stg + BM * 64, acc3, wm, wn, fr, fq);
            sums(acc3, false, st3);
            continue;
        }
        stream_mma<64, BM, BN, WM, WN>(sW, stg, acc3, wm, wn, fr, fq);
        stream_mma<64, BM, BN, WM, WN>(sW + BN * 64, stg + BM * 64, accd, wm, wn, fr, fq);
        size_t poff[MI];           // this lane's pixels in the padded outputs (out, or dz3 / dzd: same geometry)
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = mtile * BM + wm * WTM + b * 16 + fr;
            int bi, yy, xx;
            pix_split(ps, m, bi, yy, xx);
            if (MODE == 1) poff[b] = ((size_t)(bi * p.yHp + yy + p.ypad) * p.yWp + (xx + p.ypad)) * p.yC + nw;
            else poff[b] = ((size_t)(bi * bn.dzHp + yy + bn.dzpad) * bn.dzWp + (xx + bn.dzpad)) * p.Co + nw;
        }
#pragma unroll
        for (int a0 = 0; a0 < NI; a0 += 2) {
            float4 k[2][6];
            if (MODE == 1) {
#pragma unroll
                for (int ai = 0; ai < 2; ++ai)
#pragma unroll
                    for (int q = 0; q < 4; ++q) k[ai][q] = cf(q, a0 + ai);
            } else if (MODE == 3) {
#pragma unroll
                for (int ai = 0; ai < 2; ++ai)
#pragma unroll
                    for (int q = 0; q < 6; ++q) k[ai][q] = cf(q, a0 + ai);
            }
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int m = mtile * BM + wm * WTM + b * 16 + fr;
                uint2 dv[2], o3[2], od[2];
                if (MODE != 1) {
                    dv[0] = uint2{accf.old[b].q[a0 / 2].x, accf.old[b].q[a0 / 2].y};
                    dv[1] = uint2{accf.old[b].q[a0 / 2].z, accf.old[b].q[a0 / 2].w};
                    frag_pair_swap(dv[0], dv[1]);
                }
#pragma unroll
                for (int ai = 0; ai < 2; ++ai) {
                    const int a = a0 + ai;
                    float z3[4], zd[4], v[4], w[4];
                    rounded(acc3[a][b], z3);
                    rounded(accd[a][b], zd);
                    if (MODE == 1) {
                        const float s3[4] = {k[ai][0].x, k[ai][0].y, k[ai][0].z, k[ai][0].w}, h3[4] = {k[ai][1].x, k[ai][1].y, k[ai][1].z, k[ai][1].w};
                        const float sd[4] = {k[ai][2].x, k[ai][2].y, k[ai][2].z, k[ai][2].w}, hd[4] = {k[ai][3].x, k[ai][3].y, k[ai][3].z, k[ai][3].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = __builtin_fmaf(z3[j], s3[j], h3[j]) + __builtin_fmaf(zd[j], sd[j], hd[j]);
                            v[j] = v[j] > 0.f ? v[j] : 0.f;
                        }
                        o3[ai] = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                    } else {
                        const float d[4] = {bf2f((unsigned short)(dv[ai].x & 0xffff)), bf2f((unsigned short)(dv[ai].x >> 16)),
                                            bf2f((unsigned short)(dv[ai].y & 0xffff)), bf2f((unsigned short)(dv[ai].y >> 16))};
                        const unsigned bits = (unsigned)(accf.bits[b] >> (a * 16 + 4 * fq));
                        const float A3[4] = {k[ai][0].x, k[ai][0].y, k[ai][0].z, k[ai][0].w}, B3[4] = {k[ai][1].x, k[ai][1].y, k[ai][1].z, k[ai][1].w};
                        const float D3[4] = {k[ai][2].x, k[ai][2].y, k[ai][2].z, k[ai][2].w}, Ad[4] = {k[ai][3].x, k[ai][3].y, k[ai][3].z, k[ai][3].w};
                        const float Bd[4] = {k[ai][4].x, k[ai][4].y, k[ai][4].z, k[ai][4].w}, Dd[4] = {k[ai][5].x, k[ai][5].y, k[ai][5].z, k[ai][5].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float g = ((bits >> j) & 1u) ? d[j] : 0.f;
                            v[j] = __builtin_fmaf(A3[j], g, __builtin_fmaf(B3[j], z3[j], D3[j]));
                            w[j] = __builtin_fmaf(Ad[j], g, __builtin_fmaf(Bd[j], zd[j], Dd[j]));
                        }
                        o3[ai] = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                        od[ai] = uint2{pack2bf(w[0], w[1]), pack2bf(w[2], w[3])};
                    }
                }
                const int chan = frag_pair_chan(a0, fq);
                frag_pair_swap(o3[0], o3[1]);
                const uint4 ov = uint4{o3[0].x, o3[0].y, o3[1].x, o3[1].y};
                if (MODE == 1) {
                    vpd_store16<VPD_CP_EPI>(p.y + poff[b] + chan, ov);
                    if (bn.mask_out) bn.mask_out[(size_t)m * (p.Co >> 3) + ((nw + chan) >> 3)] = (unsigned char)stream_relu_bits(ov);
                } else {
                    vpd_store16<VPD_CP_EPI>(bn.dz + poff[b] + chan, ov);
                    frag_pair_swap(od[0], od[1]);
                    vpd_store16<VPD_CP_EPI>(bb.dz + poff[b] + chan, uint4{od[0].x, od[0].y, od[1].x, od[1].y});
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();                                 // END
    if (MODE == 2) {
        stream_stats_flush<BN, WM, WN>(bn.rows_out, VPD_FUSED_ROWS, p.Co, n0, st1, st2, smem);
        __syncthreads();
        stream_stats_flush<BN, WM, WN>(bb.rows_out, VPD_FUSED_ROWS, p.Co, n0, st1, st3, smem);
    }
}

int stream_cu_count() { return vpd_cu_budget(); }

// tile shape for (Kc, Co): wide channel tiles take 64-pixel tiles (64 accumulator registers beside the prefetched fragments)
void stream_shape(int Kc, int Co, int* bm, int* bn) {
    int n = Co % 256 == 0 ? 256 : Co % 128 == 0 ? 128 : 64;      // (only the instantiated widths: Co = 192 runs on 64-wide tiles)
    if (Kc == 256 && n > 128) n = 128;      // 64 KB of weights + the ring
    *bn = n;
    // 256 input channels: 32 KB per 64 pixels -- small tiles keep 96 KB per CU in flight (128-pixel tiles with two stages
    // and 64 x 128 with three: 4.0-4.3 TB/s; these: +0.9 % on the ResNet-50 step, same box)
    *bm = n == 256 ? 64 : (Kc == 256 ? (n == 64 ? 64 : 32) : 128);
}

template <int KC, int BM, int BN, int NSA>
hipError_t launch_stream(const ConvParams& p, hipStream_t stream) {
    StreamGeo sg;
    sg.mtiles = p.M / BM;
    sg.tiles_per_img = (p.Hs * p.Ws) / BM;
    sg.rows_per_tile = BM / p.Ws;
    const int NT = p.Co / BN;
    int lanes = stream_cu_count() / NT;
    lanes -= lanes % 8;                      // blocks b and b + 8 share an XCD: the NT channel tiles of a pixel tile meet in its L2
    if (lanes < 8) lanes = 8;
    if (lanes > sg.mtiles) lanes = sg.mtiles;
    const dim3 grid(lanes, NT);
    const size_t lds = (size_t)(KC / 64) * 64 * (BN + NSA * BM) * sizeof(bf16_t);
    switch (conv_ep_mode(p)) {
        case 0: VPD_LAUNCH((conv1x1_stream_kernel<KC, BM, BN, NSA, 0>), grid, dim3(512), lds, stream, p, sg); break;
        case 1: VPD_LAUNCH((conv1x1_stream_kernel<KC, BM, BN, NSA, 1>), grid, dim3(512), lds, stream, p, sg); break;
        case 2: VPD_LAUNCH((conv1x1_stream_kernel<KC, BM, BN, NSA, 2>), grid, dim3(512), lds, stream, p, sg); break;
        case 3: VPD_LAUNCH((conv1x1_stream_kernel<KC, BM, BN, NSA, 3>), grid, dim3(512), lds, stream, p, sg); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

// 1x1, one tap, one class, <= 256 input channels, whole-row pixel tiles, enough tiles for every CU to stream
bool vpd_conv1x1_stream_eligible(const ConvParams& p) {
    static const int on = getenv("VPD_CONV1X1_STREAM") ? atoi(getenv("VPD_CONV1X1_STREAM")) : 1;
    if (!on) return false;
    if (p.taps.nr != 1 || p.taps.nc != 1 || p.ncls > 1 || p.alt_w || p.x2 || p.bst_z || p.bst_z2 || p.pool_y) return false;
    if (p.osub != 1 || p.oph != 0 || p.opw != 0 || (p.istr != 1 && p.istr != 2)) return false;
    if (p.Kc != 64 && p.Kc != 128 && p.Kc != 256) return false;
    if (p.xC != p.Kc || p.Co % 64 != 0) return false;
    if (p.accumulate && (p.ypad != 0 || p.yWp != p.Ws || p.yHp != p.Hs)) return false;      // (prefetched old values: dense y)
    int bm, bn;
    stream_shape(p.Kc, p.Co, &bm, &bn);
    if (p.Co % bn != 0) return false;
    if (p.Ws <= 0 || bm % p.Ws != 0 || (p.Hs * p.Ws) % bm != 0) return false;
    if (p.M != p.N * p.Hs * p.Ws || p.M / bm < 2 * stream_cu_count()) return false;
    if (p.M >= VPD_FDIV_MAX) return false;
    return (long)p.N * p.xHp * p.xWp * p.xC < (1l << 31) && (long)p.Co * p.Kc * (p.taps.w0 + 1) < (1l << 31);
}

hipError_t vpd_launch_conv1x1_stream(const ConvParams& p, hipStream_t stream) {
    int bm, bn;
    stream_shape(p.Kc, p.Co, &bm, &bn);
    switch (p.Kc) {
        case 64:
            if (bn == 64) return launch_stream<64, 128, 64, 6>(p, stream);
            if (bn == 128) return launch_stream<64, 128, 128, 6>(p, stream);
            return launch_stream<64, 64, 256, 8>(p, stream);
        case 128:
            if (bn == 64) return launch_stream<128, 128, 64, 4>(p, stream);
            if (bn == 128) return launch_stream<128, 128, 128, 4>(p, stream);
            return launch_stream<128, 64, 256, 6>(p, stream);
        case 256:
            if (bn == 64) return launch_stream<256, 64, 64, 4>(p, stream);
            return launch_stream<256, 32, 128, 6>(p, stream);
        default: break;
    }
    return hipErrorInvalidValue;
}

// ---- the closing 1x1 convolution of a Bottleneck with its BatchNorm (conv1x1_bn_stream_kernel) ----
bool vpd_conv1x1_bn_eligible(const ConvParams& p) {
    static const int on = getenv("VPD_BNECK_RECOMPUTE") ? atoi(getenv("VPD_BNECK_RECOMPUTE")) : 1;
    if (!on || !vpd_conv1x1_stream_eligible(p)) return false;
    if (p.istr != 1 || (p.Kc != 64 && p.Kc != 128) || p.Co % 256 != 0 || p.accumulate || p.ep_scale) return false;
    return p.M % 64 == 0 && 64 % p.Ws == 0 && (p.Hs * p.Ws) % 64 == 0;
}

namespace {
template <int KC, int NSA>
hipError_t launch_bn_stream(const ConvParams& p, const StreamBn& bn, int mode, hipStream_t stream) {
    constexpr int BM = 64, BN = 256;
    StreamGeo sg;
    sg.mtiles = p.M / BM;
    sg.tiles_per_img = (p.Hs * p.Ws) / BM;
    sg.rows_per_tile = BM / p.Ws;
    const int NT = p.Co / BN;
    int lanes = stream_cu_count() / NT;
    lanes -= lanes % 8;
    if (lanes < 8) lanes = 8;
    if (lanes > sg.mtiles) lanes = sg.mtiles;
    const dim3 grid(lanes, NT);
    const size_t lds = (size_t)(KC / 64) * 64 * (BN + NSA * BM) * sizeof(bf16_t) + 3 * BN * sizeof(float);
    switch (mode) {
        // (four MFMA waves as the storing launch: the same pixels per lane, so the same fp32 partial sums -- the fused forward is
        //  bit-identical to conv + BatchNorm launches; eight waves were no faster here, 16.4-18.2 vs 17.8-19.4 us)
        case 0: VPD_LAUNCH((conv1x1_stream_kernel<KC, BM, BN, NSA, 4>), grid, dim3(512), lds, stream, p, sg); break;
        case 1: VPD_LAUNCH((conv1x1_bn_stream_kernel<KC, NSA, 1>), grid, dim3(768), lds, stream, p, sg, bn); break;
        case 2: VPD_LAUNCH((conv1x1_bn_stream_kernel<KC, NSA, 2>), grid, dim3(768), lds, stream, p, sg, bn); break;
        case 3: VPD_LAUNCH((conv1x1_bn_stream_kernel<KC, NSA, 3>), grid, dim3(768), lds, stream, p, sg, bn); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}  // namespace

// mode 0: statistics of z = conv(x) into p.stats (VPD_FUSED_ROWS rows), nothing stored
// mode 1: out = relu(BatchNorm(z) + p.res) into p.y (padded) + ReLU bit map; finalizes f.rows (mean / rstd / scale / shift / running)
// mode 2: sums of the BatchNorm backward (g = p.y * p.acc_mask; p.y dense d(out)) into f.rows
// mode 3: dz = A g + B z + D into `dz` (padded by dzpad); dgamma / dbeta
hipError_t vpd_launch_conv1x1_bn(const ConvParams& p, const BnFusedFwd* fwd, const BnFusedBwd* bwd, const float* mean,
                                 const float* rstd, unsigned char* mask_out, bf16_t* dz, int dzpad, int mode, hipStream_t stream) {
    if (!vpd_conv1x1_bn_eligible(p)) return hipErrorInvalidValue;
    StreamBn bn;
    memset(&bn, 0, sizeof bn);
    if (mode == 0) {
        if (!p.stats || p.stat_rows != VPD_FUSED_ROWS) return hipErrorInvalidValue;
    } else if (mode == 1) {
        if (!fwd || !p.res || !p.y || p.ypad != 1 || p.yC != p.Co || p.rC != p.Co) return hipErrorInvalidValue;
        bn.rows = fwd->rows; bn.count = fwd->count; bn.gamma = fwd->gamma; bn.beta = fwd->beta;
        bn.mean = fwd->mean; bn.rstd = fwd->rstd; bn.scale = fwd->scale; bn.shift = fwd->shift;
        bn.rm = fwd->rm; bn.rv = fwd->rv; bn.momentum = fwd->momentum; bn.eps = fwd->eps;
        bn.mask_out = mask_out;
    } else {
        if (!bwd || !p.y || !p.acc_mask || p.ypad != 0 || p.yC != p.Co || p.yHp != p.Hs || p.yWp != p.Ws) return hipErrorInvalidValue;
        if (mode == 2) bn.rows_out = bwd->rows;
        else {
            if (!dz || !mean || !rstd) return hipErrorInvalidValue;
            bn.rows = bwd->rows; bn.count = bwd->count; bn.gamma = bwd->gamma; bn.dgamma = bwd->dgamma; bn.dbeta = bwd->dbeta;
            bn.mean = const_cast<float*>(mean); bn.rstd = const_cast<float*>(rstd);
            bn.dz = dz; bn.dzHp = p.Hs + 2 * dzpad; bn.dzWp = p.Ws + 2 * dzpad; bn.dzpad = dzpad;
        }
    }
    if (p.Kc == 64) return launch_bn_stream<64, 8>(p, bn, mode, stream);
    return launch_bn_stream<128, 5>(p, bn, mode, stream);
}

// ---- ... of a down-sampling Bottleneck with two 64-channel stride-1 1x1 convolutions (conv1x1_bn2_stream_kernel): p.x / p.w =
// the closing convolution, p.x2 / p.w2 = the branch (same padded input geometry).  fwd: rows / ... = BatchNorm3, rows2 / ... = the
// branch's; bwd3 / bwdD likewise.  Modes 1..3 as vpd_launch_conv1x1_bn (the statistics passes are two mode-0 launches of that).
bool vpd_conv1x1_bn2_eligible(const ConvParams& p) {
    if (!(p.x2 && p.w2 && p.Kc == 64 && p.Kc2 == 64 && p.Co == 256)) return false;
    ConvParams q = p;
    q.x2 = nullptr; q.w2 = nullptr; q.Kc2 = 0;
    return vpd_conv1x1_bn_eligible(q);
}
hipError_t vpd_launch_conv1x1_bn2(const ConvParams& p, const BnFusedFwd* fwd, const BnFusedBwd* bwd3, const BnFusedBwd* bwdD,
                                  const float* mean3, const float* rstd3, const float* meanD, const float* rstdD,
                                  unsigned char* mask_out, bf16_t* dz3, bf16_t* dzD, int dzpad, int mode, hipStream_t stream) {
    if (!vpd_conv1x1_bn2_eligible(p)) return hipErrorInvalidValue;
    StreamBn bn;
    StreamBnB bb;
    memset(&bn, 0, sizeof bn);
    memset(&bb, 0, sizeof bb);
    if (mode == 1) {
        if (!fwd || !fwd->rows2 || !p.y || p.ypad != 1 || p.yC != p.Co) return hipErrorInvalidValue;
        bn.rows = fwd->rows; bn.count = fwd->count; bn.gamma = fwd->gamma; bn.beta = fwd->beta;
        bn.mean = fwd->mean; bn.rstd = fwd->rstd; bn.scale = fwd->scale; bn.shift = fwd->shift;
        bn.rm = fwd->rm; bn.rv = fwd->rv; bn.momentum = fwd->momentum; bn.eps = fwd->eps; bn.mask_out = mask_out;
        bb.rows = fwd->rows2; bb.gamma = fwd->gamma2; bb.beta = fwd->beta2;
        bb.mean = fwd->mean2; bb.rstd = fwd->rstd2; bb.scale = fwd->scale2; bb.shift = fwd->shift2; bb.rm = fwd->rm2; bb.rv = fwd->rv2;
    } else if (mode == 2 || mode == 3) {
        if (!bwd3 || !bwdD || !p.y || !p.acc_mask || p.ypad != 0 || p.yC != p.Co || p.yHp != p.Hs || p.yWp != p.Ws) return hipErrorInvalidValue;
        bn.count = bwd3->count;
        if (mode == 2) { bn.rows_out = bwd3->rows; bb.rows_out = bwdD->rows; }
        else {
            if (!dz3 || !dzD || !mean3 || !rstd3 || !meanD || !rstdD) return hipErrorInvalidValue;
            bn.rows = bwd3->rows; bn.gamma = bwd3->gamma; bn.dgamma = bwd3->dgamma; bn.dbeta = bwd3->dbeta;
            bn.mean = const_cast<float*>(mean3); bn.rstd = const_cast<float*>(rstd3);
            bn.dz = dz3; bn.dzHp = p.Hs + 2 * dzpad; bn.dzWp = p.Ws + 2 * dzpad; bn.dzpad = dzpad;
            bb.rows = bwdD->rows; bb.gamma = bwdD->gamma; bb.dgamma = bwdD->dgamma; bb.dbeta = bwdD->dbeta;
            bb.mean = const_cast<float*>(meanD); bb.rstd = const_cast<float*>(rstdD); bb.dz = dzD;
        }
    } else return hipErrorInvalidValue;
    constexpr int BM = 64, BN = 256, NSA = 5;
    StreamGeo sg;
    sg.mtiles = p.M / BM;
    sg.tiles_per_img = (p.Hs * p.Ws) / BM;
    sg.rows_per_tile = BM / p.Ws;
    int lanes = stream_cu_count();
    lanes -= lanes % 8;
    if (lanes < 8) lanes = 8;
    if (lanes > sg.mtiles) lanes = sg.mtiles;
    const dim3 grid(lanes, 1);
    const size_t lds = (size_t)2 * 64 * (BN + NSA * BM) * sizeof(bf16_t) + 6 * BN * sizeof(float);
    switch (mode) {
        case 1: VPD_LAUNCH((conv1x1_bn2_stream_kernel<NSA, 1>), grid, dim3(768), lds, stream, p, sg, bn, bb); break;
        case 2: VPD_LAUNCH((conv1x1_bn2_stream_kernel<NSA, 2>), grid, dim3(768), lds, stream, p, sg, bn, bb); break;
        default: VPD_LAUNCH((conv1x1_bn2_stream_kernel<NSA, 3>), grid, dim3(768), lds, stream, p, sg, bn, bb); break;
    }
    return hipGetLastError();
}
