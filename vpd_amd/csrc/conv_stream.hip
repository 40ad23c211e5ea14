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

template <int KC, int BM, int BN, int NSA, int EPM>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(const ConvParams p, const StreamGeo sg) {
    constexpr int KCH = KC / 64;
    constexpr int WN = BN >= 256 ? 4 : BN / 64;
    constexpr int WM = 4 / WN;
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

    if (wave >= 4) {
        const int lw = wave - 4;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        // weights: [BN][Kc] rows n0 .. of the one tap, chunk by chunk
        const bf16_t* const wsrc = p.w + (size_t)p.taps.w0 * p.Co * p.Kc;
#pragma unroll
        for (int cc = 0; cc < KCH; ++cc)
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const int n = (lw + 4 * i) * 8 + lrow;
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)(n0 + n) * p.Kc + cc * 64 + ((piece ^ (n & 7)) << 3)),
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
            const bf16_t* const src = p.x + (size_t)((b * p.xHp + r0 * p.istr) * p.xWp) * p.xC + tap_off;
            bf16_t* const st = ring + (it % NSA) * ASTAGE;
#pragma unroll
            for (int cc = 0; cc < KCH; ++cc)
#pragma unroll
                for (int i = 0; i < A_PER; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + aoff[i] + cc * 64),
                                                     (lptr_t)(st + (cc * BM + (lw + 4 * i) * 8) * 64), 16, 0, 0);
        };
#pragma unroll
        for (int it = 0; it < AHEAD; ++it)
            if (it < ntile) issue(it);
        for (int it = 0; it < ntile; ++it) {
            // tile `it` (and the weights, issued before everything) must have landed; the tiles issued behind it may be in flight
            int behind = ntile - 1 - it;
            behind = behind < AHEAD - 1 ? behind : AHEAD - 1;
            stream_wait_tiles<PER_TILE, AHEAD - 1>(behind);
            __builtin_amdgcn_s_barrier();                         // READY_it (stage (it - 1) % NSA is free again)
            if (it + AHEAD < ntile) issue(it + AHEAD);
        }
        if (ntile == 0) stream_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                             // END
        if (EPM == 1) __builtin_amdgcn_s_barrier();               // matches the barrier inside the statistics flush
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
        const bf16_t* const cA = ring + (it % NSA) * ASTAGE;
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
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfm[b], acc[a][b], 0, 0, 0);
            }
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
    if (EPM == 1) {
        // per-lane partial sums -> 16 pixel lanes (DPP) -> WM pixel waves (LDS) -> ONE fp64 atomic per channel and block
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
        if (p.stats) {
            const int rmask = (p.stat_rows ? p.stat_rows : VPD_STAT_ROWS) - 1;
            for (int i = tid; i < 2 * BN; i += 256) {
                const int which = i / BN;
                const int c = i - which * BN;
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + c];
                atomicAdd(&p.stats[((size_t)((int)blockIdx.x & rmask) * 2 + which) * p.Co + n0 + c], (double)t);
            }
        }
    }
}

int stream_cu_count() {
    static const int n = [] {
        int dev = 0, cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cu = pr.multiProcessorCount;
        }
        return cu;
    }();
    return n;
}

// tile shape for (Kc, Co): wide channel tiles take 64-pixel tiles (64 accumulator registers beside the prefetched fragments)
void stream_shape(int Kc, int Co, int* bm, int* bn) {
    int n = Co >= 256 ? 256 : Co;
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
