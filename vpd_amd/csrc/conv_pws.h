// Persistent warp-specialised 3x3 stride-1 convolution with point-to-point LDS flags (round 3).
//
// conv3x3_ws_kernel (conv_igemm.hip) is one tile per block with one workgroup barrier per 64-deep K-step: its four MFMA
// waves meet the four loader waves ~36-72 times per tile, all four MFMA waves then hit the LDS together, nothing of the
// next tile overlaps the epilogue, and every fragment read sits in front of the MFMAs that need it.  This kernel keeps
// the data layout (halo tile per 64-channel chunk, weight tiles through a ring, XOR swizzle on the DMA source) and
// changes the control structure:
//
//   * persistent blocks: grid = (pixel-tile lanes) x (channel tiles), at most one block per CU; a block keeps its
//     channel tile and walks pixel tiles, so the loader stream simply CONTINUES into the next tile's halo and weights
//     while the MFMA waves run the epilogue, and the per-channel statistics are flushed once per block;
//   * no barrier in the K loop.  The K-steps of a block are numbered 0, 1, 2 ... across its tiles; step i lives in ring
//     stage i % NS.  Every loader wave keeps ONE word READY[l] = number of steps whose bytes it has landed, every MFMA
//     wave ONE word DONE[w] = number of steps whose fragments it holds in registers (plain LDS stores of growing
//     values; MI355X_MICROARCH.md, price list row "ring-gemm").  A consumer may read step s when min READY > s, a
//     loader may refill the stage of step i when min DONE > i - NS.  Both sides CACHE the last minimum they saw, so a
//     wave that is behind never polls: it reads the flags again only when its cached value no longer covers the step;
//   * a loader that finds its stage still busy first drains its own transfers (vmcnt 0) and publishes everything it
//     has issued, then spins; a loader that is not blocked publishes behind a counted vmcnt wait that leaves AHEAD
//     bundles in flight;
//   * the MFMA waves run a two-set fragment pipeline ACROSS K-steps: while the MFMAs of half-step h issue, the
//     ds_read_b128 of half-step h + 1 are in flight (PIPE);
//   * the halo of the next 64-channel chunk (or of the next tile's first chunk) rides in the weight bundles of taps
//     NS - 1 .. 8 of the current chunk: the ring's own DONE wait then also proves that the halo buffer it overwrites is
//     no longer read (its last reader is at or before the step whose stage the bundle re-fills), and in-order vmcnt
//     retirement proves that it has landed before the first step that reads it is published.  The loader loop is
//     unrolled over the nine taps, so every bundle's size is a compile-time constant and no filler transfer is needed:
//     the LDS-DMA path of a CU (~65 GB/s out of L2) is the scarcest resource of these kernels.
// Every spin is bounded; a time-out is counted in ConvParams::err (sticky, read by vpd_plan_sync_errors) and the wave
// free-runs to the end: wrong numbers that the host sees, never a hung GPU.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"

#define PWS_SPIN_LIMIT (1 << 22)

struct PwsGrid {
    int lanes;      // pixel-tile lanes: a block takes pixel tiles lane, lane + lanes, ...
    int NT;         // channel tiles
    int MT;         // pixel tiles
    int xcd;        // 1: blocks b, b + 8, ... (one XCD under round-robin placement) take the channel tiles of ONE lane group
};

template <int N>
static __device__ __forceinline__ void pws_vmwait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS-DMA, 16 bytes per lane, as assembly: opaque to hipcc's wait-count pass, which would otherwise drain every
// outstanding transfer in front of the loader's next LDS access (its flag polls)
static __device__ __forceinline__ void pws_dma16(const void* gsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// Flag words: plain LDS stores by the owning wave (all lanes store the same value: one LDS write, no atomics, no EXEC
// games), 16-byte LDS loads by the waiting waves.  The LDS address space is explicit: through a generic pointer hipcc
// turns a volatile access into flat_load ... sc0 sc1 + s_waitcnt vmcnt(0).
typedef volatile unsigned __attribute__((address_space(3)))* pws_flag_t;
typedef const volatile u32x4 __attribute__((address_space(3)))* pws_flag4_t;
static __device__ __forceinline__ void pws_store(unsigned* w, unsigned v) { *(pws_flag_t)w = v; }
static __device__ __forceinline__ unsigned pws_min4(const u32x4& a) {
    const unsigned m = a.x < a.y ? a.x : a.y;
    const unsigned n = a.z < a.w ? a.z : a.w;
    return m < n ? m : n;
}
// s_waitcnt lgkmcnt(0) as the BUILTIN (vmcnt / expcnt fields at their maxima): hipcc's wait-count pass sees it and does
// not wait again for the fragments it covers (behind an asm wait it put lgkmcnt(4..1) in front of the next MFMAs, i.e.
// it waited for half of the reads issued a moment ago)
static __device__ __forceinline__ void pws_lgkm0() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
}
// smallest of the NW words at w (NW = 4 or 8), wave-uniform
template <int NW>
static __device__ __forceinline__ unsigned pws_min_words(const unsigned* w) {
    const u32x4 a = *(pws_flag4_t)w;
    unsigned m = pws_min4(a);
    if constexpr (NW == 8) {
        const u32x4 b = *(pws_flag4_t)(w + 4);
        const unsigned k = pws_min4(b);
        m = m < k ? m : k;
    }
    return __builtin_amdgcn_readfirstlane(m);
}
// spin until the minimum reaches `target`; returns the minimum seen (on time-out: `dead` is set and target is returned)
template <int NW>
static __device__ __forceinline__ unsigned pws_spin(const unsigned* w, unsigned target, bool& dead, unsigned* err) {
    if (dead) return target;
    for (int spin = 0; spin < PWS_SPIN_LIMIT; ++spin) {
        const unsigned m = pws_min_words<NW>(w);
        if (m >= target) return m;
    }
    dead = true;
    if ((threadIdx.x & 63) == 0 && err) atomicAdd(err, 1u);
    return target;
}

// DMA instructions per loader wave in the bundle of tap t (weights of one K-step + this tap's share of the next chunk's halo),
// and the vmcnt immediates that follow from them
template <int W_PER, int HPASS, int HOFF, int HT, int AHEAD>
struct PwsSched {
    static constexpr int cnt(int u) { return HPASS / HT + (u < HPASS % HT ? 1 : 0); }              // u-th halo-carrying bundle
    static constexpr int first(int u) { int k = 0; for (int v = 0; v < u; ++v) k += cnt(v); return k; }   // its first slice
    static constexpr int per(int t) { return W_PER + (t >= HOFF ? cnt(t - HOFF) : 0); }
    // outstanding instructions allowed after the bundle of tap t when everything up to the bundle `ahead` steps back must have landed
    static constexpr int inflight(int t, int ahead) { int n = 0; for (int k = 0; k < ahead; ++k) n += per(((t - k) % 9 + 9) % 9); return n; }
    static constexpr int max_inflight() { int m = 0; for (int t = 0; t < 9; ++t) m = inflight(t, AHEAD) > m ? inflight(t, AHEAD) : m; return m; }
};

template <int BM, int BN, int HROWS, int NS, int AHEAD, int EPM, int NMW, bool PIPE>
__global__ __launch_bounds__((NMW + 4) * 64, (NMW + 4) / 4) void conv3x3_pws_kernel(const ConvParams p, const HaloGeom g,
                                                                                   const PwsGrid sg) {
    constexpr int WN = BN / 64;
    constexpr int WM = NMW / WN;
    constexpr int WTM = BM / WM;
    constexpr int MI = WTM / 16, NI = 4;
    constexpr int WSTAGE = BN * 64;                  // elements
    constexpr int HBUF = HROWS * 64;
    constexpr int W_PER = BN / 32;                   // weight-tile DMA instructions per loader wave and step
    constexpr int HPASS = HROWS / 32;                // halo DMA instructions per loader wave and chunk
    // halo slices of the NEXT chunk ride in the bundles of taps HOFF .. 8 of the current one (HT of them), spread evenly:
    // PwsSched<...>::cnt(u) DMA instructions in the u-th of those bundles, none elsewhere -- every bundle's size is a
    // compile-time constant of its tap, so the counted vmcnt waits need no filler transfers
    constexpr int HOFF = NS - 1;
    constexpr int HT = 9 - HOFF;
    using SC = PwsSched<W_PER, HPASS, HOFF, HT, AHEAD>;
    static_assert(HROWS % 32 == 0 && HT >= 1 && NS >= AHEAD + 1 && AHEAD >= 1 && SC::max_inflight() < 64, "ring / vmcnt geometry");
    static_assert(EPM == 0 || EPM == 1 || EPM == 2 || EPM == 3 || EPM == 6 || EPM == 7 || EPM == 8, "epilogue mode");
    static_assert(NMW == 4 || NMW == 8, "flag words");
    constexpr unsigned OFF_W = 2u * HBUF * 2u;                       // bytes
    constexpr unsigned OFF_DUMP = OFF_W + NS * WSTAGE * 2u;
    constexpr unsigned OFF_FLAG = OFF_DUMP + 1024u;
    constexpr unsigned OFF_RED = OFF_FLAG + 256u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sH = reinterpret_cast<bf16_t*>(smem);                    // [2][HBUF]
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem + OFF_W);            // [NS][WSTAGE]
    unsigned* ready = reinterpret_cast<unsigned*>(smem + OFF_FLAG);  // [4]: steps landed by loader wave l
    unsigned* done = ready + 16;                                     // [NMW]: steps read by MFMA wave w (own 64-byte line)
    unsigned char* red = smem + OFF_RED;

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.Ws, H = p.Hs, Wp = W + 2;
    const int Ci = p.Kc;
    const int nchunks = Ci >> 6;
    const int nsteps = nchunks * 9;

    // block -> (lane group, channel tile)
    int lane0, nt;
    {
        const int b = blockIdx.x;
        if (sg.xcd) { const int k = b >> 3; nt = k % sg.NT; lane0 = (k / sg.NT) * 8 + (b & 7); }
        else { nt = b % sg.NT; lane0 = b / sg.NT; }
    }
    const int n0 = nt * BN;
    const int njobs = lane0 < sg.MT ? (sg.MT - lane0 + sg.lanes - 1) / sg.lanes : 0;
    if (njobs == 0) return;                                          // (whole block, before any barrier)
    if (tid < 64) ready[tid] = 0u;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;            // LDS byte address of the dynamic segment

    if (wave >= NMW) {
        // ------------------------------ loader waves ------------------------------
        const int lw = wave - NMW;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        bool dead = false;
        auto tile_gp0 = [&](int mtile) __attribute__((always_inline)) {
            const int gr0 = mtile * g.TR;
            int prow0;
            if (g.multi) prow0 = (gr0 / H) * (H + 2);
            else { const int b = gr0 / H; prow0 = b * (H + 2) + (gr0 - b * H); }
            return prow0 * Wp;
        };
        auto halo_instr = [&](int gp0, int cc, int buf, int k) __attribute__((always_inline)) {
            const int hp = (lw + 4 * k) * 8 + lrow;
            int gp = gp0 + hp;
            gp = gp < g.total_pix ? gp : g.total_pix - 1;
            const bf16_t* src = p.x + (size_t)gp * Ci + cc * 64 + ((piece ^ (hp & 7)) << 3);
            pws_dma16(src, lds0 + (unsigned)buf * (HBUF * 2u) + (unsigned)(lw + 4 * k) * 1024u);
        };
        // per-lane element offset of this wave's W_PER weight rows inside a [Co][Ci] tap slice
        int wrow[W_PER];
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int n = (lw + 4 * i) * 8 + lrow;
            wrow[i] = (n0 + n) * Ci + ((piece ^ (n & 7)) << 3);
        }
        // first tile, first chunk: the whole halo ahead of bundle 0 (retired in order with it)
        {
            const int gp0 = tile_gp0(lane0);
#pragma unroll
            for (int k = 0; k < HPASS; ++k) halo_instr(gp0, 0, 0, k);
        }
        unsigned st_i = 0;                                           // ring stage of the bundle being issued (i % NS)
        unsigned i = 0;                                              // bundle = K-step index in the block's sequence
        unsigned kd = 0;                                             // cached min DONE
        unsigned pub = 0;                                            // steps this wave has published
        const int total_chunks = njobs * nchunks;
        int w_cc = 0;                                                // chunk of the weight stream inside its tile
        int h_job = nchunks > 1 ? 0 : 1, h_cc = nchunks > 1 ? 1 : 0; // the chunk whose halo rides in this chunk's bundles
        for (int c = 0; c < total_chunks; ++c) {
            const bool has_next = c + 1 < total_chunks;
            const int h_gp0 = has_next ? tile_gp0(lane0 + h_job * sg.lanes) : 0;
            const int h_buf = (c + 1) & 1;
            // one chunk = nine bundles, unrolled: tap t's weights and its fixed share of the next chunk's halo
            auto bundle = [&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                // the stage last held step i - NS: every MFMA wave must have read it (DONE > i - NS)
                if (i >= (unsigned)NS && kd < i - NS + 1u) {
                    kd = pws_min_words<NMW>(done);
                    if (kd < i - NS + 1u) {
                        // blocked: nothing to issue, so land and publish everything issued so far, then wait
                        pws_vmwait<0>();
                        if (pub < i) { pub = i; pws_store(ready + lw, pub); }
                        kd = pws_spin<NMW>(done, i - NS + 1u, dead, p.err);
                    }
                }
                asm volatile("" ::: "memory");
                if (!VPD_ABL(p, 1)) {
                    const int wsl = p.taps.w0 + (t / 3) * p.taps.wrs + (t % 3) * p.taps.wcs;
                    const bf16_t* wbp = p.w + (size_t)wsl * p.Co * Ci + w_cc * 64;
#pragma unroll
                    for (int k = 0; k < W_PER; ++k)
                        pws_dma16(wbp + wrow[k], lds0 + OFF_W + st_i * (WSTAGE * 2u) + (unsigned)(lw + 4 * k) * 1024u);
                }
                if constexpr (t >= HOFF) {
                    if (!VPD_ABL(p, 4)) {
#pragma unroll
                        for (int u = 0; u < SC::cnt(t - HOFF); ++u) {
                            if (has_next) halo_instr(h_gp0, h_cc, h_buf, SC::first(t - HOFF) + u);
                            else pws_dma16(p.w, lds0 + OFF_DUMP);    // last chunk of the block only: keeps the counts
                        }
                    }
                }
                // not blocked: everything up to the bundle AHEAD steps back has landed
                pws_vmwait<SC::inflight(t, AHEAD)>();
                if (i >= (unsigned)AHEAD && pub < i + 1u - AHEAD) { pub = i + 1u - AHEAD; pws_store(ready + lw, pub); }
                ++i;
                if (++st_i == NS) st_i = 0;
            };
            bundle(std::integral_constant<int, 0>{}); bundle(std::integral_constant<int, 1>{}); bundle(std::integral_constant<int, 2>{});
            bundle(std::integral_constant<int, 3>{}); bundle(std::integral_constant<int, 4>{}); bundle(std::integral_constant<int, 5>{});
            bundle(std::integral_constant<int, 6>{}); bundle(std::integral_constant<int, 7>{}); bundle(std::integral_constant<int, 8>{});
            if (++w_cc == nchunks) w_cc = 0;
            if (++h_cc == nchunks) { h_cc = 0; ++h_job; }
        }
        pws_vmwait<0>();
        pws_store(ready + lw, i);                                    // everything has landed
        __builtin_amdgcn_s_barrier();                                // END
        if (EPM == 1 || EPM == 6 || EPM == 7 || EPM == 8) __builtin_amdgcn_s_barrier();      // inside conv_stats_flush
        if (EPM == 8) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }         // second flush
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
    // weight fragment a of K-half kk: element offset wa[kk] + a * 16 * 64 inside a stage (row wn*64 + a*16 + fr: r & 7 == fr & 7)
    int wa[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int r = wn * 64 + fr;
        wa[kk] = r * 64 + (((kk * 4 + fq) ^ (r & 7)) << 3);
    }
    float st1[NI][4], st2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; }
    constexpr bool BST = EPM == 6 || EPM == 7 || EPM == 8;
    BstPair<NI, VPD_BST_MB(MI)> pr;
    if (EPM == 8) {
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) pr.s3[a][j] = 0.f;
    }

    bool dead = false;
    unsigned stage = 0;                                              // ring stage of the step being consumed (sgl % NS)
    unsigned sgl = 0;                                                // its index in the block's step sequence
    unsigned kr = 0;                                                 // cached min READY
    int gch = 0;                                                     // global chunk index: halo buffer gch & 1
    for (int job = 0; job < njobs; ++job) {
        const int mtile = lane0 + job * sg.lanes;
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 af0[NI], bf0[MI], af1[NI], bf1[MI];
        auto load_a = [&](bf16x8 (&af)[NI], unsigned stg, int kk) __attribute__((always_inline)) {
            const bf16_t* cW = sW + stg * WSTAGE + wa[kk];
#pragma unroll
            for (int a = 0; a < NI; ++a) af[a] = *reinterpret_cast<const bf16x8*>(cW + a * 16 * 64);
        };
        // pixel fragments of one K-half: halo rows hbase[b] + toff (toff: the tap's shift in halo pixels, wave-uniform)
        auto load_b = [&](bf16x8 (&bfm)[MI], const bf16_t* cH, int toff, int kk) __attribute__((always_inline)) {
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int r = hbase[b] + toff;
                bfm[b] = *reinterpret_cast<const bf16x8*>(cH + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
        };
        auto mfma_set = [&](bf16x8 (&af)[NI], bf16x8 (&bfm)[MI]) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfm[b], acc[a][b], 0, 0, 0);
        };

        // the tile's first step: an exposed wait (everything the loaders could run ahead is already there)
        if (kr < sgl + 1u) kr = pws_spin<4>(ready, sgl + 1u, dead, p.err);
        asm volatile("" ::: "memory");
        const bf16_t* cH = sH + (gch & 1) * HBUF;
        // tap walk (rolled: unrolled, hipcc hoists the fragment addresses of all nine taps out of the loops and spills)
        const int tstep_c = p.taps.dxs, tstep_r = p.taps.dys * Wp - 2 * p.taps.dxs;
        const int toff0 = p.taps.dy0 * Wp + p.taps.dx0;
        int tap = 0, tcol = 0, toff = toff0;
        if constexpr (PIPE) {
            load_a(af0, stage, 0); load_b(bf0, cH, toff, 0);
            if (VPD_ABL(p, 2)) { load_a(af1, stage, 1); load_b(bf1, cH, toff, 1); }      // (defined values for the epilogue)
        }
#pragma nounroll
        for (int s = 0; s < nsteps; ++s) {
            const bool last = s + 1 == nsteps;
            const unsigned nstage = stage + 1 == NS ? 0u : stage + 1;
            // the next step's tap: shift and halo buffer
            const bool wrap = tap == 8;
            const int ntoff = wrap ? toff0 : toff + (tcol == 2 ? tstep_r : tstep_c);
            const bf16_t* nH = wrap ? sH + ((gch + 1) & 1) * HBUF : cH;
            if constexpr (PIPE) {
                if (!VPD_ABL(p, 2)) { load_a(af1, stage, 1); load_b(bf1, cH, toff, 1); }
                // the next step's bytes: only when the cached count does not cover them is READY read again -- speculatively,
                // in front of the MFMAs that hide the round trip, and checked behind them
                const bool need = !last && kr < sgl + 2u;
                u32x4 fv = {0u, 0u, 0u, 0u};
                if (need) fv = *(pws_flag4_t)ready;
                __builtin_amdgcn_sched_barrier(0);
                if (!VPD_ABL(p, 2)) mfma_set(af0, bf0);
                __builtin_amdgcn_sched_barrier(0);
                pws_lgkm0();                                            // every fragment of this stage is in registers
                pws_store(done + wave, sgl + 1u);
                if (!last) {
                    if (need) {
                        kr = __builtin_amdgcn_readfirstlane(pws_min4(fv));
                        if (kr < sgl + 2u) kr = pws_spin<4>(ready, sgl + 2u, dead, p.err);
                    }
                    asm volatile("" ::: "memory");
                    if (!VPD_ABL(p, 2)) { load_a(af0, nstage, 0); load_b(bf0, nH, ntoff, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!VPD_ABL(p, 2)) mfma_set(af1, bf1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                if (kr < sgl + 1u) kr = pws_spin<4>(ready, sgl + 1u, dead, p.err);
                asm volatile("" ::: "memory");
                load_a(af0, stage, 0); load_b(bf0, cH, toff, 0);
                load_a(af1, stage, 1); load_b(bf1, cH, toff, 1);
                mfma_set(af0, bf0);
                pws_lgkm0();
                pws_store(done + wave, sgl + 1u);
                mfma_set(af1, bf1);
            }
            stage = nstage; ++sgl;
            toff = ntoff; cH = nH;
            tcol = tcol == 2 ? 0 : tcol + 1;
            if (wrap) { tap = 0; ++gch; } else ++tap;
        }
        // ---- epilogue of the tile (the loaders are already filling the ring and the other halo buffer for the next one) ----
        if (VPD_ABL(p, 8)) continue;
        BstFrag<NI, VPD_BST_MB(MI)> bst;
        if (BST) conv_bst_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, bst);
        if (EPM == 8) conv_bst2_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, pr);
        if constexpr (EPM == 8) conv_epilogue_pre2<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst, pr);
        else if constexpr (BST) conv_epilogue_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst);
        else conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo);
    }
    __builtin_amdgcn_s_barrier();                                    // END
    if constexpr (EPM == 1 || BST) {
        conv_stats_flush<BM, BN, WM, WN>(p, st1, st2, blockIdx.x, n0, red);
        if constexpr (EPM == 8) {
            __builtin_amdgcn_s_barrier();                            // the first flush has read the scratch
            conv_stats_flush<BM, BN, WM, WN>(p, st1, pr.s3, blockIdx.x, n0, red, p.stats2);
        }
    }
}
