// Persistent warp-specialised 3x3 stride-1 convolution, barrier-synchronised (round 3; the point-to-point LDS flag ring it was
// first built with lost to the barrier it replaced, see below -- ConvParams::err is only used by -DPWS_STAMPS builds).
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
//   * ONE workgroup barrier per K-step is kept -- measured (tools/probe/flag_probe.hip, profiles/r03_flag_probe.txt): an
//     s_barrier costs ~33 cycles, one LDS flag hop (store -> seen by a spinning wave) ~210, so the point-to-point flag
//     ring this kernel was first built with (READY / DONE words per wave, profiles/r03_pws_flag_variant.txt) lost to the
//     barrier it replaced.  What the barrier guarantees is moved instead: the K-steps of a block are numbered
//     0, 1, 2 ... across its tiles, step i lives in ring stage i % NS, NS = A + 2; barrier READY_s means "steps s AND
//     s + 1 have landed, and every wave has finished reading step s - 1".  Behind it the loaders issue the bundle of
//     step s + 1 + A into the stage of step s - 1; in front of the next one they wait with a counted vmcnt that leaves
//     the A - 1 youngest bundles in flight;
//   * because step s + 1 is known to be there, the MFMA waves run a two-set fragment pipeline ACROSS the barrier: while
//     the MFMAs of half-step h issue, the ds_read_b128 of half-step h + 1 are in flight (PIPE) -- in conv3x3_ws_kernel
//     every group of reads sits exposed in front of its MFMAs, twice per K-step, on all four waves at once;
//   * the loaders sit at READY of the next tile's first step while the MFMA waves run the epilogue, with that tile's
//     first A + 1 weight bundles and its halo already on their way;
//   * the halo of the next 64-channel chunk (or of the next tile's first chunk) rides in the weight bundles issued
//     behind READY of taps 0 .. 8 - A of the current chunk: after READY of tap 0 nobody reads the buffer it overwrites
//     (the previous chunk's last fragments were read before that barrier), and the bundle issued behind tap 8 - A has
//     landed at READY of tap 8, the first step whose pipeline reads the new chunk.  The loader loop is unrolled over the
//     nine taps, so every bundle's size is a compile-time constant and no filler transfer is needed: the LDS-DMA path
//     of a CU (~65 GB/s out of L2) is the scarcest resource of these kernels.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"

struct HaloGeom {
    int TR;        // output rows per tile (BM / W)
    int multi;     // tile spans TR/H whole images (TR > H)
    int HR;        // padded rows in the halo
    int NHP;       // halo pixels = HR * (W + 2)
    int total_pix; // N * (H+2) * (W+2): clamp for the last (ragged) tile
    // conv3x3_pws_kernel only: swizzle key of halo pixel (row hr, column xp of the padded tile) = (xp & kmask) ^ ((hr & rowmask) << kshift)
    int kmask, kshift, rowmask;
    float rH, rW, rWp;   // correctly rounded 1 / H, 1 / W, 1 / (W + 2) from the host (vpd_fdiv)
    // Chunk rotation (round 6; conv3x3_pws_kernel and conv3x3_ws_kernel): pixel tile t walks the 64-channel chunks of the K
    // dimension starting at chunk t % nchunks instead of 0.  Every block of a launch reads chunk cc of 512-byte (layer3) or
    // 1-KB (layer4) pixels and weight rows at the same time -- a quarter / an eighth of the 128-byte lines; staggered, the
    // launch's requests cover all of them (layer3 / layer4 launches 4-6 % shorter, +1.3 % on the step:
    // profiles/r06_ab_chunk_rotation.txt).  The order is a function of the tile index alone, so results do not depend on
    // the grid; each accumulator's fp32 summation order over chunks is rotated accordingly.  Training launches only: the inference
    // forward keeps chunk order 0, 1, 2 ... for every tile, so that an embedding does not depend on the crop's position in its batch.
    int rot;             // 1: on (VPD_PWS_ROT=0 turns it off)
    float rnch;          // 1 / nchunks
};

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

int pws_cu_count();                                            // conv_igemm.hip (VPD_PWS_BLOCKS override)
// ring depths NS of the tiles (timing classes 1, 2, 3, 6 of vpd_conv_kernel_class): weight stages in LDS = K-steps the loaders run ahead + 2
#ifndef PWS_NS_C1
#define PWS_NS_C1 4
#endif
#ifndef PWS_NS_C2
#define PWS_NS_C2 5
#endif
#ifndef PWS_NS_C3
#define PWS_NS_C3 9      // (round 6, with the geo K loop: 7 -> 9 stages, layer4 class -3 %, profiles/r06_ab_ring_depth.txt)
#endif
#ifndef PWS_NS_C6
#define PWS_NS_C6 5
#endif

int vpd_conv_kernel_class(const ConvParams& p, HaloGeom* g);   // conv_igemm.hip

struct PwsGrid {
    int lanes;      // pixel-tile lanes: a block takes pixel tiles lane, lane + lanes, ...
    int NT;         // channel tiles
    int MT;         // pixel tiles
    int xcd;        // 1: blocks b, b + 8, ... (one XCD under round-robin placement) take the channel tiles of ONE lane group
    // IEEE reciprocals from the host (vpd_fdiv needs the correctly rounded 1 / d; in the kernel `1.0f / x` is a 12-instruction
    // division sequence behind a kernel-argument round trip -- three of them stood in front of the first LDS-DMA)
    float rNT, rlanes;
};

// -DPWS_STAMPS (diagnostic builds only; tools/probe): ConvParams::err points at 16 x u64 per block; MFMA wave 0 stamps
// slots 0..7, loader wave 0 slots 8..15 with s_memtime at the points named at the call sites
#ifdef PWS_STAMPS
#define PWS_STAMP(slot) do { if (p.err && lane == 0 && (wave == 0 || wave == NMW)) \
    reinterpret_cast<unsigned long long*>(p.err)[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// the constant 100 MHz counter beside a stamp: (delta s_memtime) / (delta s_memrealtime) x 100 MHz = the clock the shader held
#define PWS_STAMP_RT(slot) do { if (p.err && lane == 0 && (wave == 0 || wave == NMW)) \
    reinterpret_cast<unsigned long long*>(p.err)[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PWS_STAMP(slot) do { } while (0)
#define PWS_STAMP_RT(slot) do { } while (0)
#endif

template <int N>
static __device__ __forceinline__ void pws_vmwait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS-DMA, 16 bytes per lane, as assembly: opaque to hipcc's wait-count pass, which would otherwise drain every
// outstanding transfer in front of a wave's next LDS access
static __device__ __forceinline__ void pws_dma16(const void* gsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// ... with the wave-uniform part of the source address in SGPRs and a 32-bit per-lane byte offset (the SADDR form): the loaders'
// per-lane offsets are constants of the launch, so a transfer costs them no vector ALU instruction at all
static __device__ __forceinline__ void pws_dma16s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// Kernel arguments live in memory the host has just written: the first scalar load of every 64-byte line of them is a miss that
// goes all the way out (~1 us under load), and hipcc loads them group by group as the control flow needs them -- three dependent
// round trips stood in front of the loaders' first transfer.  One dword of every line at once, one wait: the compiler's own
// loads behind it hit the scalar cache.
// (ONE asm statement since round 5: written as sixteen volatile loads, hipcc issued them in two groups with a wait in between
//  when the argument block shrank -- +0.3 us on every launch.  All loads land in one scratch SGPR: only the lines matter.)
template <int NBYTES>
static __device__ __forceinline__ void pws_kernarg_touch() {
    static_assert(NBYTES <= 16 * 64, "kernel-argument block larger than sixteen lines");
    const char __attribute__((address_space(4)))* ka =
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LAST = (NBYTES - 4) & ~3;
#define PWS_KA(k) ((k) * 64 < LAST ? (k) * 64 : LAST)
    unsigned scratch;
    asm volatile(
        "s_load_dword %0, %1, %2\n\ts_load_dword %0, %1, %3\n\ts_load_dword %0, %1, %4\n\ts_load_dword %0, %1, %5\n\t"
        "s_load_dword %0, %1, %6\n\ts_load_dword %0, %1, %7\n\ts_load_dword %0, %1, %8\n\ts_load_dword %0, %1, %9\n\t"
        "s_load_dword %0, %1, %10\n\ts_load_dword %0, %1, %11\n\ts_load_dword %0, %1, %12\n\ts_load_dword %0, %1, %13\n\t"
        "s_load_dword %0, %1, %14\n\ts_load_dword %0, %1, %15\n\ts_load_dword %0, %1, %16\n\ts_load_dword %0, %1, %17\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(scratch)
        : "s"(ka), "n"(PWS_KA(0)), "n"(PWS_KA(1)), "n"(PWS_KA(2)), "n"(PWS_KA(3)), "n"(PWS_KA(4)), "n"(PWS_KA(5)), "n"(PWS_KA(6)),
          "n"(PWS_KA(7)), "n"(PWS_KA(8)), "n"(PWS_KA(9)), "n"(PWS_KA(10)), "n"(PWS_KA(11)), "n"(PWS_KA(12)), "n"(PWS_KA(13)),
          "n"(PWS_KA(14)), "n"(PWS_KA(15))
        : "memory");
#undef PWS_KA
}
// s_waitcnt lgkmcnt(0) as the BUILTIN (vmcnt / expcnt fields at their maxima): hipcc's wait-count pass sees it and does
// not wait again for the fragments it covers
static __device__ __forceinline__ void pws_lgkm0() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
}

// L2 touch of the lines a tile's epilogue will read (z / mask of the consuming BatchNorm, the old y of the accumulate modes, the
// eval residual): one LDS-DMA instruction per pixel group into the 1-KiB dump -- no VGPR destination, so nothing stays reserved
// while the K loop runs, and the MFMA waves have no other vector-memory traffic whose vmcnt order it could disturb.  Issued at
// the start of the tile's last 64-channel chunk (~9 K-steps ahead): the epilogue's own loads then hit L2 instead of exposing an
// HBM round trip per tile (in-step stamps: 8,300 cycles of mode-6 epilogue on 64 x 64 wave tiles against 3,700 for mode 1).
// NEGATIVE: the step got slower with it (the epilogue's exposure is not an L2 miss; the extra DMA competes with the loaders).
// Lane (fr, fq): pixel fr of group b; fq picks the tensor (every tensor's 64-channel slice of a pixel is one 128-byte line).
#ifndef PWS_RES_EARLY
#define PWS_RES_EARLY 1      // eval: residual fragments requested at the start of a tile's last chunk
#endif
#ifndef PWS_BST_EARLY4
#define PWS_BST_EARLY4 0     // 64 x 64 wave tiles too: z fragments + mask bits of modes 6 / 7 requested one chunk ahead (40 registers: they
                             // fit, 248 / 250 VGPRs, since the statistics partials are tile-local).  MEASURED NEGATIVE, same box, kernel trace:
                             // mode 6 26.3 vs 25.4 us, mode 7 28.5 vs 28.1 (profiles/r04_early_z_negative.txt) -- 32 KB of HBM-cold z per
                             // block requested inside the K loop sit in the CU's memory queue in front of the loaders' L2-hit transfers
#endif
#ifndef PWS_BST_ROWS4
#define PWS_BST_ROWS4 3      // ... how many tap rows (3 K-steps each) before the tile's end (1, 2, 3: all measured, none gains)
#endif
#ifndef PWS_ACC_PRE
#define PWS_ACC_PRE 1        // accumulate modes: the old values of y of a whole tile requested before its epilogue
#endif
#ifndef PWS_EVAL_COEF_LDS
#define PWS_EVAL_COEF_LDS 1      // eval: the block's BN scale / shift in LDS for all its tiles (0: 2 * NI global loads per tile epilogue)
#endif
#ifndef PWS_TOUCH
#define PWS_TOUCH 0      // measured: -1.2 % (256 crops), -1.5 % (512), -1.6 % (apply) same-box -- profiles/r03_epilogue_touch_negative.txt
#endif
template <int BM, int BN, int WM, int WN, int EPM, bool ZTOO>
static __device__ __forceinline__ void pws_epilogue_touch(const ConvParams& p, int mtile, int n0, const ConvGeo& geo,
                                                          unsigned dump) {
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16;
    constexpr bool ACC = EPM == 2 || EPM == 7 || EPM == 8;
    constexpr bool BSTM = EPM == 6 || EPM == 7 || EPM == 8;
    if constexpr (!(ACC || BSTM || EPM == 3)) return;
    if (EPM == 3 && !p.res) return;
    if (EPM == 6 && !ZTOO) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int fr = lane & 15, fq = lane >> 4;
    const PixSplit ps = pix_split_init(p, geo);
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mtile * BM + wm * WTM + b * 16 + fr;
        const int mc = m < geo.M ? m : geo.M - 1;
        const void* src;
        if constexpr (EPM == 3) {
            int bi, yy, xx;
            pix_split(ps, mc, bi, yy, xx);
            src = p.res + ((size_t)(bi * p.rHp + yy + p.rpad) * p.rWp + (xx + p.rpad)) * p.rC + n0 + wn * WTN + fq * 8;
        } else {
            size_t yoff;
            if (ps.dense) yoff = (size_t)mc * p.yC;
            else {
                int bi, yy, xx;
                pix_split(ps, mc, bi, yy, xx);
                yoff = ((size_t)(bi * p.yHp + yy * p.osub + geo.oph + p.ypad) * p.yWp + (xx * p.osub + geo.opw + p.ypad)) * p.yC;
            }
            const size_t e = yoff + n0 + wn * WTN;
            const bf16_t* t0 = ACC ? p.y : p.bst_z;                               // (mode 2 has no z; every mode here has t0)
            const bf16_t* t1 = (BSTM && ZTOO) ? p.bst_z : t0;
            const bf16_t* t2 = EPM == 8 ? p.bst_z2 : t1;
            const unsigned char* mk = BSTM ? p.bst_mask : p.acc_mask;
            src = fq == 0 ? (const void*)(t0 + e) : fq == 1 ? (const void*)(t1 + e) : (const void*)(t2 + e);
            if (fq == 3 && mk && (BSTM ? ZTOO : true)) src = (const void*)((size_t)(mk + (e >> 3)) & ~(size_t)15);
        }
        pws_dma16(src, dump);
    }
}

// DMA instructions per loader wave in the bundle issued behind READY of tap t (the weights of step s + 1 + A and this
// tap's share of the next chunk's halo), and the vmcnt immediates that follow from them
template <int W_PER, int HPASS, int HT, int A>
struct PwsSched {
    static constexpr int cnt(int t) { return t < HT ? HPASS / HT + (t < HPASS % HT ? 1 : 0) : 0; }   // halo slices behind READY of tap t
    static constexpr int first(int t) { int k = 0; for (int v = 0; v < t; ++v) k += cnt(v); return k; }
    static constexpr int per(int t) { return W_PER + cnt(t); }
    // in front of READY of tap t the weights of step s + 1 must have landed: they were issued behind READY of step s - A, so
    // the bundles issued behind the READYs of steps s - A + 1 .. s - 1 may stay in flight
    static constexpr int inflight(int t) { int n = 0; for (int k = 1; k < A; ++k) n += per(((t - k) % 9 + 9) % 9); return n; }
    static constexpr int max_inflight() { int m = 0; for (int t = 0; t < 9; ++t) m = inflight(t) > m ? inflight(t) : m; return m; }
};

// WC > 0 (round 6): the image width is a compile-time constant (WC = W; FLIP: the data gradient's mirrored taps), so a pixel
// fragment's LDS address is a per-lane register of the launch + an IMMEDIATE per tap -- see "compile-time geometry" below
template <int BM, int BN, int HROWS, int NS, int EPM, int NMW, bool PIPE, int WC = 0, bool FLIP = false>
static __device__ __forceinline__ void pws_body(const ConvParams& p, const HaloGeom& g, const PwsGrid& sg) {
    constexpr int WN = BN / 64;
    constexpr int WM = NMW / WN;
    constexpr int WTM = BM / WM;
    constexpr int MI = WTM / 16, NI = 4;
    constexpr int WSTAGE = BN * 64;                  // elements
    constexpr int HBUF = HROWS * 64;
    constexpr int W_PER = BN / 32;                   // weight-tile DMA instructions per loader wave and step
    constexpr int HINSTR = HROWS / 8;                // 1-KiB LDS-DMA instructions per halo
    constexpr int HPASS = (HINSTR + 3) / 4;          // ... per loader wave (the last one may be a filler when HINSTR % 4 != 0)
    constexpr int A = NS - 2;                        // weight bundles beyond the two readable steps
    constexpr int HT = 9 - A;                        // READYs of a chunk behind which the next chunk's halo slices may be issued
    using SC = PwsSched<W_PER, HPASS, HT, A>;
    static_assert(HROWS % 8 == 0 && A >= 1 && HT >= 1 && SC::max_inflight() < 64, "ring / vmcnt geometry");
    static_assert(EPM == 0 || EPM == 1 || EPM == 2 || EPM == 3 || EPM == 6 || EPM == 7 || EPM == 8, "epilogue mode");
    static_assert(WC == 0 || NMW == 4 || NMW == 8, "compile-time geometry: four or eight MFMA waves (always the two-set fragment pipeline)");
    constexpr unsigned OFF_W = 2u * HBUF * 2u;                       // bytes
    constexpr unsigned OFF_DUMP = OFF_W + NS * WSTAGE * 2u;
    constexpr unsigned OFF_RED = OFF_DUMP + 1024u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][HBUF] halos, [NS][WSTAGE] weight ring, dump, scratch
    unsigned char* red = smem + OFF_RED;
    pws_kernarg_touch<sizeof(ConvParams) + sizeof(HaloGeom) + sizeof(PwsGrid)>();

    const ConvGeo geo = {p.Hs, p.Ws, p.M, p.oph, p.opw};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.Ws, H = p.Hs, Wp = W + 2;
    const int Ci = p.Kc;
    const int nchunks = Ci >> 6;
    const int nsteps = nchunks * 9;

    PWS_STAMP(wave == 0 ? 0 : 8);                                    // kernel entry
    // block -> (lane group, channel tile)
    int lane0, nt;
    {
        const int b = blockIdx.x;
        const float rNT = sg.rNT;
        if (sg.xcd) { const int k = b >> 3; const int q = vpd_fdiv(k, rNT); nt = k - q * sg.NT; lane0 = q * 8 + (b & 7); }
        else { lane0 = vpd_fdiv(b, rNT); nt = b - lane0 * sg.NT; }
    }
    const int n0 = nt * BN;
    const int njobs = lane0 < sg.MT ? vpd_fdiv(sg.MT - lane0 + sg.lanes - 1, sg.rlanes) : 0;
    if (njobs == 0) return;                                          // (whole block, before any barrier)
    const int total = njobs * nsteps;                                // K-steps of this block = READY barriers
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;            // LDS byte address of the dynamic segment

    if (wave >= NMW) {
        // ------------------------------ loader waves ------------------------------
#ifdef PWS_LOADER_PRIO      // (-DPWS_LOADER_PRIO=1|3, measured neutral: the loaders are the younger half of each SIMD's two waves and lose
                            //  every issue arbitration, but they issue so little that it does not matter -- 72.1 / 72.3 vs 72.1 k crops/s)
        __builtin_amdgcn_s_setprio(PWS_LOADER_PRIO);
#endif
        const int lw = wave - NMW;
        const int piece = lane & 7;
        const int lrow = lane >> 3;
        // A transfer's source = a wave-uniform base in SGPRs (tile / chunk / tap) + a per-lane byte offset that is a constant of
        // the launch: hoff[k] for the k-th halo instruction of this wave (halo pixel hp = (lw + 4k) * 8 + lrow, row hr, column xp
        // of the padded tile: hp * Ci elements + the 16-byte piece XOR-ed with key(hr, xp), HaloGeom), wrow[k] for its k-th weight
        // row.  They are computed once, in the prologue, each right in front of its first use.
        unsigned hoff[HPASS];
        auto halo_off = [&](int k) __attribute__((always_inline)) {
            const int hp = (lw + 4 * k) * 8 + lrow;
            const int hr = vpd_fdiv(hp, g.rWp);
            const int key = ((hp - hr * Wp) & g.kmask) ^ ((hr & g.rowmask) << g.kshift);
            return (unsigned)(hp * Ci * 2 + ((piece ^ key) << 4));
        };
        // first padded pixel of a tile's halo (gr0 = its first global output row b * H + y; < 2^21 for every shape the
        // launcher admits, so the float-reciprocal division is exact); wave-uniform, brought to an SGPR
        auto tile_gp0 = [&](int mtile) __attribute__((always_inline)) {
            const int gr0 = mtile * g.TR;
            const int b = vpd_fdiv(gr0, g.rH);
            const int prow0 = g.multi ? b * (H + 2) : b * (H + 2) + (gr0 - b * H);
            return __builtin_amdgcn_readfirstlane(prow0 * Wp);
        };
        // chunk rotation (HaloGeom::rot): tile t starts at chunk t % nchunks; the chunk a loader instruction fetches = eff(logical
        // chunk, its tile's rotation) -- the MFMA waves only ever see "the next chunk"
        auto tile_rot = [&](int mtile) __attribute__((always_inline)) {
            if (!g.rot) return 0;
            return __builtin_amdgcn_readfirstlane(mtile - vpd_fdiv(mtile, g.rnch) * nchunks);
        };
        auto effcc = [&](int cc, int r) __attribute__((always_inline)) { const int e = cc + r; return e >= nchunks ? e - nchunks : e; };
        auto halo_instr = [&](int gp0, int cc, int buf, int k) __attribute__((always_inline)) {      // (cc: effective chunk)
            if (HINSTR % 4 != 0 && lw + 4 * k >= HINSTR) { pws_dma16(p.w, lds0 + OFF_DUMP); return; }      // filler: keeps the counts
            const unsigned dst = lds0 + (unsigned)buf * (HBUF * 2u) + (unsigned)(lw + 4 * k) * 1024u;
            if (gp0 + HINSTR * 8 <= g.total_pix) {               // (wave-uniform) the whole halo lies inside the tensor
                pws_dma16s(reinterpret_cast<const char*>(p.x) + ((size_t)gp0 * Ci + cc * 64) * 2, hoff[k], dst);
            } else {                                             // the tensor's last tile: rows beyond it re-read its last pixel
                const int hp = (lw + 4 * k) * 8 + lrow;
                const int gp = gp0 + hp < g.total_pix ? gp0 + hp : g.total_pix - 1;
                const unsigned swz = (hoff[k] - (unsigned)(hp * Ci * 2)) >> 1;       // (piece ^ key) << 3, in elements
                pws_dma16(p.x + (size_t)gp * Ci + cc * 64 + swz, dst);
            }
        };
        // per-lane byte offset of this wave's W_PER weight rows inside a [Co][Ci] tap slice
        unsigned wrow[W_PER];
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int n = (lw + 4 * i) * 8 + lrow;
            wrow[i] = (unsigned)(((n0 + n) * Ci + ((piece ^ (n & 7)) << 3)) * 2);
        }
        auto issue_w = [&](int tap, int cc, unsigned stg) __attribute__((always_inline)) {               // (cc: effective chunk)
            const int wsl = p.taps.w0 + (tap / 3) * p.taps.wrs + (tap % 3) * p.taps.wcs;
            const char* wbp = reinterpret_cast<const char*>(p.w) + ((size_t)wsl * p.Co * Ci + cc * 64) * 2;
#pragma unroll
            for (int k = 0; k < W_PER; ++k)
                pws_dma16s(wbp, wrow[k], lds0 + OFF_W + stg * (WSTAGE * 2u) + (unsigned)(lw + 4 * k) * 1024u);
        };
        // prologue: the first tile's first halo, then the weights of steps 0 .. A (retired in this order)
        int w_job = 0, wrot = tile_rot(lane0);                       // tile of the next weight bundle and its rotation
        {
            const int gp0 = tile_gp0(lane0);
            PWS_STAMP(12);                                           // first tile's origin known
#pragma unroll
            for (int k = 0; k < HPASS; ++k) { hoff[k] = halo_off(k); halo_instr(gp0, effcc(0, wrot), 0, k); }
        }
        PWS_STAMP(9);                                                // first halo issued
        int w_tap = 0, w_cc = 0;                                     // tap / chunk-in-tile of the next weight bundle
        unsigned w_st = 0;                                           // ... and its ring stage
        int w_step = 0;                                              // ... and its step index
#pragma unroll
        for (int k = 0; k <= A; ++k) {
            if (w_step < total && !VPD_ABL(p, 1)) issue_w(w_tap, effcc(w_cc, wrot), w_st);
            ++w_step;
            if (++w_tap == 9) { w_tap = 0; if (++w_cc == nchunks) { w_cc = 0; wrot = tile_rot(lane0 + ++w_job * sg.lanes); } }
            if (++w_st == NS) w_st = 0;
        }
        PWS_STAMP_RT(13);                                            // (100 MHz counter at the loader's start of streaming)
        const int total_chunks = njobs * nchunks;
        int h_job = nchunks > 1 ? 0 : 1, h_cc = nchunks > 1 ? 1 : 0; // the chunk whose halo is issued during this chunk
        int hrot = tile_rot(lane0 + h_job * sg.lanes);
        for (int c = 0; c < total_chunks; ++c) {
            const bool has_next = c + 1 < total_chunks;
            const int h_gp0 = has_next ? tile_gp0(lane0 + h_job * sg.lanes) : 0;
            const int h_buf = (c + 1) & 1;
            const int h_cce = effcc(h_cc, hrot);
            // one chunk = nine READY barriers, unrolled: every bundle's size is a constant of its tap
            auto step = [&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                // steps s and s + 1 have landed; on the block's last chunk (bundles shrink: no next halo, no more weights) everything
                if (has_next) pws_vmwait<SC::inflight(t)>();
                else pws_vmwait<0>();
                if (c == 0 && t == 0) PWS_STAMP(10);                 // first two steps landed
                if (!VPD_ABL(p, 16) || (c == 0 && t == 0)) __builtin_amdgcn_s_barrier();      // READY_s (ablation 16: one tile per block only)
                if (w_step < total) {
                    if (!VPD_ABL(p, 1)) issue_w(w_tap, effcc(w_cc, wrot), w_st);
                    ++w_step;
                    if (++w_tap == 9) { w_tap = 0; if (++w_cc == nchunks) { w_cc = 0; wrot = tile_rot(lane0 + ++w_job * sg.lanes); } }
                    if (++w_st == NS) w_st = 0;
                }
                if constexpr (SC::cnt(t) > 0) {
                    if (has_next && !VPD_ABL(p, 4)) {
#pragma unroll
                        for (int u = 0; u < SC::cnt(t); ++u) halo_instr(h_gp0, h_cce, h_buf, SC::first(t) + u);
                    }
                }
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            if (++h_cc == nchunks) { h_cc = 0; ++h_job; hrot = tile_rot(lane0 + h_job * sg.lanes); }
        }
        pws_vmwait<0>();
        PWS_STAMP(11);                                               // loader done
        PWS_STAMP_RT(15);                                            // (100 MHz counter at the loader's end)
        __builtin_amdgcn_s_barrier();                                // END
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    // Pixel-fragment addresses.  Halo pixel r = hbase + toff (toff = tdy * Wp + tdx, the tap's shift) of K-half 0 lives at
    // hb + r * 128 + ((key ^ fq) << 4); key (HaloGeom) depends on the pixel's COLUMN and row parity only, so the lane-dependent
    // part is one of three precomputed words per fragment -- lo[ic][b] = hbase * 128 + (((column + dx_ic) & kmask) ^ rowkey ^ fq) * 16
    // for the tap columns ic = 0, 1, 2 -- and the address of a step is ONE v_add with the wave-uniform hb + toff * 128
    // (^ the tap row's parity bit when rowmask is set).  With the tap loop unrolled over ic, that is all the VALU a step needs:
    // an MFMA wave has room for ~2 other instructions per 16-cycle MFMA, and the generic r -> (r * 8 + ((r & 7) ^ fq)) << 4
    // arithmetic (5 VALU per fragment) plus the tap bookkeeping took ~85 instruction slots per 32-MFMA step.
    unsigned lo[3][MI];
    {
        const float rW = g.rW, rH = g.rH;
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = wm * WTM + b * 16 + fr;
            const int lr = vpd_fdiv(m, rW);
            const int xx = m - lr * W;
            const int li = vpd_fdiv(lr, rH);
            const int hrow = g.multi ? li * (H + 2) + (lr - li * H) : lr;
            const unsigned rowkey = (unsigned)(((hrow & g.rowmask) << g.kshift) ^ fq);
#pragma unroll
            for (int ic = 0; ic < 3; ++ic)
                lo[ic][b] = (unsigned)((hrow * Wp + xx) * 128) +
                            ((((unsigned)(xx + p.taps.dx0 + ic * p.taps.dxs) & (unsigned)g.kmask) ^ rowkey) << 4);
        }
    }
    // weight fragment a of K-half kk: LDS byte offset wa0 (kk = 0) + a * 2048 inside a stage (row wn*64 + a*16 + fr:
    // r & 7 == fr & 7); K-half 1 is the same address with bit 6 flipped (the 16-byte piece index kk*4 + fq flips bit 2)
    unsigned wa0;
    {
        const int r = wn * 64 + fr;
        wa0 = (unsigned)(r * 128 + ((fq ^ (r & 7)) << 4));
    }
    constexpr bool BST = EPM == 6 || EPM == 7 || EPM == 8;
    constexpr bool STATS = EPM == 1 || BST;
    constexpr bool BST_EARLY = PIPE && (EPM == 6 || EPM == 7) && (MI <= 2 || PWS_BST_EARLY4);
    BstPair<NI, VPD_BST_MB(MI)> pr;
    // Per-channel sums (statistics of the stored values; modes 6 / 7 / 8: sum g, sum g * z, sum g * z2): the per-lane partials
    // live only inside a tile's epilogue -- across the K loop they were 32 registers that the early request of the z fragments
    // (40) needs.  At the end of every tile they are reduced over the 16 pixel lanes and added to the block's running sums in
    // LDS, [WM][3][BN] floats with exactly one owner lane per word (fr == 0 of the wave that holds the channel), always in tile
    // order: the result does not depend on timing.  Single-tile blocks do the same work as before.
    float* const redf = reinterpret_cast<float*>(red);
    auto red_word = [&](int which, int a, int j) __attribute__((always_inline)) {
        return (wm * 3 + which) * BN + wn * 64 + a * 16 + 4 * fq + j;
    };
    // eval (mode 3): the block's 2 * BN scale / shift values into the statistics scratch (unused in this mode), once -- every tile's
    // epilogue then starts with LDS reads instead of a global round trip (conv_epilogue_impl, lds_coef)
    const float* const eval_coef = (EPM == 3 && PWS_EVAL_COEF_LDS) ? redf : nullptr;
    if (EPM == 3 && PWS_EVAL_COEF_LDS) {
        if (tid < 2 * BN) redf[tid] = tid < BN ? p.ep_scale[n0 + tid] : p.ep_shift[n0 + tid - BN];
        pws_lgkm0();                                                 // (written before this wave's first READY)
    }
    if (STATS && fr == 0) {
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                redf[red_word(0, a, j)] = 0.f; redf[red_word(1, a, j)] = 0.f;
                if (EPM == 8) redf[red_word(2, a, j)] = 0.f;
            }
    }

#ifdef PWS_STAMPS
    unsigned long long xs_store = 0;                                 // cycles this wave spent issuing the activation / bit-map stores
#endif
    unsigned stage = 0;                                              // ring stage of the step being consumed
    int gch = 0;                                                     // global chunk index: halo buffer gch & 1
    // compile-time geometry: per-lane fragment bases (see the K loop); pb starts in halo buffer 0
    unsigned pb[WC > 0 ? 3 : 1][2][MI];
    unsigned wv0 = 0, wv1 = 0, hbsel = 0;
    if constexpr (WC > 0) {
#pragma unroll
        for (int ic = 0; ic < 3; ++ic)
#pragma unroll
            for (int b = 0; b < MI; ++b) { pb[ic][0][b] = lds0 + lo[ic][b]; pb[ic][1][b] = (lds0 + lo[ic][b]) ^ 64u; }
        wv0 = lds0 + OFF_W + wa0;
        wv1 = wv0 ^ 64u;
    }
    for (int job = 0; job < njobs; ++job) {
        const int mtile = lane0 + job * sg.lanes;
        f32x4 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 af0[NI], bf0[MI], af1[NI], bf1[MI];
        typedef const bf16x8 __attribute__((address_space(3)))* frag_t;
        // fragments by LDS byte address.  A: `wst` = the stage's base + wa0 (^ 64 for K-half 1).  B: halo pixel r of K-half 0 is at
        // hb + r * 128 + ((key ^ fq) << 4) = hb + ((r * 8 + (key ^ fq)) << 4), K-half 1 at that address ^ 64 (buffers are
        // 128-byte aligned).  key (HaloGeom) is made of the pixel's column and row parity, not of r: the 16 pixels of a fragment
        // span 16 / W image rows whose r differ by the two padding columns, and keyed on r & 7 two of those rows meet in the
        // same banks (2-way conflicts on every pixel read of layer3 / layer4: the LDS pipe was ~95 % busy at full MFMA rate)
        auto load_a = [&](bf16x8 (&af)[NI], unsigned addr) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < NI; ++a) af[a] = *(frag_t)(size_t)(addr + a * 2048u);
        };
        auto load_b = [&](bf16x8 (&bfm)[MI], const unsigned (&addr)[MI]) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < MI; ++b) bfm[b] = *(frag_t)(size_t)addr[b];
        };
        // K-half 0 addresses of the pixel fragments of tap column ic in the tap row whose wave-uniform terms are S (hb + toff * 128
        // of its first column + ic * dxs * 128) and P (row-parity bit of the key, as an address bit)
        auto b_addr = [&](unsigned (&addr)[MI], int ic, unsigned S, unsigned P) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < MI; ++b) addr[b] = (lo[ic][b] + S) ^ P;
        };
        auto mfma_set = [&](bf16x8 (&af)[NI], bf16x8 (&bfm)[MI]) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
        };
        // Interleave of one half-step region: its MFMAs (NI * MI) with the address arithmetic and the NI + MI fragment reads of
        // the NEXT half-step, which the source places in front of them.  Left to itself hipcc issues address math, reads and
        // MFMAs as three blocks (and sched_barrier pins exactly that): ~200 cycles of VALU + LDS issue exposed per half-step.
        auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NI + MI; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);      // up to three VALU / SALU
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NI * MI - (NI + MI), 0);
        };

        // modes 6 / 7: the epilogue's z fragments and mask bits; eval with a residual (mode 3): its fragments -- requested at the start of
        // the tile's LAST chunk where the registers are there (see the generic loop below)
        BstFrag<NI, VPD_BST_MB(MI)> bst;
        constexpr bool RES_EARLY = PIPE && EPM == 3 && PWS_RES_EARLY;
        ResFrag<NI, MI> resf;
        const bool res_pre = RES_EARLY && p.res != nullptr;
        // ---- compile-time geometry (WC > 0) -------------------------------------------------------------------------------
        // What bounds the generic K loop below is INSTRUCTION ISSUE, not the LDS array (profiles/r06_lds_counters.txt: no bank
        // conflicts, the array 15-30 % busy): one wave per SIMD issues in order, a 16x16x32 MFMA holds the vector issue port for
        // 8 of its 16 cycles, and hipcc placed the ~45 address / bookkeeping / read instructions of a K-step in blocks of up to
        // 22 between two MFMAs -- the matrix pipe idles behind each block.  With W known at compile time a K-step needs 16
        // ds_read_b128 with immediate offsets + 2 VALU + ~4 SALU, and they are pinned two MFMAs apart:
        //   pb[ic][k][b] = halo buffer + pixel base of fragment b for tap COLUMN ic (the swizzle key depends on the column
        //                  read), K-half k (address bit 6); the tap's row / column shift (tdy * (W + 2) + tdx) * 128 is the
        //                  instruction's offset field; where the key has a row-parity bit (W < 8) it IS address bit 6, so an odd
        //                  tap row swaps the two K-half registers (compile time)
        //   wv[k]        = weight ring base + this lane's row / piece, K-half k; + stage * 8 KB (one v_add per K-half)
        // pb follows the halo double buffer by +- HBUF * 2 per chunk, column ic switched right after its last read of the chunk.
        if constexpr (WC > 0) {
            constexpr int Wpc = WC + 2;
            constexpr bool PAR = WC < 8;
            constexpr unsigned HB2 = HBUF * 2u;
            const int nchunks_ = nchunks;
            auto rd_a = [&](bf16x8 (&af)[NI], unsigned base) __attribute__((always_inline)) {
#pragma unroll
                for (int a = 0; a < NI; ++a) af[a] = *(frag_t)(size_t)(base + a * 2048u);
            };
            auto rd_b = [&](bf16x8 (&bfm)[MI], auto icc, auto kc, auto immc) __attribute__((always_inline)) {
                constexpr int ic = decltype(icc)::value, k = decltype(kc)::value;
                constexpr unsigned imm = decltype(immc)::value;
#pragma unroll
                for (int b = 0; b < MI; ++b) bfm[b] = *(frag_t)(size_t)(pb[ic][k][b] + imm);
            };
            // one region: Q = NI * MI MFMAs with R = NI + MI reads (and NV vector-ALU instructions) spread between them
            auto pin = [&](auto nvc) __attribute__((always_inline)) {
                constexpr int NV = decltype(nvc)::value;
                constexpr int Q = NI * MI, R = NI + MI, PER = Q / R;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    if (k < NV) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if constexpr (Q - PER * R > 0) __builtin_amdgcn_sched_group_barrier(0x008, Q - PER * R, 0);
            };
            // pixel-fragment-major MFMA order: the reads of a region are issued weights first, pixels last, so the first MFMAs of
            // the next region must not be the ones that need the youngest read (each accumulator still sees the same K order)
            auto mfma_set_b = [&](bf16x8 (&af)[NI], bf16x8 (&bfm)[MI]) __attribute__((always_inline)) {
#pragma unroll
                for (int b = 0; b < MI; ++b)
#pragma unroll
                    for (int a = 0; a < NI; ++a)
                        acc[a][b] = VPD_MFMA16(af[a], bfm[b], acc[a][b]);
            };
            auto tap_imm = [](int t) constexpr { const int ir = t / 3, ic = t % 3; return (unsigned)((((FLIP ? 2 - ir : ir)) * Wpc + (FLIP ? 2 - ic : ic)) * 128); };
            auto tap_par = [](int t) constexpr { const int ir = t / 3; return PAR ? ((FLIP ? 2 - ir : ir) & 1) : 0; };
            if (job == 0) PWS_STAMP(1);
            __builtin_amdgcn_s_barrier();                            // the tile's first READY
            if (job == 0) PWS_STAMP(2);
#ifdef PWS_GEO_DUMMY
            bf16x8 afd[NI], bfd[MI];
            rd_a(afd, wv0); 
#pragma unroll
            for (int b = 0; b < MI; ++b) bfd[b] = afd[b % NI];
#endif
            rd_a(af0, wv0 + stage * (WSTAGE * 2u));
            rd_b(bf0, std::integral_constant<int, 0>{}, std::integral_constant<int, tap_par(0)>{}, std::integral_constant<unsigned, tap_imm(0)>{});
#pragma nounroll
            for (int cc = 0; cc < nchunks_; ++cc) {
                if (BST_EARLY && cc == nchunks_ - 1) conv_bst_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, bst);
                if (RES_EARLY && res_pre && cc == nchunks_ - 1) conv_res_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, resf);
                const unsigned hdelta = hbsel ? 0u - HB2 : HB2;          // to the other halo buffer
                auto tap = [&](auto tc) __attribute__((always_inline)) {
                    constexpr int t = decltype(tc)::value, ic = t % 3;
                    constexpr int nt = (t + 1) % 9, nic = nt % 3;
                    if (t != 0 || cc != 0) __builtin_amdgcn_s_barrier();         // READY_s: this step and the next have landed
                    // region 1: K-half 1 of this step is requested while K-half 0 multiplies
                    rd_a(af1, wv1 + stage * (WSTAGE * 2u));
                    rd_b(bf1, std::integral_constant<int, ic>{}, std::integral_constant<int, 1 ^ tap_par(t)>{}, std::integral_constant<unsigned, tap_imm(t)>{});
#ifdef PWS_GEO_DUMMY      // diagnostic: the reads are issued, nothing waits for them (the MFMAs multiply registers no read targets)
                    mfma_set_b(afd, bfd);
#else
                    mfma_set_b(af0, bf0);
#endif
                    pin(std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    // region 2: K-half 0 of the NEXT step while K-half 1 multiplies; behind the chunk's last read of tap column ic its
                    // bases move to the other halo buffer (t = 6, 7, 8: columns 0, 1, 2; the next chunk's tap 0 reads column 0)
                    const unsigned nstage = stage + 1 == NS ? 0u : stage + 1;
                    if constexpr (t >= 6) {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int b = 0; b < MI; ++b) pb[ic][k][b] += hdelta;
                    }
                    rd_a(af0, wv0 + nstage * (WSTAGE * 2u));
                    rd_b(bf0, std::integral_constant<int, nic>{}, std::integral_constant<int, tap_par(nt)>{}, std::integral_constant<unsigned, tap_imm(nt)>{});
#ifdef PWS_GEO_DUMMY
                    mfma_set_b(afd, bfd);
#else
                    mfma_set_b(af1, bf1);
#endif
                    pin(std::integral_constant<int, (t >= 6 ? 2 * MI + 1 : 1)>{});
                    __builtin_amdgcn_sched_barrier(0);
                    stage = nstage;
                };
                tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
                tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
                tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
                hbsel ^= 1u;
            }
            gch += nchunks_;
#ifdef PWS_GEO_DUMMY
#pragma unroll
            for (int a = 0; a < NI; ++a) { asm volatile("" ::"v"(af0[a]), "v"(af1[a])); }
#pragma unroll
            for (int b = 0; b < MI; ++b) { asm volatile("" ::"v"(bf0[b]), "v"(bf1[b])); }
#endif
        } else {
        const unsigned hb0 = lds0, hb1 = lds0 + HBUF * 2u;           // the two halo buffers
        unsigned hb = (gch & 1) ? hb1 : hb0;
        // tap walk: one kernel ROW per iteration (rolled: fully unrolled, hipcc hoists the addresses of all nine taps out of the
        // loops and spills), its three columns unrolled
        const int dxs128 = p.taps.dxs * 128;
        auto row_S = [&](unsigned hbuf, int tdy) __attribute__((always_inline)) {
            return hbuf + (unsigned)((tdy * Wp + p.taps.dx0) * 128);
        };
        auto row_P = [&](int tdy) __attribute__((always_inline)) { return (unsigned)(((tdy & g.rowmask) << g.kshift) << 4); };
        int ir = 0, tdy = p.taps.dy0;
        unsigned S0 = row_S(hb, tdy), P = row_P(tdy);
        unsigned ba[MI];                                             // K-half 0 addresses of the current step's pixel fragments
        // the tile's first READY; its first fragments are the one exposed LDS round trip of the tile
        if (job == 0) PWS_STAMP(1);                                  // set-up done, waiting for the first bytes
        __builtin_amdgcn_s_barrier();
        if (job == 0) PWS_STAMP(2);                                  // first READY
        if constexpr (PIPE) {
            b_addr(ba, 0, S0, P);
            load_a(af0, lds0 + OFF_W + stage * (WSTAGE * 2u) + wa0); load_b(bf0, ba);
        }
        const int nrows = nchunks * 3;
        // modes 6 / 7: the epilogue's z fragments and mask bits (first 4 pixel groups) are requested at the start of the tile's
        // LAST chunk -- nine K-steps (~3 us) ahead of their use; requested behind the K loop (conv3x3_ws_kernel: its 256-pixel
        // tile has no registers for them) they cost the data gradients ~2 us of exposed memory latency per tile (in-step stamps:
        // 8,400 cycles of epilogue against 3,700 for the plain forward store).  Mode 8 has no registers left for it.
#pragma nounroll
        for (int row = 0; row < nrows; ++row) {
            if (BST_EARLY && row == nrows - (MI <= 2 ? 3 : PWS_BST_ROWS4)) conv_bst_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, bst);
            if (RES_EARLY && res_pre && row == nrows - 3) conv_res_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, resf);
            if (PWS_TOUCH && row == nrows - 3) pws_epilogue_touch<BM, BN, WM, WN, EPM, !BST_EARLY>(p, mtile, n0, geo, lds0 + OFF_DUMP);
            // the next tap row: the same chunk's, or the first one of the next chunk (other halo buffer)
            const bool wrap = ir == 2;
            const unsigned nhb = wrap ? (hb == hb0 ? hb1 : hb0) : hb;
            const int ntdy = wrap ? p.taps.dy0 : tdy + p.taps.dys;
            const unsigned nS0 = row_S(nhb, ntdy), nP = row_P(ntdy);
#pragma unroll
            for (int ic = 0; ic < 3; ++ic) {
                const unsigned nstage = stage + 1 == NS ? 0u : stage + 1;
                if ((row != 0 || ic != 0) && !VPD_ABL(p, 16)) __builtin_amdgcn_s_barrier();      // READY_s: this step and the next have landed
                if constexpr (PIPE) {
                    if (!VPD_ABL(p, 2)) {
                        // region 1: K-half 1 of this step is requested while K-half 0 multiplies
                        unsigned ba1[MI];
#pragma unroll
                        for (int b = 0; b < MI; ++b) ba1[b] = ba[b] ^ 64u;
                        if (!VPD_ABL(p, 32)) { load_a(af1, (lds0 + OFF_W + stage * (WSTAGE * 2u) + wa0) ^ 64u); load_b(bf1, ba1); }
                        if (!VPD_ABL(p, 64)) mfma_set(af0, bf0);
                        interleave();
                        __builtin_amdgcn_sched_barrier(0);
                        // region 2: K-half 0 of the NEXT step (landed: READY_s covers it; behind a tile's last step these are the
                        // next tile's first fragments or stale bytes, never used) while K-half 1 multiplies
                        if (ic < 2) b_addr(ba, ic + 1, S0 + (unsigned)((ic + 1) * dxs128), P);
                        else b_addr(ba, 0, nS0, nP);
                        if (!VPD_ABL(p, 32)) { load_a(af0, lds0 + OFF_W + nstage * (WSTAGE * 2u) + wa0); load_b(bf0, ba); }
                        if (!VPD_ABL(p, 64)) mfma_set(af1, bf1);
                        interleave();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    b_addr(ba, ic, S0 + (unsigned)(ic * dxs128), P);
                    unsigned ba1[MI];
#pragma unroll
                    for (int b = 0; b < MI; ++b) ba1[b] = ba[b] ^ 64u;
                    const unsigned wst = lds0 + OFF_W + stage * (WSTAGE * 2u) + wa0;
                    load_a(af0, wst); load_b(bf0, ba);
                    load_a(af1, wst ^ 64u); load_b(bf1, ba1);
                    mfma_set(af0, bf0);
                    mfma_set(af1, bf1);
                }
                stage = nstage;
            }
            hb = nhb; tdy = ntdy; S0 = nS0; P = nP;
            if (wrap) { ir = 0; ++gch; } else ++ir;
        }
        }      // (generic K loop)
        if (job == 0) PWS_STAMP(3);                                  // first tile's K loop done
        // ---- epilogue of the tile (the loaders are already filling the ring and the other halo buffer for the next one) ----
        if (VPD_ABL(p, 8)) continue;
        float st1[NI][4], st2[NI][4];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) { st1[a][j] = 0.f; st2[a][j] = 0.f; if (EPM == 8) pr.s3[a][j] = 0.f; }
        if (BST && !BST_EARLY) conv_bst_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, bst);
        if (EPM == 8) conv_bst2_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, pr);
        // accumulate modes: ALL old values of the tile requested up front (inside the epilogue the stores to y keep every
        // pixel group's reload of y behind the previous group's stores: four exposed round trips per tile on 64 x 64 wave tiles)
        constexpr bool ACC_PRE = PWS_ACC_PRE && NMW == 4 && (EPM == 2 || EPM == 7 || (EPM == 8 && MI <= 2));
        AccFrag<NI, MI> accf;
        if constexpr (ACC_PRE) conv_acc_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, accf);
        if constexpr (ACC_PRE && EPM == 8) conv_epilogue_acc_pre2<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst, pr, accf);
        else if constexpr (ACC_PRE && EPM == 7) conv_epilogue_acc_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst, accf);
        else if constexpr (ACC_PRE && EPM == 2) conv_epilogue_acc_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst, accf);
        else if constexpr (EPM == 8) conv_epilogue_pre2<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst, pr);
        else if constexpr (BST) conv_epilogue_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, bst);
        else if (RES_EARLY && res_pre) conv_epilogue_res_pre<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, resf, 0, eval_coef);
        else conv_epilogue<BM, BN, WM, WN, EPM>(p, acc, mtile, n0, st1, st2, geo, 0, eval_coef);
        if constexpr (STATS) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u = row16_sum(st1[a][j]), v = row16_sum(st2[a][j]);
                    const float w = EPM == 8 ? row16_sum(pr.s3[a][j]) : 0.f;
                    if (fr == 0) {      // (the owner lane: LDS float add, no return value)
                        atomicAdd(&redf[red_word(0, a, j)], u);
                        atomicAdd(&redf[red_word(1, a, j)], v);
                        if (EPM == 8) atomicAdd(&redf[red_word(2, a, j)], w);
                    }
                }
        }
        if (job == 0) PWS_STAMP(4);                                  // first tile's epilogue issued
    }
    PWS_STAMP(5);                                                    // all tiles done
    if constexpr (STATS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS adds are done
    __builtin_amdgcn_s_barrier();                                    // END
    PWS_STAMP(14);                                                   // every wave's epilogue issued
    if constexpr (STATS) {
        // the block's sums over its WM pixel-waves, then ONE fp64 atomic per channel and block into the BatchNorm's accumulator
        // rows (conv_stats_flush: the rows are fp64 so that the blocks' arrival order cannot reach the fp32 results)
        if (tid < 2 * BN) {
            const int which = tid / BN;
            const int c = tid - which * BN;
            const int rmask = (p.stat_rows ? p.stat_rows : VPD_STAT_ROWS) - 1;
            if (p.stats) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) t += redf[(w * 3 + which) * BN + c];
                atomicAdd(&p.stats[((size_t)(blockIdx.x & rmask) * 2 + which) * p.Co + n0 + c], (double)t);
            }
            if (EPM == 8 && p.stats2) {      // the second BatchNorm: sum g again, sum g * z2
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) t += redf[(w * 3 + (which ? 2 : 0)) * BN + c];
                atomicAdd(&p.stats2[((size_t)(blockIdx.x & rmask) * 2 + which) * p.Co + n0 + c], (double)t);
            }
        }
    }
    PWS_STAMP(6);                                                    // statistics flushed
#ifdef PWS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PWS_STAMP(7);                                                    // stores drained (diagnostic wait)
#endif
}

template <int BM, int BN, int HROWS, int NS, int EPM, int NMW, bool PIPE, int WC = 0, bool FLIP = false>
__global__ __launch_bounds__((NMW + 4) * 64, (NMW + 4) / 4) void conv3x3_pws_kernel(const ConvParams p, const HaloGeom g,
                                                                                   const PwsGrid sg) {
    pws_body<BM, BN, HROWS, NS, EPM, NMW, PIPE, WC, FLIP>(p, g, sg);
}

// conv_pws_geo.hip: the compile-time-geometry instantiations (W = 16 / 8 on 256 x 64 tiles, W = 4 on 128 x 64 tiles; forward taps or
// the data gradient's mirrored ones).  Returns false when (tile, W, taps, epilogue mode) has no such instantiation: the caller
// launches the generic kernel.
bool vpd_launch_pws_geo(int bm, int bn, int hrows, int ns, int nmw, const ConvParams& q, const HaloGeom& g, const PwsGrid& sg, dim3 grid,
                        dim3 block, size_t lds, hipStream_t stream);
