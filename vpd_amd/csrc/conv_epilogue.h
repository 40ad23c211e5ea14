// Epilogue shared by the convolution kernels (bf16 NHWC store, eval scale/shift/residual/ReLU,
// accumulate, per-channel statistics).
#pragma once
#include "common.h"

// Shared epilogue: acc[a][b][j] = out[channel n0 + wn*WTN + a*16 + 4*fq + j][pixel m0 + wm*WTM + b*16 + fr].
// Stores bf16 NHWC (8 B per lane), optional eval epilogue (scale/shift/residual/ReLU), optional
// read-modify-write accumulate, optional per-channel sum / sum-of-squares of the stored values.
// s1 / s2: the caller's per-lane partial sum / sum of squares of the stored values (accumulated here, reduced and
// published by conv_stats_flush -- once per tile, or once per block in the persistent kernel).
// EPM: epilogue mode fixed at compile time (dead branches cost registers and issue slots in every conv kernel):
//   0 plain store, 1 store + per-channel statistics (train forward), 2 accumulate onto y (data gradient on top of
//   the identity path), 3 eval epilogue (scale/shift, residual, ReLU); -1 decides at run time (legacy kernels).
//   6 / 7 = 0 / 2 + the sums of the consuming BatchNorm's backward (ConvParams::bst_z); 8 = 7 + a second BatchNorm (bst_z2).
//   4 = the statistics of mode 1 WITHOUT the store (conv_stream.hip: a Bottleneck's closing 1x1 conv is computed twice instead of
//   written and read back; never returned by conv_ep_mode).
static __host__ __device__ __forceinline__ int conv_ep_mode(const ConvParams& p) {
    if (p.bst_z) return p.accumulate ? (p.bst_z2 ? 8 : 7) : 6;
    return p.ep_scale ? 3 : (p.accumulate ? 2 : (p.stats ? 1 : 0));
}

// Output pixel m of the sub-grid -> element offset of its first channel in y.  Dense outputs (train-mode z, data gradients:
// y is [M][yC] in sub-grid order) need no split at all; otherwise image / row / column by float-reciprocal division.
struct PixSplit {
    float rHW, rW; int HW, W; bool dense, fast;
};
static __device__ __forceinline__ PixSplit pix_split_init(const ConvParams& p, const ConvGeo& geo) {
    PixSplit s;
    s.HW = geo.Hs * geo.Ws; s.W = geo.Ws;
    s.rHW = 1.0f / (float)s.HW; s.rW = 1.0f / (float)s.W;
    s.dense = p.ypad == 0 && p.osub == 1 && geo.oph == 0 && geo.opw == 0 && p.yWp == geo.Ws && p.yHp == geo.Hs;
    s.fast = geo.M < VPD_FDIV_MAX;
    return s;
}
static __device__ __forceinline__ void pix_split(const PixSplit& s, int mc, int& bi, int& yy, int& xx) {
    if (s.fast) {
        bi = vpd_fdiv(mc, s.rHW);
        const int r = mc - bi * s.HW;
        yy = vpd_fdiv(r, s.rW);
        xx = r - yy * s.W;
    } else {
        bi = mc / s.HW;
        const int r = mc - bi * s.HW;
        yy = r / s.W;
        xx = r - yy * s.W;
    }
}

// ---- 16-byte epilogue accesses (round 4; cdna_hip_programming.md T21 for the 16x16 fragment) ----
// A lane (fr, fq) of the swapped-operand 16x16 MFMA holds, for channel group a, the four channels 16a + 4fq .. + 3 of pixel fr:
// 8 bytes, so a 64-channel wave tile used to leave (and read z / old y / the residual) as 16 eight-byte instructions per
// lane, each touching 32 bytes of 16 different pixels -- store-ISSUE-bound (in-step stamps: 3,600 ticks of epilogue for the
// plain forward store, 8,300 with the BatchNorm-sum loads).  v_permlane16_swap_b32 exchanges the odd lane rows of one register
// with the even rows of another: applied to the packed registers X (group a) and Y (group a + 1) it leaves lane row fq with
// 8 CONSECUTIVE channels of group a + (fq & 1), starting at channel (fq >> 1) * 8, in (X, Y) -- one 16-byte access per lane and
// PAIR of groups, 64 contiguous bytes per pixel and instruction.  The exchange is an involution: data loaded 16 bytes wide in
// that layout is brought back to the MFMA layout by the same two swaps.
static __device__ __forceinline__ void frag_pair_swap(uint2& X, uint2& Y) {
    auto r0 = __builtin_amdgcn_permlane16_swap(X.x, Y.x, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(X.y, Y.y, false, false);
    X.x = r0[0]; Y.x = r0[1];
    X.y = r1[0]; Y.y = r1[1];
}
// element offset, inside the wave's channel range, of this lane's 16 bytes of the pair (a, a + 1), a even
static __device__ __forceinline__ int frag_pair_chan(int a, int fq) { return (a + (fq & 1)) * 16 + (fq >> 1) * 8; }
// One pixel's fragments of a bf16 tensor for NI channel groups, as loaded: pairs 16 bytes wide in the exchanged layout, a
// trailing odd group 8 bytes wide in the MFMA layout.  `base` = the pixel's first channel of this wave's channel range.
template <int NI>
struct FragRow {
    uint4 q[(NI + 1) / 2];
};
template <int NI>
static __device__ __forceinline__ void frag_row_load(FragRow<NI>& f, const bf16_t* base, int fq) {
#pragma unroll
    for (int a = 0; a + 1 < NI; a += 2) f.q[a / 2] = *reinterpret_cast<const uint4*>(base + frag_pair_chan(a, fq));
    if constexpr ((NI & 1) != 0) {
        const uint2 t = *reinterpret_cast<const uint2*>(base + (NI - 1) * 16 + 4 * fq);
        f.q[NI / 2] = uint4{t.x, t.y, 0u, 0u};
    }
}
// ... back to the MFMA layout: v[a] = channels 16a + 4fq .. + 3 (call once the loads have landed)
template <int NI>
static __device__ __forceinline__ void frag_row_unpack(const FragRow<NI>& f, uint2 (&v)[NI]) {
#pragma unroll
    for (int a = 0; a + 1 < NI; a += 2) {
        v[a] = uint2{f.q[a / 2].x, f.q[a / 2].y};
        v[a + 1] = uint2{f.q[a / 2].z, f.q[a / 2].w};
        frag_pair_swap(v[a], v[a + 1]);
    }
    if constexpr ((NI & 1) != 0) v[NI - 1] = uint2{f.q[NI / 2].x, f.q[NI / 2].y};
}
// ... and the store of NI packed groups (MFMA layout in, clobbered)
template <int NI>
static __device__ __forceinline__ void frag_row_store(bf16_t* base, uint2 (&v)[NI], int fq, bool valid) {
#pragma unroll
    for (int a = 0; a + 1 < NI; a += 2) {
        frag_pair_swap(v[a], v[a + 1]);
        if (valid) vpd_store16<VPD_CP_EPI>(base + frag_pair_chan(a, fq), uint4{v[a].x, v[a].y, v[a + 1].x, v[a + 1].y});
    }
    if constexpr ((NI & 1) != 0) {
        if (valid) *reinterpret_cast<uint2*>(base + (NI - 1) * 16 + 4 * fq) = v[NI - 1];
    }
}

// EPM 6 / 7: this lane's z fragments and ReLU bits of the consuming BatchNorm (ConvParams::bst_z / bst_mask), fetched ahead
// of their use -- all loads of a tile in flight together (inside the epilogue the stores to y keep hipcc from hoisting
// them), and in the persistent kernel before the tile's MFMA loop, which hides their latency altogether.
template <int NI, int MI>
struct BstFrag {
    FragRow<NI> z[MI];
    unsigned long long bits[MI];
};
// Accumulate modes (EPM 2 / 7) of the persistent kernel: the old values of y and their ReLU bits (ConvParams::acc_mask),
// requested before the tile's MFMA loop like the fragments above (y dense, ypad 0).
template <int NI, int MI>
struct AccFrag {
    FragRow<NI> old[MI];
    unsigned long long bits[MI];
};
template <int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void conv_acc_prefetch(const ConvParams& p, int mtile, int n0, const ConvGeo& geo,
                                                         AccFrag<BN / WN / 16, BM / WM / 16>& f) {
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int fr = lane & 15, fq = lane >> 4;
    const PixSplit ps = pix_split_init(p, geo);
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mtile * BM + wm * WTM + b * 16 + fr;
        const int mc = m < geo.M ? m : geo.M - 1;
        f.bits[b] = ~0ull;
        if (p.acc_mask) {      // (a mask implies a dense y)
            const unsigned char* mp = p.acc_mask + (size_t)mc * (p.yC >> 3) + ((n0 + wn * WTN) >> 3);
            if (WTN == 64) f.bits[b] = *reinterpret_cast<const unsigned long long*>(mp);
            else if (WTN == 32) f.bits[b] = *reinterpret_cast<const unsigned*>(mp);
            else f.bits[b] = *reinterpret_cast<const unsigned short*>(mp);
        }
        size_t yoff;
        if (ps.dense) yoff = (size_t)mc * p.yC;
        else {
            int bi, yy, xx;
            pix_split(ps, mc, bi, yy, xx);
            yoff = ((size_t)(bi * p.yHp + yy * p.osub + geo.oph + p.ypad) * p.yWp + (xx * p.osub + geo.opw + p.ypad)) * p.yC;
        }
        frag_row_load<NI>(f.old[b], p.y + yoff + n0 + wn * WTN, fq);
    }
}
// Eval epilogue (EPM 3) with a residual: the residual fragments of the tile, requested ahead of the epilogue (the persistent
// kernels: before or during the tile's MFMA loop) -- inside the epilogue they are one exposed round trip per tile.
template <int NI, int MI>
struct ResFrag {
    FragRow<NI> r[MI];
};
template <int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void conv_res_prefetch(const ConvParams& p, int mtile, int n0, const ConvGeo& geo,
                                                         ResFrag<BN / WN / 16, BM / WM / 16>& f, int wave_base = 0) {
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
    const int lane = threadIdx.x & 63;
    const int wave = (threadIdx.x >> 6) - wave_base;
    const int wm = wave % WM, wn = wave / WM;
    const int fr = lane & 15, fq = lane >> 4;
    const PixSplit ps = pix_split_init(p, geo);
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mtile * BM + wm * WTM + b * 16 + fr;
        const int mc = m < geo.M ? m : geo.M - 1;
        int bi, yy, xx;
        pix_split(ps, mc, bi, yy, xx);
        const size_t roff = ((size_t)(bi * p.rHp + yy + p.rpad) * p.rWp + (xx + p.rpad)) * p.rC;
        frag_row_load<NI>(f.r[b], p.res + roff + n0 + wn * WTN, fq);
    }
}
// (at most 4 pixel groups at a time: 8 of them are 64 registers on top of 128 accumulators)
#define VPD_BST_MB(MI) ((MI) > 4 ? 4 : (MI))
template <int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void conv_bst_prefetch(const ConvParams& p, int mtile, int n0, const ConvGeo& geo,
                                                         BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& f,
                                                         int wave_base = 0, int b0 = 0) {
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = VPD_BST_MB(WTM / 16), NI = WTN / 16;
    const int lane = threadIdx.x & 63;
    const int wave = (threadIdx.x >> 6) - wave_base;
    const int wm = wave % WM, wn = wave / WM;
    const int fr = lane & 15, fq = lane >> 4;
    const PixSplit ps = pix_split_init(p, geo);
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mtile * BM + wm * WTM + (b0 + b) * 16 + fr;
        const int mc = m < geo.M ? m : geo.M - 1;
        // y is dense ([pixel][yC], ypad 0): yoff = pixel index * yC, which also indexes z and (in bits) the mask
        size_t yoff;
        if (ps.dense) yoff = (size_t)mc * p.yC;
        else {
            int bi, yy, xx;
            pix_split(ps, mc, bi, yy, xx);
            yoff = ((size_t)(bi * p.yHp + yy * p.osub + geo.oph) * p.yWp + (xx * p.osub + geo.opw)) * p.yC;
        }
        const unsigned char* mp = p.bst_mask + ((yoff + n0 + wn * WTN) >> 3);
        if (WTN == 64) f.bits[b] = *reinterpret_cast<const unsigned long long*>(mp);
        else if (WTN == 32) f.bits[b] = *reinterpret_cast<const unsigned*>(mp);
        else f.bits[b] = *reinterpret_cast<const unsigned short*>(mp);
        frag_row_load<NI>(f.z[b], p.bst_z + yoff + n0 + wn * WTN, fq);
    }
}

// Mode 8: the second BatchNorm's z fragments and its per-lane sum g * z2
template <int NI, int MI>
struct BstPair {
    FragRow<NI> z2[MI];
    float s3[NI][4];
};
template <int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void conv_bst2_prefetch(const ConvParams& p, int mtile, int n0, const ConvGeo& geo,
                                                          BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& f, int b0 = 0) {
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = VPD_BST_MB(WTM / 16), NI = WTN / 16;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int fr = lane & 15, fq = lane >> 4;
    const PixSplit ps = pix_split_init(p, geo);
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mtile * BM + wm * WTM + (b0 + b) * 16 + fr;
        const int mc = m < geo.M ? m : geo.M - 1;
        size_t yoff;
        if (ps.dense) yoff = (size_t)mc * p.yC;
        else {
            int bi, yy, xx;
            pix_split(ps, mc, bi, yy, xx);
            yoff = ((size_t)(bi * p.yHp + yy * p.osub + geo.oph) * p.yWp + (xx * p.osub + geo.opw)) * p.yC;
        }
        frag_row_load<NI>(f.z2[b], p.bst_z2 + yoff + n0 + wn * WTN, fq);
    }
}

// PRE: `own` already holds the first VPD_BST_MB pixel groups (the caller ran conv_bst_prefetch ahead of time)
// APRE: `accf` holds the old values of y and their mask bits (conv_acc_prefetch; accumulate modes, dense y)
template <int BM, int BN, int WM, int WN, int EPM, bool PRE, bool APRE = false>
static __device__ __forceinline__ void conv_epilogue_impl(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                          int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                          float (&s2)[BN / WN / 16][4], const ConvGeo& geo, int wave_base,
                                                          BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& own,
                                                          const AccFrag<BN / WN / 16, BM / WM / 16>* accf,
                                                          BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& pr,
                                                          const ResFrag<BN / WN / 16, BM / WM / 16>* resf = nullptr,
                                                          const float* lds_coef = nullptr) {
    constexpr int WTM = BM / WM;
    constexpr int WTN = BN / WN;
    constexpr int MI = WTM / 16;
    constexpr int NI = WTN / 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (tid >> 6) - wave_base;      // wave_base: first wave of this MFMA wave group (kernels with several groups)
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const int m0 = mtile * BM;
    const PixSplit ps = pix_split_init(p, geo);
    const bool do_eval = EPM < 0 ? (p.ep_scale != nullptr) : (EPM == 3);
    const bool do_acc = EPM < 0 ? (p.accumulate != 0) : (EPM == 2 || EPM == 7 || EPM == 8);
    const bool do_stats = EPM < 0 ? (p.stats != nullptr) : (EPM == 1 || EPM == 4);
    constexpr bool do_store = EPM != 4;
    constexpr bool do_bst = EPM == 6 || EPM == 7 || EPM == 8;
    constexpr bool do_pair = EPM == 8;
    // acc[a][b][j] = out[channel n0 + wn*WTN + a*16 + 4*fq + j][pixel m0 + wm*WTM + b*16 + fr]

    // eval epilogue: this lane's 4 * NI scale / shift values once, not once per pixel group (the stores in between keep hipcc
    // from merging the reloads)
    constexpr int MB = VPD_BST_MB(MI);
    float4 esc[NI], esh[NI];
    if (do_eval) {
        // lds_coef (persistent kernels): the block's BN scale / shift values (scale[BN], shift[BN]) were put into LDS once, at its
        // start -- a block keeps its channel tile, and 2 * NI global loads (L2 hits) stood at the head of every tile's epilogue.
        // Measured small: apply forward 360.3 vs 359.3 k crops/s same box (the loads overlapped the epilogue's address set-up)
        if (lds_coef) {
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                esc[a] = *reinterpret_cast<const float4*>(lds_coef + wn * WTN + a * 16 + 4 * fq);
                esh[a] = *reinterpret_cast<const float4*>(lds_coef + BN + wn * WTN + a * 16 + 4 * fq);
            }
        } else {
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                esc[a] = *reinterpret_cast<const float4*>(p.ep_scale + n0 + wn * WTN + a * 16 + 4 * fq);
                esh[a] = *reinterpret_cast<const float4*>(p.ep_shift + n0 + wn * WTN + a * 16 + 4 * fq);
            }
        }
    }
    const int nw = n0 + wn * WTN;                 // first channel of this wave's range
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        // (PRE: the caller fetched the first MB groups; the second half of an 8-group tile is fetched here)
        if (do_bst && b % MB == 0 && (!PRE || b > 0)) conv_bst_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, own, wave_base, b);
        if (do_pair && b % MB == 0 && (!PRE || b > 0)) conv_bst2_prefetch<BM, BN, WM, WN>(p, mtile, n0, geo, pr, b);
        const int m = m0 + wm * WTM + b * 16 + fr;
        const bool valid = m < geo.M;
        const int mc = valid ? m : geo.M - 1;
        size_t yoff, roff = 0;
        if (ps.dense && !(do_eval && p.res)) yoff = (size_t)mc * p.yC;
        else {
            int bi, yy, xx;
            pix_split(ps, mc, bi, yy, xx);
            yoff = ((size_t)(bi * p.yHp + yy * p.osub + geo.oph + p.ypad) * p.yWp + (xx * p.osub + geo.opw + p.ypad)) * p.yC;
            if (do_eval && p.res) roff = ((size_t)(bi * p.rHp + yy + p.rpad) * p.rWp + (xx + p.rpad)) * p.rC;
        }
        // accumulate with a ReLU bit map: ONE load per pixel covers this wave's WTN channels (WTN / 8 bytes, 8-byte aligned
        // for WTN = 64; a byte load per 4-channel group doubled the epilogue's memory instructions: measured +79 us per step)
        unsigned long long mbits = 0;
        if (APRE) mbits = accf->bits[b];
        if (!APRE && do_acc && p.acc_mask && valid) {
            const unsigned char* mp = p.acc_mask + (size_t)mc * (p.yC >> 3) + (nw >> 3);
            if (WTN == 64) mbits = *reinterpret_cast<const unsigned long long*>(mp);
            else if (WTN == 32) mbits = *reinterpret_cast<const unsigned*>(mp);
            else mbits = *reinterpret_cast<const unsigned short*>(mp);
        }
        const BstFrag<NI, MB>& bf = own;
        const unsigned long long bbits = do_bst ? bf.bits[b % MB] : 0ull;      // ReLU bits of the consuming BatchNorm's activation
        bf16_t* const dpix = p.y + yoff + nw;
        // this pixel's fragments of the other tensors: 16 bytes wide as loaded, brought back to the MFMA layout pair by pair
        // (one pair's worth of temporaries live at a time: the eight-wave tile has 168 registers)
        FragRow<NI> rest, oldt;
        if (do_eval && p.res && !resf) frag_row_load<NI>(rest, p.res + roff + nw, fq);
        if (do_acc && !APRE) frag_row_load<NI>(oldt, dpix, fq);      // (rows beyond M re-read row M - 1: in bounds, never stored)
        const FragRow<NI>& resr = (do_eval && p.res && resf) ? resf->r[b] : rest;
        const FragRow<NI>& oldr = APRE ? accf->old[b] : oldt;
#pragma unroll
        for (int a0 = 0; a0 < NI; a0 += 2) {
            const bool pair = a0 + 1 < NI;
            uint2 resv[2], oldv[2], zv[2], z2v[2], ovs[2];
            auto unpack2 = [&](const FragRow<NI>& f, uint2 (&v)[2]) __attribute__((always_inline)) {
                v[0] = uint2{f.q[a0 / 2].x, f.q[a0 / 2].y};
                v[1] = uint2{f.q[a0 / 2].z, f.q[a0 / 2].w};
                if (pair) frag_pair_swap(v[0], v[1]);
            };
            if (do_eval && p.res) unpack2(resr, resv);
            if (do_acc) unpack2(oldr, oldv);
            if (do_bst) unpack2(bf.z[b % MB], zv);
            if (do_pair) unpack2(pr.z2[b % MB], z2v);
#pragma unroll
            for (int ai = 0; ai < 2; ++ai) {
                const int a = a0 + ai;
                if (a >= NI) break;
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                if (do_eval) {
                    const float4 sc = esc[a];
                    const float4 sh = esh[a];
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                    v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    if (p.res) {
                        const uint2 rv = resv[ai];
                        v[0] += bf2f((unsigned short)(rv.x & 0xffff)); v[1] += bf2f((unsigned short)(rv.x >> 16));
                        v[2] += bf2f((unsigned short)(rv.y & 0xffff)); v[3] += bf2f((unsigned short)(rv.y >> 16));
                    }
                    if (p.ep_relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                }
                if (do_acc) {
                    const uint2 ov = oldv[ai];
                    float o0 = bf2f((unsigned short)(ov.x & 0xffff)), o1 = bf2f((unsigned short)(ov.x >> 16));
                    float o2 = bf2f((unsigned short)(ov.y & 0xffff)), o3 = bf2f((unsigned short)(ov.y >> 16));
                    if (APRE || p.acc_mask) {      // y dense [M][yC]: channel n0 + wn*WTN + k of pixel m = bit k of mbits (APRE: all ones without a mask)
                        const unsigned bits = (unsigned)(mbits >> (a * 16 + 4 * fq));
                        o0 = (bits & 1u) ? o0 : 0.f; o1 = (bits & 2u) ? o1 : 0.f;
                        o2 = (bits & 4u) ? o2 : 0.f; o3 = (bits & 8u) ? o3 : 0.f;
                    }
                    v[0] += o0; v[1] += o1; v[2] += o2; v[3] += o3;
                }
                uint2 ov;
                ov.x = pack2bf(v[0], v[1]);
                ov.y = pack2bf(v[2], v[3]);
                ovs[ai] = ov;
                if (do_bst && valid) {
                    // g = stored d * mask; sum g and sum g * z (the BatchNorm backward's finalize turns the latter into sum g * xhat)
                    const uint2 zr = zv[ai];
                    const unsigned bits = (unsigned)(bbits >> (a * 16 + 4 * fq));
                    const float q0 = (bits & 1u) ? bf2f((unsigned short)(ov.x & 0xffff)) : 0.f;
                    const float q1 = (bits & 2u) ? bf2f((unsigned short)(ov.x >> 16)) : 0.f;
                    const float q2 = (bits & 4u) ? bf2f((unsigned short)(ov.y & 0xffff)) : 0.f;
                    const float q3 = (bits & 8u) ? bf2f((unsigned short)(ov.y >> 16)) : 0.f;
                    s1[a][0] += q0; s2[a][0] += q0 * bf2f((unsigned short)(zr.x & 0xffff));
                    s1[a][1] += q1; s2[a][1] += q1 * bf2f((unsigned short)(zr.x >> 16));
                    s1[a][2] += q2; s2[a][2] += q2 * bf2f((unsigned short)(zr.y & 0xffff));
                    s1[a][3] += q3; s2[a][3] += q3 * bf2f((unsigned short)(zr.y >> 16));
                    if (do_pair) {
                        const uint2 z2 = z2v[ai];
                        pr.s3[a][0] += q0 * bf2f((unsigned short)(z2.x & 0xffff));
                        pr.s3[a][1] += q1 * bf2f((unsigned short)(z2.x >> 16));
                        pr.s3[a][2] += q2 * bf2f((unsigned short)(z2.y & 0xffff));
                        pr.s3[a][3] += q3 * bf2f((unsigned short)(z2.y >> 16));
                    }
                }
                if (do_stats && valid) {
                    // statistics are taken over the bf16-rounded values actually stored
                    const float q0 = bf2f((unsigned short)(ov.x & 0xffff)), q1 = bf2f((unsigned short)(ov.x >> 16));
                    const float q2 = bf2f((unsigned short)(ov.y & 0xffff)), q3 = bf2f((unsigned short)(ov.y >> 16));
                    s1[a][0] += q0; s2[a][0] += q0 * q0;
                    s1[a][1] += q1; s2[a][1] += q1 * q1;
                    s1[a][2] += q2; s2[a][2] += q2 * q2;
                    s1[a][3] += q3; s2[a][3] += q3 * q3;
                }
            }
            // the pair leaves 16 bytes wide (64 contiguous bytes per pixel and instruction); a trailing odd group 8 bytes wide
            if (!do_store) {
            } else if (pair) {
                frag_pair_swap(ovs[0], ovs[1]);
                if (valid) vpd_store16<VPD_CP_EPI>(dpix + frag_pair_chan(a0, fq), uint4{ovs[0].x, ovs[0].y, ovs[1].x, ovs[1].y});
            } else if (valid) {
                *reinterpret_cast<uint2*>(dpix + a0 * 16 + 4 * fq) = ovs[0];
            }
        }
    }

}

template <int BM, int BN, int WM, int WN, int EPM = -1>
static __device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                     int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                     float (&s2)[BN / WN / 16][4], const ConvGeo& geo, int wave_base = 0,
                                                     const float* lds_coef = nullptr) {
    BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)> own;
    BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)> pr;
    conv_epilogue_impl<BM, BN, WM, WN, EPM, false>(p, acc, mtile, n0, s1, s2, geo, wave_base, own, nullptr, pr, nullptr, lds_coef);
}
// ... eval epilogue with the residual fragments fetched by the caller (conv_res_prefetch; p.res != null)
template <int BM, int BN, int WM, int WN, int EPM>
static __device__ __forceinline__ void conv_epilogue_res_pre(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                             int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                             float (&s2)[BN / WN / 16][4], const ConvGeo& geo,
                                                             const ResFrag<BN / WN / 16, BM / WM / 16>& resf, int wave_base = 0,
                                                             const float* lds_coef = nullptr) {
    BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)> own;
    BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)> pr;
    conv_epilogue_impl<BM, BN, WM, WN, EPM, false>(p, acc, mtile, n0, s1, s2, geo, wave_base, own, nullptr, pr, &resf, lds_coef);
}
// ... with the consuming BatchNorm's z fragments / mask bits fetched by the caller (EPM 6 / 7 only)
template <int BM, int BN, int WM, int WN, int EPM>
static __device__ __forceinline__ void conv_epilogue_pre(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                         int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                         float (&s2)[BN / WN / 16][4], const ConvGeo& geo,
                                                         BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& bst) {
    BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)> pr;
    conv_epilogue_impl<BM, BN, WM, WN, EPM, true>(p, acc, mtile, n0, s1, s2, geo, 0, bst, nullptr, pr);
}
// ... mode 8: both BatchNorms' fragments fetched by the caller; pr.s3 collects the second BatchNorm's sum g * z2
template <int BM, int BN, int WM, int WN, int EPM>
static __device__ __forceinline__ void conv_epilogue_pre2(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                          int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                          float (&s2)[BN / WN / 16][4], const ConvGeo& geo,
                                                          BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& bst,
                                                          BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& pr) {
    conv_epilogue_impl<BM, BN, WM, WN, EPM, true>(p, acc, mtile, n0, s1, s2, geo, 0, bst, nullptr, pr);
}

// ... accumulate modes of the persistent kernel: old values (and, EPM 7, the BatchNorm fragments) fetched by the caller
template <int BM, int BN, int WM, int WN, int EPM>
static __device__ __forceinline__ void conv_epilogue_acc_pre(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                             int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                             float (&s2)[BN / WN / 16][4], const ConvGeo& geo,
                                                             BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& bst,
                                                             const AccFrag<BN / WN / 16, BM / WM / 16>& accf) {
    BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)> pr;
    conv_epilogue_impl<BM, BN, WM, WN, EPM, EPM == 7, true>(p, acc, mtile, n0, s1, s2, geo, 0, bst, &accf, pr);
}
// ... mode 8: the old values, both BatchNorms' fragments fetched by the caller
template <int BM, int BN, int WM, int WN, int EPM>
static __device__ __forceinline__ void conv_epilogue_acc_pre2(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                              int mtile, int n0, float (&s1)[BN / WN / 16][4],
                                                              float (&s2)[BN / WN / 16][4], const ConvGeo& geo,
                                                              BstFrag<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& bst,
                                                              BstPair<BN / WN / 16, VPD_BST_MB(BM / WM / 16)>& pr,
                                                              const AccFrag<BN / WN / 16, BM / WM / 16>& accf) {
    conv_epilogue_impl<BM, BN, WM, WN, EPM, true, true>(p, acc, mtile, n0, s1, s2, geo, 0, bst, &accf, pr);
}

// Reduce the per-lane partial statistics over the 16 pixel lanes and the WM pixel-waves, then ONE atomic per
// channel and block into accumulator row `row`.  Contains a block barrier: call from all MFMA-layout threads.
template <int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void conv_stats_flush(const ConvParams& p, float (&s1)[BN / WN / 16][4],
                                                        float (&s2)[BN / WN / 16][4], int row, int n0,
                                                        unsigned char* smem, double* rows_override = nullptr) {
    double* const rows = rows_override ? rows_override : p.stats;
    constexpr int WTN = BN / WN;
    constexpr int NI = WTN / 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    if (rows) {
        // reduce over the 16 pixel lanes, then over the WM pixel-waves through LDS
        float* red = reinterpret_cast<float*>(smem);      // [WM][2][BN] (staging LDS is free now)
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = row16_sum(s1[a][j]), v = row16_sum(s2[a][j]);
                if (fr == 0) {
                    const int c = wn * WTN + a * 16 + 4 * fq + j;
                    red[(wm * 2 + 0) * BN + c] = u;
                    red[(wm * 2 + 1) * BN + c] = v;
                }
            }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN;
            const int c = tid - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + c];
            // VPD_STAT_ROWS accumulator rows spread the atomic traffic; bn_finalize sums and re-zeroes them.  The rows
            // are fp64: the order in which blocks arrive then perturbs a sum at the 1e-16 level, far below the fp32
            // rounding of mean / rstd, so the statistics (and with them the whole step) repeat run to run
            const int rmask = (p.stat_rows ? p.stat_rows : VPD_STAT_ROWS) - 1;
            atomicAdd(&rows[((size_t)(row & rmask) * 2 + which) * p.Co + n0 + c], (double)t);
        }
    }
}
