// Shared device helpers for the VPD student kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

// Element type of activations / packed weights / activation gradients.  The library is built twice from these sources
// (vpd_amd/csrc/Makefile): libvpdhip.so with bf16 elements (training and inference), libvpdhip_f16.so with -DVPD_ELEM_F16 = IEEE fp16
// elements for INFERENCE (apply_vpd_model.py --dtype fp16): the reference's own GPU precision (fp16 autocast, train_vpd_model.py:79),
// the same MFMA rate (v_mfma_f32_16x16x32_f16), 8x finer rounding (11 significant bits against 8).  The type NAMES stay bf16_t /
// bf16x8 in both builds; every conversion goes through bf2f / f2bf / pack2bf below and every matrix instruction through VPD_MFMA16.
#ifdef VPD_ELEM_F16
typedef _Float16 bf16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
#define VPD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define VPD_ELEM_NAME "fp16"
#else
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#define VPD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define VPD_ELEM_NAME "bf16"
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define VPD_WAVE 64

// CU budget of the persistent grids (round 6): the device's CU count minus VPD_RESERVE_CUS (default 0) -- every kernel that sizes its
// grid by "one block per CU" (conv3x3_pws / c64 / stem / streaming 1x1 kernels, the persistent weight-gradient launch, the grid-barrier
// BatchNorm launches) asks here, so that R CUs stay free of LDS-heavy blocks for whatever co-runs with the step (RCCL's kernels under
// data parallelism).  vpd_cu_budget_override(): a launch sequence's own budget (the weight-gradient side stream).
inline int& vpd_cu_budget_override() { static thread_local int v = 0; return v; }
inline int vpd_cu_budget() {
    if (vpd_cu_budget_override() > 0) return vpd_cu_budget_override();
    static const int n = [] {
        int dev = 0, cu = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        const char* r = getenv("VPD_RESERVE_CUS");
        int res = r ? atoi(r) : 0;
        if (res < 0) res = 0;
        res = (res + 7) & ~7;                       // whole octets: one CU of every XCD per 8
        if (cu - res < 8) res = cu > 8 ? cu - 8 : 0;
        return cu - res;
    }();
    return n;
}
// Diagnostics (tools/bench_conv.py, VPD_ABLATE env) are compiled in only with -DVPD_ENABLE_ABLATE: even never-taken
// branches cost registers and issue slots in the hot loops.
#ifdef VPD_ENABLE_ABLATE
#define VPD_ABL(p, bit) ((p).ablate & (bit))
#else
#define VPD_ABL(p, bit) 0
#endif
// per-channel statistics are accumulated into this many [2][C] fp32 rows (row = producer block % rows)
#define VPD_STAT_ROWS 16
// the fused BatchNorm passes (bn.hip) give every BatchNorm its OWN rows, fewer of them: every consumer block sums them itself
#define VPD_FUSED_ROWS 4

static __device__ __forceinline__ float bf2f(unsigned short u) {
#ifdef VPD_ELEM_F16
    return (float)__builtin_bit_cast(_Float16, u);
#else
    return __builtin_bit_cast(float, ((unsigned)u) << 16);
#endif
}
static __device__ __forceinline__ unsigned short f2bf(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    return __builtin_bit_cast(unsigned short, (bf16_t)f);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#ifdef VPD_ELEM_F16
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;      // (v_cvt_pk_f16_f32: RNE)
#else
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#endif
static __device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    // ONE v_cvt_pk_bf16_f32 for the pair (RNE, NaN stays NaN).  Written as two scalar casts + shift + or, hipcc emitted two
    // half-empty conversions, a shift and an or per dword: four instructions where one does -- in every epilogue and BatchNorm pass
    const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}
static __device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf2f((unsigned short)(v.x & 0xffff)); f[1] = bf2f((unsigned short)(v.x >> 16));
    f[2] = bf2f((unsigned short)(v.y & 0xffff)); f[3] = bf2f((unsigned short)(v.y >> 16));
    f[4] = bf2f((unsigned short)(v.z & 0xffff)); f[5] = bf2f((unsigned short)(v.z >> 16));
    f[6] = bf2f((unsigned short)(v.w & 0xffff)); f[7] = bf2f((unsigned short)(v.w >> 16));
}
static __device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}
// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15); every lane gets the total.  Four v_add_f32_dpp
// (xor 1, xor 2 inside the quad, then the mirrored half and the mirrored row) instead of four ds_bpermute + add.
template <int CTRL>
static __device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);      // row_half_mirror
    v += dpp_move<0x140>(v);      // row_mirror
    return v;
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exact m / d for 0 <= m < 2^21 and 1 <= d <= 2^21 by one multiplication with the float reciprocal (rcp = 1.0f / d, IEEE):
// (m + 0.5) / d lies at least 0.5 / d away from an integer and the two roundings move it by < q * 1.2e-7.  hipcc's integer
// division is ~45 VALU instructions; a tile's pixel -> (image, row, column) split used to cost the conv epilogues ~1,600
// cycles per tile and the persistent kernels ~2,000 cycles of set-up (profiles/r03_pws_stamps_timeline.txt).
#define VPD_FDIV_MAX (1 << 21)
static __device__ __forceinline__ int vpd_fdiv(int m, float rcp) { return (int)(((float)m + 0.5f) * rcp); }

// Activation tensor descriptor: bf16 NHWC with a zero border of `pad` pixels.
// Element offset of interior pixel (b, y, x): ((b*Hp + y + pad)*Wp + x + pad)*C.
struct TensorDesc {
    int H, W, C, pad;
};
static __host__ __device__ __forceinline__ int td_hp(const TensorDesc& t) { return t.H + 2 * t.pad; }
static __host__ __device__ __forceinline__ int td_wp(const TensorDesc& t) { return t.W + 2 * t.pad; }

// Tap set of one conv launch: a product of an arithmetic row list and an
// arithmetic column list (no per-tap tables, so nothing is indexed dynamically
// out of the kernel arguments).  Tap (ir, ic), ir < nr, ic < nc:
//   input offset  dy = dy0 + ir*dys, dx = dx0 + ic*dxs   (padded coordinates)
//   weight slice  w0 + ir*wrs + ic*wcs
struct TapSet {
    int nr, nc;
    int dy0, dys, dx0, dxs;
    int w0, wrs, wcs;
};

// Output sub-grid of one launch (or of one parity class of a merged stride-2 data-gradient launch).
struct ConvGeo {
    int Hs, Ws, M, oph, opw;
};
struct ConvClass {
    ConvGeo geo;
    TapSet taps;
};

// One implicit-GEMM convolution launch (forward conv, or data-gradient conv).
// Output pixels are enumerated on a sub-grid (Hs x Ws per image); output pixel
// (y, x) of the sub-grid lands at (y*osub+oph, x*osub+opw) of tensor Y and
// gathers, for tap i, input pixel (y*istr + dy[i], x*istr + dx[i]) in the
// PADDED coordinate space of X.  All gathers are in bounds by construction of
// the zero borders ("data layout").
struct ConvParams {
    const bf16_t* x; int xHp, xWp, xC;          // padded dims, pixel stride (elements)
    const bf16_t* w;                            // [tap][Co][Kc] bf16
    bf16_t* y; int yHp, yWp, yC, ypad;
    double* stats;                              // [stat_rows][2][Co] accumulators of sum / sum of squares (fp64 atomics), or null
    int stat_rows;                              // power of two; 0 = VPD_STAT_ROWS
    const float* ep_scale; const float* ep_shift;   // eval epilogue: y = relu?(scale*acc+shift(+res))
    const bf16_t* res; int rHp, rWp, rC, rpad;  // residual for the eval epilogue (padded act) or null
    int ep_relu;
    int N, Hs, Ws, osub, oph, opw, istr;
    int Kc, Co;
    int M;                                      // N*Hs*Ws
    int accumulate;                             // y += result
    // accumulate only: the old value of y is first multiplied by this ReLU mask ([M][yC/8] bytes, one bit per element; y dense):
    // y holds d(block output) and the identity path carries d * [out > 0] -- the mask the BatchNorm backward used too
    const unsigned char* acc_mask;
    // data gradients only: also take the sums of the BatchNorm backward that consumes this launch's output d -- g = d * mask
    // (mask: ReLU bit map [M][yC/8] of that BatchNorm's activation), sum g and sum g * z (z: the BatchNorm's dense input
    // [M][yC]) per channel into `stats` (that BatchNorm's own rows).  The BatchNorm backward is then finalize + apply only.
    const bf16_t* bst_z; const unsigned char* bst_mask;
    // ... and of a SECOND BatchNorm fed with the same g (the 1x1 branch of a down-sampling block: same gradient, same ReLU
    // mask): sum g * z2 (and sum g again) into `stats2`.  conv3x3_ws_kernel, accumulate mode only (epilogue mode 8).
    const bf16_t* bst_z2; double* stats2;
    unsigned* err;                              // -DPWS_STAMPS builds: 16 x u64 s_memtime stamps per block (conv_pws.h); else unused
    int ablate;                                 // diagnostics only (VPD_ABLATE env): 1 skip weight loads, 2 skip MFMAs, 4 skip halo loads
    TapSet taps;
    // gather kernel only: extra parity classes of a stride-2 data gradient, selected by blockIdx.z (class 0 is
    // described by the fields above); ncls == 0 or 1 means a single class
    int ncls;
    ConvClass cls[3];
    // gather kernel only: a SECOND convolution of the same input with the same output geometry and channel count in the
    // same launch (a BasicBlock's 1x1 down-sampling branch beside its first 3x3): the blocks with blockIdx.y >= alt_y0
    // compute it -- own weights, taps, output, statistics rows and eval epilogue
    const bf16_t* alt_w; bf16_t* alt_y; double* alt_stats; int alt_y0; TapSet alt_taps;
    const float* alt_ep_scale; const float* alt_ep_shift; int alt_ep_relu;
    // gather kernel only: extra K-steps of class 0 from a second input tensor of the same geometry and channel count
    // (the data gradient of the down-sampling branch lands on the even-even pixels, i.e. on class 0 of the 3x3's)
    const bf16_t* x2; const bf16_t* w2; int Kc2;
    // stem kernel, eval only (epilogue mode 9): scale / shift / ReLU (ep_*) and the 3x3 stride-2 max-pool applied in the
    // epilogue; the pooled activation goes to `pool_y` (padded by 1, [N][Hs/2 + 2][Ws/2 + 2][64]) and y is not written
    bf16_t* pool_y;
};

// One channel of a train-mode BatchNorm from its fp64 accumulator rows [VPD_FUSED_ROWS][2][C] (sum z, sum z^2): mean, 1 / std,
// scale = gamma / std, shift = beta - mean * scale (bn_fwd_fused_kernel)
static __device__ __forceinline__ void bn_finalize_channel(const double* rows, int C, int ch, float count, float eps,
                                                           float gamma, float beta, float* mu_o, float* r_o, float* sc_o,
                                                           float* sh_o, double* var_o) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < VPD_FUSED_ROWS; ++t) {
        s1 += rows[((size_t)t * 2) * C + ch];
        s2 += rows[((size_t)t * 2 + 1) * C + ch];
    }
    // (one fp64 division per thread, not two per channel plus an fp64 square root: every block of the fused BatchNorm launch
    //  runs this in front of its first byte of real work; the cancellation-prone part, E[z^2] - mean^2, stays in fp64,
    //  1 / sqrt is v_rsq_f32 -- 1 ulp)
    const double inv = 1.0 / (double)count;
    const double mu = s1 * inv;
    double var = s2 * inv - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const float r = __builtin_amdgcn_rsqf((float)(var + (double)eps));
    const float sc = gamma * r;
    *mu_o = (float)mu; *r_o = r; *sc_o = sc; *sh_o = beta - (float)mu * sc; *var_o = var;
}

// One channel of a BatchNorm backward whose sums (sum g, sum g * z; g = dy * ReLU mask) sit in its fp64 rows [VPD_FUSED_ROWS][2][C]:
// dz = A g + B z + D with A = gamma rstd, B = -A rstd mean(g xhat), D = -A mean(g) - B mean; dgamma = sum g xhat, dbeta = sum g
// (bn_bwd_apply_fused_kernel, conv1x1_bn_stream_kernel).
static __device__ __forceinline__ void bn_bwd_apply_coef(const double* rows, int C, int ch, float count, float gamma, float mean,
                                                         float rstd, float* A, float* B, float* D, float* dgamma, float* dbeta,
                                                         bool write, int oi = -1) {
    if (oi < 0) oi = ch;                                           // A / B / D are indexed by oi (default: the channel)
    double s1 = 0.0, sz = 0.0;
#pragma unroll
    for (int t = 0; t < VPD_FUSED_ROWS; ++t) {
        s1 += rows[((size_t)t * 2) * C + ch];
        sz += rows[((size_t)t * 2 + 1) * C + ch];
    }
    const double mu = (double)mean, rs = (double)rstd;
    const double sx = (sz - mu * s1) * rs;                         // sum g * xhat
    const double a = (double)gamma * rs;
    const double b = -a * rs * (sx / (double)count);
    A[oi] = (float)a; B[oi] = (float)b;
    D[oi] = (float)(-a * (s1 / (double)count) - b * mu);
    if (write) { dbeta[ch] = (float)s1; dgamma[ch] = (float)sx; }
}
// Pixel-chunk split of the halo weight-gradient kernel (64-pixel chunks): shared by the launcher and by the
// bucket-level reduce so both agree on the number of slabs without a host->device hand-off.
static __host__ __device__ __forceinline__ int vpd_wgrad_split(int M, int Co, int Kc, int* cpb_out) {
    const int tiles = (Co / 64) * (Kc / 64);
    const int nchunks = (M + 63) / 64;
    int ksplit = 256 / tiles;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > nchunks) ksplit = nchunks;
    const int cpb = (nchunks + ksplit - 1) / ksplit;
    if (cpb_out) *cpb_out = cpb;
    return (nchunks + cpb - 1) / cpb;
}

// ---------------------------------------------------------------------------
// Cache policy of the step's big streams (round 5; profiles/r05_floor_probe.txt).  Purely a hint: every flavour stores / loads the
// same bytes.  Store flavours: 0 plain, 1 `sc1` (write-through, line dropped from the XCD's L2), 2 `sc0 sc1`, 3 `nt`; load
// flavours 0 plain, 3 `nt`.  MEASURED NEUTRAL (sc1 epilogue / BatchNorm stores + nt layer1 stores + nt BatchNorm loads: +0.6 % on a
// box where round 4's library runs 75.9 k crops/s, -0.1 % on one where it runs 73.3 k; all-nt: -0.3 ... -1 %), so every site
// defaults to plain; the per-site macros stay as the A/B hook (-DVPD_CP_xxx=n through tools/build_variant_lib.sh):
//   VPD_CP_EPI  conv epilogue outputs      VPD_CP_BNF / _BNF64  BatchNorm-forward outputs (C > 64 / C = 64)
//   VPD_CP_BNB / _BNB64  BatchNorm-backward dz      VPD_CP_STEM stem pooling outputs      VPD_CL_BN  BatchNorm operand loads
// ---------------------------------------------------------------------------
#ifndef VPD_CP_ALL
#define VPD_CP_ALL 0
#endif
#ifndef VPD_CP_EPI
#define VPD_CP_EPI VPD_CP_ALL
#endif
#ifndef VPD_CP_BNF
#define VPD_CP_BNF VPD_CP_ALL
#endif
#ifndef VPD_CP_BNB
#define VPD_CP_BNB VPD_CP_ALL
#endif
#ifndef VPD_CP_BNF64
#define VPD_CP_BNF64 VPD_CP_BNF
#endif
#ifndef VPD_CP_BNB64
#define VPD_CP_BNB64 VPD_CP_BNB
#endif
#ifndef VPD_CP_STEM
#define VPD_CP_STEM VPD_CP_ALL
#endif
#ifndef VPD_CL_BN
#define VPD_CL_BN 0
#endif
typedef __attribute__((ext_vector_type(4))) unsigned int vpd_u32x4_cp;
template <int CP> static __device__ __forceinline__ void vpd_store16(void* p, const uint4& v) {
    if (CP == 0) { *reinterpret_cast<uint4*>(p) = v; return; }
    const vpd_u32x4_cp w = {v.x, v.y, v.z, v.w};
    // (s_nop 1: a store of more than 8 bytes followed by a vector-ALU write of its data registers needs wait states that hipcc's
    //  hazard recognizer inserts for its own instructions but not behind inline assembly -- without it the stored bits are wrong,
    //  and a step with wrong bits ran 5 % FASTER on power-limited boxes, which first read as a win of the write-through stores)
    if (CP == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
    if (CP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
    if (CP == 3) __builtin_nontemporal_store(w, reinterpret_cast<vpd_u32x4_cp*>(p));
}
template <int CP> static __device__ __forceinline__ uint4 vpd_load16(const void* p) {
    if (CP == 0) return *reinterpret_cast<const uint4*>(p);
    const vpd_u32x4_cp w = __builtin_nontemporal_load(reinterpret_cast<const vpd_u32x4_cp*>(p));
    return uint4{w.x, w.y, w.z, w.w};
}

// One weight-gradient launch: dw[tap][co][kc] += sum_m dz[m][co] * x[gather(m,tap)][kc]
struct WgradParams {
    const bf16_t* dz; int dzHp, dzWp, dzC, dzpad;
    const bf16_t* x; int xHp, xWp, xC;
    float* dw;                                  // [ntaps][Co][Kc] fp32 (atomically accumulated)
    float* slab;                                // fp32 scratch for split partials (vpd_wgrad_slab_bytes()) or null
    int defer_reduce;                           // leave the partials in the slab (a later bucket-level kernel sums them)
    int ablate;                                 // diagnostics (VPD_ABLATE): 1 no LDS-DMA, 2 no MFMA loop, 8 no slab store, 16 no reduce
    int N, Hs, Ws, istr;
    int Kc, Co;
    int M;
    int chunks_per_block;                       // 128-pixel chunks handled by one block
    TapSet taps;                                // w0 + ir*wrs + ic*wcs indexes the dw slice
    // halo kernel, set by its launchers: a 1x1 (pad 0) convolution runs as the CENTRE tap of the 3x3 halo form -- taps
    // holds the 3x3 geometry, only tap 4 is accumulated and dw / the slab have ONE slice
    int one_by_one;
    int prefer_halo_1x1;                        // caller's wish for a 1x1 conv: the halo kernel (no atomics) instead of conv_wgrad_kernel
    // 1x1 tasks of conv_wgrad128_persistent_kernel only: the problem is stated with the operands SWAPPED (dz := the convolution's
    // input activation, x := its dz; Co := input channels, Kc := output channels) because the convolution has fewer than 128 output
    // channels but >= 128 input channels; the tile is stored transposed, dw[Kc][Co] = the convolution's own [Co][Ci] layout
    int transposed;
};

// ---------------------------------------------------------------------------
// Timing hook (bench.py's roofline): while a TimeScope is armed, the NEXT matrix-kernel launch of this thread carries
// the scope's start / stop events in its own dispatch packet (hipExtLaunchKernelGGL), so their elapsed time is the
// kernel's duration as a profiler reports it -- an event pair recorded around a launch also contains ~5 us of
// event-packet processing.  Unarmed (always, outside the instrumented steps) it is a plain launch.
// ---------------------------------------------------------------------------
struct VpdLaunchEvents { hipEvent_t start, stop; };
inline VpdLaunchEvents& vpd_launch_events() {
    static thread_local VpdLaunchEvents e = {nullptr, nullptr};
    return e;
}
#define VPD_LAUNCH(kernel, grid, block, lds, stream, ...)                                                          \
    do {                                                                                                          \
        VpdLaunchEvents& le_ = vpd_launch_events();                                                               \
        if (le_.start) {                                                                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, le_.start, le_.stop, 0, __VA_ARGS__);         \
            le_.start = nullptr;                                                                                  \
        } else {                                                                                                  \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                    \
        }                                                                                                         \
    } while (0)
