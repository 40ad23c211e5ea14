// Weight-gradient convolution on bf16 MFMA (gfx950).
//
//   dw[tap][co][kc] += sum_m dz[m][co] * x[gather(m, tap)][kc]      (fp32)
//
// GEMM view: rows = co, cols = kc (input channels of one tap), K = output pixels.
// Both operands are stored channel-contiguous (NHWC) while the reduction runs
// over pixels, so both MFMA fragments are fetched from LDS with the CDNA4
// transposing read ds_read_b64_tr_b16 (no software transpose).  LDS tiles are
// [128 pixels][64 channels] bf16 (128-B rows); 32-B granules are XOR-swizzled
// so the 8 rows a 32-lane half touches land on 8 distinct bank groups.
//
// One block = one (co tile 64, kc tile 64, tap) over `chunks_per_block`
// 128-pixel chunks; the 4 waves split every chunk's pixels (32 each) and their
// 64x64 partial tiles are summed through LDS before one fp32 atomic per
// element.
#include "common.h"

static __device__ __forceinline__ int wg_swz(int r, int c16) {
    // element offset of 16-B piece c16 (0..7) of row r in a [128][64] bf16 tile
    const int f = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    return r * 64 + ((((c16 >> 1) ^ f) << 4) | ((c16 & 1) << 3));
}

static __device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int rbase, int ctile, int lane) {
    // MFMA 16x16x32 operand with the K index on LDS rows:
    //   element j of lane l = tile[row rbase + 8*(l>>4) + j][col ctile*16 + (l&15)]
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int col = ctile * 16 + 4 * pp;               // 4 consecutive columns = 8 bytes
    const int r0 = rbase + 8 * g + q;
    const int r1 = r0 + 4;
    const int f0 = ((r0 >> 1) & 1) | (((r0 >> 3) & 1) << 1);
    const int f1 = ((r1 >> 1) & 1) | (((r1 >> 3) & 1) << 1);
    const int o0 = r0 * 64 + ((((col >> 4) ^ f0) << 4) | (col & 15));
    const int o1 = r1 * 64 + ((((col >> 4) ^ f1) << 4) | (col & 15));
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + o1));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sZ = reinterpret_cast<bf16_t*>(smem);        // [2][128*64]  dz tile
    bf16_t* sX = sZ + 2 * 128 * 64;                      // [2][128*64]  x tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int piece = tid & 7;
    const int row0 = tid >> 3;

    const int kct = p.Kc >> 6;
    const int cot = p.Co >> 6;
    int bx = blockIdx.x;
    const int tap = bx / (cot * kct);
    bx -= tap * cot * kct;
    const int co0 = (bx / kct) * 64;
    const int kc0 = (bx % kct) * 64;

    const int HW = p.Hs * p.Ws;
    const int chunk_begin = blockIdx.y * p.chunks_per_block;
    const int nchunks_total = (p.M + 127) >> 7;
    int chunk_end = chunk_begin + p.chunks_per_block;
    chunk_end = chunk_end < nchunks_total ? chunk_end : nchunks_total;
    if (chunk_begin >= chunk_end) return;

    const int tir = tap / p.taps.nc;
    const int tic = tap - tir * p.taps.nc;
    const int toff = ((p.taps.dy0 + tir * p.taps.dys) * p.xWp + (p.taps.dx0 + tic * p.taps.dxs)) * p.xC + kc0 + piece * 8;
    const int wsl = p.taps.w0 + tir * p.taps.wrs + tic * p.taps.wcs;

    u32x4 rz[4], rx[4];
    auto load_chunk = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = ch * 128 + row0 + 32 * i;
            if (m < p.M) {
                const int b = m / HW;
                const int r = m - b * HW;
                const int yy = r / p.Ws;
                const int xx = r - yy * p.Ws;
                const size_t zo = ((size_t)(b * p.dzHp + yy + p.dzpad) * p.dzWp + xx + p.dzpad) * p.dzC + co0 + piece * 8;
                rz[i] = *reinterpret_cast<const u32x4*>(p.dz + zo);
                const size_t xo = ((size_t)(b * p.xHp + yy * p.istr) * p.xWp + xx * p.istr) * p.xC + toff;
                rx[i] = *reinterpret_cast<const u32x4*>(p.x + xo);
            } else {
                rz[i] = u32x4{0u, 0u, 0u, 0u};
                rx[i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        bf16_t* dZ = sZ + buf * 128 * 64;
        bf16_t* dX = sX + buf * 128 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + 32 * i;
            const int o = wg_swz(r, piece);
            *reinterpret_cast<u32x4*>(dZ + o) = rz[i];
            *reinterpret_cast<u32x4*>(dX + o) = rx[i];
        }
    };

    f32x4 acc[4][4];     // [co tile][kc tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_chunk(chunk_begin);
    store_chunk(0);
    __syncthreads();

    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
        const int buf = (ch - chunk_begin) & 1;
        if (ch + 1 < chunk_end) load_chunk(ch + 1);
        const bf16_t* cZ = sZ + buf * 128 * 64;
        const bf16_t* cX = sX + buf * 128 * 64;
        bf16x8 az[4], bx8[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) az[a] = tr_frag(cZ, wave * 32, a, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) bx8[b] = tr_frag(cX, wave * 32, b, lane);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(az[a], bx8[b], acc[a][b], 0, 0, 0);
        if (ch + 1 < chunk_end) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // cross-wave reduction through LDS: red[wave][co 64][kc 64] fp32 = 64 KiB
    float* red = reinterpret_cast<float*>(smem);
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = a * 16 + 4 * fq + j;
                const int kc = b * 16 + fr;
                red[(wave * 64 + co) * 64 + kc] = acc[a][b][j];
            }
    __syncthreads();
    float* out = p.dw + ((size_t)wsl * p.Co + co0) * p.Kc + kc0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;          // 0..4095
        const int co = e >> 6, kc = e & 63;
        const float v = red[e] + red[4096 + e] + red[8192 + e] + red[12288 + e];
        atomicAdd(out + (size_t)co * p.Kc + kc, v);
    }
}

hipError_t vpd_launch_wgrad(const WgradParams& p0, hipStream_t stream) {
    if (p0.Kc % 64 != 0 || p0.Co % 64 != 0 || p0.M <= 0) return hipErrorInvalidValue;
    WgradParams p = p0;
    const int tiles = (p.Co / 64) * (p.Kc / 64) * p.taps.nr * p.taps.nc;
    const int nchunks = (p.M + 127) / 128;
    // aim for ~1024 blocks; at least 4 chunks per block to amortise the atomics
    int ksplit = (1024 + tiles - 1) / tiles;
    if (ksplit > nchunks) ksplit = nchunks;
    if (ksplit < 1) ksplit = 1;
    int cpb = (nchunks + ksplit - 1) / ksplit;
    if (cpb < 4) cpb = nchunks < 4 ? nchunks : 4;
    ksplit = (nchunks + cpb - 1) / cpb;
    p.chunks_per_block = cpb;
    dim3 grid(tiles, ksplit);
    const size_t lds = 64 * 1024;     // max(staging 2*2*16 KiB, reduction 64 KiB)
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Probe for the parity tests: stages one [128][64] bf16 tile exactly like the
// wgrad kernel and dumps the transposed-read fragments, so the
// ds_read_b64_tr_b16 lane mapping is checked on hardware in isolation.
// out: [4 waves][4 ctiles][64 lanes][8] bf16
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tr_read_probe_kernel(const bf16_t* tile, bf16_t* out) {
    __shared__ __attribute__((aligned(16))) bf16_t s[128 * 64];
    const int tid = threadIdx.x, piece = tid & 7, row0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 32 * i;
        *reinterpret_cast<uint4*>(s + wg_swz(r, piece)) = *reinterpret_cast<const uint4*>(tile + r * 64 + piece * 8);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 f = tr_frag(s, wave * 32, ct, lane);
        *reinterpret_cast<bf16x8*>(out + ((size_t)(wave * 4 + ct) * 64 + lane) * 8) = f;
    }
}
extern "C" int vpd_op_tr_read_probe(const void* tile, void* out, void* stream) {
    hipLaunchKernelGGL(tr_read_probe_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)tile,
                       (bf16_t*)out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
