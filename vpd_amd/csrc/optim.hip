// Layout conversion at the reference boundary (fp32 NCHW / OIHW <-> internal
// bf16 NHWC / packed-tap weights) and the fused AdamW step.
#include "common.h"
#include "kernels.h"

// f32 [N][C][H][W] -> bf16 [N][Hp][Wp][Cp] interior (zero border is pre-set and never written).
// A thread converts 4 consecutive pixels of a row: one float4 per channel plane in, four 16-byte pixels out.
// CT: the channel count as a compile-time constant (3, 5, 6), or 0 = run-time C with CLAMPED unconditional loads -- a
// per-channel "load or zero" on a run-time condition makes hipcc branch around every load and wait for each one
// (54 us for 84 MB; /opt/skills/guides/cdna_hip_programming.md, "three .s-level traps" (c))
template <int CT>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* x, int N, int C, int H, int W, bf16_t* out,
                                                         int Hp, int Wp, int pad, int Cp) {
    const int W4 = W >> 2;
    const long total = (long)N * H * W4;
    const long plane = (long)H * W;
    const int Cn = CT ? CT : C;
    // (item -> (image, row, pixel quad): the 64-bit divisions of the generic form were most of an item's vector instructions -- 40 us
    //  for 157 MB; item counts below 2^21 split exactly with the float reciprocal, vpd_fdiv)
    const bool fast = total < VPD_FDIV_MAX;
    const float rW4 = 1.0f / (float)W4, rH = 1.0f / (float)H;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        int b, y, x0;
        if (fast) {
            const int q = vpd_fdiv((int)it, rW4);              // b * H + y
            x0 = ((int)it - q * W4) << 2;
            b = vpd_fdiv(q, rH);
            y = q - b * H;
        } else {
            b = (int)(it / ((long)H * W4));
            const long r4 = it - (long)b * H * W4;
            y = (int)(r4 / W4);
            x0 = (int)(r4 - (long)y * W4) << 2;
        }
        const float* src = x + (size_t)b * Cn * plane + (size_t)y * W + x0;
        float4 f[8];
        if (CT) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                f[c] = c < CT ? *reinterpret_cast<const float4*>(src + (size_t)c * plane) : float4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int cc = c < Cn ? c : Cn - 1;
                f[c] = *reinterpret_cast<const float4*>(src + (size_t)cc * plane);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c >= Cn) f[c] = float4{0.f, 0.f, 0.f, 0.f};
        }
        float v[4][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[0][c] = f[c].x; v[1][c] = f[c].y; v[2][c] = f[c].z; v[3][c] = f[c].w; }
        bf16_t* dst = out + ((size_t)(b * Hp + y + pad) * Wp + x0 + pad) * Cp;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(dst + q * Cp) = pack8(v[q]);
    }
}
// widths that are not a multiple of 4: one pixel per thread
__global__ __launch_bounds__(256) void pack_input_px_kernel(const float* x, int N, int C, int H, int W, bf16_t* out,
                                                            int Hp, int Wp, int pad, int Cp) {
    const long total = (long)N * H * W;
    const long plane = (long)H * W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int b = (int)(it / plane);
        const long r = it - (long)b * plane;
        const int y = (int)(r / W);
        const int xx = (int)(r - (long)y * W);
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? x[((size_t)b * C + c) * plane + r] : 0.f;
        *reinterpret_cast<uint4*>(out + ((size_t)(b * Hp + y + pad) * Wp + xx + pad) * Cp) = pack8(v);
    }
}
// Rows of a multiple of 64 pixels (the 128 x 128 crops of every BASELINE config): a wave takes 256 consecutive pixels of an image --
// one float4 per lane and channel plane in (1 KB per instruction), through its own 8 KB of LDS ([plane][256 pixels]: written four
// pixels per lane, read one pixel per lane), one 16-byte pixel per lane out: every store instruction covers 1 KB of one padded row.
// (pack_input_kernel's lanes write 64-byte pieces 64 bytes apart with four instructions that each touch 64 different lines.)
template <int CT>
__global__ __launch_bounds__(256) void pack_input_rows_kernel(const float* x, int N, int H, int W, bf16_t* out, int Hp, int Wp,
                                                              int pad) {
    __shared__ float lds[4][CT][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long plane = (long)H * W;
    const long nwork = (long)N * plane / 256;                          // (H * W % 256 == 0: checked by the launcher)
    const int per_img = (int)(plane / 256);
    float (*my)[256] = lds[wave];
    for (long it = (long)blockIdx.x * 4 + wave; it < nwork; it += (long)gridDim.x * 4) {
        const int b = (int)(it / per_img);
        const int p0 = (int)(it - (long)b * per_img) * 256;            // first pixel of the image
        const float* src = x + (size_t)b * CT * plane + p0 + 4 * lane;
        float4 f[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) f[c] = *reinterpret_cast<const float4*>(src + (size_t)c * plane);
#pragma unroll
        for (int c = 0; c < CT; ++c) *reinterpret_cast<float4*>(&my[c][4 * lane]) = f[c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + q * 64 + lane;
            const int y = p / W, xx = p - y * W;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = c < CT ? my[c][q * 64 + lane] : 0.f;
            *reinterpret_cast<uint4*>(out + ((size_t)(b * Hp + y + pad) * Wp + xx + pad) * 8) = pack8(v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
hipError_t vpd_launch_pack_input(const float* x, int N, int C, int H, int W, bf16_t* out, int Hp, int Wp, int pad,
                                 int Cp, hipStream_t s) {
    if (Cp != 8 || C > 8) return hipErrorInvalidValue;
    const bool quad = (W & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
    long items = quad ? (long)N * H * (W / 4) : (long)N * H * W;
    long g = (items + 255) / 256;
    if (g > 8192) g = 8192;
    if (!quad) {
        hipLaunchKernelGGL(pack_input_px_kernel, dim3(g < 1 ? 1 : (int)g), dim3(256), 0, s, x, N, C, H, W, out, Hp, Wp, pad, Cp);
        return hipGetLastError();
    }
    if ((W & 63) == 0 && ((long)H * W) % 256 == 0 && (C == 5 || C == 3 || C == 6)) {
        long gb = ((long)N * H * W / 256 + 3) / 4;
        if (gb > 4096) gb = 4096;
        const dim3 gr((unsigned)(gb < 1 ? 1 : gb));
        if (C == 5) hipLaunchKernelGGL(pack_input_rows_kernel<5>, gr, dim3(256), 0, s, x, N, H, W, out, Hp, Wp, pad);
        else if (C == 3) hipLaunchKernelGGL(pack_input_rows_kernel<3>, gr, dim3(256), 0, s, x, N, H, W, out, Hp, Wp, pad);
        else hipLaunchKernelGGL(pack_input_rows_kernel<6>, gr, dim3(256), 0, s, x, N, H, W, out, Hp, Wp, pad);
        return hipGetLastError();
    }
    const dim3 grid(g < 1 ? 1 : (int)g);
    if (C == 5) hipLaunchKernelGGL(pack_input_kernel<5>, grid, dim3(256), 0, s, x, N, C, H, W, out, Hp, Wp, pad, Cp);
    else if (C == 3) hipLaunchKernelGGL(pack_input_kernel<3>, grid, dim3(256), 0, s, x, N, C, H, W, out, Hp, Wp, pad, Cp);
    else if (C == 6) hipLaunchKernelGGL(pack_input_kernel<6>, grid, dim3(256), 0, s, x, N, C, H, W, out, Hp, Wp, pad, Cp);
    else hipLaunchKernelGGL(pack_input_kernel<0>, grid, dim3(256), 0, s, x, N, C, H, W, out, Hp, Wp, pad, Cp);
    return hipGetLastError();
}

#define PACK_CHUNK 1024

// master fp32 OIHW -> bf16 forward layout [tap][Co][Ci] and dgrad layout [tap][Ci][Co].
// One block transposes a 32(co) x 32(ci) x taps tile through LDS: the OIHW reads are 32 contiguous runs of
// 32*taps floats, the forward layout is written in 64-byte runs along ci and the dgrad layout in 64-byte runs
// along co (a plain element-wise gather re-fetched every source line ~5x: FETCH_SIZE 410 MB for 85 MB of weights).
// The stem (7x7, Ci <= 8, row-tap packing with zero fill) keeps the element-wise form.
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackDesc* descs, const int* blockmap,
                                                           const float* master, bf16_t* arena) {
    __shared__ float tile[32][32 * 9 + 1];
    const PackDesc d = descs[blockmap[2 * blockIdx.x]];
    const int chunk = blockmap[2 * blockIdx.x + 1];
    const float* src = master + d.src_off;
    const int khw = d.kh * d.kw;
    if (d.stem) {
        const long e0 = (long)chunk * PACK_CHUNK;
        const long nf = (long)d.ntaps * d.Co * d.Kc;
        for (long e = e0 + threadIdx.x; e < e0 + PACK_CHUNK && e < nf; e += 256) {
            const int kc = (int)(e % d.Kc);
            const long q = e / d.Kc;
            const int co = (int)(q % d.Co);
            const int tap = (int)(q / d.Co);      // tap = kernel row r; kc = t*8 + c
            const int t = kc >> 3, c = kc & 7;
            float v = 0.f;
            if (t < d.kw && c < d.Ci) v = src[((size_t)(co * d.Ci + c) * d.kh + tap) * d.kw + t];
            arena[d.fwd_off + e] = (bf16_t)v;
        }
        return;
    }
    const int tci = d.Ci >> 5;
    const int co0 = (chunk / tci) * 32, ci0 = (chunk % tci) * 32;
    const int run = 32 * khw;                      // contiguous floats per co row of the tile
    if ((reinterpret_cast<size_t>(src) & 15) == 0) {   // 16-byte loads (run and the row starts are multiples of 4 floats)
        const int run4 = run >> 2;
        for (int i = threadIdx.x; i < 32 * run4; i += 256) {
            const int r = i / run4, k = (i - r * run4) << 2;
            const float4 f = *reinterpret_cast<const float4*>(src + ((size_t)(co0 + r) * d.Ci + ci0) * khw + k);
            tile[r][k] = f.x; tile[r][k + 1] = f.y; tile[r][k + 2] = f.z; tile[r][k + 3] = f.w;
        }
    } else {
        for (int i = threadIdx.x; i < 32 * run; i += 256) {
            const int r = i / run, k = i - r * run;
            tile[r][k] = src[((size_t)(co0 + r) * d.Ci + ci0) * khw + k];
        }
    }
    __syncthreads();
    // forward: dst[(tap*Co + co)*Ci + ci]: a lane converts 8 consecutive ci -> one 16-byte store, 4 lanes per 64-byte run
    for (int i = threadIdx.x; i < 128 * khw; i += 256) {
        const int q = i & 3, co = (i >> 2) & 31, tap = i >> 7;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = tile[co][(q * 8 + j) * khw + tap];
        *reinterpret_cast<uint4*>(arena + d.fwd_off + ((size_t)tap * d.Co + co0 + co) * d.Ci + ci0 + q * 8) = pack8(f);
    }
    // dgrad: dst[(tap*Ci + ci)*Co + co]: 8 consecutive co per lane
    if (d.dgr_off >= 0)
        for (int i = threadIdx.x; i < 128 * khw; i += 256) {
            const int q = i & 3, ci = (i >> 2) & 31, tap = i >> 7;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = tile[q * 8 + j][ci * khw + tap];
            *reinterpret_cast<uint4*>(arena + d.dgr_off + ((size_t)tap * d.Ci + ci0 + ci) * d.Co + co0 + q * 8) = pack8(f);
        }
}
hipError_t vpd_launch_pack_weights(const PackDesc* d_descs, int ndesc, const int* d_blockmap, int nblocks,
                                   const float* master, bf16_t* arena, hipStream_t s) {
    (void)ndesc;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nblocks), dim3(256), 0, s, d_descs, d_blockmap, master, arena);
    return hipGetLastError();
}

// Gradient of every conv of a bucket -> flat gradient buffer in the reference's OIHW order, one launch.
// Every conv's weight gradient sits in the wgrad scratch as [tap][Co][Kc] fp32: gather into OIHW (element-wise: the
// strided 4-byte reads are absorbed by L2; an LDS-tiled transpose like pack_weights_kernel measured 48 % slower, one
// thread per (co, ci) pair -- coalesced reads, 36-byte-strided stores -- 40 % slower).
__global__ __launch_bounds__(256) void unpack_grads_kernel(const PackDesc* descs, const int* blockmap, const float* wg,
                                                           float* grads) {
    const PackDesc d = descs[blockmap[2 * blockIdx.x]];
    const long e0 = (long)blockmap[2 * blockIdx.x + 1] * PACK_CHUNK;
    const int khw = d.kh * d.kw;
    const long ns = (long)d.Co * d.Ci * khw;
    const float* src = wg + d.wg_off;
    for (long e = e0 + threadIdx.x; e < e0 + PACK_CHUNK && e < ns; e += 256) {
        const int tap = (int)(e % khw);
        const long q = e / khw;
        const int ci = (int)(q % d.Ci);
        const int co = (int)(q / d.Ci);
        float v;
        if (d.stem) {
            const int r = tap / d.kw, t = tap - r * d.kw;
            v = src[((size_t)r * d.Co + co) * d.Kc + t * 8 + ci];
        } else {
            v = src[((size_t)tap * d.Co + co) * d.Kc + ci];
        }
        grads[d.src_off + e] = v;
    }
}
hipError_t vpd_launch_unpack_grads(const PackDesc* d_descs, int ndesc, const int* d_blockmap, int nblocks,
                                   const float* wg, float* grads, hipStream_t s) {
    (void)ndesc;
    hipLaunchKernelGGL(unpack_grads_kernel, dim3(nblocks), dim3(256), 0, s, d_descs, d_blockmap, wg, grads);
    return hipGetLastError();
}

// Fused AdamW over the whole flat parameter buffer (every tensor shares the
// hyper-parameters: train_vpd_model.py:104 uses one param group, wd on all).
// (explicit fma placement: the flat kernel and the fused AdamW + repack kernel must round identically)
static __device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, const AdamHyper& h) {
#pragma clang fp contract(off)
    g *= h.gscale;      // 1 / loss scale: exact for the powers of two a LossScaler uses, and for 1
    p *= h.decay;
    m = fmaf(g - m, h.omb1, m);
    v = fmaf(h.omb2 * g, g, v * h.b2);
    const float den = fmaf(sqrtf(v), h.inv_sqrt_bc2, h.eps);
    p = fmaf(-h.step_size, m / den, p);
}
static __device__ __forceinline__ void adamw4(float4& pp, const float4& gg, float4& mm, float4& vv, const AdamHyper& h) {
    adamw1(pp.x, gg.x, mm.x, vv.x, h); adamw1(pp.y, gg.y, mm.y, vv.y, h);
    adamw1(pp.z, gg.z, mm.z, vv.z, h); adamw1(pp.w, gg.w, mm.w, vv.w, h);
}
static AdamHyper adam_hyper(double lr, double b1, double b2, double eps, double wd, int step, float gscale) {
    const double bc1 = 1.0 - pow(b1, step);
    const double bc2 = 1.0 - pow(b2, step);
    return AdamHyper{(float)(1.0 - lr * wd), (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(lr / bc1),
                     (float)(1.0 / sqrt(bc2)), (float)eps, gscale};
}

__global__ __launch_bounds__(256) void adamw_kernel(float4* p, const float4* g, float4* m, float4* v, long n4,
                                                    const AdamHyper h) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adamw4(pp, gg, mm, vv, h);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}
hipError_t vpd_launch_adamw(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                            double eps, double wd, int step, hipStream_t s, float gscale) {
    if (n % 4) return hipErrorInvalidValue;
    const long n4 = n / 4;
    long gsz = (n4 + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3(gsz < 1 ? 1 : (int)gsz), dim3(256), 0, s, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, n4, adam_hyper(lr, b1, b2, eps, wd, step, gscale));
    return hipGetLastError();
}

// AdamW and the repack of the bf16 weight layouts in ONE pass over the parameters (the train step's optimizer):
// a block updates a 32(co) x 32(ci) x taps tile of a conv weight in place (p, m, v: 16-byte accesses of the OIHW
// runs), keeps the new values in LDS and writes both bf16 layouts from there -- pack_weights_kernel's tile without
// re-reading the 85 MB of master weights.  Ranges that are not conv weights (BatchNorm, fc, motion head) are
// `stem == 2` descriptors updated element-wise; so is the stem conv, whose row-tap packing gathers across the whole
// tensor and is done by a 28-block pack_weights_kernel launch afterwards.
#define ADAM_PLAIN_CHUNK 2048
// wg != null: the conv weight gradients are read where the weight-gradient kernels left them -- the fp32 scratch in
// [tap][Co][Ci] order (PackDesc::wg_off) -- instead of from the OIHW gradient buffer: the unpack pass in between (170 MB of
// traffic, 67 us per step) disappears.  The stem and every non-conv tensor still come from `g`.
template <int KHW>
static __device__ __forceinline__ float4 adamw_gather_g(const float* wgc, int Co, int Kc, int co, int ci0, int k) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        const int ci = kk / KHW, tap = kk - ci * KHW;
        v[j] = wgc[((size_t)tap * Co + co) * Kc + ci0 + ci];
    }
    return float4{v[0], v[1], v[2], v[3]};
}
__global__ __launch_bounds__(256) void adamw_pack_kernel(const PackDesc* descs, const int* blockmap, float* p,
                                                         const float* g, float* m, float* v, bf16_t* arena,
                                                         const AdamHyper h, const float* wg) {
    __shared__ unsigned short tile[32][32 * 9 + 2];      // the new weights, already rounded to bf16 (18 KB: 8 blocks per CU)
    const PackDesc d = descs[blockmap[2 * blockIdx.x]];
    const int chunk = blockmap[2 * blockIdx.x + 1];
    const int khw = d.kh * d.kw;
    if (d.stem == 2) {
        const long e0 = d.src_off + (long)chunk * ADAM_PLAIN_CHUNK;
        const long e1 = d.src_off + d.numel;
        for (long e = e0 + threadIdx.x; e < e0 + ADAM_PLAIN_CHUNK && e < e1; e += 256) {
            float pp = p[e], mm = m[e], vv = v[e];
            adamw1(pp, g[e], mm, vv, h);
            p[e] = pp; m[e] = mm; v[e] = vv;
        }
        return;
    }
    const int tci = d.Ci >> 5;
    const int co0 = (chunk / tci) * 32, ci0 = (chunk % tci) * 32;
    const int run4 = (32 * khw) >> 2;
    const float* wgc = wg ? wg + d.wg_off : nullptr;
    for (int i = threadIdx.x; i < 32 * run4; i += 256) {
        const int r = i / run4, k = (i - r * run4) << 2;
        const size_t e = d.src_off + ((size_t)(co0 + r) * d.Ci + ci0) * khw + k;
        float4 pp = *reinterpret_cast<const float4*>(p + e);
        float4 mm = *reinterpret_cast<const float4*>(m + e);
        float4 vv = *reinterpret_cast<const float4*>(v + e);
        float4 gg;
        if (!wgc) gg = *reinterpret_cast<const float4*>(g + e);
        else if (khw == 9) gg = adamw_gather_g<9>(wgc, d.Co, d.Kc, co0 + r, ci0, k);
        else if (khw == 1) gg = *reinterpret_cast<const float4*>(wgc + (size_t)(co0 + r) * d.Kc + ci0 + k);
        else {
            float t4[4];
            for (int j = 0; j < 4; ++j) {
                const int ci = (k + j) / khw, tap = (k + j) - ci * khw;
                t4[j] = wgc[((size_t)tap * d.Co + co0 + r) * d.Kc + ci0 + ci];
            }
            gg = float4{t4[0], t4[1], t4[2], t4[3]};
        }
        adamw4(pp, gg, mm, vv, h);
        *reinterpret_cast<float4*>(p + e) = pp;
        *reinterpret_cast<float4*>(m + e) = mm;
        *reinterpret_cast<float4*>(v + e) = vv;
        tile[r][k] = __builtin_bit_cast(unsigned short, (bf16_t)pp.x);
        tile[r][k + 1] = __builtin_bit_cast(unsigned short, (bf16_t)pp.y);
        tile[r][k + 2] = __builtin_bit_cast(unsigned short, (bf16_t)pp.z);
        tile[r][k + 3] = __builtin_bit_cast(unsigned short, (bf16_t)pp.w);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 128 * khw; i += 256) {
        const int q = i & 3, co = (i >> 2) & 31, tap = i >> 7;
        unsigned int w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = (unsigned int)tile[co][(q * 8 + 2 * j) * khw + tap] | ((unsigned int)tile[co][(q * 8 + 2 * j + 1) * khw + tap] << 16);
        *reinterpret_cast<uint4*>(arena + d.fwd_off + ((size_t)tap * d.Co + co0 + co) * d.Ci + ci0 + q * 8) = uint4{w[0], w[1], w[2], w[3]};
    }
    if (d.dgr_off >= 0)
        for (int i = threadIdx.x; i < 128 * khw; i += 256) {
            const int q = i & 3, ci = (i >> 2) & 31, tap = i >> 7;
            unsigned int w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = (unsigned int)tile[q * 8 + 2 * j][ci * khw + tap] | ((unsigned int)tile[q * 8 + 2 * j + 1][ci * khw + tap] << 16);
            *reinterpret_cast<uint4*>(arena + d.dgr_off + ((size_t)tap * d.Ci + ci0 + ci) * d.Co + co0 + q * 8) = uint4{w[0], w[1], w[2], w[3]};
        }
}
hipError_t vpd_launch_adamw_pack(const PackDesc* d_descs, const int* d_blockmap, int nblocks, float* p, const float* g,
                                 float* m, float* v, bf16_t* arena, double lr, double b1, double b2, double eps, double wd,
                                 int step, hipStream_t s, const float* wg, float gscale) {
    if ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
         reinterpret_cast<size_t>(v)) & 15)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(adamw_pack_kernel, dim3(nblocks), dim3(256), 0, s, d_descs, d_blockmap, p, g, m, v, arena,
                       adam_hyper(lr, b1, b2, eps, wd, step, gscale), wg);
    return hipGetLastError();
}

// Zero up to ZR_MAX float ranges in ONE launch (accumulator rows of the BN statistics + the fp32 weight-gradient
// ranges the atomics kernel adds into): hipMemsetAsync costs 5-12 us per call on this runtime.
__global__ __launch_bounds__(256) void zero_ranges_kernel(const ZeroRanges z) {
    float4* p = reinterpret_cast<float4*>(z.ptr[blockIdx.y]);
    const long n4 = z.n4[blockIdx.y];
    const float4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = zero;
}
hipError_t vpd_launch_zero_ranges(const ZeroRanges& z, hipStream_t s) {
    if (z.count <= 0) return hipSuccess;
    long mx = 0;
    for (int i = 0; i < z.count; ++i) mx = z.n4[i] > mx ? z.n4[i] : mx;
    long gx = (mx + 255) / 256;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)(gx < 1 ? 1 : gx), z.count), dim3(256), 0, s, z);
    return hipGetLastError();
}
