// Layout conversion at the reference boundary (fp32 NCHW / OIHW <-> internal
// bf16 NHWC / packed-tap weights) and the fused AdamW step.
#include "common.h"
#include "kernels.h"

// f32 [N][C][H][W] -> bf16 [N][Hp][Wp][Cp] interior (zero border is pre-set and never written).
// A thread converts 4 consecutive pixels of a row: one float4 per channel plane in, four 16-byte pixels out.
__global__ __launch_bounds__(256) void pack_input_kernel(const float* x, int N, int C, int H, int W, bf16_t* out,
                                                         int Hp, int Wp, int pad, int Cp) {
    const int W4 = W >> 2;
    const long total = (long)N * H * W4;
    const long plane = (long)H * W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int b = (int)(it / ((long)H * W4));
        const long r4 = it - (long)b * H * W4;
        const int y = (int)(r4 / W4);
        const int x0 = (int)(r4 - (long)y * W4) << 2;
        float v[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[q][c] = 0.f;
        for (int c = 0; c < C; ++c) {
            const float4 f = *reinterpret_cast<const float4*>(x + ((size_t)b * C + c) * plane + (size_t)y * W + x0);
            v[0][c] = f.x; v[1][c] = f.y; v[2][c] = f.z; v[3][c] = f.w;
        }
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
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < C; ++c) v[c] = x[((size_t)b * C + c) * plane + r];
        *reinterpret_cast<uint4*>(out + ((size_t)(b * Hp + y + pad) * Wp + xx + pad) * Cp) = pack8(v);
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
    hipLaunchKernelGGL(pack_input_kernel, dim3(g < 1 ? 1 : (int)g), dim3(256), 0, s, x, N, C, H, W, out, Hp, Wp, pad, Cp);
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
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int r = i / run, k = i - r * run;
        tile[r][k] = src[((size_t)(co0 + r) * d.Ci + ci0) * khw + k];
    }
    __syncthreads();
    // forward: dst[(tap*Co + co)*Ci + ci]; lanes run along ci
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
        arena[d.fwd_off + ((size_t)tap * d.Co + co0 + co) * d.Ci + ci0 + ci] = (bf16_t)tile[co][ci * khw + tap];
    }
    // dgrad: dst[(tap*Ci + ci)*Co + co]; lanes run along co
    if (d.dgr_off >= 0)
        for (int i = threadIdx.x; i < 32 * run; i += 256) {
            const int co = i & 31, ci = (i >> 5) & 31, tap = i >> 10;
            arena[d.dgr_off + ((size_t)tap * d.Ci + ci0 + ci) * d.Co + co0 + co] = (bf16_t)tile[co][ci * khw + tap];
        }
}
hipError_t vpd_launch_pack_weights(const PackDesc* d_descs, int ndesc, const int* d_blockmap, int nblocks,
                                   const float* master, bf16_t* arena, hipStream_t s) {
    (void)ndesc;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nblocks), dim3(256), 0, s, d_descs, d_blockmap, master, arena);
    return hipGetLastError();
}

// Gradient of every conv of a bucket -> flat gradient buffer in the reference's OIHW order, one launch.
// Every conv's weight gradient sits in the wgrad scratch as [tap][Co][Kc] fp32: gather into OIHW.
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
__global__ __launch_bounds__(256) void adamw_kernel(float4* p, const float4* g, float4* m, float4* v, long n4,
                                                    float decay, float omb1, float b2, float omb2, float step_size,
                                                    float inv_sqrt_bc2, float eps) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define ADAMW1(f)                                                         \
        pp.f *= decay;                                                    \
        mm.f += (gg.f - mm.f) * omb1;                                     \
        vv.f = vv.f * b2 + omb2 * gg.f * gg.f;                            \
        pp.f -= step_size * (mm.f / (sqrtf(vv.f) * inv_sqrt_bc2 + eps));
        ADAMW1(x) ADAMW1(y) ADAMW1(z) ADAMW1(w)
#undef ADAMW1
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}
hipError_t vpd_launch_adamw(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                            double eps, double wd, int step, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const double bc1 = 1.0 - pow(b1, step);
    const double bc2 = 1.0 - pow(b2, step);
    const long n4 = n / 4;
    long gsz = (n4 + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3(gsz < 1 ? 1 : (int)gsz), dim3(256), 0, s, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, n4, (float)(1.0 - lr * wd), (float)(1.0 - b1), (float)b2,
                       (float)(1.0 - b2), (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)eps);
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
