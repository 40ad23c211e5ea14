// BatchNorm (train + eval), ReLU, residual add, stem max-pool and their
// backward passes.  All HBM-bound: 16-byte (8 x bf16) accesses per lane,
// per-channel reductions done as per-block fp32 partials (fixed order inside a
// block) added with fp64 atomics into 16 accumulator rows that a tiny finalize
// kernel sums in double precision: the arrival order of the blocks moves a sum
// by ~1e-16 relative, which does not survive the fp32 rounding of the results.
#include "common.h"
#include "kernels.h"

// ---------------------------------------------------------------------------
// finalize of forward statistics: partials [T][2][C] -> mean, rstd, scale,
// shift; running-stat update (momentum, unbiased var) as nn.BatchNorm2d.
// ---------------------------------------------------------------------------
// One wave per 64 channels, one channel per lane: the 2*VPD_STAT_ROWS loads are independent and go out
// back to back, so the kernel costs one memory round trip (it is latency, not bandwidth, bound).
static __device__ __forceinline__ void stat_rows_take(double* partials, int C, int c, double* s1, double* s2) {
    double v1[VPD_STAT_ROWS], v2[VPD_STAT_ROWS];
#pragma unroll
    for (int t = 0; t < VPD_STAT_ROWS; ++t) {
        v1[t] = partials[((size_t)t * 2) * C + c];
        v2[t] = partials[((size_t)t * 2 + 1) * C + c];
    }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int t = 0; t < VPD_STAT_ROWS; ++t) {
        a += v1[t]; b += v2[t];
        partials[((size_t)t * 2) * C + c] = 0.0;          // leave the accumulator rows zeroed for the next producer
        partials[((size_t)t * 2 + 1) * C + c] = 0.0;
    }
    *s1 = a; *s2 = b;
}

__global__ __launch_bounds__(64) void bn_finalize_kernel(
    double* partials, int T, int C, float count,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* running_mean, float* running_var, float momentum, float eps,
    float* mean, float* rstd, float* scale, float* shift) {
    (void)T;
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s1, s2;
    stat_rows_take(partials, C, c, &s1, &s2);
    const double mu = s1 / (double)count;
    double var = s2 / (double)count - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    rstd[c] = r;
    const float sc = gamma[c] * r;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
    if (running_mean) {
        const double unb = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

hipError_t vpd_launch_bn_finalize(double* partials, int T, int C, float count, const float* gamma,
                                  const float* beta, float* rm, float* rv, float momentum, float eps,
                                  float* mean, float* rstd, float* scale, float* shift, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, s, partials, T, C, count, gamma,
                       beta, rm, rv, momentum, eps, mean, rstd, scale, shift);
    return hipGetLastError();
}

// eval-mode fold: scale = gamma / sqrt(rv + eps), shift = beta - rm * scale
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}
hipError_t vpd_launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                              float* scale, float* shift, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, rm, rv, eps, scale, shift, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// forward apply: out(padded) = relu?( scale*z + shift (+ residual) )
//   res_kind 0: none; 1: padded activation tensor; 2: dense z_d with its own
//   scale/shift (the downsample branch: conv1x1 -> BN, no ReLU)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyParams p) {
    const int cv = p.C >> 3;
    const long total = (long)p.M * cv;
    const int HW = p.H * p.W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int m = (int)(it / cv);
        const int c = (int)(it - (long)m * cv) << 3;
        const int b = m / HW;
        const int r = m - b * HW;
        const int y = r / p.W;
        const int x = r - y * p.W;
        float v[8], sc[8], sh[8];
        unpack8(*reinterpret_cast<const uint4*>(p.z + (size_t)m * p.C + c), v);
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(p.scale + c);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(p.scale + c + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(p.shift + c);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(p.shift + c + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
        if (p.res_kind == 1) {
            float rr[8];
            const size_t ro = ((size_t)(b * p.rHp + y + p.rpad) * p.rWp + x + p.rpad) * p.C + c;
            unpack8(*reinterpret_cast<const uint4*>(p.res + ro), rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j];
        } else if (p.res_kind == 2) {
            float rr[8];
            unpack8(*reinterpret_cast<const uint4*>(p.res + (size_t)m * p.C + c), rr);
            *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(p.rscale + c);
            *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(p.rscale + c + 4);
            *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(p.rshift + c);
            *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(p.rshift + c + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j] * sc[j] + sh[j];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        const size_t oo = ((size_t)(b * p.oHp + y + p.opad) * p.oWp + x + p.opad) * p.C + c;
        vpd_store16<VPD_CP_BNF>(p.out + oo, pack8(v));
    }
}

static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t vpd_launch_bn_apply(const BnApplyParams& p, hipStream_t s) {
    if (p.C % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid((long)p.M * (p.C / 8))), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stem: BN + ReLU + MaxPool 3x3 s2 p1 fused.  z dense [N][Hz][Wz][C] ->
// out padded [N][Ho+2*opad][Wo+2*opad][C]; idx (train) u8 dense [N][Ho][Wo][C]
// holds the window position r*3+t of the FIRST maximum (row-major scan, as
// torch's max_pool2d).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_pool_kernel(const StemPoolParams p) {
    const int cv = p.C >> 3;
    const int Ho = p.Ho, Wo = p.Wo;
    const long total = (long)p.N * Ho * Wo * cv;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % cv) << 3;
        long t = it / cv;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float sc[8], sh[8], best[8];
        int bi[8];
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(p.scale + c);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(p.scale + c + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(p.shift + c);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(p.shift + c + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
        for (int r = 0; r < 3; ++r) {
            const int y = 2 * oy - 1 + r;
            if (y < 0 || y >= p.Hz) continue;
            for (int tt = 0; tt < 3; ++tt) {
                const int x = 2 * ox - 1 + tt;
                if (x < 0 || x >= p.Wz) continue;
                float v[8];
                unpack8(*reinterpret_cast<const uint4*>(p.z + ((size_t)(b * p.Hz + y) * p.Wz + x) * p.C + c), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = v[j] * sc[j] + sh[j];
                    a = a > 0.f ? a : 0.f;
                    // compare the bf16-rounded activation (what a materialised tensor would hold)
                    a = bf2f(f2bf(a));
                    if (a > best[j]) { best[j] = a; bi[j] = r * 3 + tt; }
                }
            }
        }
        const size_t oo = ((size_t)(b * (Ho + 2 * p.opad) + oy + p.opad) * (Wo + 2 * p.opad) + ox + p.opad) * p.C + c;
        vpd_store16<VPD_CP_STEM>(p.out + oo, pack8(best));
        if (p.idx) {
            uint2 iv;
            iv.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
            iv.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
            *reinterpret_cast<uint2*>(p.idx + ((size_t)(b * Ho + oy) * Wo + ox) * p.C + c) = iv;
        }
    }
}
// The same for a horizontal PAIR of outputs per thread (even Wo): their windows share a column, so 15 z pixels are
// loaded -- all up front, independent of each other -- instead of 2 x 9 behind boundary tests.
__global__ __launch_bounds__(256) void stem_pool_pair_kernel(const StemPoolParams p) {
    const int cv = p.C >> 3;
    const int Ho = p.Ho, Wo = p.Wo, Wk = p.Wo >> 1;
    const long total = (long)p.N * Ho * Wk * cv;
    const int c = (int)(threadIdx.x % cv) << 3;            // blockDim is a multiple of cv: fixed per thread
    float sc[8], sh[8];
    *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(p.scale + c);
    *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(p.scale + c + 4);
    *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(p.shift + c);
    *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(p.shift + c + 4);
    // (item -> (image, row, output pair): 64-bit divisions cost more than the pooling itself; below 2^21 items the float
    //  reciprocal splits exactly, and cv is a power of two for every ResNet stem)
    const bool fast = total < VPD_FDIV_MAX && (cv & (cv - 1)) == 0;
    const int cvs = 31 - __builtin_clz(cv);
    const float rWk = 1.0f / (float)Wk, rHo = 1.0f / (float)Ho;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        int kx, oy, b;
        if (fast) {
            const int t = (int)it >> cvs;
            const int q = vpd_fdiv(t, rWk);
            kx = t - q * Wk;
            b = vpd_fdiv(q, rHo);
            oy = q - b * Ho;
        } else {
            long t = it / cv;
            kx = (int)(t % Wk); t /= Wk;
            oy = (int)(t % Ho);
            b = (int)(t / Ho);
        }
        uint4 zr[3][5];
        bool ok[3][5];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int y = 2 * oy - 1 + r, x = 4 * kx - 1 + q;
                ok[r][q] = y >= 0 && y < p.Hz && x >= 0 && x < p.Wz;
                const int yc = y < 0 ? 0 : (y >= p.Hz ? p.Hz - 1 : y), xc = x < 0 ? 0 : (x >= p.Wz ? p.Wz - 1 : x);
                zr[r][q] = *reinterpret_cast<const uint4*>(p.z + ((size_t)(b * p.Hz + yc) * p.Wz + xc) * p.C + c);
            }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            // running maximum as ONE unsigned word per channel: the candidate's bf16 bits (post-ReLU, so non-negative: unsigned order
            // = value order) above 15 - tap index -- v_max_u32 then keeps the larger value and, on a tie, the EARLIER tap, exactly
            // what `if (a > best) { best = a; bi = idx; }` over the bf16-rounded activation did with a compare and two selects
            unsigned key[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) key[j] = 0u;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) {
                    if (!ok[r][2 * o + tt]) continue;
                    float v[8];
                    unpack8(zr[r][2 * o + tt], v);
                    const unsigned low = 15u - (unsigned)(r * 3 + tt);
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        float a0 = v[j] * sc[j] + sh[j], a1 = v[j + 1] * sc[j + 1] + sh[j + 1];
                        a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f;
                        const unsigned w = pack2bf(a0, a1);              // (the bf16-rounded activation, what a materialised tensor would hold)
                        const unsigned k0 = (w << 16) | low, k1 = (w & 0xffff0000u) | low;
                        key[j] = k0 > key[j] ? k0 : key[j];
                        key[j + 1] = k1 > key[j + 1] ? k1 : key[j + 1];
                    }
                }
            const int ox = 2 * kx + o;
            const size_t oo = ((size_t)(b * (Ho + 2 * p.opad) + oy + p.opad) * (Wo + 2 * p.opad) + ox + p.opad) * p.C + c;
            uint4 ov;
            ov.x = (key[0] >> 16) | (key[1] & 0xffff0000u); ov.y = (key[2] >> 16) | (key[3] & 0xffff0000u);
            ov.z = (key[4] >> 16) | (key[5] & 0xffff0000u); ov.w = (key[6] >> 16) | (key[7] & 0xffff0000u);
            vpd_store16<VPD_CP_STEM>(p.out + oo, ov);
            if (p.idx) {
                unsigned bi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bi[j] = 15u - (key[j] & 15u);
                uint2 iv;
                iv.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
                iv.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
                *reinterpret_cast<uint2*>(p.idx + ((size_t)(b * Ho + oy) * Wo + ox) * p.C + c) = iv;
            }
        }
    }
}
hipError_t vpd_launch_stem_pool(const StemPoolParams& p, hipStream_t s) {
    static const bool pair = !(getenv("VPD_STEM_PAIR") && !atoi(getenv("VPD_STEM_PAIR")));
    const int cv = p.C / 8;
    if (pair && !(p.Wo & 1) && 256 % cv == 0) {
        // the grid-stride loop must keep a thread's channel slice fixed: total stride a multiple of cv (256 is)
        hipLaunchKernelGGL(stem_pool_pair_kernel, dim3(ew_grid((long)p.N * p.Ho * (p.Wo / 2) * cv)), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(stem_pool_kernel, dim3(ew_grid((long)p.N * p.Ho * p.Wo * cv)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// BN backward, pass 1: per-block partial sums of g and g*xhat, where
//   g = dy * [act > 0]  (mask from the padded post-ReLU activation, or none)
// Each block owns `ppb` consecutive pixels; a thread owns 8 channels.
// partials layout [T][2][C].
// ---------------------------------------------------------------------------
template <int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdParams p) {
    __shared__ float sh[256][17];
    const int cv = p.C >> 3;
    const int ppi = 256 / cv;                       // pixels per block iteration
    const int c8 = threadIdx.x % cv;
    const int pl = threadIdx.x / cv;
    const int c = c8 << 3;
    const int HW = p.H * p.W;
    float mu[8], rs[8], a1[8], a2[8], msc[8], msh[8];
    *reinterpret_cast<float4*>(mu) = *reinterpret_cast<const float4*>(p.mean + c);
    *reinterpret_cast<float4*>(mu + 4) = *reinterpret_cast<const float4*>(p.mean + c + 4);
    *reinterpret_cast<float4*>(rs) = *reinterpret_cast<const float4*>(p.rstd + c);
    *reinterpret_cast<float4*>(rs + 4) = *reinterpret_cast<const float4*>(p.rstd + c + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) { msc[j] = 0.f; msh[j] = 1.f; }
    if (MASK == 2) {
        *reinterpret_cast<float4*>(msc) = *reinterpret_cast<const float4*>(p.mscale + c);
        *reinterpret_cast<float4*>(msc + 4) = *reinterpret_cast<const float4*>(p.mscale + c + 4);
        *reinterpret_cast<float4*>(msh) = *reinterpret_cast<const float4*>(p.mshift + c);
        *reinterpret_cast<float4*>(msh + 4) = *reinterpret_cast<const float4*>(p.mshift + c + 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    const int mbeg = blockIdx.x * p.ppb;
    int mend = mbeg + p.ppb;
    mend = mend < p.M ? mend : p.M;
    if (pl < ppi)
        for (int m = mbeg + pl; m < mend; m += ppi) {
            float g[8], z[8];
            unpack8(*reinterpret_cast<const uint4*>(p.dy + (size_t)m * p.C + c), g);
            unpack8(*reinterpret_cast<const uint4*>(p.z + (size_t)m * p.C + c), z);
            if (MASK == 1) {
                const int b = m / HW;
                const int r = m - b * HW;
                const int y = r / p.W;
                const int x = r - y * p.W;
                float a[8];
                unpack8(*reinterpret_cast<const uint4*>(
                            p.act + ((size_t)(b * p.aHp + y + p.apad) * p.aWp + x + p.apad) * p.C + c), a);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = a[j] > 0.f ? g[j] : 0.f;
            } else if (MASK == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = (z[j] * msc[j] + msh[j]) > 0.f ? g[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a1[j] += g[j];
                a2[j] += g[j] * ((z[j] - mu[j]) * rs[j]);
            }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[threadIdx.x][j] = a1[j]; sh[threadIdx.x][8 + j] = a2[j]; }
    __syncthreads();
    // thread t < C sums channel t (c8 = t>>3, j = t&7) over the pixel groups, for both sums
    for (int t = threadIdx.x; t < 2 * p.C; t += 256) {
        const int which = t / p.C;
        const int ch = t - which * p.C;
        const int q8 = ch >> 3, j = ch & 7;
        float tot = 0.f;
        for (int g = 0; g < ppi; ++g) tot += sh[g * cv + q8][which * 8 + j];
        atomicAdd(&p.partials[((size_t)(blockIdx.x & (VPD_STAT_ROWS - 1)) * 2 + which) * p.C + ch], (double)tot);
    }
}

// pass 1b: partials -> dgamma, dbeta (fp32 grads) and the apply coefficients
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(
    double* partials, int T, int C, float count, const float* __restrict__ gamma,
    const float* __restrict__ rstd, float* dgamma, float* dbeta, float* coef) {
    (void)T;
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s1, s2;
    stat_rows_take(partials, C, c, &s1, &s2);
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = gamma[c] * rstd[c];
    coef[C + c] = (float)(s1 / (double)count);
    coef[2 * C + c] = (float)(s2 / (double)count);
}

// pass 2: dz = c1 * (g - c2 - xhat * c3); optionally write g back over dy
template <int MASK, int WRITE_G>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdParams p) {
    const int cv = p.C >> 3;
    const long total = (long)p.M * cv;
    const int HW = p.H * p.W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int m = (int)(it / cv);
        const int c = (int)(it - (long)m * cv) << 3;
        const int b = m / HW;
        const int r = m - b * HW;
        const int y = r / p.W;
        const int x = r - y * p.W;
        float g[8], z[8], mu[8], rs[8], c1[8], c2[8], c3[8];
        unpack8(*reinterpret_cast<const uint4*>(p.dy + (size_t)m * p.C + c), g);
        unpack8(*reinterpret_cast<const uint4*>(p.z + (size_t)m * p.C + c), z);
        if (MASK == 1) {
            float a[8];
            unpack8(*reinterpret_cast<const uint4*>(
                        p.act + ((size_t)(b * p.aHp + y + p.apad) * p.aWp + x + p.apad) * p.C + c), a);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = a[j] > 0.f ? g[j] : 0.f;
            if (WRITE_G) *reinterpret_cast<uint4*>(p.dy_rw + (size_t)m * p.C + c) = pack8(g);
        } else if (MASK == 2) {
            float msc[8], msh[8];
            *reinterpret_cast<float4*>(msc) = *reinterpret_cast<const float4*>(p.mscale + c);
            *reinterpret_cast<float4*>(msc + 4) = *reinterpret_cast<const float4*>(p.mscale + c + 4);
            *reinterpret_cast<float4*>(msh) = *reinterpret_cast<const float4*>(p.mshift + c);
            *reinterpret_cast<float4*>(msh + 4) = *reinterpret_cast<const float4*>(p.mshift + c + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = (z[j] * msc[j] + msh[j]) > 0.f ? g[j] : 0.f;
        }
#define LD8(dst, src) \
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src); \
        *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
        LD8(mu, p.mean + c) LD8(rs, p.rstd + c) LD8(c1, p.coef + c) LD8(c2, p.coef + p.C + c) LD8(c3, p.coef + 2 * p.C + c)
#undef LD8
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = c1[j] * (g[j] - c2[j] - (z[j] - mu[j]) * rs[j] * c3[j]);
        const size_t oo = ((size_t)(b * p.dzHp + y + p.dzpad) * p.dzWp + x + p.dzpad) * p.C + c;
        vpd_store16<VPD_CP_BNB>(p.dz + oo, pack8(o));
    }
}

int vpd_bn_bwd_blocks(int M, int C, int* ppb_out) {
    // pixels per block: enough blocks to hide the load latency (the small layers were latency-bound at one block per
    // CU with the earlier 8 iterations / 1024 blocks: +1.5 % end to end), at least four pixel iterations per thread so
    // that the per-block LDS reduce + 2C atomics stay amortised
    const int min_iter = 4, max_blocks = 2048;
    const int ppi = 256 / (C / 8);
    int ppb = (M + max_blocks - 1) / max_blocks;
    if (ppb < ppi * min_iter) ppb = ppi * min_iter;
    ppb = ((ppb + ppi - 1) / ppi) * ppi;
    if (ppb_out) *ppb_out = ppb;
    return (M + ppb - 1) / ppb;
}

hipError_t vpd_launch_bn_bwd(const BnBwdParams& p0, float count, const float* gamma, float* dgamma, float* dbeta,
                             hipStream_t s, bool reduce_done) {
    BnBwdParams p = p0;
    if (p.C % 8 || p.C > 2048 || 256 % (p.C / 8)) return hipErrorInvalidValue;
    const int T = vpd_bn_bwd_blocks(p.M, p.C, &p.ppb);
    // reduce_done: the producing data-gradient kernel already masked dy and accumulated (sum g, sum g*xhat)
    const int mask = p.act ? 1 : (p.mscale ? 2 : 0);
    if (!reduce_done) {
        if (mask == 1) hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(T), dim3(256), 0, s, p);
        else if (mask == 2) hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, dim3(T), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, dim3(T), dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((p.C + 63) / 64), dim3(64), 0, s, p.partials,
                       VPD_STAT_ROWS, p.C, count,
                       gamma, p.rstd, dgamma, dbeta, p.coef);
    const dim3 ag(ew_grid((long)p.M * (p.C / 8)));
    if (mask == 1 && p.write_g) hipLaunchKernelGGL((bn_bwd_apply_kernel<1, 1>), ag, dim3(256), 0, s, p);
    else if (mask == 1) hipLaunchKernelGGL((bn_bwd_apply_kernel<1, 0>), ag, dim3(256), 0, s, p);
    else if (mask == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<2, 0>), ag, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<0, 0>), ag, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stem backward: max-pool routing + ReLU mask -> g0 (dense bf16) and the BN
// backward partial sums in one pass.  d_pool is dense [N][Ho][Wo][C].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(const StemPoolBwdParams p) {
    __shared__ float sh[256][17];
    const int cv = p.C >> 3;
    const int ppi = 256 / cv;
    const int c8 = threadIdx.x % cv;
    const int pl = threadIdx.x / cv;
    const int c = c8 << 3;
    const int HWz = p.Hz * p.Wz;
    float mu[8], rs[8], sc[8], shf[8], a1[8], a2[8];
#define LD8(dst, src) \
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src); \
    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
    LD8(mu, p.mean + c) LD8(rs, p.rstd + c) LD8(sc, p.scale + c) LD8(shf, p.shift + c)
    float c1[8], c2[8], c3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { c1[j] = 0.f; c2[j] = 0.f; c3[j] = 0.f; }
    if (p.pass == 2) { LD8(c1, p.coef + c) LD8(c2, p.coef + p.C + c) LD8(c3, p.coef + 2 * p.C + c) }
#undef LD8
#pragma unroll
    for (int j = 0; j < 8; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    const int mbeg = blockIdx.x * p.ppb;
    int mend = mbeg + p.ppb;
    mend = mend < p.M ? mend : p.M;
    if (pl < ppi)
        for (int m = mbeg + pl; m < mend; m += ppi) {
            const int b = m / HWz;
            const int r = m - b * HWz;
            const int y = r / p.Wz;
            const int x = r - y * p.Wz;
            float z[8], g[8], o[8];
            unpack8(*reinterpret_cast<const uint4*>(p.z + (size_t)m * p.C + c), z);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = 0.f;
            // windows (oy, ox) that contain (y, x): oy in {y/2 (r=1 or 2)} and, for odd y, (y+1)/2 (r=0)
            for (int wy = 0; wy < 2; ++wy) {
                const int oy = (y >> 1) + wy;
                const int rr = y - (2 * oy - 1);
                if (rr < 0 || rr > 2 || oy >= p.Ho) continue;
                for (int wx = 0; wx < 2; ++wx) {
                    const int ox = (x >> 1) + wx;
                    const int tt = x - (2 * ox - 1);
                    if (tt < 0 || tt > 2 || ox >= p.Wo) continue;
                    const size_t po = ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.C + c;
                    const uint2 iv = *reinterpret_cast<const uint2*>(p.idx + po);
                    float d[8];
                    unpack8(*reinterpret_cast<const uint4*>(p.dpool + po), d);
                    const unsigned want = (unsigned)(rr * 3 + tt);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned sel = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
                        if (sel == want) g[j] += d[j];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = z[j] * sc[j] + shf[j];
                g[j] = a > 0.f ? g[j] : 0.f;
                const float xh = (z[j] - mu[j]) * rs[j];
                a1[j] += g[j];
                a2[j] += g[j] * xh;
                o[j] = c1[j] * (g[j] - c2[j] - xh * c3[j]);      // pass 2 only (coefficients are 0 in pass 1)
            }
            if (p.pass == 2) *reinterpret_cast<uint4*>(p.dz + (size_t)m * p.C + c) = pack8(o);
        }
    if (p.pass == 2) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[threadIdx.x][j] = a1[j]; sh[threadIdx.x][8 + j] = a2[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * p.C; t += 256) {
        const int which = t / p.C;
        const int ch = t - which * p.C;
        const int q8 = ch >> 3, j = ch & 7;
        float tot = 0.f;
        for (int gI = 0; gI < ppi; ++gI) tot += sh[gI * cv + q8][which * 8 + j];
        atomicAdd(&p.partials[((size_t)(blockIdx.x & (VPD_STAT_ROWS - 1)) * 2 + which) * p.C + ch], (double)tot);
    }
}

// Pass 2 of the stem backward over 2x2 blocks of z pixels (even Hz, Wz; 3x3 stride-2 pad-1 pooling, Ho = Hz/2):
// the block (2ky..2ky+1, 2kx..2kx+1) lies in the windows (ky..ky+1, kx..kx+1) only, so a thread loads four windows
// (dpool + arg-max bytes) and four z pixels ONCE -- all eight loads independent -- instead of the 9 window visits a
// pixel-at-a-time pass makes for the same four pixels, each behind its own z load.
__global__ __launch_bounds__(256) void stem_pool_bwd_quad_kernel(const StemPoolBwdParams p, long items) {
    const int cv = p.C >> 3;
    const int c8 = threadIdx.x % cv;
    const int c = c8 << 3;
    const int per = 256 / cv;                       // items per block iteration
    const int Hk = p.Hz >> 1, Wk = p.Wz >> 1;
    float mu[8], rs[8], sc[8], shf[8], c1[8], c2[8], c3[8];
#define LD8(dst, src) \
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src); \
    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
    LD8(mu, p.mean + c) LD8(rs, p.rstd + c) LD8(sc, p.scale + c) LD8(shf, p.shift + c)
    LD8(c1, p.coef + c) LD8(c2, p.coef + p.C + c) LD8(c3, p.coef + 2 * p.C + c)
#undef LD8
    // item -> (image, quad row, quad column): two 64-bit divisions + two remainders were ~480 of an item's ~1,100 vector instructions
    // (the launch is ALU-bound: 60 us for 318 MB); item counts below 2^21 split exactly with the float reciprocal (vpd_fdiv)
    const bool fast = items < VPD_FDIV_MAX;
    const float rWk = 1.0f / (float)Wk, rHk = 1.0f / (float)Hk;
    for (long it = (long)blockIdx.x * per + threadIdx.x / cv; it < items; it += (long)gridDim.x * per) {
        int kx, ky, b;
        if (fast) {
            const int q = vpd_fdiv((int)it, rWk);
            kx = (int)it - q * Wk;
            b = vpd_fdiv(q, rHk);
            ky = q - b * Hk;
        } else {
            kx = (int)(it % Wk);
            const long q = it / Wk;
            ky = (int)(q % Hk);
            b = (int)(q / Hk);
        }
        // windows w[wy][wx] = (ky + wy, kx + wx); the far ones may fall off the pooled map
        uint4 dw[2][2];
        uint2 iw[2][2];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = ky + wy, ox = kx + wx;
                dw[wy][wx] = uint4{0u, 0u, 0u, 0u};
                iw[wy][wx] = uint2{0xffffffffu, 0xffffffffu};                 // matches no tap
                if (oy < p.Ho && ox < p.Wo) {
                    const size_t po = ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.C + c;
                    dw[wy][wx] = *reinterpret_cast<const uint4*>(p.dpool + po);
                    iw[wy][wx] = *reinterpret_cast<const uint2*>(p.idx + po);
                }
            }
        uint4 zr[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                zr[dy][dx] = *reinterpret_cast<const uint4*>(
                    p.z + ((size_t)(b * p.Hz + 2 * ky + dy) * p.Wz + 2 * kx + dx) * p.C + c);
        float d[2][2][8];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) unpack8(dw[wy][wx], d[wy][wx]);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float z[8], o[8];
                unpack8(zr[dy][dx], z);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float g = 0.f;
                    // pixel row 2ky+dy is tap row dy+1 of window row ky and, for dy = 1, tap row 0 of window row ky+1
#pragma unroll
                    for (int wy = 0; wy <= dy; ++wy)
#pragma unroll
                        for (int wx = 0; wx <= dx; ++wx) {
                            const unsigned want = (unsigned)((dy + 1 - 2 * wy) * 3 + (dx + 1 - 2 * wx));
                            const unsigned word = j < 4 ? iw[wy][wx].x : iw[wy][wx].y;
                            const unsigned sel = (word >> (8 * (j & 3))) & 0xffu;
                            g += sel == want ? d[wy][wx][j] : 0.f;
                        }
                    // (explicit fma nesting, as in bn_bwd_apply_fused_kernel: left to the compiler the contraction of these lines -- and
                    //  with it the stem's dz in the last bit -- changed when the STORE below became a non-temporal one)
                    const float a = __builtin_fmaf(z[j], sc[j], shf[j]);
                    g = a > 0.f ? g : 0.f;
                    const float xh = (z[j] - mu[j]) * rs[j];
                    o[j] = c1[j] * __builtin_fmaf(-xh, c3[j], g - c2[j]);
                }
                vpd_store16<VPD_CP_STEM>(p.dz + ((size_t)(b * p.Hz + 2 * ky + dy) * p.Wz + 2 * kx + dx) * p.C + c, pack8(o));
            }
    }
}

// BatchNorm-backward sums of the stem from the pooled side.  The gradient of a pooling window goes to its arg-max pixel
// only, and survives the ReLU iff the pooled value is positive; there a = gamma * xhat + beta, so
//   sum g = sum_windows d [a > 0],   sum g * xhat = sum_windows d [a > 0] (a - beta) / gamma
// needs dpool and the pooled activation only (67 MB) instead of a pass over z with its 2.25 windows per pixel
// (260 MB, 90 us).  a is the stored bf16 activation: xhat differs from the z-based value by its rounding (2^-9
// relative, zero mean), far below the bf16 noise of the gradients themselves.
__global__ __launch_bounds__(256) void stem_pool_bwd_sums_kernel(const StemPoolBwdParams p) {
    __shared__ float sh[256][17];
    const int cv = p.C >> 3;
    const int ppi = 256 / cv;
    const int c8 = threadIdx.x % cv;
    const int pl = threadIdx.x / cv;
    const int c = c8 << 3;
    const int HWo = p.Ho * p.Wo;
    const int Mo = p.M / (p.Hz * p.Wz) * HWo;                      // pooled pixels of the batch
    float ig[8], be[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gm = p.gamma_p[c + j];
        ig[j] = fabsf(gm) > 1e-20f ? 1.f / gm : 0.f;
        be[j] = p.beta_p[c + j];
        a1[j] = 0.f; a2[j] = 0.f;
    }
    const int pHp = p.Ho + 2 * p.ppad, pWp = p.Wo + 2 * p.ppad;
    const int mbeg = blockIdx.x * p.ppb;
    int mend = mbeg + p.ppb;
    mend = mend < Mo ? mend : Mo;
    const bool fast = Mo < VPD_FDIV_MAX;               // (two integer divisions per item were ~90 of its ~130 vector instructions)
    const float rHWo = 1.0f / (float)HWo, rWo = 1.0f / (float)p.Wo;
    if (pl < ppi)
        for (int m = mbeg + pl; m < mend; m += ppi) {
            const int b = fast ? vpd_fdiv(m, rHWo) : m / HWo;
            const int r = m - b * HWo;
            const int oy = fast ? vpd_fdiv(r, rWo) : r / p.Wo;
            const int ox = r - oy * p.Wo;
            float d[8], a[8];
            unpack8(*reinterpret_cast<const uint4*>(p.dpool + (size_t)m * p.C + c), d);
            unpack8(*reinterpret_cast<const uint4*>(p.pooled + ((size_t)(b * pHp + oy + p.ppad) * pWp + ox + p.ppad) * p.C + c), a);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = a[j] > 0.f ? d[j] : 0.f;
                a1[j] += g;
                a2[j] += g * ((a[j] - be[j]) * ig[j]);
            }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[threadIdx.x][j] = a1[j]; sh[threadIdx.x][8 + j] = a2[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * p.C; t += 256) {
        const int which = t / p.C;
        const int ch = t - which * p.C;
        const int q8 = ch >> 3, j = ch & 7;
        float tot = 0.f;
        for (int gI = 0; gI < ppi; ++gI) tot += sh[gI * cv + q8][which * 8 + j];
        atomicAdd(&p.partials[((size_t)(blockIdx.x & (VPD_STAT_ROWS - 1)) * 2 + which) * p.C + ch], (double)tot);
    }
}

hipError_t vpd_launch_stem_pool_bwd(const StemPoolBwdParams& p0, float count, const float* gamma, float* dgamma,
                                    float* dbeta, float* coef, bf16_t* dz, hipStream_t s) {
    // Two passes over (d_pool, argmax, z): pass 1 = the BN-backward sums of g (max-pool routing + ReLU mask, never
    // materialised); finalize; pass 2 recomputes g and writes dz.  Saves writing and re-reading the 134 MB g tensor.
    StemPoolBwdParams p = p0;
    if (p.C % 8 || 256 % (p.C / 8)) return hipErrorInvalidValue;
    const int T = vpd_bn_bwd_blocks(p.M, p.C, &p.ppb);
    p.pass = 1; p.coef = coef; p.dz = dz;
    static const bool pooled_sums = !(getenv("VPD_STEM_POOLSUMS") && !atoi(getenv("VPD_STEM_POOLSUMS")));
    if (pooled_sums && p.pooled) {
        StemPoolBwdParams q = p;
        const int Mo = p.M / (p.Hz * p.Wz) * p.Ho * p.Wo;
        const int To = vpd_bn_bwd_blocks(Mo, p.C, &q.ppb);
        hipLaunchKernelGGL(stem_pool_bwd_sums_kernel, dim3(To), dim3(256), 0, s, q);
    } else {
        hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(T), dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((p.C + 63) / 64), dim3(64), 0, s, p.partials,
                       VPD_STAT_ROWS, p.C, count,
                       gamma, p.rstd, dgamma, dbeta, coef);
    p.pass = 2;
    static const bool quad = !(getenv("VPD_STEM_QUAD") && !atoi(getenv("VPD_STEM_QUAD")));
    if (quad && !(p.Hz & 1) && !(p.Wz & 1) && p.Ho == p.Hz / 2 && p.Wo == p.Wz / 2) {
        const int per = 256 / (p.C / 8);
        const long items = (long)(p.M / (p.Hz * p.Wz)) * (p.Hz / 2) * (p.Wz / 2);
        long blocks = (items + per - 1) / per;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(stem_pool_bwd_quad_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, p, items);
    } else {
        hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(T), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// ===========================================================================
// Fused BatchNorm passes (one launch per BatchNorm and direction instead of two / three).
//
// Forward:  bn_fwd_fused_kernel = finalize (per-channel sums -> mean, rstd, scale, shift, running statistics) + apply.
//   The sums are complete when the kernel starts (the producing convolution is the previous launch), so every block
//   finalizes the channels itself from the BatchNorm's OWN accumulator rows (VPD_FUSED_ROWS fp64 rows, 4x fewer than the
//   shared rows of the separate finalize kernel: a block reads 2*C*4 doubles, <= 32 KB) into LDS; block 0 also stores
//   mean / rstd / scale / shift for backward and updates the running statistics.  The residual of a down-sampling
//   block (conv1x1 -> BN, res_kind 2) is finalized in the same prologue.
// Backward: bn_bwd_fused_kernel = reduce (sum g, sum g*xhat) + finalize + apply in ONE launch with an in-launch grid
//   barrier (sync.h).  One 1024-thread block per CU owns a contiguous pixel range; g = dy*[mask] (and z when it fits)
//   stays in LDS across the barrier, so dy / z / act are read once instead of twice and two launches (~5 us of fixed
//   cost each at these sizes) disappear.  Per-block partial sums are reduced in a fixed order and added with fp64
//   atomics to the BatchNorm's own rows, as before: results do not depend on the arrival order beyond 1e-16.
// ===========================================================================
#include "sync.h"

static int vpd_bn_xcd_on() {      // VPD_BN_XCD=0: plain grid-stride order (same-box A/B; bit-identical either way)
    static const int on = getenv("VPD_BN_XCD") ? atoi(getenv("VPD_BN_XCD")) : 1;
    return on;
}

// XCD-affine block order of the fused BatchNorm launches (round 5).  conv3x3_pws_kernel runs pixel tile t (BM pixels) on XCD t % 8
// (blocks b and b + 8 share an XCD), writes its z there and reads its halo from there; a BatchNorm launch whose blocks take the
// items of a grid-stride iteration in plain order spreads every such tile over all eight XCDs, so each of the two hand-overs
// (z: conv -> BatchNorm, activation: BatchNorm -> conv) goes through memory.  Here the R = tile_px / (pixels per block and
// iteration) consecutive virtual blocks that cover one tile run on physical blocks of XCD t % 8: both hand-overs can hit that
// XCD's L2 (layer3: 1.0 + 1.6 MB of z + activation per XCD, layer4 half of that).  Same items, same arithmetic: bit-identical.
static __device__ __forceinline__ int vpd_bn_virtual_block(int b, int grid, int R) {
    if (R <= 0 || grid != 256) return b;
    const int x = b & 7, jj = b >> 3;
    const int u = jj / R, r = jj - u * R;
    return (u * 8 + x) * R + r;
}
// R for a launch of `grid` blocks of 1024 threads over items of 8 channels, or 0 when the tile does not map (host side)
static int vpd_bn_xcd_r(int tile_px, int C, int grid) {
    const int cv = C / 8;
    if (tile_px <= 0 || grid != 256 || cv <= 0 || 1024 % cv != 0) return 0;
    const int pb = 1024 / cv;                                   // pixels per block and iteration
    if (tile_px % pb != 0) return 0;
    const int R = tile_px / pb;
    return (R >= 1 && R <= 32 && 32 % R == 0) ? R : 0;
}

struct BnFusedFwdArgs {
    double* rows; float count;                 // this BatchNorm's accumulator rows [VPD_FUSED_ROWS][2][C]
    const float* gamma; const float* beta; float* rm; float* rv;
    float* mean; float* rstd; float* scale; float* shift;
    // residual BatchNorm (res_kind 2) or rows2 == null
    double* rows2; float count2;
    const float* gamma2; const float* beta2; float* rm2; float* rv2;
    float* mean2; float* rstd2; float* scale2; float* shift2;
    float momentum, eps;
    int xcd_r;                                 // vpd_bn_virtual_block
};

__global__ __launch_bounds__(1024) void bn_fwd_fused_kernel(const BnApplyParams p, const BnFusedFwdArgs f) {
    extern __shared__ float sm[];                      // scale[C] shift[C] (rscale[C] rshift[C])
    const int C = p.C;
    float* s_sc = sm; float* s_sh = sm + C; float* s_rsc = sm + 2 * C; float* s_rsh = sm + 3 * C;
    // The first item's operands are requested BEFORE the finalize prologue: on the small tensors (layer3 / layer4: one or two
    // items per thread) the launch is a chain of dependent round trips -- rows -> fp64 finalize -> barrier -> z -> store --
    // and z / the residual do not depend on the first three.
    const int cv = C >> 3;
    const long total = (long)p.M * cv;
    const int HW = p.H * p.W;
    const long stride = (long)gridDim.x * blockDim.x;
    // item -> (pixel, channel slice) -> (image, row, column): the generic form is a 64-bit and two 32-bit integer divisions,
    // ~130 of the ~190 VALU instructions of an item -- at one 1024-thread block per CU that made the launch VALU-bound
    // (~3.4 TB/s on layer1's 71 MB).  Every ResNet width has a power-of-two C / 8 (shift / mask), and pixel counts below
    // 2^21 split exactly with the float reciprocal (vpd_fdiv).
    const bool fast = (cv & (cv - 1)) == 0 && p.M < VPD_FDIV_MAX && total < (1l << 31);
    const int cvs = 31 - __builtin_clz(cv);
    const float rHW = 1.0f / (float)HW, rW = 1.0f / (float)p.W;
    auto load_item = [&](long i, uint4& zv, uint4& rv, size_t& o, int& cc) __attribute__((always_inline)) {
        int m, b, y, x;
        if (fast) {
            m = (int)i >> cvs;
            cc = ((int)i & (cv - 1)) << 3;
            b = vpd_fdiv(m, rHW);
            const int r = m - b * HW;
            y = vpd_fdiv(r, rW);
            x = r - y * p.W;
        } else {
            m = (int)(i / cv);
            cc = (int)(i - (long)m * cv) << 3;
            b = m / HW;
            const int r = m - b * HW;
            y = r / p.W;
            x = r - y * p.W;
        }
        zv = vpd_load16<VPD_CL_BN>(p.z + (size_t)m * C + cc);
        if (p.res_kind == 1) rv = vpd_load16<VPD_CL_BN>(p.res + ((size_t)(b * p.rHp + y + p.rpad) * p.rWp + x + p.rpad) * C + cc);
        else if (p.res_kind == 2) rv = vpd_load16<VPD_CL_BN>(p.res + (size_t)m * C + cc);
        o = ((size_t)(b * p.oHp + y + p.opad) * p.oWp + x + p.opad) * C + cc;
    };
    long it = (long)vpd_bn_virtual_block(blockIdx.x, gridDim.x, f.xcd_r) * blockDim.x + threadIdx.x;
    bool have = it < total;
    uint4 zc = uint4{0u, 0u, 0u, 0u}, rc = zc; size_t oo = 0; int c = 0;
    if (have) load_item(it, zc, rc, oo, c);
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        float mu, r, sc, sh; double var;
        bn_finalize_channel(f.rows, C, ch, f.count, f.eps, f.gamma[ch], f.beta[ch], &mu, &r, &sc, &sh, &var);
        s_sc[ch] = sc; s_sh[ch] = sh;
        if (blockIdx.x == 0) {
            f.mean[ch] = mu; f.rstd[ch] = r; f.scale[ch] = sc; f.shift[ch] = sh;
            if (f.rm) {
                const double unb = f.count > 1.f ? var * (double)f.count / ((double)f.count - 1.0) : var;
                f.rm[ch] = (1.f - f.momentum) * f.rm[ch] + f.momentum * mu;
                f.rv[ch] = (1.f - f.momentum) * f.rv[ch] + f.momentum * (float)unb;
            }
        }
        if (f.rows2) {
            bn_finalize_channel(f.rows2, C, ch, f.count2, f.eps, f.gamma2[ch], f.beta2[ch], &mu, &r, &sc, &sh, &var);
            s_rsc[ch] = sc; s_rsh[ch] = sh;
            if (blockIdx.x == 0) {
                f.mean2[ch] = mu; f.rstd2[ch] = r; f.scale2[ch] = sc; f.shift2[ch] = sh;
                if (f.rm2) {
                    const double unb = f.count2 > 1.f ? var * (double)f.count2 / ((double)f.count2 - 1.0) : var;
                    f.rm2[ch] = (1.f - f.momentum) * f.rm2[ch] + f.momentum * mu;
                    f.rv2[ch] = (1.f - f.momentum) * f.rv2[ch] + f.momentum * (float)unb;
                }
            }
        }
    }
    __syncthreads();
    // fast path: the stride (grid x 1024) is a multiple of C / 8, so a thread keeps its channel slice for every item it takes -- its
    // sixteen coefficients are read from the LDS once, not per item
    float ksc[8], ksh[8];
    if (fast && have) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { ksc[j] = s_sc[c + j]; ksh[j] = s_sh[c + j]; }
    }
    while (have) {
        // the next item's operands are requested before this one is computed (two items in flight per thread)
        const long nx = it + stride;
        const bool hn = nx < total;
        uint4 zn = zc, rn = rc; size_t on = oo; int cn = c;
        if (hn) load_item(nx, zn, rn, on, cn);
        float v[8];
        unpack8(zc, v);
        if (fast) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * ksc[j] + ksh[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * s_sc[c + j] + s_sh[c + j];
        }
        if (p.res_kind == 1) {
            float rr[8];
            unpack8(rc, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j];
        } else if (p.res_kind == 2) {
            float rr[8];
            unpack8(rc, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j] * s_rsc[c + j] + s_rsh[c + j];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        const uint4 ov = pack8(v);
        if (C > 64) vpd_store16<VPD_CP_BNF>(p.out + oo, ov); else vpd_store16<VPD_CP_BNF64>(p.out + oo, ov);
        if (p.mask_out) {      // bit j = the STORED bf16 value is > 0 (values are >= 0 after the ReLU: non-zero bits)
            const unsigned w[4] = {ov.x, ov.y, ov.z, ov.w};
            unsigned bits = 0;
            if (p.relu) {
                // after the ReLU a stored value is +0 or positive: [half != 0] = min(half, 1) as unsigned 16-bit lanes (one instruction
                // per word instead of two compares and two selects per element); the four words' flags side by side -- low halves at
                // bits 0, 2, 4, 6, high halves at 16, 18, 20, 22 -- then the high halves folded one bit above the low ones
                unsigned acc = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned f1;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(f1) : "v"(w[j]), "v"(0x00010001u));
                    acc |= f1 << (2 * j);
                }
                bits = (acc | (acc >> 15)) & 0xffu;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bits |= ((w[j] & 0x7fffu) != 0u && !(w[j] & 0x8000u)) ? (1u << (2 * j)) : 0u;
                    bits |= ((w[j] & 0x7fff0000u) != 0u && !(w[j] & 0x80000000u)) ? (2u << (2 * j)) : 0u;
                }
            }
            p.mask_out[it] = (unsigned char)bits;      // it = m * (C / 8) + c / 8
        }
        it = nx; have = hn; zc = zn; rc = rn; oo = on; c = cn;
    }
}

hipError_t vpd_launch_bn_fwd_fused(const BnApplyParams& p, const BnFusedFwd& f0, hipStream_t s) {
    if (p.C % 8 || p.C > 4096) return hipErrorInvalidValue;
    BnFusedFwdArgs f;
    f.rows = f0.rows; f.count = f0.count; f.gamma = f0.gamma; f.beta = f0.beta; f.rm = f0.rm; f.rv = f0.rv;
    f.mean = f0.mean; f.rstd = f0.rstd; f.scale = f0.scale; f.shift = f0.shift;
    f.rows2 = f0.rows2; f.count2 = f0.count2; f.gamma2 = f0.gamma2; f.beta2 = f0.beta2; f.rm2 = f0.rm2; f.rv2 = f0.rv2;
    f.mean2 = f0.mean2; f.rstd2 = f0.rstd2; f.scale2 = f0.scale2; f.shift2 = f0.shift2;
    f.momentum = f0.momentum; f.eps = f0.eps;
    const long items = (long)p.M * (p.C / 8);
    long g = (items + 1023) / 1024;
    // every block pays the finalize prologue (2*C*VPD_FUSED_ROWS doubles): few, fat blocks -- one per CU
    if (g > 256) g = 256;      // (one 1024-thread block per CU: same-box -10 us per step against two; 192 or fewer: +110 us)
    if (g < 1) g = 1;
    const int xcd_on = vpd_bn_xcd_on();
    f.xcd_r = xcd_on ? vpd_bn_xcd_r(p.xcd_tile_px, p.C, (int)g) : 0;
#ifdef VPD_ENABLE_ABLATE      // tools/bench_bn_chain.py: the operator-level entry point knows no neighbouring convolution
    if (const char* e = getenv("VPD_BN_XCD_FORCE")) f.xcd_r = vpd_bn_xcd_r(atoi(e), p.C, (int)g);
#endif
    hipLaunchKernelGGL(bn_fwd_fused_kernel, dim3((unsigned)g), dim3(1024), (size_t)4 * p.C * sizeof(float), s, p, f);
    return hipGetLastError();
}

struct BnFusedBwdArgs {
    double* rows;                              // this BatchNorm's accumulator rows [VPD_FUSED_ROWS][2][C], zeroed
    GridSync* sync;                            // zeroed
    unsigned* err;                             // sticky time-out counter
    const float* gamma; float* dgamma; float* dbeta;
    float count;
    int keep_g, keep_z, iters;                 // LDS residency of g / z across the barrier; pixel iterations per thread
};

template <int MASK, int WRITE_G>
__global__ __launch_bounds__(1024) void bn_bwd_fused_kernel(const BnBwdParams p, const BnFusedBwdArgs f) {
    extern __shared__ uint4 smem4[];
    const int T = 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, cv = C >> 3;
    const int ppi = T / cv;
    const int c8 = tid % cv, pl = tid / cv, c = c8 << 3;
    const int HW = p.H * p.W;
    uint4* sG = smem4;
    uint4* sZ = sG + (f.keep_g ? (size_t)f.iters * T : 0);
    float* red = reinterpret_cast<float*>(sZ + (f.keep_z ? (size_t)f.iters * T : 0));      // [16 waves][2][C], then coef [2][C]

    float mu[8], rs[8], msc[8], msh[8], a1[8], a2[8];
#define LD8(dst, src) \
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src); \
    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
    LD8(mu, p.mean + c) LD8(rs, p.rstd + c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { msc[j] = 0.f; msh[j] = 1.f; a1[j] = 0.f; a2[j] = 0.f; }
    if (MASK == 2) { LD8(msc, p.mscale + c) LD8(msh, p.mshift + c) }

    const int mbeg = blockIdx.x * p.ppb;
    int mend = mbeg + p.ppb;
    mend = mend < p.M ? mend : p.M;
    // ---- phase 1: g = dy * mask, per-thread partial sums; g (and z) parked in LDS ----
    int it = 0;
    for (int m = mbeg + pl; m < mend; m += ppi, ++it) {
        float g[8], z[8];
        const uint4 zr = *reinterpret_cast<const uint4*>(p.z + (size_t)m * C + c);
        uint4 gr;
        if (MASK == 3 && !WRITE_G && p.dy_pooled) {         // dy = gradient of the global average pool, produced here
            const float* d = p.dy_pooled + (size_t)(m / HW) * C + c;
            const float4 d0 = *reinterpret_cast<const float4*>(d), d1 = *reinterpret_cast<const float4*>(d + 4);
            const float dv[8] = {d0.x * p.dy_pool_scale, d0.y * p.dy_pool_scale, d0.z * p.dy_pool_scale, d0.w * p.dy_pool_scale,
                                 d1.x * p.dy_pool_scale, d1.y * p.dy_pool_scale, d1.z * p.dy_pool_scale, d1.w * p.dy_pool_scale};
            gr = pack8(dv);
            *reinterpret_cast<uint4*>(p.dy_rw + (size_t)m * C + c) = gr;
        } else {
            gr = *reinterpret_cast<const uint4*>(p.dy + (size_t)m * C + c);
        }
        unpack8(gr, g);
        unpack8(zr, z);
        if (MASK == 1) {
            const int b = m / HW;
            const int r = m - b * HW;
            const int y = r / p.W;
            const int x = r - y * p.W;
            float a[8];
            unpack8(*reinterpret_cast<const uint4*>(p.act + ((size_t)(b * p.aHp + y + p.apad) * p.aWp + x + p.apad) * C + c), a);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = a[j] > 0.f ? g[j] : 0.f;
            gr = pack8(g);                                  // exact: g is dy or 0
            if (WRITE_G) *reinterpret_cast<uint4*>(p.dy_rw + (size_t)m * C + c) = gr;
        } else if (MASK == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = (z[j] * msc[j] + msh[j]) > 0.f ? g[j] : 0.f;
            gr = pack8(g);
        } else if (MASK == 3) {
            const unsigned bits = p.mask_bits[(size_t)m * cv + c8];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = ((bits >> j) & 1u) ? g[j] : 0.f;
            gr = pack8(g);
        }
        if (f.keep_g) sG[(size_t)it * T + tid] = gr;
        if (f.keep_z) sZ[(size_t)it * T + tid] = zr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a1[j] += g[j];
            a2[j] += g[j] * ((z[j] - mu[j]) * rs[j]);
        }
    }
    // ---- block reduction in a fixed order: lanes of a wave that share a channel group, then the 16 waves ----
    if (cv < 64) {
        for (int o = cv; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a1[j] += __shfl_xor(a1[j], o, 64); a2[j] += __shfl_xor(a2[j], o, 64); }
        }
    }
    // waves that hold the same channel groups: all 16 when cv <= 64, every (cv/64)-th otherwise (waves j, j + wstep, ...: their
    // sums go to rows 0, 1, ... of `red`, [16 / wstep][2][C] floats -- 64 KB at most, also for 2,048 channels)
    const int wstep = cv <= 64 ? 1 : cv / 64;
    if (lane < cv) {                                        // (cv >= 64: every lane; its channel group is c8)
        const int rrow = wave / wstep;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(size_t)(rrow * 2 + 0) * C + c + j] = a1[j];
            red[(size_t)(rrow * 2 + 1) * C + c + j] = a2[j];
        }
    }
    __syncthreads();
    for (int t = tid; t < 2 * C; t += T) {
        const int which = t / C;
        const int ch = t - which * C;
        float tot = 0.f;
        for (int k = 0; k < 16 / wstep; ++k) tot += red[(size_t)(k * 2 + which) * C + ch];
        atomicAdd(&f.rows[((size_t)(blockIdx.x & (VPD_FUSED_ROWS - 1)) * 2 + which) * C + ch], (double)tot);
    }
    vpd_grid_barrier(f.sync, false, f.err, blockIdx.x, gridDim.x);
    // ---- finalize: every block sums the rows of all channels (2*C*VPD_FUSED_ROWS doubles) ----
    for (int t = tid; t < 2 * C; t += T) {
        const int which = t / C;
        const int ch = t - which * C;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < VPD_FUSED_ROWS; ++r)
            s += __hip_atomic_load(&f.rows[((size_t)r * 2 + which) * C + ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[t] = (float)(s / (double)f.count);
        if (blockIdx.x == 0) {
            if (which == 0) f.dbeta[ch] = (float)s;
            else f.dgamma[ch] = (float)s;
        }
    }
    __syncthreads();
    float c1[8], c2[8], c3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c1[j] = f.gamma[c + j] * rs[j];
        c2[j] = red[c + j];
        c3[j] = red[C + c + j];
    }
    // ---- phase 2: dz = c1 * (g - c2 - xhat * c3) ----
    it = 0;
    for (int m = mbeg + pl; m < mend; m += ppi, ++it) {
        float g[8], z[8];
        const uint4 zr = f.keep_z ? sZ[(size_t)it * T + tid] : *reinterpret_cast<const uint4*>(p.z + (size_t)m * C + c);
        unpack8(zr, z);
        if (f.keep_g) {
            unpack8(sG[(size_t)it * T + tid], g);
        } else {
            // not resident: g was written back over dy by this very thread (WRITE_G), or is recomputed from dy and z
            unpack8(*reinterpret_cast<const uint4*>(p.dy + (size_t)m * C + c), g);
            if (MASK == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = (z[j] * msc[j] + msh[j]) > 0.f ? g[j] : 0.f;
            } else if (MASK == 3) {
                const unsigned bits = p.mask_bits[(size_t)m * cv + c8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = ((bits >> j) & 1u) ? g[j] : 0.f;
            }
        }
        const int b = m / HW;
        const int r = m - b * HW;
        const int y = r / p.W;
        const int x = r - y * p.W;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = c1[j] * (g[j] - c2[j] - (z[j] - mu[j]) * rs[j] * c3[j]);
        const size_t oo = ((size_t)(b * p.dzHp + y + p.dzpad) * p.dzWp + x + p.dzpad) * C + c;
        vpd_store16<VPD_CP_BNB>(p.dz + oo, pack8(o));
    }
#undef LD8
}

// BatchNorm backward, finalize + apply only: the data-gradient convolution that produced dy has added sum g and sum g * z
// (g = dy * mask) to this BatchNorm's rows in its epilogue (conv_epilogue.h, EPM 6 / 7), so the reduction pass, the atomics
// and the grid barrier of bn_bwd_fused_kernel are gone.  sum g * xhat = (sum g*z - mean * sum g) * rstd (fp64), and
// dz = c1 * (g - c2 - xhat * c3) is evaluated as A*g + B*z + D with per-channel A = gamma*rstd, B = -A*rstd*c3,
// D = -A*c2 - B*mean.
struct BnBwdApplyArgs {
    const double* rows; float count;
    const float* gamma; const float* mean; const float* rstd; float* dgamma; float* dbeta;
    // PAIR: a second BatchNorm fed with the same g (the 1x1 branch of a down-sampling block)
    const double* rows2; const float* gamma2; const float* mean2; const float* rstd2; float* dgamma2; float* dbeta2;
    const bf16_t* z2; bf16_t* dz2;
    int xcd_r;                                 // vpd_bn_virtual_block
};
template <bool PAIR>
__global__ __launch_bounds__(1024) void bn_bwd_apply_fused_kernel(const BnBwdParams p, const BnBwdApplyArgs f) {
    extern __shared__ float sm[];                      // A[C] B[C] D[C] (A2[C] B2[C] D2[C])
    const int C = p.C;
    float* sA = sm; float* sB = sm + C; float* sD = sm + 2 * C;
    // as in bn_fwd_fused_kernel: the first item's operands are requested before the coefficient prologue, the next item's
    // before the current one is computed
    const int cv = C >> 3;
    const long total = (long)p.M * cv;
    const int HW = p.H * p.W;
    const long stride = (long)gridDim.x * blockDim.x;
    const bool fast = (cv & (cv - 1)) == 0 && p.M < VPD_FDIV_MAX && total < (1l << 31);      // as in bn_fwd_fused_kernel
    const int cvs = 31 - __builtin_clz(cv);
    const float rHW = 1.0f / (float)HW, rW = 1.0f / (float)p.W;
    auto load_item = [&](long i, uint4& gv, uint4& zv, uint4& z2v, unsigned& bits, size_t& o, int& cc) __attribute__((always_inline)) {
        int m, b, y, x;
        if (fast) {
            m = (int)i >> cvs;
            cc = ((int)i & (cv - 1)) << 3;
            b = vpd_fdiv(m, rHW);
            const int r = m - b * HW;
            y = vpd_fdiv(r, rW);
            x = r - y * p.W;
        } else {
            m = (int)(i / cv);
            cc = (int)(i - (long)m * cv) << 3;
            b = m / HW;
            const int r = m - b * HW;
            y = r / p.W;
            x = r - y * p.W;
        }
        gv = vpd_load16<VPD_CL_BN>(p.dy + (size_t)m * C + cc);
        zv = vpd_load16<VPD_CL_BN>(p.z + (size_t)m * C + cc);
        if (PAIR) z2v = vpd_load16<VPD_CL_BN>(f.z2 + (size_t)m * C + cc);
        bits = p.mask_bits[i];                                     // i = m * (C / 8) + c / 8
        o = ((size_t)(b * p.dzHp + y + p.dzpad) * p.dzWp + x + p.dzpad) * C + cc;
    };
    long it = (long)vpd_bn_virtual_block(blockIdx.x, gridDim.x, f.xcd_r) * blockDim.x + threadIdx.x;
    bool have = it < total;
    uint4 gc = uint4{0u, 0u, 0u, 0u}, zc = gc, z2c = gc; unsigned bc = 0; size_t oo = 0; int c = 0;
    if (have) load_item(it, gc, zc, z2c, bc, oo, c);
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        bn_bwd_apply_coef(f.rows, C, ch, f.count, f.gamma[ch], f.mean[ch], f.rstd[ch], sA, sB, sD, f.dgamma, f.dbeta,
                          blockIdx.x == 0);
        if (PAIR)
            bn_bwd_apply_coef(f.rows2, C, ch, f.count, f.gamma2[ch], f.mean2[ch], f.rstd2[ch], sA + 3 * C, sB + 3 * C,
                              sD + 3 * C, f.dgamma2, f.dbeta2, blockIdx.x == 0);
    }
    __syncthreads();
    // fast path: the stride (grid x 1024) is a multiple of C / 8 -- a thread's channel slice, and with it its 24 (48) coefficients, is
    // the same for every item it takes: read from the LDS once
    float kA[8], kB[8], kD[8], kA2[8], kB2[8], kD2[8];
    if (fast && have) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kA[j] = sA[c + j]; kB[j] = sB[c + j]; kD[j] = sD[c + j];
            if (PAIR) { kA2[j] = sA[3 * C + c + j]; kB2[j] = sB[3 * C + c + j]; kD2[j] = sD[3 * C + c + j]; }
        }
    }
    while (have) {
        const long nx = it + stride;
        const bool hn = nx < total;
        uint4 gn = gc, zn = zc, z2n = z2c; unsigned bn_ = bc; size_t on = oo; int cn = c;
        if (hn) load_item(nx, gn, zn, z2n, bn_, on, cn);
        float g[8], z[8], o[8];
        unpack8(gc, g);
        unpack8(zc, z);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = ((bc >> j) & 1u) ? g[j] : 0.f;
        // (explicit fma nesting in both branches: left to the compiler, the contraction of A g + B z + D depended on where the
        //  operands came from, and every bit of dz with it)
        if (fast) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(kA[j], g[j], __builtin_fmaf(kB[j], z[j], kD[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(sA[c + j], g[j], __builtin_fmaf(sB[c + j], z[j], sD[c + j]));
        }
        if (C > 64) vpd_store16<VPD_CP_BNB>(p.dz + oo, pack8(o)); else vpd_store16<VPD_CP_BNB64>(p.dz + oo, pack8(o));
        if (PAIR) {      // same g, the branch's own z and coefficients, same padded geometry
            unpack8(z2c, z);
            if (fast) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(kA2[j], g[j], __builtin_fmaf(kB2[j], z[j], kD2[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = __builtin_fmaf(sA[3 * C + c + j], g[j], __builtin_fmaf(sB[3 * C + c + j], z[j], sD[3 * C + c + j]));
            }
            vpd_store16<VPD_CP_BNB>(f.dz2 + oo, pack8(o));
        }
        it = nx; have = hn; gc = gn; zc = zn; z2c = z2n; bc = bn_; oo = on; c = cn;
    }
}

hipError_t vpd_launch_bn_bwd_apply_fused(const BnBwdParams& p, const BnFusedBwd& f0, hipStream_t s, const BnFusedBwd* fB,
                                         const bf16_t* zB, const float* meanB, const float* rstdB, bf16_t* dzB) {
    if (p.C % 8 || p.C > 4096 || !p.mask_bits) return hipErrorInvalidValue;
    BnBwdApplyArgs f;
    f = BnBwdApplyArgs{};
    f.rows = f0.rows; f.count = f0.count; f.gamma = f0.gamma; f.mean = p.mean; f.rstd = p.rstd;
    f.dgamma = f0.dgamma; f.dbeta = f0.dbeta;
    const long items = (long)p.M * (p.C / 8);
    long g = (items + 1023) / 1024;
    if (g > 256) g = 256;      // (one 1024-thread block per CU: same-box -10 us per step against two; 192 or fewer: +110 us)
    if (g < 1) g = 1;
    const int xcd_on = vpd_bn_xcd_on();
    f.xcd_r = xcd_on ? vpd_bn_xcd_r(p.xcd_tile_px, p.C, (int)g) : 0;
    if (fB) {
        f.rows2 = fB->rows; f.gamma2 = fB->gamma; f.mean2 = meanB; f.rstd2 = rstdB; f.dgamma2 = fB->dgamma; f.dbeta2 = fB->dbeta;
        f.z2 = zB; f.dz2 = dzB;
        hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<true>, dim3((unsigned)g), dim3(1024), (size_t)6 * p.C * sizeof(float), s, p, f);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<false>, dim3((unsigned)g), dim3(1024), (size_t)3 * p.C * sizeof(float), s, p, f);
    }
    return hipGetLastError();
}

// Two BatchNorm backwards that share their output gradient in one launch: the block-output BatchNorm of a down-sampling
// BasicBlock (A: conv2's, ReLU mask from the stored block output) and the BatchNorm of its 1x1 branch (B: no ReLU of its
// own, fed with the same masked gradient g).  One read of dy / act, three sums per channel (sum g shared), one grid
// barrier, two dz outputs; g itself is not needed afterwards (both paths continue through convolutions).
struct BnFusedBwd2Args {
    double* rowsA; double* rowsB;              // [VPD_FUSED_ROWS][2][C] each, zeroed: A = (sum g, sum g xhatA), B = (-, sum g xhatB)
    GridSync* sync; unsigned* err;
    const bf16_t* zB; const float* meanB; const float* rstdB;
    const float* gammaA; float* dgammaA; float* dbetaA;
    const float* gammaB; float* dgammaB; float* dbetaB;
    bf16_t* dzB;                               // same padded geometry as p.dz
    float count;
    int keep_g, keep_zA, keep_zB, iters;
};

__global__ __launch_bounds__(1024) void bn_bwd_fused2_kernel(const BnBwdParams p, const BnFusedBwd2Args f) {
    extern __shared__ uint4 smem4[];
    const int T = 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, cv = C >> 3;
    const int ppi = T / cv;
    const int c8 = tid % cv, pl = tid / cv, c = c8 << 3;
    const int HW = p.H * p.W;
    uint4* sG = smem4;
    uint4* sZA = sG + (f.keep_g ? (size_t)f.iters * T : 0);
    uint4* sZB = sZA + (f.keep_zA ? (size_t)f.iters * T : 0);
    float* red = reinterpret_cast<float*>(sZB + (f.keep_zB ? (size_t)f.iters * T : 0));      // [16 waves][3][C], then coef [3][C]

    float muA[8], rsA[8], muB[8], rsB[8], a1[8], a2[8], a3[8];
#define LD8(dst, src) \
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src); \
    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
    LD8(muA, p.mean + c) LD8(rsA, p.rstd + c) LD8(muB, f.meanB + c) LD8(rsB, f.rstdB + c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { a1[j] = 0.f; a2[j] = 0.f; a3[j] = 0.f; }
    const int mbeg = blockIdx.x * p.ppb;
    int mend = mbeg + p.ppb;
    mend = mend < p.M ? mend : p.M;
    int it = 0;
    for (int m = mbeg + pl; m < mend; m += ppi, ++it) {
        float g[8], zA[8], zB[8], a[8];
        const uint4 zAr = *reinterpret_cast<const uint4*>(p.z + (size_t)m * C + c);
        const uint4 zBr = *reinterpret_cast<const uint4*>(f.zB + (size_t)m * C + c);
        uint4 gr = *reinterpret_cast<const uint4*>(p.dy + (size_t)m * C + c);
        const int b = m / HW;
        const int r = m - b * HW;
        const int y = r / p.W;
        const int x = r - y * p.W;
        unpack8(*reinterpret_cast<const uint4*>(p.act + ((size_t)(b * p.aHp + y + p.apad) * p.aWp + x + p.apad) * C + c), a);
        unpack8(gr, g); unpack8(zAr, zA); unpack8(zBr, zB);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = a[j] > 0.f ? g[j] : 0.f;
        gr = pack8(g);                                      // exact: g is dy or 0
        if (f.keep_g) sG[(size_t)it * T + tid] = gr;
        else *reinterpret_cast<uint4*>(p.dy_rw + (size_t)m * C + c) = gr;      // parked in dy itself (this thread re-reads it)
        if (f.keep_zA) sZA[(size_t)it * T + tid] = zAr;
        if (f.keep_zB) sZB[(size_t)it * T + tid] = zBr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a1[j] += g[j];
            a2[j] += g[j] * ((zA[j] - muA[j]) * rsA[j]);
            a3[j] += g[j] * ((zB[j] - muB[j]) * rsB[j]);
        }
    }
    if (cv < 64) {
        for (int o = cv; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a1[j] += __shfl_xor(a1[j], o, 64); a2[j] += __shfl_xor(a2[j], o, 64); a3[j] += __shfl_xor(a3[j], o, 64);
            }
        }
    }
    const int wstep = cv <= 64 ? 1 : cv / 64;      // (rows of `red` by the waves that share a channel group: as bn_bwd_fused_kernel)
    if (lane < cv) {
        const int rrow = wave / wstep;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(size_t)(rrow * 3 + 0) * C + c + j] = a1[j];
            red[(size_t)(rrow * 3 + 1) * C + c + j] = a2[j];
            red[(size_t)(rrow * 3 + 2) * C + c + j] = a3[j];
        }
    }
    __syncthreads();
    for (int t = tid; t < 3 * C; t += T) {
        const int which = t / C;
        const int ch = t - which * C;
        float tot = 0.f;
        for (int k = 0; k < 16 / wstep; ++k) tot += red[(size_t)(k * 3 + which) * C + ch];
        double* rows = which == 2 ? f.rowsB : f.rowsA;
        const int slot = which == 2 ? 1 : which;
        atomicAdd(&rows[((size_t)(blockIdx.x & (VPD_FUSED_ROWS - 1)) * 2 + slot) * C + ch], (double)tot);
    }
    vpd_grid_barrier(f.sync, false, f.err, blockIdx.x, gridDim.x);
    for (int t = tid; t < 3 * C; t += T) {
        const int which = t / C;
        const int ch = t - which * C;
        const double* rows = which == 2 ? f.rowsB : f.rowsA;
        const int slot = which == 2 ? 1 : which;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < VPD_FUSED_ROWS; ++r)
            s += __hip_atomic_load(&rows[((size_t)r * 2 + slot) * C + ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[t] = (float)(s / (double)f.count);
        if (blockIdx.x == 0) {
            if (which == 0) { f.dbetaA[ch] = (float)s; f.dbetaB[ch] = (float)s; }
            else if (which == 1) f.dgammaA[ch] = (float)s;
            else f.dgammaB[ch] = (float)s;
        }
    }
    __syncthreads();
    float c1A[8], c1B[8], c2[8], c3A[8], c3B[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c1A[j] = f.gammaA[c + j] * rsA[j];
        c1B[j] = f.gammaB[c + j] * rsB[j];
        c2[j] = red[c + j];
        c3A[j] = red[C + c + j];
        c3B[j] = red[2 * C + c + j];
    }
    it = 0;
    for (int m = mbeg + pl; m < mend; m += ppi, ++it) {
        float g[8], zA[8], zB[8];
        unpack8(f.keep_zA ? sZA[(size_t)it * T + tid] : *reinterpret_cast<const uint4*>(p.z + (size_t)m * C + c), zA);
        unpack8(f.keep_zB ? sZB[(size_t)it * T + tid] : *reinterpret_cast<const uint4*>(f.zB + (size_t)m * C + c), zB);
        unpack8(f.keep_g ? sG[(size_t)it * T + tid] : *reinterpret_cast<const uint4*>(p.dy + (size_t)m * C + c), g);
        const int b = m / HW;
        const int r = m - b * HW;
        const int y = r / p.W;
        const int x = r - y * p.W;
        float oA[8], oB[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            oA[j] = c1A[j] * (g[j] - c2[j] - (zA[j] - muA[j]) * rsA[j] * c3A[j]);
            oB[j] = c1B[j] * (g[j] - c2[j] - (zB[j] - muB[j]) * rsB[j] * c3B[j]);
        }
        const size_t oo = ((size_t)(b * p.dzHp + y + p.dzpad) * p.dzWp + x + p.dzpad) * C + c;
        *reinterpret_cast<uint4*>(p.dz + oo) = pack8(oA);
        *reinterpret_cast<uint4*>(f.dzB + oo) = pack8(oB);
    }
#undef LD8
}

bool vpd_bn_bwd_fused2_ok(int M, int C) {
    static const int off = getenv("VPD_FUSED_BN") ? !atoi(getenv("VPD_FUSED_BN")) : 0;
    static const int off2 = getenv("VPD_BN_PAIR") ? !atoi(getenv("VPD_BN_PAIR")) : 0;
    return !(off || off2 || C % 8 || C < 64 || C > 2048 || 1024 % (C / 8)) && M >= 1;
}

// p: BatchNorm A as for vpd_launch_bn_bwd_fused (dy, act, z, mean, rstd, dz + geometry); fA / fB: rows, gamma, dgamma, dbeta
// of the two BatchNorms (fA.sync / err / count are used); zB / meanB / rstdB / dzB: BatchNorm B's tensors
hipError_t vpd_launch_bn_bwd_fused2(const BnBwdParams& p0, const BnFusedBwd& fA, const BnFusedBwd& fB, const bf16_t* zB,
                                    const float* meanB, const float* rstdB, bf16_t* dzB, hipStream_t s) {
    BnBwdParams p = p0;
    if (!p.act) return hipErrorInvalidValue;
    const int ncu = vpd_cu_budget();
    const int cv = p.C / 8, ppi = 1024 / cv;
    int G = ncu;
    int ppb = (p.M + G - 1) / G;
    ppb = ((ppb + ppi - 1) / ppi) * ppi;
    G = (p.M + ppb - 1) / ppb;
    p.ppb = ppb;
    BnFusedBwd2Args f;
    f.rowsA = fA.rows; f.rowsB = fB.rows; f.sync = reinterpret_cast<GridSync*>(fA.sync); f.err = fA.err;
    f.zB = zB; f.meanB = meanB; f.rstdB = rstdB;
    f.gammaA = fA.gamma; f.dgammaA = fA.dgamma; f.dbetaA = fA.dbeta;
    f.gammaB = fB.gamma; f.dgammaB = fB.dgamma; f.dbetaB = fB.dbeta;
    f.dzB = dzB; f.count = fA.count;
    f.iters = ppb / ppi;
    const size_t red_bytes = (size_t)(cv <= 64 ? 16 : 16 / (cv / 64)) * 3 * p.C * sizeof(float);
    const size_t tile = (size_t)f.iters * 1024 * 16;
    const size_t cap = 160 * 1024;
    f.keep_g = red_bytes + tile <= cap;
    f.keep_zA = f.keep_g && red_bytes + 2 * tile <= cap;
    f.keep_zB = f.keep_zA && red_bytes + 3 * tile <= cap;
    const size_t lds = red_bytes + (f.keep_g ? tile : 0) + (f.keep_zA ? tile : 0) + (f.keep_zB ? tile : 0);
    hipLaunchKernelGGL(bn_bwd_fused2_kernel, dim3(G), dim3(1024), lds, s, p, f);
    return hipGetLastError();
}

// false: this shape has to take the three-launch path (vpd_launch_bn_bwd)
bool vpd_bn_bwd_fused_ok(int M, int C, bool mask_act, bool write_g) {
    static const int off = getenv("VPD_FUSED_BN") ? !atoi(getenv("VPD_FUSED_BN")) : 0;
    if (off || C % 8 || C < 64 || C > 2048 || 1024 % (C / 8)) return false;
    if (mask_act && !write_g) return false;             // (no caller: the masked g could not be recovered in phase 2)
    return M >= 1;
}

hipError_t vpd_launch_bn_bwd_fused(const BnBwdParams& p0, const BnFusedBwd& f0, hipStream_t s) {
    BnBwdParams p = p0;
    const int ncu = vpd_cu_budget();
    const int cv = p.C / 8, ppi = 1024 / cv;
    int G = ncu;                                        // one 1024-thread block per CU: the whole grid is resident
    {
    }
    int ppb = (p.M + G - 1) / G;
    ppb = ((ppb + ppi - 1) / ppi) * ppi;
    G = (p.M + ppb - 1) / ppb;
    p.ppb = ppb;
    BnFusedBwdArgs f;
    f.rows = f0.rows; f.sync = reinterpret_cast<GridSync*>(f0.sync); f.err = f0.err; f.gamma = f0.gamma; f.dgamma = f0.dgamma; f.dbeta = f0.dbeta;
    f.count = f0.count;
    f.iters = ppb / ppi;
    const size_t red_bytes = (size_t)(cv <= 64 ? 16 : 16 / (cv / 64)) * 2 * p.C * sizeof(float);
    const size_t tile = (size_t)f.iters * 1024 * 16;    // bytes of one resident tensor slice
    const size_t cap = 160 * 1024;
    const int mask = p.mask_bits ? 3 : (p.act ? 1 : (p.mscale ? 2 : 0));
    f.keep_g = red_bytes + tile <= cap;
    f.keep_z = f.keep_g && red_bytes + 2 * tile <= cap;
    if (!f.keep_g && mask == 1 && !p.write_g) return hipErrorInvalidValue;
    const size_t lds = red_bytes + (f.keep_g ? tile : 0) + (f.keep_z ? tile : 0);
    if (mask == 3) hipLaunchKernelGGL((bn_bwd_fused_kernel<3, 0>), dim3(G), dim3(1024), lds, s, p, f);
    else if (mask == 1 && p.write_g) hipLaunchKernelGGL((bn_bwd_fused_kernel<1, 1>), dim3(G), dim3(1024), lds, s, p, f);
    else if (mask == 1) hipLaunchKernelGGL((bn_bwd_fused_kernel<1, 0>), dim3(G), dim3(1024), lds, s, p, f);
    else if (mask == 2) hipLaunchKernelGGL((bn_bwd_fused_kernel<2, 0>), dim3(G), dim3(1024), lds, s, p, f);
    else hipLaunchKernelGGL((bn_bwd_fused_kernel<0, 0>), dim3(G), dim3(1024), lds, s, p, f);
    return hipGetLastError();
}
