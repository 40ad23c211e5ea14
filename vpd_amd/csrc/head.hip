// Embedding head: global average pool, the fc / motion-MLP linears (small fp32
// GEMMs, <0.1 % of the FLOPs), sum-MSE loss and its gradient.
#include "common.h"
#include "kernels.h"

// pooled[b][c] = mean over the H*W interior pixels of a padded activation
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16_t* act, int Hp, int Wp, int pad, int H, int W, int C,
                                                      int N, float* pooled) {
    const int cv = C >> 3;
    const long total = (long)N * cv;
    const float inv = 1.f / (float)(H * W);
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int b = (int)(it / cv);
        const int c = (int)(it - (long)b * cv) << 3;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (H == 4 && W == 4) {      // (128 x 128 crops: all sixteen pixels requested before the first is added -- one round trip, not 16)
            uint4 px[16];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                px[q] = *reinterpret_cast<const uint4*>(act + ((size_t)(b * Hp + (q >> 2) + pad) * Wp + (q & 3) + pad) * C + c);
#pragma unroll
            for (int q = 0; q < 16; ++q) {      // same order of additions as the general loop
                float v[8];
                unpack8(px[q], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        } else {
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(act + ((size_t)(b * Hp + y + pad) * Wp + x + pad) * C + c), v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
        }
        float* o = pooled + (size_t)b * C + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = acc[j] * inv;
    }
}
hipError_t vpd_launch_avgpool(const bf16_t* act, int Hp, int Wp, int pad, int H, int W, int C, int N, float* pooled,
                              hipStream_t s) {
    long items = (long)N * (C / 8);
    int g = (int)((items + 255) / 256);
    hipLaunchKernelGGL(avgpool_kernel, dim3(g < 1 ? 1 : g), dim3(256), 0, s, act, Hp, Wp, pad, H, W, C, N, pooled);
    return hipGetLastError();
}

// dact[b][y][x][c] (dense bf16) = dpooled[b][c] / (H*W)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* dpooled, int HW, int C, int N, bf16_t* dact) {
    const int cv = C >> 3;
    const long total = (long)N * HW * cv;
    const float inv = 1.f / (float)HW;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % cv) << 3;
        const long m = it / cv;
        const int b = (int)(m / HW);
        float v[8];
        const float* d = dpooled + (size_t)b * C + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = d[j] * inv;
        *reinterpret_cast<uint4*>(dact + (size_t)m * C + c) = pack8(v);
    }
}
hipError_t vpd_launch_avgpool_bwd(const float* dpooled, int H, int W, int C, int N, bf16_t* dact, hipStream_t s) {
    long items = (long)N * H * W * (C / 8);
    long g = (items + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(g < 1 ? 1 : (int)g), dim3(256), 0, s, dpooled, H * W, C, N, dact);
    return hipGetLastError();
}

// Small fp32 GEMM on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fp32 fma chain, so the
// head keeps fp32 accuracy).  One wave per 16x16 output tile, operands straight from global memory (these
// matrices are at most 512 wide and L2-resident), K walked 4 at a time.
__global__ __launch_bounds__(256) void sgemm_small_kernel(const float* A, const float* B, float* Y, const float* bias,
                                                          int M, int N, int K, int ta, int tb, int relu) {
    const int lane = threadIdx.x & 63;
    const int tiles_n = (N + 15) >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    if (tm * 16 >= M) return;
    const int r = lane & 15, kq = lane >> 4;
    const int m = tm * 16 + r, n = tn * 16 + r;
    const bool mok = m < M, nok = n < N;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t a_m = ta ? (size_t)(mok ? m : 0) : (size_t)(mok ? m : 0) * K;
    const size_t a_k = ta ? (size_t)M : 1;
    const size_t b_n = tb ? (size_t)(nok ? n : 0) * K : (size_t)(nok ? n : 0);
    const size_t b_k = tb ? 1 : (size_t)N;
    // 8 k-steps (32 values of K) per trip: all 16 operand loads are issued before the first MFMA waits for one of them
    // (one load pair per MFMA made every k-step a full memory round trip: 36 us for a 256x512x128 GEMM)
    for (int k0 = 0; k0 < K; k0 += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kok = k < K;
            av[u] = (mok && kok) ? A[a_m + (size_t)k * a_k] : 0.f;
            bv[u] = (nok && kok) ? B[b_n + (size_t)k * b_k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    // acc[j] = Y[tm*16 + 4*kq + j][tn*16 + r]
    if (nok) {
        const float bb = bias ? bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = tm * 16 + 4 * kq + j;
            if (mm < M) {
                float v = acc[j] + bb;
                if (relu) v = v > 0.f ? v : 0.f;
                Y[(size_t)mm * N + n] = v;
            }
        }
    }
}
// The same GEMM for row-major A [M][K] and B stored [N][K] (the forward linears: activations x weight^T), K % 64 == 0:
// both operands are contiguous along K, so a lane loads 4 consecutive k as ONE 16-byte load per operand (the MFMA sums
// over one k per 16-lane group and any 4 distinct k do, as long as A and B agree: group q supplies k = base + 4q + j in
// MFMA j), the block's 4 waves split K and their partial tiles are added through LDS in a fixed order.  One block per
// 16x16 tile: the fc forward (256 x 128 x 512) was ONE dependent memory round trip per 32 k on 32 blocks, 31 us.
__global__ __launch_bounds__(256) void sgemm_nt_splitk_kernel(const float* A, const float* B, float* Y, const float* bias,
                                                              int M, int N, int K, int relu) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_n = (N + 15) >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int r = lane & 15, kq = lane >> 4;
    const int m = tm * 16 + r, n = tn * 16 + r;
    const bool mok = m < M, nok = n < N;
    const float4* Ap = reinterpret_cast<const float4*>(A + (size_t)(mok ? m : 0) * K);
    const float4* Bp = reinterpret_cast<const float4*>(B + (size_t)(nok ? n : 0) * K);
    const int kper = K / 4;                                   // this wave's K range, a multiple of 16
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = wave * kper; k0 < (wave + 1) * kper; k0 += 64) {
        float4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 16 * u + 4 * kq;
            const bool kok = k < (wave + 1) * kper;
            av[u] = (mok && kok) ? Ap[k >> 2] : float4{0.f, 0.f, 0.f, 0.f};
            bv[u] = (nok && kok) ? Bp[k >> 2] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][4 * kq + j][r] = acc[j];
    __syncthreads();
    if (wave == 0 && nok) {
        const float bb = bias ? bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = tm * 16 + 4 * kq + j;
            if (mm < M) {
                float v = ((red[0][4 * kq + j][r] + red[1][4 * kq + j][r]) + red[2][4 * kq + j][r]) + red[3][4 * kq + j][r] + bb;
                if (relu) v = v > 0.f ? v : 0.f;
                Y[(size_t)mm * N + n] = v;
            }
        }
    }
}

hipError_t vpd_launch_sgemm(const float* A, const float* B, float* Y, const float* bias, int M, int N, int K, int ta,
                            int tb, int relu, hipStream_t s) {
    const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
    if (!ta && tb && K % 64 == 0 && K >= 64) {
        hipLaunchKernelGGL(sgemm_nt_splitk_kernel, dim3(tiles), dim3(256), 0, s, A, B, Y, bias, M, N, K, relu);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(sgemm_small_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, A, B, Y, bias, M, N, K, ta, tb, relu);
    return hipGetLastError();
}

// out[n] = sum_m A[m][n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* A, int M, int N, float* out) {
    __shared__ float sh[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + tx;
    float a = 0.f;
    if (n < N)
        for (int m = ty; m < M; m += 8) a += A[(size_t)m * N + n];
    sh[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += sh[i][tx];
        out[n] = t;
    }
}
hipError_t vpd_launch_colsum(const float* A, int M, int N, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32), dim3(256), 0, s, A, M, N, out);
    return hipGetLastError();
}

__global__ void relu_mask_kernel(float* d, const float* act, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        d[i] = act[i] > 0.f ? d[i] : 0.f;
}
hipError_t vpd_launch_relu_mask(float* d, const float* act, long n, hipStream_t s) {
    long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(g < 1 ? 1 : (int)g), dim3(256), 0, s, d, act, n);
    return hipGetLastError();
}

// loss = sum (e - t)^2 ; de = 2 (e - t).  One block: wave shuffles + LDS, no atomics.
__global__ __launch_bounds__(1024) void mse_kernel(const float* e, const float* t, long n, float* de,
                                                   float* loss_step, double* loss_accum) {
    __shared__ float sh[16];
    float a = 0.f;
    // four elements per thread and trip, loads of a trip issued together (the batch's 32 k values are 8 trips)
    const long n4 = n >> 2;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 8 * 1024) {      // eight trips' loads in flight (one block: nothing else hides them)
        float4 ev[8], tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long i = i0 + u * 1024;
            const long ic = i < n4 ? i : i0;
            ev[u] = reinterpret_cast<const float4*>(e)[ic];
            tv[u] = reinterpret_cast<const float4*>(t)[ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {                            // (same order of additions per thread as one trip at a time)
            const long i = i0 + u * 1024;
            if (i < n4) {
                const float4 d = {ev[u].x - tv[u].x, ev[u].y - tv[u].y, ev[u].z - tv[u].z, ev[u].w - tv[u].w};
                a += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
                if (de) reinterpret_cast<float4*>(de)[i] = float4{2.f * d.x, 2.f * d.y, 2.f * d.z, 2.f * d.w};
            }
        }
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 1024) {
        const float d = e[i] - t[i];
        a += d * d;
        if (de) de[i] = 2.f * d;
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += sh[i];
        if (loss_step) loss_step[0] = tot;
        if (loss_accum) loss_accum[0] += (double)tot;
    }
}
hipError_t vpd_launch_mse(const float* e, const float* t, long n, float* de, float* loss_step, double* loss_accum,
                          hipStream_t s) {
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, s, e, t, n, de, loss_step, loss_accum);
    return hipGetLastError();
}


// x *= s over n floats: the loss scale of fp16 training on d(loss)/d(pred) at the start of the backward pass (every gradient of the
// pass is then S x its value -- backward is linear -- and the optimizer step multiplies by 1 / S: AdamHyper::gscale)
__global__ __launch_bounds__(256) void scale_kernel(float* x, long n, float s) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= s;
}
hipError_t vpd_launch_scale(float* x, long n, float s, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, n, s);
    return hipGetLastError();
}
