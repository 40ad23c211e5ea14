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


// ===========================================================================
// Fused embedding head of the train step without the motion MLP (train_vpd_model.py:82-91 with motion=False):
//   head_fwd_fused_kernel: global average pool -> fc -> sum-MSE loss + d(emb), one block per crop (the batch's loss is
//     summed by the last block to finish, in crop order: no float atomics, the value repeats run to run);
//   head_bwd_fused_kernel: d(pooled) = d(emb) W broadcast over the H*W pixels (crop blocks), dW = d(emb)^T pooled and
//     db = column sums of d(emb) (weight blocks) -- one launch.
// Replaces avgpool + sgemm + mse and sgemm + colsum + sgemm + avgpool_bwd: 7 launches of 5-9 us at a batch of 256.
// ===========================================================================
struct HeadFwdArgs {
    const bf16_t* act; int Hp, Wp, pad, H, W, C;     // padded activation of the last block
    const float* Wt; const float* bias; int D;       // fc weight [D][C], bias [D]
    const float* target;                             // [N][D] or null (no loss)
    float* pooled; float* emb; float* demb;          // [N][C], [N][D], [N][D] (demb null: no gradient)
    float* partial; unsigned* counter;               // [N] per-crop losses, arrival counter (zero between launches)
    float* loss_step; double* loss_accum; int N;
};
__global__ __launch_bounds__(256) void head_fwd_fused_kernel(const HeadFwdArgs a) {
    extern __shared__ float sp[];                    // pooled[C], then 4 wave partials
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, D = a.D;
    const float inv = 1.f / (float)(a.H * a.W);
    for (int c = tid * 8; c < C; c += 256 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = 0; y < a.H; ++y)
            for (int x = 0; x < a.W; ++x) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4*>(a.act + ((size_t)(n * a.Hp + y + a.pad) * a.Wp + x + a.pad) * C + c), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[j] *= inv; sp[c + j] = acc[j]; }
        float* o = a.pooled + (size_t)n * C + c;
        *reinterpret_cast<float4*>(o) = float4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<float4*>(o + 4) = float4{acc[4], acc[5], acc[6], acc[7]};
    }
    __syncthreads();
    float lsum = 0.f;                                // lane 0 of each wave: sum of squared differences of its outputs
    for (int d = wave; d < D; d += 4) {
        const float* wr = a.Wt + (size_t)d * C;
        float t = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 w = *reinterpret_cast<const float4*>(wr + c);
            t += w.x * sp[c] + w.y * sp[c + 1] + w.z * sp[c + 2] + w.w * sp[c + 3];
        }
        t = wave_sum(t);
        if (lane == 0) {
            const float e = t + a.bias[d];
            a.emb[(size_t)n * D + d] = e;
            if (a.target) {
                const float df = e - a.target[(size_t)n * D + d];
                lsum += df * df;
                if (a.demb) a.demb[(size_t)n * D + d] = 2.f * df;
            }
        }
    }
    if (!a.target) return;
    __syncthreads();                                 // (pooled in sp is dead)
    if (lane == 0) sp[wave] = lsum;
    __syncthreads();
    __shared__ int last;
    if (tid == 0) {
        a.partial[n] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        __threadfence();
        last = atomicAdd(a.counter, 1u) == (unsigned)(a.N - 1);
    }
    __syncthreads();
    if (!last || wave != 0) return;
    __threadfence();
    float t = 0.f;
    for (int i = lane; i < a.N; i += 64) t += __hip_atomic_load(a.partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = wave_sum(t);
    if (lane == 0) {
        if (a.loss_step) a.loss_step[0] = t;
        if (a.loss_accum) a.loss_accum[0] += (double)t;
        *a.counter = 0u;
    }
}
hipError_t vpd_launch_head_fwd_fused(const bf16_t* act, int Hp, int Wp, int pad, int H, int W, int C, int N, const float* Wt,
                                     const float* bias, int D, const float* target, float* pooled, float* emb, float* demb,
                                     float* partial, unsigned* counter, float* loss_step, double* loss_accum, hipStream_t s) {
    if (C % 256 || N < 1) return hipErrorInvalidValue;      // (a lane's float4 of the fc row, 8 channels per pooling thread)
    HeadFwdArgs a{act, Hp, Wp, pad, H, W, C, Wt, bias, D, target, pooled, emb, demb, partial, counter, loss_step, loss_accum, N};
    hipLaunchKernelGGL(head_fwd_fused_kernel, dim3(N), dim3(256), (size_t)(C + 4) * sizeof(float), s, a);
    return hipGetLastError();
}

struct HeadBwdArgs {
    const float* demb; const float* Wt; const float* pooled;      // [N][D], [D][C], [N][C]
    float* dW; float* db; bf16_t* dact;                           // [D][C], [D], dense [N][HW][C]
    int N, D, C, HW;
};
#define HEAD_BWD_DT 4                                             // fc rows per weight block
__global__ __launch_bounds__(256) void head_bwd_fused_kernel(const HeadBwdArgs a) {
    extern __shared__ float sd[];                                 // crop blocks: demb row [D]
    const int tid = threadIdx.x;
    const int N = a.N, D = a.D, C = a.C;
    int b = blockIdx.x;
    if (b < N) {                                                  // ---- d(pooled) of crop b, broadcast to its pixels ----
        for (int d = tid; d < D; d += 256) sd[d] = a.demb[(size_t)b * D + d];
        __syncthreads();
        const float inv = 1.f / (float)a.HW;
        for (int c = tid * 2; c < C; c += 512) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll 8
            for (int d = 0; d < D; ++d) {
                const float2 w = *reinterpret_cast<const float2*>(a.Wt + (size_t)d * C + c);
                t0 += sd[d] * w.x; t1 += sd[d] * w.y;
            }
            const unsigned v = pack2bf(t0 * inv, t1 * inv);
            for (int m = 0; m < a.HW; ++m)
                *reinterpret_cast<unsigned*>(a.dact + ((size_t)b * a.HW + m) * C + c) = v;
        }
        return;
    }
    b -= N;
    const int nwb = (D + HEAD_BWD_DT - 1) / HEAD_BWD_DT;
    if (b < nwb) {                                                // ---- dW rows d0 .. d0+DT-1 ----
        const int d0 = b * HEAD_BWD_DT;
        for (int c = tid * 2; c < C; c += 512) {
            float acc[HEAD_BWD_DT][2];
#pragma unroll
            for (int j = 0; j < HEAD_BWD_DT; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; }
#pragma unroll 4
            for (int n = 0; n < N; ++n) {
                const float2 pv = *reinterpret_cast<const float2*>(a.pooled + (size_t)n * C + c);
#pragma unroll
                for (int j = 0; j < HEAD_BWD_DT; ++j) {
                    const float de = d0 + j < D ? a.demb[(size_t)n * D + d0 + j] : 0.f;
                    acc[j][0] += de * pv.x; acc[j][1] += de * pv.y;
                }
            }
#pragma unroll
            for (int j = 0; j < HEAD_BWD_DT; ++j)
                if (d0 + j < D) *reinterpret_cast<float2*>(a.dW + (size_t)(d0 + j) * C + c) = float2{acc[j][0], acc[j][1]};
        }
        return;
    }
    for (int d = tid; d < D; d += 256) {                          // ---- db ----
        float t = 0.f;
        for (int n = 0; n < N; ++n) t += a.demb[(size_t)n * D + d];
        a.db[d] = t;
    }
}
hipError_t vpd_launch_head_bwd_fused(const float* demb, const float* Wt, const float* pooled, float* dW, float* db,
                                     bf16_t* dact, int N, int D, int C, int HW, hipStream_t s) {
    if (C % 2 || N < 1) return hipErrorInvalidValue;
    HeadBwdArgs a{demb, Wt, pooled, dW, db, dact, N, D, C, HW};
    const int grid = N + (D + HEAD_BWD_DT - 1) / HEAD_BWD_DT + 1;
    hipLaunchKernelGGL(head_bwd_fused_kernel, dim3(grid), dim3(256), (size_t)D * sizeof(float), s, a);
    return hipGetLastError();
}
