// Conv + BatchNorm-forward behind the in-launch grid barrier against two launches -- the SKELETON of both, timed.
//
// What a fused launch would do behind its K loop, per block (one block per CU, 512 threads, one 256 x 64 output tile in registers):
//   A  add the tile's per-channel sums to the BatchNorm's fp64 rows (128 atomics), store the tile's z (32 KB, needed by backward)
//   B  grid barrier (vpd_grid_barrier of vpd_amd/csrc/sync.h; no release needed: only the atomics are read behind it)
//   C  read the 4 x 2 x 64 fp64 rows, finalize scale / shift
//   D  apply to the registers, store the activation (32 KB)
// and what the two launches do instead: launch 1 = A, launch 2 = every block reads the rows + finalizes (C) and streams the tensor
// (z in, activation out: 32 KB + 32 KB per block).  Both skeletons move the bytes the real kernels move and compute nothing else.
// Stamps (s_memtime, 100 MHz s_memrealtime beside it for the clock) per block: t0 entry, t1 stores issued, t2 barrier passed,
// t3 coefficients ready, t4 activation stores issued; the host prints medians / maxima, and the wall time per repetition of
//   chain F = [fused] x N      chain T = [launch 1, launch 2] x N      (hipGraph, created stream)
// Build: hipcc --offload-arch=gfx950 -O3 -I ../../vpd_amd/csrc -o barrier_probe barrier_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#include "sync.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Args {
    double* rows;            // [4][2][C] fp64, C = 256 (4 channel tiles of 64)
    u32x4* z; u32x4* act;    // [blocks][2048] 16-byte items = 32 KB per block each
    float* coef;             // [2][C]
    GridSync* gs; unsigned* err;
    unsigned long long* stamps;      // [blocks][8]
    int C;
};
static __device__ __forceinline__ void phase_a(const Args& a, int tid, int bid, u32x4 (&tile)[4]) {
    // statistics: 128 fp64 atomics per block (threads 0..127: sum / sum of squares of this block's 64 channels), row = block & 3
    const int n0 = (bid & 3) * 64;
    if (tid < 128) atomicAdd(&a.rows[((size_t)(bid & 3) * 2 + (tid >> 6)) * a.C + n0 + (tid & 63)], (double)(tid + 1) * 1e-3);
#pragma unroll
    for (int k = 0; k < 4; ++k) { tile[k] = u32x4{(unsigned)tid, (unsigned)bid, (unsigned)k, 0x3f803f80u}; a.z[(size_t)bid * 2048 + k * 512 + tid] = tile[k]; }
}
static __device__ __forceinline__ void phase_c(const Args& a, int tid, int bid, float* s_coef) {
    const int n0 = (bid & 3) * 64;
    if (tid < 64) {
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1 += __hip_atomic_load(&a.rows[((size_t)r * 2 + 0) * a.C + n0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s2 += __hip_atomic_load(&a.rows[((size_t)r * 2 + 1) * a.C + n0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double mu = s1 * (1.0 / 16384.0);
        double var = s2 * (1.0 / 16384.0) - mu * mu;
        var = var > 0 ? var : 0;
        const float r = __builtin_amdgcn_rsqf((float)var + 1e-5f);
        s_coef[tid] = r; s_coef[64 + tid] = -(float)mu * r;
        if (bid < 4) { a.coef[n0 + tid] = r; a.coef[a.C + n0 + tid] = -(float)mu * r; }
    }
    __syncthreads();
}
static __device__ __forceinline__ void stamp(const Args& a, int bid, int k) {
    if (threadIdx.x == 0) a.stamps[(size_t)bid * 8 + k] = __builtin_amdgcn_s_memtime();
}


// the same barrier in two halves (best case for the fused form: the z stores leave BEHIND the arrival and fly while the block polls)
static __device__ __forceinline__ void barrier_arrive(GridSync* gs, unsigned bid, unsigned nb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the statistics atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = bid & 7u;
        const unsigned in_group = (nb + 7u - g) >> 3;
        const unsigned ngroups = nb < 8u ? nb : 8u;
        const unsigned t = __hip_atomic_fetch_add(&gs->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == in_group - 1u) {
            const unsigned tt = __hip_atomic_fetch_add(&gs->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tt == ngroups - 1u)
                for (int k = 0; k < 8; ++k) __hip_atomic_store(&gs->gen[k][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
static __device__ __forceinline__ void barrier_wait(GridSync* gs, unsigned* err, unsigned bid) {
    if (threadIdx.x == 0) {
        if (!vpd_spin_until_nonzero(&gs->gen[bid & 7u][0])) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int SPLIT>
__global__ __launch_bounds__(512) void fused_kernel(const Args a) {
    __shared__ float s_coef[128];
    const int tid = threadIdx.x, bid = blockIdx.x;
    if (tid == 0) a.stamps[(size_t)bid * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    stamp(a, bid, 0);
    u32x4 tile[4];
    if (SPLIT) {
        const int n0 = (bid & 3) * 64;
        if (tid < 128) atomicAdd(&a.rows[((size_t)(bid & 3) * 2 + (tid >> 6)) * a.C + n0 + (tid & 63)], (double)(tid + 1) * 1e-3);
        barrier_arrive(a.gs, (unsigned)bid, gridDim.x);
#pragma unroll
        for (int k = 0; k < 4; ++k) { tile[k] = u32x4{(unsigned)tid, (unsigned)bid, (unsigned)k, 0x3f803f80u}; a.z[(size_t)bid * 2048 + k * 512 + tid] = tile[k]; }
        stamp(a, bid, 1);
        barrier_wait(a.gs, a.err, (unsigned)bid);
    } else {
        phase_a(a, tid, bid, tile);
        stamp(a, bid, 1);
        vpd_grid_barrier(a.gs, false, a.err, (unsigned)bid, gridDim.x);
    }
    stamp(a, bid, 2);
    phase_c(a, tid, bid, s_coef);
    stamp(a, bid, 3);
    const float sc = s_coef[tid & 63];
#pragma unroll
    for (int k = 0; k < 4; ++k) { tile[k].x += (unsigned)sc; a.act[(size_t)bid * 2048 + k * 512 + tid] = tile[k]; }
    stamp(a, bid, 4);
    if (tid == 0) a.stamps[(size_t)bid * 8 + 7] = __builtin_amdgcn_s_memrealtime();
}
__global__ __launch_bounds__(512) void first_kernel(const Args a) {
    const int tid = threadIdx.x, bid = blockIdx.x;
    u32x4 tile[4];
    phase_a(a, tid, bid, tile);
}
// the BatchNorm launch's shape: 1024-thread blocks, one per CU, first item requested before the prologue
__global__ __launch_bounds__(1024) void second_kernel(const Args a) {
    __shared__ float s_coef[128];
    const int tid = threadIdx.x, bid = blockIdx.x;
    u32x4 t0 = a.z[(size_t)bid * 2048 + tid], t1 = a.z[(size_t)bid * 2048 + 1024 + tid];
    phase_c(a, tid & 511, bid, s_coef);
    const float sc = s_coef[tid & 63];
    t0.x += (unsigned)sc; t1.x += (unsigned)sc;
    a.act[(size_t)bid * 2048 + tid] = t0; a.act[(size_t)bid * 2048 + 1024 + tid] = t1;
}
__global__ void zero_kernel(unsigned* p, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0; }

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    Args a{};
    a.C = 256;
    CHECK(hipMalloc(&a.rows, 4 * 2 * 256 * 8)); CHECK(hipMemset(a.rows, 0, 4 * 2 * 256 * 8));
    CHECK(hipMalloc(&a.z, (size_t)G * 32768)); CHECK(hipMalloc(&a.act, (size_t)G * 32768)); CHECK(hipMalloc(&a.coef, 2 * 256 * 4));
    const int NREP = 100;
    GridSync* gs; CHECK(hipMalloc(&gs, sizeof(GridSync) * NREP)); CHECK(hipMemset(gs, 0, sizeof(GridSync) * NREP));
    CHECK(hipMalloc(&a.err, 256)); CHECK(hipMemset(a.err, 0, 256));
    CHECK(hipMalloc(&a.stamps, (size_t)G * 64));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("# barrier_probe: %d CUs; fused = stats atomics + 32 KB z stores -> grid barrier -> rows + finalize -> 32 KB activation stores\n", G);
    for (int split = 0; split < 2; ++split) {
    printf("---- fused launch, %s ----\n", split ? "z stores BEHIND the arrival (they fly while the block polls)" : "z stores in front of the barrier (its first step waits for them)");
    // ---- stamps of one fused launch (warm) ----
    std::vector<unsigned long long> h((size_t)G * 8);
    std::vector<double> d[5];
    for (int rep = 0; rep < 20; ++rep) {
        a.gs = gs;
        hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, s, (unsigned*)gs, (int)(sizeof(GridSync) / 4));
        if (split) hipLaunchKernelGGL(fused_kernel<1>, dim3(G), dim3(512), 0, s, a);
        else hipLaunchKernelGGL(fused_kernel<0>, dim3(G), dim3(512), 0, s, a);
        CHECK(hipStreamSynchronize(s));
        if (rep < 5) continue;
        CHECK(hipMemcpy(h.data(), a.stamps, (size_t)G * 64, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull;
        for (int b = 0; b < G; ++b) first = std::min(first, h[(size_t)b * 8]);
        for (int b = 0; b < G; ++b) {
            const double ghz = (double)(h[b * 8 + 4] - h[b * 8]) / ((double)(h[b * 8 + 7] - h[b * 8 + 6]) * 10.0);      // ticks per ns
            for (int k = 1; k <= 4; ++k) d[k].push_back((double)(h[b * 8 + k] - h[b * 8 + k - 1]));
            d[0].push_back(ghz);
        }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
    const double ghz = med(d[0]);
    printf("shader clock inside the launch: %.2f GHz (s_memtime ticks per ns, median over blocks)\n", ghz);
    const char* names[5] = {"", split ? "A  atomics, arrival, z stores" : "A  atomics + z stores issued", split ? "B  poll + acquire" : "B  grid barrier", "C  rows + finalize", "D  activation stores issued"};
    for (int k = 1; k <= 4; ++k) printf("  %-30s median %6.2f us   max %6.2f us\n", names[k], med(d[k]) / ghz * 1e-3, mx(d[k]) / ghz * 1e-3);
    unsigned herr = 0; CHECK(hipMemcpy(&herr, a.err, 4, hipMemcpyDeviceToHost));
    printf("  barrier time-outs: %u\n", herr);
    }
    unsigned herr = 0;
    // ---- chains ----
    auto chain = [&](int kind) {
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipMemsetAsync(gs, 0, sizeof(GridSync) * NREP, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < NREP; ++r) {
            Args b = a; b.gs = gs + r;
            if (kind == 0) hipLaunchKernelGGL(fused_kernel<0>, dim3(G), dim3(512), 0, s, b);
            else if (kind == 2) hipLaunchKernelGGL(fused_kernel<1>, dim3(G), dim3(512), 0, s, b);
            else { hipLaunchKernelGGL(first_kernel, dim3(G), dim3(512), 0, s, b); hipLaunchKernelGGL(second_kernel, dim3(G), dim3(1024), 0, s, b); }
        }
        CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemsetAsync(gs, 0, sizeof(GridSync) * NREP, s));
            CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) best = std::min(best, (double)ms * 1e3 / NREP);
        }
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
        return best;
    };
    const double f = chain(0), f2 = chain(2), t = chain(1);
    printf("hipGraph chains, us per repetition: fused launch %.2f (stores behind the arrival: %.2f)   |   two launches %.2f\n", f, f2, t);
    CHECK(hipMemcpy(&herr, a.err, 4, hipMemcpyDeviceToHost));
    printf("barrier time-outs after the chains: %u\n", herr);
    return 0;
}
