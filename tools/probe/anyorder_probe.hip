// Probe: does hipExtAnyOrderLaunch (AQL barrier bit cleared) let a dependent kernel's blocks start while its
// predecessor drains, and is an in-kernel completion counter (release / acquire at agent scope) a correct hand-over?
//   mode 0: plain in-order launches                         (baseline: launch-to-launch cost)
//   mode 1: any-order launches, NO dependency handling      (upper bound of what overlap buys; results not checked)
//   mode 2: any-order launches + completion counter chain   (kernel i waits until all blocks of kernel i-1 have signalled)
// Each block works ~T us (+ skew by block), then writes buf[i&1][block] = i; kernel i checks buf[(i-1)&1][(block+97)%grid] == i-1.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512) void k_chain(int* buf, unsigned* done, unsigned* errs, int iter, int grid, int work_ticks,
                                               int skew_ticks, int chain) {
    extern __shared__ int sm[];
    const int b = blockIdx.x;
    if (chain && iter > 0) {
        if (threadIdx.x == 0) {
            const unsigned want = (unsigned)iter * (unsigned)grid;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { atomicAdd(errs + 1, 1u); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (iter > 0 && threadIdx.x == 0) {
        const int v = buf[((iter - 1) & 1) * grid + (b + 97) % grid];
        if (v != iter - 1) atomicAdd(errs, 1u);
    }
    // "work"
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long dt = (unsigned long long)(work_ticks + (b & 7) * skew_ticks);
    while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) buf[(iter & 1) * grid + b] = iter;
    if (work_ticks < 0) sm[threadIdx.x] = buf[threadIdx.x];
    if (chain) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    const int lds = argc > 2 ? atoi(argv[2]) : 150 * 1024;
    int* buf; unsigned* ctr;
    hipMalloc(&buf, 2 * grid * sizeof(int));
    hipMalloc(&ctr, 4096);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int N = 400;
    for (int work_us : {0, 5, 20}) for (int skew_us : {0, 1}) for (int mode = 0; mode < 3; ++mode) {
        hipMemsetAsync(buf, 0xff, 2 * grid * sizeof(int), s);
        hipMemsetAsync(ctr, 0, 4096, s);
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int i = 0; i < N; ++i) {
            const int flags = mode == 0 ? 0 : hipExtAnyOrderLaunch;
            hipExtLaunchKernelGGL(k_chain, dim3(grid), dim3(512), lds, s, nullptr, nullptr, flags, buf, ctr, ctr + 64, i, grid,
                                  work_us * 100, skew_us * 100, mode == 2 ? 1 : 0);
        }
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned h[66]; hipMemcpy(h, ctr, sizeof h, hipMemcpyDeviceToHost);
        printf("grid %d lds %d work %2d us skew %d us mode %d : %.2f us per launch, check errors %u, timeouts %u\n", grid, lds,
               work_us, skew_us, mode, ms * 1000 / N, h[64], h[65]);
    }
    return 0;
}
