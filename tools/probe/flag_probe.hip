// LDS flag hand-off latency between waves of one workgroup (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 -o flag_probe flag_probe.hip
// Ping-pong: wave 0 stores k to word A, wave W spins on A, stores k to word B, wave 0 spins on B.  Cycles per round trip by
// s_memtime, for several store / load forms, with 0 / 6 other waves of the block spinning on a third word (LDS contention).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef volatile unsigned __attribute__((address_space(3)))* lds_vu;
typedef const volatile unsigned __attribute__((address_space(3)))* lds_cvu;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const volatile u32x4 __attribute__((address_space(3)))* lds_cv4;

template <int MODE>   // 0: all-lane b32 store + b32 load; 1: lane-0 store + b32 load; 2: all-lane store + b128 load; 3: s_barrier pair
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int partner, int noisy) {
    __shared__ __attribute__((aligned(16))) unsigned flags[64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) flags[threadIdx.x] = 0;
    __syncthreads();
    unsigned* A = flags, *B = flags + 16, *Q = flags + 32;
    auto st = [&](unsigned* w, unsigned v) {
        if (MODE == 1) { if (lane == 0) *(lds_vu)w = v; }
        else *(lds_vu)w = v;
    };
    auto ld = [&](unsigned* w) -> unsigned {
        if (MODE == 2) { u32x4 a = *(lds_cv4)w; return __builtin_amdgcn_readfirstlane(a.x); }
        return __builtin_amdgcn_readfirstlane(*(lds_cvu)w);
    };
    if (MODE == 3) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < iters; ++k) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
        return;
    }
    if (wave == 0) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 1; k <= iters; ++k) {
            st(A, k);
            int guard = 0;
            while (ld(B) != (unsigned)k && ++guard < (1 << 20)) {}
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        st(Q, 1);
        if (lane == 0) out[blockIdx.x] = t1 - t0;
    } else if (wave == partner) {
        for (int k = 1; k <= iters; ++k) {
            int guard = 0;
            while (ld(A) != (unsigned)k && ++guard < (1 << 20)) {}
            st(B, k);
        }
    } else if (noisy) {
        int guard = 0;
        while (ld(Q) == 0 && ++guard < (1 << 24)) {}
    }
}

int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 8);
    unsigned long long h[256];
    const int iters = 2000;
    const char* names[4] = {"all-lane b32 store, b32 load", "lane-0 store, b32 load", "all-lane store, b128 load", "two s_barrier"};
    for (int mode = 0; mode < 4; ++mode)
        for (int partner : {1, 4})           // same SIMD pair? waves go to SIMDs in the order 0,2,1,3: wave 4 shares wave 0's SIMD
            for (int noisy : {0, 1}) {
                for (int rep = 0; rep < 2; ++rep) {
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, d, iters, partner, noisy); break;
                        case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, d, iters, partner, noisy); break;
                        case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, d, iters, partner, noisy); break;
                        default: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, d, iters, partner, noisy); break;
                    }
                    hipDeviceSynchronize();
                }
                hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
                double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
                printf("%-32s partner wave %d noisy %d: %.0f cycles per round trip (s_memtime ticks)\n", names[mode], partner, noisy, s / 256 / iters);
            }
    return 0;
}
