// Launch-cost probe: back-to-back launches of an (almost) empty kernel for several block shapes / LDS sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(int* p, int n) {
    extern __shared__ int sm[];
    if (n < 0) { sm[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[0] = sm[0]; }   // never taken: keeps LDS allocated
}
__global__ void k_barriers(int* p, int n) {
    extern __shared__ int sm[];
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_barrier();
    if (n < 0) { sm[threadIdx.x] = p[threadIdx.x]; p[0] = sm[0]; }
}
int main() {
    int* d; hipMalloc(&d, 4096);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int lds_sizes[] = {0, 65536, 122880, 157696};
    const int threads[] = {64, 256, 512, 768};
    const int grids[] = {256, 512, 2048};
    for (int g : grids) for (int t : threads) for (int l : lds_sizes) {
        hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty, dim3(g), dim3(t), l, s, d, 0);
        hipEventRecord(a, s);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(g), dim3(t), l, s, d, 0);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("grid %4d threads %3d lds %6d : %.2f us per launch\n", g, t, l, ms * 1000 / 200);
    }
    hipFuncSetAttribute((const void*)k_barriers, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nb : {0, 36, 72}) {
        hipEventRecord(a, s);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_barriers, dim3(256), dim3(512), 122880, s, d, nb);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("barriers %2d (256 x 512, 120 KB) : %.2f us per launch\n", nb, ms * 1000 / 200);
    }
    return 0;
}
