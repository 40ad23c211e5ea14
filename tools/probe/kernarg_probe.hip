// Does kernarg preload (first dwords of the kernel arguments delivered in SGPRs by the dispatcher) shorten a short dependent
// kernel?  Chain of N launches of a kernel that reads ONE value through a pointer argument and adds it to an output;
// variant A takes a 256-byte struct by value (s_load from the kernarg segment), variant B leading scalar arguments
// (compiled with -mllvm -amdgpu-kernarg-preload-count=8).  Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
struct P { const float* a; float* b; int n; float s; int pad[58]; };
__global__ void k_struct(const P p) { if (threadIdx.x == 0 && blockIdx.x == 0) p.b[0] += p.a[0] * p.s + (float)p.pad[57]; }
__global__ void k_scalar(const float* a, float* b, int n, float s, const P rest) { if (threadIdx.x == 0 && blockIdx.x == 0) b[0] += a[0] * s + (float)n; }
int main() {
    float *a, *b; hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMemset(a, 0, 4096); hipMemset(b, 0, 4096);
    P p{}; p.a = a; p.b = b; p.n = 0; p.s = 1.f;
    hipStream_t s; hipStreamCreate(&s);
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            const int N = 2000;
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, s, p);
                else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, s, (const float*)a, b, 0, 1.f, p);
            }
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s rep %d: %.3f us per launch\n", variant ? "scalar+preload" : "struct        ", rep, us / N);
        }
    return 0;
}
