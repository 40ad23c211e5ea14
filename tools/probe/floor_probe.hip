// What is the per-launch floor of a dependent kernel chain on MI355X made of?  
//
// The 256-crop train step is 166 dependent launches; the ones that do nothing take 4.7-5.4 us in rocprofv3's trace while
// the guide prices a boundary at 1.45-1.9 us "+ B / 6 TB/s when the predecessor leaves B bytes dirty".  This probe times
//   writer<FLAVOUR>(B bytes)  ->  successor
// for B in {0, 4, 16, 32, 128} MB written with plain / sc1 / sc0 sc1 / nt 16-byte stores, and four successors:
//   empty1     one 64-thread block that only stamps the clock
//   empty256   256 blocks x 256 threads that only stamp the clock
//   finalize   one wave reading 16 rows x 64 channels of fp64 and writing 4 x 64 floats (bn_finalize_kernel's shape)
//   stream     256 blocks x 1024 threads: 16 MB of what the writer wrote -> 16 MB elsewhere (a BatchNorm apply's shape)
// on the null stream or on a hipStreamNonBlocking stream.  Three clocks per pair:
//   (a) start/stop events carried by the kernel's own dispatch packet (hipExtLaunchKernelGGL) = what rocprofv3 reports;
//   (b) s_memrealtime (100 MHz) stamps: last writer wave after its vmcnt(0)  ->  first successor wave = the boundary itself;
//   (c) rocprofv3 --kernel-trace of the same binary with --noevents (tools/probe/floor_table.py pairs rows by order).
// Build: hipcc --offload-arch=gfx950 -O3 -o floor_probe floor_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <utility>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
enum { PLAIN = 0, SC1 = 1, SC0SC1 = 2, NT = 3 };
static const char* FNAME[4] = {"plain", "sc1", "sc0sc1", "nt"};

template <int F> static __device__ __forceinline__ void store16(void* p, u32x4 v) {
    if (F == PLAIN)  asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (F == SC1)    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (F == SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (F == NT)     asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ void stamp(unsigned long long* slot) {
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1 store: leaves nothing dirty
}

// 256 blocks x 256 threads; every block writes `per_block` bytes (multiple of 4096) as 16-byte lane-contiguous stores.
template <int F> __global__ __launch_bounds__(256) void writer(char* buf, long per_block, unsigned long long* end_stamp, unsigned seed) {
    char* p = buf + (long)blockIdx.x * per_block + threadIdx.x * 16;
    const u32x4 v = {seed, threadIdx.x, blockIdx.x, 0x3f803f80u};
    for (long o = 0; o < per_block; o += 4096) store16<F>(p + o, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) stamp(end_stamp + blockIdx.x);
}
__global__ void empty_kernel(unsigned long long* start_stamp) {
    if (threadIdx.x == 0) stamp(start_stamp + blockIdx.x);
}
__global__ void finalize_kernel(const double* rows, float* out, unsigned long long* start_stamp) {
    if (threadIdx.x == 0) stamp(start_stamp);
    double s = 0, q = 0;
    for (int r = 0; r < 8; ++r) { s += rows[r * 64 + threadIdx.x]; q += rows[(8 + r) * 64 + threadIdx.x]; }
    const double mean = s * (1.0 / 65536.0), var = q * (1.0 / 65536.0) - mean * mean;
    const float rstd = rsqrtf((float)var + 1e-5f);
    out[threadIdx.x] = (float)mean; out[64 + threadIdx.x] = rstd; out[128 + threadIdx.x] = rstd * 1.1f; out[192 + threadIdx.x] = -(float)mean * rstd;
}
// reads `bytes` of src (what the writer wrote), writes them scaled to dst: 16-byte items, grid-stride, two in flight
__global__ __launch_bounds__(1024) void stream_kernel(const uint4* src, uint4* dst, long items, unsigned long long* start_stamp) {
    if (threadIdx.x == 0) stamp(start_stamp + blockIdx.x);
    const long stride = (long)gridDim.x * 1024;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < items; i += 2 * stride) {
        uint4 a = src[i], b = (i + stride < items) ? src[i + stride] : a;
        a.x += 1; b.x += 1;
        dst[i] = a; if (i + stride < items) dst[i + stride] = b;
    }
}

struct Ctx {
    hipStream_t s; bool events; char *buf, *dst, *dst2; double* rows; float* fout; unsigned long long *wend, *sstart;
    hipEvent_t e[6];
};
template <typename K, typename... A> static void launch(Ctx& c, K k, dim3 g, dim3 b, hipEvent_t e0, hipEvent_t e1, A... a) {
    if (c.events) hipExtLaunchKernelGGL(k, g, b, 0, c.s, e0, e1, 0, a...);
    else hipLaunchKernelGGL(k, g, b, 0, c.s, a...);
}
static void launch_writer(Ctx& c, int f, long per_block, unsigned seed) {
    switch (f) {
    case PLAIN:  launch(c, writer<PLAIN>,  dim3(256), dim3(256), c.e[0], c.e[1], c.buf, per_block, c.wend, seed); break;
    case SC1:    launch(c, writer<SC1>,    dim3(256), dim3(256), c.e[0], c.e[1], c.buf, per_block, c.wend, seed); break;
    case SC0SC1: launch(c, writer<SC0SC1>, dim3(256), dim3(256), c.e[0], c.e[1], c.buf, per_block, c.wend, seed); break;
    default:     launch(c, writer<NT>,     dim3(256), dim3(256), c.e[0], c.e[1], c.buf, per_block, c.wend, seed); break;
    }
}
static void launch_succ(Ctx& c, int kind) {
    switch (kind) {
    case 0: launch(c, empty_kernel, dim3(1), dim3(64), c.e[2], c.e[3], c.sstart); break;
    case 1: launch(c, empty_kernel, dim3(256), dim3(256), c.e[2], c.e[3], c.sstart); break;
    case 2: launch(c, finalize_kernel, dim3(1), dim3(64), c.e[2], c.e[3], (const double*)c.rows, c.fout, c.sstart); break;
    default: launch(c, stream_kernel, dim3(256), dim3(1024), c.e[2], c.e[3], (const uint4*)c.buf, (uint4*)c.dst, (long)(16 << 20) / 16, c.sstart); break;
    }
}

// ---- part 2: marginal costs inside a busy stream, by differences of hipGraph-captured chains (no host in the loop) ----
// rw<F>: 256 blocks x 1024 threads read `bytes` of src and write them to dst with 16-byte stores of flavour F, two items in
// flight per thread (the shape of a BatchNorm apply launch).  Chains per flavour and size, N repetitions each:
//   P1 = [rw(a->b), rw(b->a)]            every launch reads what its predecessor wrote
//   P2 = P1 with an empty 256-block launch behind every rw      -> (P2 - P1) / 2N = an empty launch behind a streaming one
//   P3 = P1 with a one-wave finalize launch behind every rw     -> the same for bn_finalize_kernel's shape
//   P4 = P1 with TWO empty launches behind every rw             -> (P4 - P2) / 2N = an empty launch behind an empty one
template <int F> __global__ __launch_bounds__(1024) void rw_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long items) {
    const long stride = (long)gridDim.x * 1024;
    long i = (long)blockIdx.x * 1024 + threadIdx.x;
    u32x4 a = {0, 0, 0, 0}, b = a;
    if (i < items) a = src[i];
    while (i < items) {
        const long n = i + stride;
        if (n < items) b = src[n];
        a.x += 1u;
        store16<F>(dst + i, a);
        a = b; i = n;
    }
}
static void launch_rw(hipStream_t s, int f, const void* src, void* dst, long bytes) {
    const long items = bytes / 16;
    switch (f) {
    case PLAIN:  hipLaunchKernelGGL(rw_kernel<PLAIN>,  dim3(256), dim3(1024), 0, s, (const u32x4*)src, (u32x4*)dst, items); break;
    case SC1:    hipLaunchKernelGGL(rw_kernel<SC1>,    dim3(256), dim3(1024), 0, s, (const u32x4*)src, (u32x4*)dst, items); break;
    case SC0SC1: hipLaunchKernelGGL(rw_kernel<SC0SC1>, dim3(256), dim3(1024), 0, s, (const u32x4*)src, (u32x4*)dst, items); break;
    default:     hipLaunchKernelGGL(rw_kernel<NT>,     dim3(256), dim3(1024), 0, s, (const u32x4*)src, (u32x4*)dst, items); break;
    }
}
static double time_chain(Ctx& c, hipStream_t s, int f, long bytes, int pattern, int N, bool eager = false) {
    hipGraph_t g; hipGraphExec_t ge;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    double best = 1e30;
    for (int rep = 0; rep < (eager ? 4 : 1); ++rep) {
    if (eager) { CHECK(hipStreamSynchronize(s)); CHECK(hipEventRecord(a, s)); }
    else CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < N; ++r)
        for (int half = 0; half < 2; ++half) {
            launch_rw(s, f, half ? c.dst2 : c.buf, half ? c.buf : c.dst2, bytes);
            if (pattern == 1 || pattern == 3) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, c.sstart);
            if (pattern == 3) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, c.sstart);
            if (pattern == 2) hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)c.rows, c.fout, c.sstart);
        }
    if (eager) { CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b)); float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); if (rep > 0) best = std::min(best, (double)ms * 1e3); }
    }
    if (eager) { CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b)); return best; }
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0) best = std::min(best, (double)ms * 1e3);
    }
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return best;
}
static void part2(Ctx& c, bool eager, bool nullstream) {
    hipStream_t s = nullptr; if (!nullstream) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!c.dst2) CHECK(hipMalloc(&c.dst2, 64l << 20));
    const int N = 100;
    printf("# part 2: %s chains of %d x [rw(a->b), rw(b->a)] (+ small launches) on %s, us per launch by difference\n", eager ? "EAGER" : "hipGraph", N, nullstream ? "the null stream" : "a created stream");
    printf("%-8s %4s %12s %12s %14s %14s %14s\n", "flavour", "MB", "rw_us", "rw_TB/s", "empty_after_rw", "final_after_rw", "empty_after_empty");
    for (int mb : {2, 8, 16, 32, 64})
        for (int f : {PLAIN, SC1, SC0SC1, NT}) {
            const long bytes = (long)mb << 20;
            const double p1 = time_chain(c, s, f, bytes, 0, N, eager), p2 = time_chain(c, s, f, bytes, 1, N, eager), p3 = time_chain(c, s, f, bytes, 2, N, eager), p4 = time_chain(c, s, f, bytes, 3, N, eager);
            printf("%-8s %4d %12.2f %12.2f %14.2f %14.2f %14.2f\n", FNAME[f], mb, p1 / (2 * N), 2.0 * bytes / (p1 / (2 * N)) * 1e-6, (p2 - p1) / (2 * N), (p3 - p1) / (2 * N), (p4 - p2) / (2 * N));
            fflush(stdout);
        }
}

// ---- part 3: is the excess of the real step's small launches (4.7-5.4 us) cold INSTRUCTION fetch or cold DATA? ----
// cold_kernel<ID, KB>: one wave per block, KB KiB of straight-line 8-byte scalar instructions executed once, then a stamp.
// 32 instantiations are separate functions at separate addresses; launched round robin behind 32-MB rw launches (which sweep
// the L2s) every launch finds its code cold; launched as ID 0 every time the code is as warm as it gets.
template <int ID, int KB> __global__ void cold_kernel(unsigned long long* start_stamp, unsigned* out) {
    unsigned acc = ID;
    if (KB > 0) asm volatile(".rept %1\n s_add_u32 %0, %0, 0x12345678\n .endr" : "+s"(acc) : "n"(KB * 128));
    if (threadIdx.x == 0) { stamp(start_stamp + blockIdx.x); if (acc == 17u) out[0] = acc; }
}
template <int KB, int... IDS> static void launch_cold_table(hipStream_t s, int id, int nblocks, unsigned long long* st, unsigned* out, std::integer_sequence<int, IDS...>) {
    using Fn = void (*)(unsigned long long*, unsigned*);
    static const Fn table[] = {cold_kernel<IDS, KB>...};
    hipLaunchKernelGGL(table[id], dim3(nblocks), dim3(64), 0, s, st, out);
}
template <int KB> static double cold_chain(Ctx& c, hipStream_t s, int N, bool rotate, int nblocks, int data_mode) {
    // data_mode 0: cold_kernel; 1: finalize reading rows nobody wrote; 2: finalize reading rows INSIDE what the preceding rw wrote (plain); 3: same, sc1 stores
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipStreamSynchronize(s)); CHECK(hipEventRecord(a, s));
        for (int r = 0; r < N; ++r) {
            char* dst = (r & 1) ? c.buf : c.dst2; const char* src = (r & 1) ? c.dst2 : c.buf;
            launch_rw(s, data_mode == 3 ? SC1 : PLAIN, src, dst, 32l << 20);
            if (nblocks == 0) continue;
            if (data_mode == 0) launch_cold_table<KB>(s, rotate ? (r & 31) : 0, nblocks, c.sstart, (unsigned*)c.fout, std::make_integer_sequence<int, 32>{});
            else hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, s, data_mode == 1 ? (const double*)c.rows : (const double*)(dst + (16l << 20) + (r % 61) * 65536), c.fout, c.sstart);
        }
        CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0) best = std::min(best, (double)ms * 1e3);
    }
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return best;
}
static void part3(Ctx& c) {
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!c.dst2) CHECK(hipMalloc(&c.dst2, 64l << 20));
    const int N = 256;
    const double base = cold_chain<0>(c, s, N, false, 0, 0);
    printf("# part 3: EAGER chains of %d x [rw 32 MB, small launch] on a created stream; rw alone %.2f us per launch; marginal us of the small launch\n", N, base / N);
    printf("%-34s %10s %10s %12s %12s\n", "small launch", "warm,1blk", "cold,1blk", "warm,256blk", "cold,256blk");
#define ROW(KB) printf("%-34s %10.2f %10.2f %12.2f %12.2f\n", "straight-line code " #KB " KiB", (cold_chain<KB>(c, s, N, false, 1, 0) - base) / N, (cold_chain<KB>(c, s, N, true, 1, 0) - base) / N, (cold_chain<KB>(c, s, N, false, 256, 0) - base) / N, (cold_chain<KB>(c, s, N, true, 256, 0) - base) / N); fflush(stdout);
    ROW(0) ROW(1) ROW(2) ROW(4) ROW(8) ROW(16)
    printf("%-34s %10.2f\n", "finalize, rows nobody wrote", (cold_chain<0>(c, s, N, false, 1, 1) - base) / N);
    printf("%-34s %10.2f\n", "finalize, rows fresh from rw plain", (cold_chain<0>(c, s, N, false, 1, 2) - base) / N);
    const double base_sc1 = cold_chain<0>(c, s, N, false, 0, 3);
    printf("%-34s %10.2f   (rw sc1 alone %.2f us per launch)\n", "finalize, rows fresh from rw sc1", (cold_chain<0>(c, s, N, false, 1, 3) - base_sc1) / N, base_sc1 / N);
}

// ---- part 4: do a kernel's plain stores survive in its XCD's L2 for the NEXT kernel?  rwx reads what block (b + shift) % 256 of its
// predecessor wrote and writes its own slice: shift 0 / 8 = producer on the same XCD (round-robin placement), shift 1 = another XCD ----
__global__ __launch_bounds__(1024) void rwx_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long per_block_items, int shift) {
    const long rb = (long)((blockIdx.x + shift) % gridDim.x) * per_block_items, wb = (long)blockIdx.x * per_block_items;
    u32x4 v[4];
    for (long i = threadIdx.x; i < per_block_items; i += 4096) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 1024 < per_block_items) v[k] = src[rb + i + k * 1024];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 1024 < per_block_items) { v[k].x += 1u; dst[wb + i + k * 1024] = v[k]; }
    }
}
static void part4(Ctx& c) {
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!c.dst2) CHECK(hipMalloc(&c.dst2, 64l << 20));
    const int N = 100;
    printf("# part 4: hipGraph chains of %d x [rwx(a->b), rwx(b->a)], 256 blocks x 1024 threads, contiguous slice per block; us per launch\n", N);
    printf("%6s %12s %12s %12s\n", "MB", "shift 0", "shift 8", "shift 1");
    for (int mb : {2, 4, 8, 16, 32}) {
        double t[3];
        int sh[3] = {0, 8, 1};
        for (int q = 0; q < 3; ++q) {
            const long per = ((long)mb << 20) / 16 / 256;
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int r = 0; r < N; ++r) {
                hipLaunchKernelGGL(rwx_kernel, dim3(256), dim3(1024), 0, s, (const u32x4*)c.buf, (u32x4*)c.dst2, per, sh[q]);
                hipLaunchKernelGGL(rwx_kernel, dim3(256), dim3(1024), 0, s, (const u32x4*)c.dst2, (u32x4*)c.buf, per, sh[q]);
            }
            CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(a, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep > 0) best = std::min(best, (double)ms * 1e3 / (2 * N));
            }
            CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
            t[q] = best;
        }
        printf("%6d %12.2f %12.2f %12.2f\n", mb, t[0], t[1], t[2]);
    }
}

// ---- part 5: which XCD does block b run on?  (HW_REG_XCC_ID, hwreg 20 on gfx940+) for the two launch shapes of the step ----
__global__ void xcc_kernel(unsigned* out) {
    extern __shared__ unsigned char dyn[];
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) { out[blockIdx.x] = v; if (dyn[0] == 77) out[0] = 0; }
}
static void part5(Ctx& c) {
    unsigned* d; CHECK(hipMalloc(&d, 4096 * 4));
    std::vector<unsigned> h(4096);
    struct Shape { int grid, threads, lds; const char* what; } shapes[] = {
        {256, 1024, 4096, "BatchNorm launch: 256 x 1024 threads, 4 KB LDS"},
        {256, 512, 150 * 1024, "3x3 launch: 256 x 512 threads, 150 KB LDS"},
        {256, 768, 150 * 1024, "3x3 launch, eight MFMA waves: 256 x 768 threads, 150 KB LDS"},
        {512, 256, 0, "512 x 256 threads"}};
    for (auto& sh : shapes) {
        CHECK(hipMemset(d, 0xff, 4096 * 4));
        (void)hipFuncSetAttribute((const void*)xcc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(xcc_kernel, dim3(sh.grid), dim3(sh.threads), sh.lds, c.s, d);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, sh.grid * 4, hipMemcpyDeviceToHost));
        int match = 0;
        for (int b = 0; b < sh.grid; ++b) match += (int)(h[b] & 0xf) == (b & 7);
        printf("%-62s XCC_ID == blockIdx %% 8 for %d of %d blocks; first 16:", sh.what, match, sh.grid);
        for (int b = 0; b < 16; ++b) printf(" %u", h[b] & 0xf);
        printf("\n");
    }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    bool created = false, events = true, chains_only = false; int reps = 25;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--created")) created = true;
        if (!strcmp(argv[i], "--noevents")) events = false;
        if (!strcmp(argv[i], "--chains")) chains_only = true;
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
    }
    Ctx c{}; c.events = events;
    if (created) CHECK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking)); else c.s = nullptr;
    const long MAXB = 128l << 20;
    CHECK(hipMalloc(&c.buf, MAXB)); CHECK(hipMalloc(&c.dst, 16 << 20)); CHECK(hipMalloc(&c.rows, 16 * 64 * 8)); CHECK(hipMalloc(&c.fout, 4096));
    CHECK(hipMalloc(&c.wend, 256 * 8)); CHECK(hipMalloc(&c.sstart, 256 * 8));
    CHECK(hipMemset(c.buf, 0, MAXB)); CHECK(hipMemset(c.rows, 0, 16 * 64 * 8));
    for (auto& e : c.e) CHECK(hipEventCreate(&e));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# floor_probe: %s, %d CUs, stream=%s, events=%d, reps=%d (medians; us)\n", prop.name, prop.multiProcessorCount, created ? "created(nonblocking)" : "null", (int)events, reps);
    if (chains_only) { part2(c, false, false); part2(c, true, false); part2(c, true, true); return 0; }
    if (argc > 1 && !strcmp(argv[1], "--cold")) { part3(c); return 0; }
    if (argc > 1 && !strcmp(argv[1], "--xcd")) { part4(c); part5(c); return 0; }
    printf("# writer_us / succ_us: dispatch-packet events (= rocprofv3 durations).  gap_us: last writer wave's vmcnt(0) -> first successor wave (s_memrealtime).\n");
    printf("%-8s %6s %-9s %10s %9s %8s %10s\n", "flavour", "MB", "successor", "writer_us", "succ_us", "gap_us", "pair_us");
    static const char* SNAME[4] = {"empty1", "empty256", "finalize", "stream"};
    const int MBS[5] = {0, 4, 16, 32, 128};
    std::vector<unsigned long long> hw(256), hs(256);
    for (int kind = 0; kind < 4; ++kind)
        for (int mi = 0; mi < 5; ++mi)
            for (int f = 0; f < 4; ++f) {
                if (MBS[mi] == 0 && f != PLAIN) continue;
                const long per_block = (long)MBS[mi] * (1 << 20) / 256;
                std::vector<double> tw, ts, tg, tp;
                for (int r = 0; r < reps + 3; ++r) {
                    // a quiet predecessor so the writer itself starts from a clean chip
                    launch(c, empty_kernel, dim3(256), dim3(256), c.e[4], c.e[5], c.sstart);
                    launch_writer(c, f, per_block, (unsigned)r);
                    launch_succ(c, kind);
                    CHECK(hipStreamSynchronize(c.s));
                    if (r < 3) continue;
                    CHECK(hipMemcpy(hw.data(), c.wend, 256 * 8, hipMemcpyDeviceToHost));
                    CHECK(hipMemcpy(hs.data(), c.sstart, 256 * 8, hipMemcpyDeviceToHost));
                    const int nsb = (kind == 0 || kind == 2) ? 1 : 256;
                    const unsigned long long wmax = *std::max_element(hw.begin(), hw.end());
                    const unsigned long long smin = *std::min_element(hs.begin(), hs.begin() + nsb);
                    tg.push_back(((double)smin - (double)wmax) * 0.01);
                    if (events) {
                        float a = 0, b = 0, p = 0;
                        CHECK(hipEventElapsedTime(&a, c.e[0], c.e[1])); CHECK(hipEventElapsedTime(&b, c.e[2], c.e[3])); CHECK(hipEventElapsedTime(&p, c.e[0], c.e[3]));
                        tw.push_back(a * 1e3); ts.push_back(b * 1e3); tp.push_back(p * 1e3);
                    }
                }
                printf("%-8s %6d %-9s %10.2f %9.2f %8.2f %10.2f\n", FNAME[f], MBS[mi], SNAME[kind], events ? median(tw) : 0.0, events ? median(ts) : 0.0, median(tg), events ? median(tp) : 0.0);
                fflush(stdout);
            }
    // a back-to-back chain of trivial kernels: the guide's "boundary" number on this box, host-timed
    for (int nb : {1, 256}) {
        CHECK(hipStreamSynchronize(c.s));
        const int N = 2000;
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a, c.s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(nb), dim3(nb == 1 ? 64 : 256), 0, c.s, c.sstart);
        CHECK(hipEventRecord(b, c.s)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("# chain of %d empty kernels of %d blocks: %.2f us per launch\n", N, nb, ms * 1e3 / N);
    }
    return 0;
}
