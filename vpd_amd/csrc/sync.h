// In-launch grid barrier for kernels whose whole grid is co-resident (gfx950 / CDNA4 only).
//
// Used where two dependent passes over the SAME per-CU slice of a tensor are separated only by a tiny all-to-all
// (the per-channel sums of a BatchNorm backward): the slice stays in LDS / registers across the barrier instead of
// taking a second and third kernel launch (each ~5 us of fixed cost at these sizes) and a second trip through memory.
//
// Protocol (placement-independent, /opt/skills/guides/cdna_hip_programming.md Guideline 16): arrivals are agent-scope
// atomic adds on counters that live on their own 128-byte lines; blocks are grouped by blockIdx % 8 (the groups that
// usually share an XCD -- speed only), the last arriver of a group adds to the top counter, the last group opens the
// eight per-group generation words that everybody polls with relaxed agent-scope (sc1) loads + s_sleep; one lane then
// runs the agent-scope acquire (buffer_inv sc1), waits for it, and the workgroup barrier releases the other waves.
// Every polled word is zeroed by the host-side launch sequence before the kernel (vpd zero_ranges launch).
//
// Residency is the CALLER's contract: gridDim.x must not exceed what the device keeps resident for this kernel
// (the launchers size the grid from the CU count and one block per CU).  Every spin is bounded (1 s of wall time):
// on time-out the barrier counts it in `err` and returns, so a mis-sized grid ends in wrong numbers that the host
// checks for (vpd_plan_sync_errors), never in a hung GPU.
#pragma once
#include <hip/hip_runtime.h>

struct GridSync {                       // 18 lines of 128 B; zero before every use
    unsigned grp[8][32];
    unsigned top[32];
    unsigned gen[8][32];
    unsigned pad[32];
};

static __device__ __forceinline__ bool vpd_spin_until_nonzero(const unsigned* word) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return false;
    }
    return true;
}

// Call from ALL threads of ALL blocks of the grid, exactly once per GridSync.  `release`: the block wrote data with
// plain stores that other blocks read after the barrier (agent-scope release = L2 write-back); atomics need none.
// `err`: a sticky word (never re-zeroed by the launch sequence) that time-outs increment; read by vpd_plan_sync_errors.
// `bid` / `nb`: this block's linear index and the number of blocks of the grid (any grid shape).
static __device__ __forceinline__ void vpd_grid_barrier(GridSync* gs, bool release, unsigned* err, unsigned bid, unsigned nb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores / atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        if (release) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned g = bid & 7u;
        const unsigned in_group = (nb + 7u - g) >> 3;       // blocks b with b % 8 == g
        const unsigned ngroups = nb < 8u ? nb : 8u;
        const unsigned t = __hip_atomic_fetch_add(&gs->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == in_group - 1u) {
            const unsigned tt = __hip_atomic_fetch_add(&gs->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tt == ngroups - 1u) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __hip_atomic_store(&gs->gen[k][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!vpd_spin_until_nonzero(&gs->gen[g][0]))
            __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed before the barrier opens
    }
    __syncthreads();
}
