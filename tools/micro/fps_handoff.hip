// Micro-benchmark: the floor of one FPS iteration's exchange -- how long a value takes from one workgroup to another
// through memory, with exactly the instructions p2n_fps_kernel uses:
//   same XCD : plain global_store_dwordx2 (stays in that XCD's L2) + polling global_load_dwordx4 sc1 (L1 bypassed)
//   any XCD  : agent-scope atomic store (write-through) + agent-scope atomic load
// Pairs of workgroups play ping-pong `iters` times; one round trip = two hand-offs.  Workgroups are dealt round-robin
// over the 8 XCDs (observed, and printed here from HW_REG_XCC_ID), so blocks b and b^8 share an XCD, b and b^1 do not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool SLEEP>   // MODE 0: same-XCD partner (b ^ 8), plain store + sc1 load; 1: cross-XCD partner (b ^ 1), agent-scope atomics;
__global__ __launch_bounds__(256) void pingpong(u64 *slots, int iters, u64 *cycles, unsigned *xcc) {   // 2: same-XCD partner, agent-scope atomics
    const int me = blockIdx.x, partner = MODE == 1 ? me ^ 1 : me ^ 8;
    const bool first = MODE == 1 ? !(me & 1) : !(me & 8);
    if (threadIdx.x != 0) return;
    xcc[me] = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 0xF;   // hwreg(HW_REG_XCC_ID, 0, 4)
    u64 *mine = slots + me * 32, *theirs = slots + partner * 32;   // 256 bytes apart
    auto put = [&](u64 v) {
        if (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(mine), "v"(v) : "memory");
        else __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto get = [&]() -> u64 {
        if (MODE == 0) {
            u32x4 a;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(theirs) : "memory");
            return ((u64)a[1] << 32) | a[0];
        }
        return __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 1; it <= iters; ++it) {
        if (first) put((u64)it);
        int spins = 0;
        while (get() != (u64)it && ++spins < (1 << 22)) { if (SLEEP) __builtin_amdgcn_s_sleep(1); }
        if (!first) put((u64)it);
    }
    cycles[me] = __builtin_amdgcn_s_memtime() - t0;
}

template <int MODE, bool SLEEP>
static void run(const char *name, int iters) {
    u64 *slots, *cyc; unsigned *xcc;
    hipMalloc(&slots, 16 * 256); hipMalloc(&cyc, 16 * 8); hipMalloc(&xcc, 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; u64 hc[16]; unsigned hx[16];
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(slots, 0, 16 * 256);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((pingpong<MODE, SLEEP>), dim3(16), dim3(256), 0, 0, slots, iters, cyc, xcc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost); }
    }
    int same = 0;
    for (int b = 0; b < 16; ++b) same += hx[b] == hx[MODE == 1 ? b ^ 1 : b ^ 8];
    u64 cmax = 0; for (int b = 0; b < 16; ++b) cmax = hc[b] > cmax ? hc[b] : cmax;
    printf("%-44s %6d round trips: %7.1f ns per hand-off by the wall clock, %6.0f shader cycles per hand-off; partner on the same XCD for %d of 16 workgroups\n",
           name, iters, 1e6 * best / (2.0 * iters), (double)cmax / (2.0 * iters), same);
    hipFree(slots); hipFree(cyc); hipFree(xcc);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    run<0, true>("same XCD, plain store + sc1 poll, s_sleep 1", iters);
    run<0, false>("same XCD, plain store + sc1 poll, no sleep", iters);
    run<2, true>("same XCD, agent-scope atomics, s_sleep 1", iters);
    run<1, true>("cross XCD, agent-scope atomics, s_sleep 1", iters);
    run<1, false>("cross XCD, agent-scope atomics, no sleep", iters);
    return 0;
}
