// Micro-benchmark: what a SIMD's second wave gets done while the first one issues a block of ready, independent
// v_mfma_f32_32x32x2_f32 (the situation of the training layer kernels at two workgroups per CU, DESIGN_HISTORY §7).
// A workgroup has 8 waves, two per SIMD.  Waves 0..3 ("matrix waves") issue MFMAs, waves 4..7 ("sibling waves") run a
// VALU chain, L2-resident global loads, or LDS reads.  Each side is timed alone and together with the other one.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_sibling.hip -o tools/micro/mfma_sibling && tools/micro/mfma_sibling
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// XVAR (matrix waves): 0 back-to-back MFMAs, 1 `s_nop 0` behind each, 2 `s_sleep 1` behind every fourth,
//                      3 as 0 with the sibling at priority 3, 4 `s_nop 7` behind each, 5 two VALU FMAs of its own behind each,
//                      6 / 7 as 0 / 5 with v_mfma_f32_32x32x16_bf16
// YKIND (sibling waves): 0 VALU chain, 1 global loads (16 B per lane, L2-resident), 2 LDS reads (16 B per lane)
template <int XVAR, int YKIND, bool SWAP>
__global__ __launch_bounds__(512) void duo(float *out, long long *cyc, int xiters, int yiters, const float *src) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float res = 0.f;
    __syncthreads();
    long long t0, t1;
    if ((wave < 4) != SWAP) {   // SWAP: the matrix waves are the younger ones (4..7)
        f32x16 acc[4];
        float own[4] = {1.f, 2.f, 3.f, 4.f};
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const float a = lane * 1e-3f, b = 1.0f + blockIdx.x;
        bf16x8 ab[4], bb;
        for (int e = 0; e < 8; ++e) {
            bb[e] = (__bf16)(b + e);
            for (int i = 0; i < 4; ++i) ab[i][e] = (__bf16)(a + i + e);
        }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < xiters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (XVAR >= 6) acc[i] = MFMAB(ab[i], bb, acc[i]);
                    else acc[i] = MFMA32(a + (float)i, b, acc[i]);
                    if (XVAR == 5 || XVAR == 7) { own[i] = __builtin_fmaf(own[i], a, b); own[(i + 1) & 3] = __builtin_fmaf(own[(i + 1) & 3], b, a); }
                    if (XVAR == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); }
                    if (XVAR == 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
                    if (XVAR == 2 && i == 3) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); }
                }
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) res += acc[i][e];
        res += own[0] + own[1] + own[2] + own[3];
    } else {
        if (XVAR == 3) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_readcyclecounter();
        if (YKIND == 0) {
            float c[8];
            for (int j = 0; j < 8; ++j) c[j] = lane + j;
            const float m = 1.0f + 1e-7f * blockIdx.x, d = 1e-3f * lane;
            for (int it = 0; it < yiters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) c[j] = __builtin_fmaf(c[j], m, d);
            }
            for (int j = 0; j < 8; ++j) res += c[j];
        } else if (YKIND == 1) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const f32x4 *p = (const f32x4 *)src + lane;
            for (int it = 0; it < yiters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __builtin_nontemporal_load(p + ((it * 16 + r) & 255) * 64);
            }
            res = s[0] + s[1] + s[2] + s[3];
        } else {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const f32x4 *p = (const f32x4 *)lds + lane;
            for (int it = 0; it < yiters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s += *(volatile const f32x4 *)(p + ((it + r) & 15) * 64);
            }
            res = s[0] + s[1] + s[2] + s[3];
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int XVAR, int YKIND, bool SWAP>
static int run(float *out, long long *cyc, const float *src, const char *xname, const char *yname) {
    // sized so that each side alone takes about the same time
    const int xiters = 256;                                   // 256 x 32 MFMAs x 64 cycles = 524k cycles
    const int yiters = YKIND == 0 ? 2048 : YKIND == 1 ? 256 : 1024;
    long long h[3][8];
    for (int leg = 0; leg < 3; ++leg) {                       // 0: matrix waves alone, 1: siblings alone, 2: both
        const int xi = leg == 1 ? 0 : xiters, yi = leg == 0 ? 0 : yiters;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((duo<XVAR, YKIND, SWAP>), dim3(256), dim3(512), 0, 0, out, cyc, xi, yi, src);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h[leg], cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost));
    }
    printf("matrix waves: %-28s sibling: %-12s | matrix alone %7lld, sibling alone %7lld | together: matrix %7lld (x%.2f), sibling %7lld (x%.2f)\n",
           xname, yname, h[0][SWAP ? 4 : 0], h[1][SWAP ? 0 : 4], h[2][SWAP ? 4 : 0], (double)h[2][SWAP ? 4 : 0] / h[0][SWAP ? 4 : 0], h[2][SWAP ? 0 : 4],
           (double)h[2][SWAP ? 0 : 4] / h[1][SWAP ? 0 : 4]);
    return 0;
}

int main() {
    float *out, *src; long long *cyc;
    CK(hipMalloc(&out, sizeof(float) * 256 * 512));
    CK(hipMalloc(&src, 256 * 64 * 16));
    CK(hipMemset(src, 0, 256 * 64 * 16));
    CK(hipMalloc(&cyc, sizeof(long long) * 256 * 8));
    printf("counter ticks of wave 0 (matrix) and wave 4 (its SIMD's sibling) of workgroup 0; 256 workgroups of 8 waves\n");
#define ROW(X, XN) \
    if (run<X, 0, false>(out, cyc, src, XN, "VALU chain")) return 1; \
    if (run<X, 1, false>(out, cyc, src, XN, "global loads")) return 1; \
    if (run<X, 2, false>(out, cyc, src, XN, "LDS reads")) return 1;
    ROW(0, "back to back")
    ROW(1, "s_nop 0 behind each")
    ROW(4, "s_nop 7 behind each")
    ROW(2, "s_sleep 1 behind every 4th")
    ROW(3, "back to back, sibling prio 3")
    ROW(5, "2 own FMAs behind each")
    printf("v_mfma_f32_32x32x16_bf16 in place of the fp32 MFMA (same count)\n");
    ROW(6, "bf16, back to back")
    ROW(7, "bf16, 2 own FMAs behind each")
    printf("the same with the matrix waves the YOUNGER ones (waves 4..7; the siblings are waves 0..3)\n");
    if (run<0, 0, true>(out, cyc, src, "back to back", "VALU chain")) return 1;
    if (run<0, 1, true>(out, cyc, src, "back to back", "global loads")) return 1;
    if (run<0, 2, true>(out, cyc, src, "back to back", "LDS reads")) return 1;
    return 0;
}
