// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 seen by one wave, with 1, 2, 3 or 4 waves per SIMD issuing
// the same stream of independent MFMAs (8 accumulators in rotation, as the layer kernels do).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NACC, int W>
__global__ __launch_bounds__(256 * W) void k(float *out, long long *cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = MFMA32(a + (float)i, b, acc[i]);   // distinct operands: no CSE
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 256 * 16);
    const int iters = 4096;   // ~0.2 s per launch at most: wall clock dominated by the MFMAs, not the launch
    for (int nacc = 8; nacc >= 1; nacc /= 2)
        for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
            const int threads = 256 * waves_per_simd;   // one workgroup per CU, waves spread over the 4 SIMDs
            float ms = 0.f;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
#define GO(A, W) hipLaunchKernelGGL((k<A, W>), dim3(256), dim3(threads), 0, 0, out, cyc, iters)
#define GOW(A) do { if (waves_per_simd == 1) GO(A, 1); else if (waves_per_simd == 2) GO(A, 2); else if (waves_per_simd == 3) GO(A, 3); else GO(A, 4); } while (0)
                if (nacc == 8) { if (waves_per_simd <= 2) GOW(8); }   // 8 x 16 accumulator registers: at most 2 waves per SIMD
                else if (nacc == 4) GOW(4);
                else if (nacc == 2) GOW(2);
                else GOW(1);
                hipEventRecord(e1, 0);
                if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
                hipEventElapsedTime(&ms, e0, e1);
            }
            if (nacc == 8 && waves_per_simd > 2) continue;
            long long h[16];
            hipMemcpy(h, cyc, sizeof(long long) * 4 * waves_per_simd, hipMemcpyDeviceToHost);
            const double per_wave = (double)h[0] / (iters * 16 * nacc);
            long long lo = h[0], hi = h[0];
            for (int w = 1; w < 4 * waves_per_simd; ++w) { lo = h[w] < lo ? h[w] : lo; hi = h[w] > hi ? h[w] : hi; }
            const double flop = 256.0 * 4 * waves_per_simd * iters * 16.0 * nacc * 4096.0;
            printf("accumulators %d, waves per SIMD %d: %.1f counter ticks per MFMA per wave (waves of workgroup 0: %lld..%lld ticks); "
                   "wall clock %.4f ms -> %.1f TFLOP/s (157.3 = peak)\n",
                   nacc, waves_per_simd, per_wave, lo, hi, ms, flop / (ms * 1e-3) * 1e-12);
        }
    return 0;
}
