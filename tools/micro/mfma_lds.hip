// Micro-benchmark: the bare loop of the weight-gradient kernel -- 2 x 2 tiles per wave, per group of four k two A and
// two B fragments read from LDS with ds_read_b128, sixteen v_mfma_f32_32x32x2_f32 -- against the same MFMAs with one
// or both operands held in registers, and with lane-linear instead of row-strided LDS addresses.
// 512 workgroups of 4 waves, 68 KiB of LDS each (two per CU, as the kernel runs).  Wall-clock TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds.hip -o tools/micro/mfma_lds && tools/micro/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define LD 68

// MODE 0: operands in registers; 1: A from LDS (row-strided), B registers; 2: A and B from LDS (row-strided, the kernel's
// addressing); 3: A and B from LDS, lane-linear addresses (lane * 16 bytes: no bank conflict possible);
// 4 / 5: as 2 with the kernel's park phase behind every step (see below)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int steps, const float *src, int ld) {
    __shared__ __attribute__((aligned(16))) float sa[2 * 64 * LD], sb[2 * 64 * LD];
    for (int i = threadIdx.x; i < 2 * 64 * LD; i += 256) { sa[i] = 1e-3f * i; sb[i] = 1.f - 1e-3f * i; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const float *pa = MODE == 3 ? sa + wr * 4096 + lane * 4 : sa + (64 * wr + r) * LD + 4 * h;
    const float *pb = MODE == 3 ? sb + wc * 4096 + lane * 4 : sb + (64 * wc + r) * LD + 4 * h;
    const int tstride = MODE == 3 ? 2048 : 32 * LD, gstride = MODE == 3 ? 256 : 8;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 2; ++q)
            for (int e = 0; e < 16; ++e) acc[i][q][e] = 0.f;
    f32x4 ra[2], rb[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) { ra[i][j] = lane * 1e-3f + i + j; rb[i][j] = 1.f + blockIdx.x + i - j; }
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int G = 0; G < 8; ++G) {
            f32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = MODE >= 1 ? *reinterpret_cast<const f32x4 *>(pa + i * tstride + (G & 7) * gstride) : ra[i];
#pragma unroll
            for (int q = 0; q < 2; ++q) bv[q] = MODE >= 2 ? *reinterpret_cast<const f32x4 *>(pb + q * tstride + (G & 7) * gstride) : rb[q];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[i][q] = MFMA32(av[i][j], bv[q][j], acc[i][q]);
        }
        asm volatile("" ::: "memory");   // the LDS reads of every step are really issued
        if (MODE >= 4) {
            // the kernel's park phase: barrier, this thread's 4 x 16 floats of the next operand blocks (loaded from
            // global memory: MODE 5; from registers: MODE 4) through BatchNorm+ReLU into LDS rows, barrier
            __syncthreads();
            const int t = threadIdx.x, row = t >> 2, seg = (t & 3) * 16;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                float *dst = (blk < 2 ? sa : sb) + (blk & 1) * 64 * LD + row * LD + seg;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = ra[blk & 1];
                    if (MODE == 5) v = *reinterpret_cast<const f32x4 *>(src + ((long)(blockIdx.x & 63) * 64 + row) * ld + ((s * 64 + seg + 4 * i) & (ld - 1)) + blk * 16);
                    // MODE 6: the kernel's real operand layout -- 256 rows of 262,144 points (1-MiB row stride, 268 MB), this
                    // workgroup's slice of 2,048 points, 64 points per step: 256 contiguous bytes per row per step
                    if (MODE == 6) v = *reinterpret_cast<const f32x4 *>(src + ((long)(((blockIdx.x & 1) * 2 + (blk & 1)) * 64 + row)) * 262144 +
                                                                        (long)(blockIdx.x >> 2) * 2048 + ((s & 31) * 64 + seg + 4 * i));
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaxf(fmaf(v[j], rb[0][j], rb[1][j]), 0.f);
                    *reinterpret_cast<f32x4 *>(dst + 4 * i) = o;
                }
            }
            __syncthreads();
        }
    }
    float t = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 2; ++q)
            for (int e = 0; e < 16; ++e) t += acc[i][q][e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
static int run(float *out, const float *src, const char *name) {
    const int steps = getenv("STEPS") ? atoi(getenv("STEPS")) : 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(256), 0, 0, out, steps, src, 4096);
        CK(hipEventRecord(e1, 0));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flop = 512.0 * 4 * steps * 128 * 4096.0;
    printf("%-64s %8.3f ms  %6.1f TFLOP/s (157.3 = peak)\n", name, ms, flop / (ms * 1e-3) * 1e-12);
    return 0;
}

int main() {
    float *out;
    CK(hipMalloc(&out, sizeof(float) * 512 * 256));
    float *src;
    CK(hipMalloc(&src, sizeof(float) * 256 * 262144));
    CK(hipMemset(src, 0, sizeof(float) * 256 * 262144));
    if (run<0>(out, src, "operands in registers")) return 1;
    if (run<1>(out, src, "A fragments from LDS (row-strided), B in registers")) return 1;
    if (run<2>(out, src, "A and B fragments from LDS (row-strided: the kernel's loop)")) return 1;
    if (run<3>(out, src, "A and B fragments from LDS (lane-linear addresses)")) return 1;
    if (run<4>(out, src, "as 2, plus per step: barrier, transform + LDS stores, barrier")) return 1;
    if (run<5>(out, src, "as 4, the parked data loaded from global memory (cache-resident)")) return 1;
    if (run<6>(out, src, "as 4, loaded in the kernel's layout: 1-MiB row stride, 268 MB")) return 1;
    return 0;
}
