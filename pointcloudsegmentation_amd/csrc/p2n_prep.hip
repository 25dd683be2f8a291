// Input normalisation on gfx950 (MI355X): reference inference.py:76-104
// (preprocess_point_cloud) -- SURVEY.md section 8f rank 2.
//   centroid = np.mean(points, axis=0)          float64, numpy adds the rows in order
//   points   = (points - centroid) / np.max(np.abs(points - centroid))
//   colors   = 0.5 if absent; colors / 255.0 if colors.max() > 1
//   features = float32([points, colors])        (1, N, 6)
// Bit-exact by construction: every operation is the same IEEE float64 operation numpy
// performs, and the only order-dependent one -- the row sum -- is replayed in numpy's
// order: one lane per coordinate adds the rows sequentially (the rest of the workgroup
// streams them into LDS ahead of it).  Max reductions are order-free.
#include "p2n_common.h"

#define PREP_CHUNK 2048   // rows per LDS chunk (48 KiB of doubles)

// ws: double[0..2] sum -> centroid, [3] scale bits (max |p - c|), [4] colour max bits
__global__ __launch_bounds__(1024) void prep_sum_kernel(const double *__restrict__ pts, long N, double *__restrict__ ws) {
    __shared__ double buf[PREP_CHUNK * 3];
    double s = 0.0;
    for (long r0 = 0; r0 < N; r0 += PREP_CHUNK) {
        const long n = (N - r0 < PREP_CHUNK ? N - r0 : PREP_CHUNK) * 3;
        __syncthreads();
        for (long i = threadIdx.x; i < n; i += 1024) buf[i] = pts[r0 * 3 + i];
        __syncthreads();
        if (threadIdx.x < 3) {
            long i = threadIdx.x;
#pragma unroll 8
            for (; i < n; i += 3) s = __dadd_rn(s, buf[i]);
        }
    }
    if (threadIdx.x < 3) ws[threadIdx.x] = s / (double)N;   // np.mean: sum / N
    if (threadIdx.x == 3) ws[3] = 0.0;
}

__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
    // non-negative doubles (and +inf) order like their bit patterns; NaN (sign clear) sorts above inf, as np.max propagates it
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o);
        v = (w > v || w != w) ? w : v;
    }
    return v;
}

// scale = max |p - c| ; colour maximum (only its comparison with 1 is used)
__global__ __launch_bounds__(256) void prep_max_kernel(const double *__restrict__ pts, const double *__restrict__ col, long N,
                                                       double *__restrict__ ws) {
    const double c0 = ws[0], c1 = ws[1], c2 = ws[2];
    double m = 0.0, cm = -__builtin_inf();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N * 3; i += (long)gridDim.x * 256) {
        const int k = (int)(i % 3);
        const double d = fabs(__dsub_rn(pts[i], k == 0 ? c0 : (k == 1 ? c1 : c2)));
        m = (d > m || d != d) ? d : m;
        if (col) {
            const double v = col[i];
            cm = (v > cm || v != v) ? v : cm;
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomic_max_nonneg(ws + 3, m);
    if (col) {
        cm = wave_max(cm);
        // only "max > 1" matters: record whether this wave saw a colour above 1 (or a NaN, for which > is false)
        if ((threadIdx.x & 63) == 0 && cm > 1.0) atomic_max_nonneg(ws + 4, 1.0);
    }
}

__global__ __launch_bounds__(256) void prep_write_kernel(const double *__restrict__ pts, const double *__restrict__ col, long N,
                                                         const double *__restrict__ ws, float *__restrict__ feat) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one (point, coordinate) pair per thread
    if (i >= N * 3) return;
    const long p = i / 3;
    const int k = (int)(i - p * 3);
    const double d = __dsub_rn(pts[i], ws[k]);
    feat[p * 6 + k] = (float)(d / ws[3]);
    float c = 0.5f;
    if (col) c = (float)(ws[4] > 0.0 ? col[i] / 255.0 : col[i]);
    feat[p * 6 + 3 + k] = c;
}

extern "C" size_t p2n_preprocess_workspace_bytes(void) { return 8 * sizeof(double); }

extern "C" int p2n_preprocess_points(const double *points, const double *colors, long N, float *features,
                                     double *centroid_scale, void *workspace, size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(N >= 1 && N <= (1L << 40), "p2n_preprocess_points: N=%ld must be >= 1", N);
    P2N_REQUIRE(points && features && workspace, "p2n_preprocess_points: null pointer");
    if (workspace_bytes < 8 * sizeof(double)) {
        p2n_set_error("p2n_preprocess_points: workspace %zu < 64 bytes", workspace_bytes);
        return P2N_EWORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    double *ws = (double *)workspace;
    const int ps = p2n_prof_begin(P2N_K_PREP, stream);
    P2N_CHECK_HIP(hipMemsetAsync(ws, 0, 8 * sizeof(double), stream));
    hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(1024), 0, stream, points, N, ws);
    long nb = (N * 3 + 255) / 256;
    hipLaunchKernelGGL(prep_max_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, points, colors, N, ws);
    hipLaunchKernelGGL(prep_write_kernel, dim3((unsigned)nb), dim3(256), 0, stream, points, colors, N, ws, features);
    if (centroid_scale)
        P2N_CHECK_HIP(hipMemcpyAsync(centroid_scale, ws, 4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
