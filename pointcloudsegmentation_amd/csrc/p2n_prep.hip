// Input normalisation on gfx950 (MI355X): reference inference.py:76-104
// (preprocess_point_cloud) -- SURVEY.md section 8f rank 2.
//   centroid = np.mean(points, axis=0)          float64, numpy adds the rows in order
//   points   = (points - centroid) / np.max(np.abs(points - centroid))
//   colors   = 0.5 if absent; colors / 255.0 if colors.max() > 1
//   features = float32([points, colors])        (1, N, 6)
// Bit-exact by construction: every operation is the same IEEE float64 operation numpy
// performs, and the only order-dependent one -- the row sum -- is replayed in numpy's
// order wherever the order can matter (see prep_sum_kernel).  Max reductions are order-free.
#include "p2n_common.h"

#define PREP_CHUNK 1920   // rows per LDS chunk = two blocks for each of the 15 loader waves; two chunks of [3][PREP_CHUNK] doubles = 90 KiB
#define PREP_BLOCK 64     // rows per block = one loader wave
#define PREP_NBLK (PREP_CHUNK / PREP_BLOCK)
#define PREP_Q_ZERO 4000      // "quantum exponent" of 0.0: no constraint
#define PREP_Q_NONFINITE (-5000)   // inf / NaN: never exact

typedef double f64x2 __attribute__((ext_vector_type(2)));

// exponent of the lowest set bit of a double (x is an integer multiple of 2^q and of no higher power of two)
__device__ __forceinline__ int prep_lsb_exp(double x) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const int e = (int)((bits >> 52) & 0x7ff);
    unsigned long long m = bits & ((1ull << 52) - 1);
    if (e == 0x7ff) return PREP_Q_NONFINITE;
    if (e == 0) return m ? -1074 + __builtin_ctzll(m) : PREP_Q_ZERO;
    m |= 1ull << 52;
    return e - 1075 + __builtin_ctzll(m);
}

// sum over the 16 lanes of a DPP row, every lane gets a total (the order differs from lane to lane: callers only use
// it where the sum is exact or where a relative error of a few ulps does not matter)
#define PREP_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, false)
__device__ __forceinline__ double prep_dpp_f64(double v, int which) {
    const long long b = __double_as_longlong(v);
    int lo = (int)b, hi = (int)(b >> 32);
    switch (which) {
        case 0: lo = PREP_DPP(lo, 0xB1); hi = PREP_DPP(hi, 0xB1); break;     // quad_perm [1,0,3,2]
        case 1: lo = PREP_DPP(lo, 0x4E); hi = PREP_DPP(hi, 0x4E); break;     // quad_perm [2,3,0,1]
        case 2: lo = PREP_DPP(lo, 0x141); hi = PREP_DPP(hi, 0x141); break;   // row_half_mirror
        default: lo = PREP_DPP(lo, 0x140); hi = PREP_DPP(hi, 0x140); break;  // row_mirror
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int prep_dpp_i32(int v, int which) {
    switch (which) {
        case 0: return PREP_DPP(v, 0xB1);
        case 1: return PREP_DPP(v, 0x4E);
        case 2: return PREP_DPP(v, 0x141);
        default: return PREP_DPP(v, 0x140);
    }
}

// 2^(53+q) less 2^-39 of itself: the limit of the block test below (0 when it cannot be represented: never exact)
__device__ __forceinline__ double prep_bound(int q) {
    int ef = 1023 + 53 + q;
    ef = ef < 1 ? 0 : (ef > 2046 ? 2046 : ef);
    return ef ? __longlong_as_double(((long long)ef << 52) - (1LL << 14)) : 0.0;
}

// ws: double[0..2] sum -> centroid, [3] scale bits (max |p - c|), [4] colour max bits
//
// numpy's np.mean(points, axis=0) adds the rows in order: ONE dependent chain of float64 adds per coordinate, and
// what it yields depends on that order wherever an add rounds.  Where no add rounds, it does not: if the running sum
// s and the next 64 values are all integer multiples of 2^q and |s| + sum|x_i| < 2^(53+q), then every partial sum on
// the way is a multiple of 2^q below 2^53 * 2^q, hence exactly representable, hence computed without rounding -- and
// s + (x_1 + ... + x_64), summed in ANY order, is that same number.  So waves 1..15 stream the rows into LDS and
// reduce every block of 64 rows per coordinate to (sum, sum of magnitudes carried a little high, limit 2^(53+q) for
// the block's lowest set bit q) on the way, and the chain (lanes 0..2 of wave 0, one per coordinate) takes a whole
// block with one add when the test holds and replays numpy's 64 adds from LDS when it does not.  The chain's state is
// (s, B) with B = the limit for the lowest bit s may carry: min(B, B_block) after an exact block, recomputed from
// the bits of s after a replayed one.  Coordinates that came from float32 files (24 significant bits) pass the test
// almost everywhere (at 166,499 points: 7-10 blocks of 2,602 replayed per coordinate); full-mantissa float64 data
// never does, and runs at the speed of the chain (8.4 ns per add).  Bit-identical to numpy either way
// (tests/test_preprocess_gpu.py covers both kinds and their mixtures).
// A loader wave owns 128 rows of a chunk = two blocks: lanes 0-31 block 2(w-1), lanes 32-63 block 2(w-1)+1, two
// consecutive rows per lane (48 contiguous bytes), reduced in the lane, then over the DPP row, then across the two
// rows of a block; rows are fetched into registers one chunk ahead of the chunk being reduced and parked.
__global__ __launch_bounds__(1024) void prep_sum_kernel(const double *__restrict__ pts, long N, double *__restrict__ ws) {
    __shared__ __attribute__((aligned(16))) double buf[2][3][PREP_CHUNK];
    __shared__ __attribute__((aligned(16))) double bsum[2][3][32], bmag[2][3][32], blim[2][3][32];   // per coordinate, over the chunk's blocks
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int myblk = 2 * (wave - 1) + (lane >> 5), myrow = myblk * PREP_BLOCK + 2 * (lane & 31);
    auto fetch = [&](long r0, double (&v)[2][3]) __attribute__((always_inline)) {
        const long rows = N - r0 < PREP_CHUNK ? N - r0 : PREP_CHUNK;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[j][k] = myrow + j < rows ? pts[(r0 + myrow + j) * 3 + k] : 0.0;
    };
    auto park = [&](long r0, int b, const double (&v)[2][3]) __attribute__((always_inline)) {
        const int rows = (int)(N - r0 < PREP_CHUNK ? N - r0 : PREP_CHUNK);
        if (2 * (wave - 1) * PREP_BLOCK >= rows) return;   // wave-uniform
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            *reinterpret_cast<f64x2 *>(&buf[b][k][myrow]) = f64x2{v[0][k], v[1][k]};
            double sm = __dadd_rn(v[0][k], v[1][k]), mg = __dadd_rn(fabs(v[0][k]), fabs(v[1][k]));
            const int q0 = prep_lsb_exp(v[0][k]), q1 = prep_lsb_exp(v[1][k]);
            int q = q0 < q1 ? q0 : q1;
#pragma unroll
            for (int lv = 0; lv < 4; ++lv) {
                sm = __dadd_rn(sm, prep_dpp_f64(sm, lv));
                mg = __dadd_rn(mg, prep_dpp_f64(mg, lv));
                const int q2 = prep_dpp_i32(q, lv);
                q = q2 < q ? q2 : q;
            }
            sm = __dadd_rn(sm, __shfl_xor(sm, 16));
            mg = __dadd_rn(mg, __shfl_xor(mg, 16));
            const int q2 = __shfl_xor(q, 16);
            q = q2 < q ? q2 : q;
            // the magnitude sum was itself rounded (seven levels): carry it a little high
            if ((lane & 31) == 0) { bsum[b][k][myblk] = sm; bmag[b][k][myblk] = mg * (1.0 + 0x1p-40); blim[b][k][myblk] = prep_bound(q); }
        }
    };
    double cur[2][3], nxt[2][3];
    if (wave > 0) {
        fetch(0, cur);
        park(0, 0, cur);
        fetch(PREP_CHUNK, cur);
    }
    __syncthreads();
    double s = 0.0, B = prep_bound(PREP_Q_ZERO);
    int b = 0;
    // the chain's wave shares its SIMD with three loader waves: every one of its (dependent) instructions would
    // otherwise queue behind theirs
    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    for (long r0 = 0; r0 < N; r0 += PREP_CHUNK, b ^= 1) {
        const int rows = (int)(N - r0 < PREP_CHUNK ? N - r0 : PREP_CHUNK);
        if (wave > 0) {
            fetch(r0 + 2 * PREP_CHUNK, nxt);   // (all lanes out of range past the end: no loads)
            if (r0 + PREP_CHUNK < N) park(r0 + PREP_CHUNK, b ^ 1, cur);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) cur[j][k] = nxt[j][k];
        } else if (t < 3) {
            const int nblk = (rows + PREP_BLOCK - 1) / PREP_BLOCK;
            // Ten blocks at a time: their figures come out of LDS together, ahead of the chain, and the ten tests and
            // adds run as straight-line code with selects (a branch per block cost more than the block's add: the
            // replay code sat between the blocks and every skip over it was an instruction-fetch round trip).  The
            // first block of a group that fails its test ends the group's fast run; the rest of the group goes
            // through the branchy loop below, which replays numpy's adds where it must.
#pragma unroll 1
            for (int g0 = 0; g0 < nblk; g0 += 10) {
                double smv[10], mgv[10], lmv[10];
#pragma unroll
                for (int j = 0; j < 10; j += 2) {
                    const f64x2 a = *reinterpret_cast<const f64x2 *>(&bsum[b][t][g0 + j]);
                    const f64x2 m2 = *reinterpret_cast<const f64x2 *>(&bmag[b][t][g0 + j]);
                    const f64x2 l2 = *reinterpret_cast<const f64x2 *>(&blim[b][t][g0 + j]);
                    smv[j] = a[0]; smv[j + 1] = a[1]; mgv[j] = m2[0]; mgv[j + 1] = m2[1]; lmv[j] = l2[0]; lmv[j + 1] = l2[1];
                }
                bool run = true;
                int taken = 0;
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const double Bmin = lmv[j] < B ? lmv[j] : B;
                    const double fast = __dadd_rn(s, smv[j]);
                    run = run && g0 + j < nblk && __dadd_rn(fabs(s), mgv[j]) < Bmin;   // the comparison is false for NaN / inf
                    s = run ? fast : s;
                    B = run ? Bmin : B;
                    taken += run ? 1 : 0;
                }
                const int gend = g0 + 10 < nblk ? g0 + 10 : nblk;
#pragma unroll 1
                for (int blk = g0 + taken; blk < gend; ++blk) {
                    const double lm = blim[b][t][blk];
                    const double Bmin = lm < B ? lm : B;
                    if (__dadd_rn(fabs(s), bmag[b][t][blk]) < Bmin) {
                        s = __dadd_rn(s, bsum[b][t][blk]);
                        B = Bmin;
                    } else {
                        const double *src = buf[b][t] + blk * PREP_BLOCK;
                        const int n = rows - blk * PREP_BLOCK < PREP_BLOCK ? rows - blk * PREP_BLOCK : PREP_BLOCK;
                        int i = 0;
                        if (n == PREP_BLOCK) {
#pragma unroll 1
                            for (; i < PREP_BLOCK; i += 16) {
                                f64x2 x[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f64x2 *>(src + i + 2 * u);
#pragma unroll
                                for (int u = 0; u < 8; ++u) { s = __dadd_rn(s, x[u][0]); s = __dadd_rn(s, x[u][1]); }
                            }
                        }
                        for (; i < n; ++i) s = __dadd_rn(s, src[i]);
                        B = prep_bound(prep_lsb_exp(s));
                    }
                }
            }
        }
        __syncthreads();   // chunk c consumed, chunk c+1 in place
    }
    if (t < 3) ws[t] = s / (double)N;   // np.mean: sum / N
    if (t == 3) ws[3] = 0.0;
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
    // one atomic per workgroup: thousands of waves hitting one address serialise (180 us at 166k points)
    __shared__ double shm[4], shc[4];
    m = wave_max(m);
    cm = wave_max(cm);
    if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = m; shc[threadIdx.x >> 6] = cm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool above = false;
        for (int w = 0; w < 4; ++w) {
            const double v = shm[w];
            m = (v > m || v != v) ? v : m;
            above = above || shc[w] > 1.0;   // only "max > 1" matters (a NaN colour compares false, as in numpy's `max() > 1`)
        }
        atomic_max_nonneg(ws + 3, m);
        if (col && above) atomic_max_nonneg(ws + 4, 1.0);
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
    P2N_REQUIRE(points && workspace, "p2n_preprocess_points: null pointer");   // features == NULL: statistics only (p2n_forward_eval_raw applies them)
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
    hipLaunchKernelGGL(prep_max_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, stream, points, colors, N, ws);
    if (features) hipLaunchKernelGGL(prep_write_kernel, dim3((unsigned)nb), dim3(256), 0, stream, points, colors, N, ws, features);
    if (centroid_scale)
        P2N_CHECK_HIP(hipMemcpyAsync(centroid_scale, ws, 4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
