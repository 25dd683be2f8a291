// Embedding clustering on gfx950 (MI355X): reference inference.py:106-142
// (cluster_embeddings) = row L2-normalisation (:124) + scikit-learn
// DBSCAN(eps, min_samples).fit_predict (:127-128) + relabel noise -> 0, clusters -> 1..
// (:131-140).  SURVEY.md section 8f rank 1; oracle/p2n_oracle.c (orc_dbscan) is the CPU
// restatement this is tested against, itself pinned on scikit-learn's output.
//
// DBSCAN is brute force here, as in scikit-learn for D > 15: two passes over the pairs of rows,
// each a Gram-matrix GEMM on the matrix pipe (d2 = |a|^2 + |b|^2 - 2 a.b), never storing
// distances or adjacency:
//   pass 1 (count) neighbourhood sizes -> core flags;
//   pass 2 (link)  core-core edges -> lock-free union-find whose roots are the LOWEST index of
//                  their component (scikit-learn numbers clusters by that index); non-core rows
//                  record their (< min_samples) core neighbours for the border rule.
// A pair whose fp32 d2 lies within `tau` of eps^2 is decided by an exact recomputation in
// double (sum_k ((double)a_k-(double)b_k)^2, k ascending, no fma): the arithmetic of the
// oracle, so labels are bit-identical to it.
//
// Which pairs: 256-row workgroups against blocks of 128 rows, and only the block pairs that can hold
// a pair within eps.  The stages before pass 1 see to that:
//   working order   rows sorted by a 32-bit code of their embedding (cl_project/near/code_kernel, rocPRIM
//                   sort), so that a block's rows are neighbours in the embedding whatever order
//                   the caller's scan came in; labels / counts / cluster numbers go back in the
//                   caller's indices (cl_minorig/owner/rank/label_kernel);
//   spheres         bounding sphere per block -- taken by cl_normalize_kernel from the rows it has staged in
//                   LDS (D <= 128; cl_sphere_kernel beyond), which also packs the centres for the next stage;
//                   the list of blocks whose spheres reach (cl_active_kernel) only with P2N_CL_REACH=0;
//   reach           any point of A within eps of B's sphere AND the other way round (cl_reach_kernel,
//                   dense on the matrix pipe; cl_relist_kernel);
//   plan            pass 1's lists cut into chunks of equal length, one workgroup each (cl_plan_kernel);
//   hits            pass 1 notes per (row block, block) whether any candidate pair showed up: pass 2's lists.
// Inside the passes a finished 32 x 64 unit is first put to a vote (any value near the threshold at all? every
// value inside?) before its values are looked at one by one (unit_step in cl_pair_kernel).
//
// Layout: the normalised rows are stored once as MFMA fragments
//   Xp[tile = point/32][g = k/8][lane = 32*h + point%32][j]  = x[point][8g + 4h + j]
// (1 KiB per fragment, lane-linear), which is both the A operand a wave keeps in registers
// (its 32 rows) and the B operand streamed through LDS by LDS-DMA, 32 fragments (one block of
// 128/64/32 columns for D <= 64/128/256) per step, double buffered.  D[row][col]: lane = column,
// register 4g'+j' of lane half h = row 8g'+4h+j'.
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>
#include "p2n_common.h"

// The Gram products run on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate) with every
// value split into two bf16 parts, x = hi + lo + r, |r| <= 2^-18 |x|: three products per k-step
// (hi.hi + hi.lo + lo.hi) leave out lo.lo and the r terms, at most 3 * 2^-18 * sum|a_k b_k| <=
// 1.2e-5 for unit vectors -- the band `tau` around eps^2 that goes to the exact test is widened by
// that much, so the decisions (and the labels) are unchanged while a 32 x 32 x 64 tile costs
// 12 x 32 instead of 32 x 64 matrix-pipe cycles.  Xb is that image:
//   Xb[tile][ks = k/16][part hi|lo][lane = 32*h + point%32][8 bf16]  = part(x[point][16 ks + 8h + 0..7])
// (1 KiB per fragment again, G fragments per 32-point tile); the fp32 image Xp stays for the exact test.
typedef __bf16 cl_bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) void cl_lds_void;
typedef const __attribute__((address_space(1))) void cl_gbl_void;

// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
#define CL_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14))
#define CL_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CL_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// ------------------------------------------------------------------ normalise + pack
// numpy's float32 pairwise sum of squares over e[0..n) (n <= 128): the order np.linalg.norm sums in
__device__ __forceinline__ float cl_np_block_sumsq(const float *e, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, __fmul_rn(e[i], e[i]));
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fmul_rn(e[j], e[j]);
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], __fmul_rn(e[i + j], e[i + j]));
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __fadd_rn(res, __fmul_rn(e[i], e[i]));
    return res;
}

// numpy recurses above 128 elements (halves rounded down to a multiple of 8); two levels cover n <= 256
template <int LEVELS>
__device__ __forceinline__ float cl_np_sumsq(const float *e, int n) {
    if (n <= 128) return cl_np_block_sumsq(e, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    if constexpr (LEVELS > 0)
        return __fadd_rn(cl_np_sumsq<LEVELS - 1>(e, n2), cl_np_sumsq<LEVELS - 1>(e + n2, n - n2));
    else
        return __builtin_nanf("");   // unreachable for n <= 256
}

// A thread per row reads its own row, and the lanes of a wave sit a whole row apart: every load instruction touches 64
// lines for 16 bytes each, and the lines are long gone from the L1 when the next piece of the row is wanted
// (cl_normalize_kernel ran at 1.3 TB/s, cl_project_kernel at 0.9).  So for D <= CL_STAGE_DMAX the workgroup first
// copies its rows into LDS, a wave per row (whole lines, each fetched once), row r at tile + r * S with S odd: a
// thread reading its own row from there meets no bank conflict.  The arithmetic is the same, value for value.
#define CL_STAGE_DMAX 128
#ifndef CL_UNIT_VOTE
#define CL_UNIT_VOTE 1   // the pair passes vote per finished unit whether it holds any candidate at all (cl_pair_kernel)
#endif
#define CL_SUM_COPIES 16   // copies of the projections' sums and near-mean counts (cl_project_kernel)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CL_STAGE_ROWS 128   // threads (= rows) per workgroup of the staged kernels: 128 * 129 floats = 64.5 KiB at most
__device__ __forceinline__ int cl_stage_stride(int D) { return D | 1; }
// a wave copies its 64 rows 16 at a time: 16 (or 32) independent loads in flight per lane, then the LDS stores (a row
// at a time the copy was a chain of 64 memory round trips)
template <int NC>   // 64-float pieces per row
__device__ __forceinline__ void cl_stage_rows_nc(float *tile, int S, const float *__restrict__ emb, const int *__restrict__ perm,
                                                 long i0, long N, int D) {
    const int lane = threadIdx.x & 63;
    const int r0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63);   // this wave's first row
    const long left = N - i0;
    const int nrows = left < (long)blockDim.x ? (int)(left < 0 ? 0 : left) : (int)blockDim.x;
    const int mine = r0 + lane < nrows ? (perm ? perm[i0 + r0 + lane] : 0) : 0;   // lane l: the source row of row r0 + l
    for (int rb = 0; rb < 64; rb += 16) {
        if (r0 + rb >= nrows) break;
        float v[16][NC];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + rb + u;
            const long src = perm ? (long)__builtin_amdgcn_readlane(mine, rb + u) : i0 + r;
            const float *e = emb + src * D;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = lane + 64 * q;
                v[u][q] = (r < nrows && c < D) ? e[c] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = lane + 64 * q;
                if (c < D) tile[(r0 + rb + u) * S + c] = v[u][q];
            }
    }
}
// rows of 32, 64 or 128 floats at 16-byte addresses: 16-byte loads, 64 >> LOG_P4 rows per instruction (four times the
// bytes in flight per wave: the staged kernels run few waves per CU, their LDS is the rows)
template <int LOG_P4>   // 16-byte pieces per row = 1 << LOG_P4
__device__ __forceinline__ void cl_stage_rows_wide(float *tile, int S, const float *__restrict__ emb, const int *__restrict__ perm,
                                                   long i0, long N, int D) {
    constexpr int P4 = 1 << LOG_P4, RPI = 64 >> LOG_P4, NI = P4, BATCH = NI < 16 ? NI : 16;
    const int lane = threadIdx.x & 63, sub = lane >> LOG_P4, c4 = lane & (P4 - 1);
    const int r0 = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63);   // this wave's first row
    const long left = N - i0;
    const int nrows = left < (long)blockDim.x ? (int)(left < 0 ? 0 : left) : (int)blockDim.x;
    const int mine = r0 + lane < nrows ? (perm ? perm[i0 + r0 + lane] : 0) : 0;   // lane l: the source row of row r0 + l
    for (int b = 0; b < NI; b += BATCH) {
        if (r0 + b * RPI >= nrows) break;
        f32x4 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int r = (b + u) * RPI + sub;
            const long src = perm ? (long)__shfl(mine, r) : i0 + r0 + r;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + r < nrows) v[u] = *reinterpret_cast<const f32x4 *>(emb + src * D + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            float *t = tile + (r0 + (b + u) * RPI + sub) * S + c4 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = v[u][j];   // (S is odd: four 4-byte stores; rows of other sub-groups land one bank on)
        }
    }
}
__device__ __forceinline__ void cl_stage_rows(float *tile, int S, const float *__restrict__ emb, const int *__restrict__ perm,
                                              long i0, long N, int D) {
    const bool al = (reinterpret_cast<size_t>(emb) & 15) == 0;
    if (al && D == 64) cl_stage_rows_wide<4>(tile, S, emb, perm, i0, N, D);
    else if (al && D == 128) cl_stage_rows_wide<5>(tile, S, emb, perm, i0, N, D);
    else if (al && D == 32) cl_stage_rows_wide<3>(tile, S, emb, perm, i0, N, D);
    else if (D <= 64) cl_stage_rows_nc<1>(tile, S, emb, perm, i0, N, D);
    else cl_stage_rows_nc<2>(tile, S, emb, perm, i0, N, D);
    __syncthreads();
}

// what the block-sphere stage leaves (cl_sphere_kernel, or the staged cl_normalize_kernel on its way out)
struct ClSphereOut {
    float *centre, *radius;   // [nb][8G], [nb]
    float *Cb, *nhc;          // the reach stage's packed centres (nullptr: no reach stage)
    int nb, nbpad;
    float eps;
};
__device__ void cl_block_sphere_from_tile(const float *tile, int S, int D, int G, int nreal, int b, const ClSphereOut &sp);

// thread per point.  out_rm (N,D) row-major and/or the fragment image Xp + fp32 squared norms.
template <bool STAGED>
__global__ __launch_bounds__(256) void cl_normalize_kernel(const float *__restrict__ emb, long N, long Npad, int D, int G,
                                                           float *__restrict__ out_rm, float *__restrict__ Xp,
                                                           float *__restrict__ nrm, float *__restrict__ Xb,
                                                           const int *__restrict__ perm, ClSphereOut sp) {
    extern __shared__ float tile[];   // STAGED: [blockDim.x][D | 1]
    const long i0 = (long)blockIdx.x * blockDim.x;
    const long i = i0 + threadIdx.x;
    const int S = cl_stage_stride(D);
    if constexpr (STAGED) cl_stage_rows(tile, S, emb, perm, i0, N, D);
    if (i < Npad && i >= N) {   // pad rows: zero vectors that are never anybody's neighbour
        if (Xp) {
            for (int k = 0; k < G * 8; k += 4) {
                const int g = k >> 3, h = (k >> 2) & 1;
                *reinterpret_cast<f32x4 *>(Xp + (((i >> 5) * G + g) * 64 + h * 32 + (i & 31)) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(Xb + (((i >> 5) * G + g) * 64 + h * 32 + (i & 31)) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            nrm[i] = -__builtin_inff();
        }
    }
    if (i < N) {
        float *mine = tile + threadIdx.x * S;
        const float *eg = emb + (perm ? (long)perm[i] : i) * D;   // row i of the working order
        float ss;
        if constexpr (STAGED) ss = cl_np_sumsq<2>(mine, D);
        else ss = cl_np_sumsq<2>(eg, D);
        // sqrt and divide through double: correctly rounded in fp32 (53 >= 2*24+2) whatever the fp32 fast paths do
        const float den = __fadd_rn((float)sqrt((double)ss), 1e-8f);
        double sq = 0.0;
        for (int k = 0; k < G * 8; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = 0.f;
                if (k + j < D) {
                    if constexpr (STAGED) x = mine[k + j];
                    else x = eg[k + j];
                }
                v[j] = (k + j < D) ? (float)((double)x / (double)den) : 0.f;
                sq += (double)v[j] * (double)v[j];
                if constexpr (STAGED) {
                    if ((out_rm || sp.centre) && k + j < D) mine[k + j] = v[j];   // (copied out by whole lines / summed per block below)
                } else {
                    if (out_rm && k + j < D) out_rm[i * D + k + j] = v[j];
                }
            }
            if (Xp) {
                const int g = k >> 3;
                float *xp = Xp + (((i >> 5) * G + g) * 64 + (i & 31)) * 4;
                *reinterpret_cast<f32x4 *>(xp) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4 *>(xp + 128) = f32x4{v[4], v[5], v[6], v[7]};
                // split image: dims k..k+7 are the 8 elements of lane half (k/8 % 2) of k-step k/16, one 16-byte piece
                const int ks = k >> 4, hb = (k >> 3) & 1;
                float *fh = Xb + (((i >> 5) * G + 2 * ks) * 64 + hb * 32 + (i & 31)) * 4;
                unsigned hi[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 h0 = (__bf16)v[2 * j], h1 = (__bf16)v[2 * j + 1];
                    const __bf16 l0 = (__bf16)(v[2 * j] - (float)h0), l1 = (__bf16)(v[2 * j + 1] - (float)h1);
                    hi[j] = (unsigned)__builtin_bit_cast(unsigned short, h0) | (unsigned)__builtin_bit_cast(unsigned short, h1) << 16;
                    lo[j] = (unsigned)__builtin_bit_cast(unsigned short, l0) | (unsigned)__builtin_bit_cast(unsigned short, l1) << 16;
                }
                *reinterpret_cast<u32x4 *>(fh) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4 *>(fh + 256) = u32x4{lo[0], lo[1], lo[2], lo[3]};   // the lo fragment follows the hi fragment (1 KiB)
            }
        }
        if (Xp) nrm[i] = (float)(-0.5 * sq);   // nh = -|x|^2/2, the accumulator start value of the pair passes
    }
    if constexpr (STAGED) {
        if (out_rm) {   // rows i0.. are one stretch of out_rm
            __syncthreads();
            const long left = N - i0;
            const int nrows = left < (long)blockDim.x ? (int)(left < 0 ? 0 : left) : (int)blockDim.x;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
            for (int r = wave; r < nrows; r += nw)
                for (int c = lane; c < D; c += 64) out_rm[(i0 + r) * D + c] = tile[r * S + c];
        }
        if (sp.centre) {   // the workgroup's CL_STAGE_ROWS = 128 rows are block blockIdx.x: its bounding sphere from the rows in LDS
            __syncthreads();
            const long left = N - i0;
            cl_block_sphere_from_tile(tile, S, D, G, left < 128 ? (int)(left < 0 ? 0 : left) : 128, (int)blockIdx.x, sp);
        }
    }
}

// ------------------------------------------------------------------ working order
// The pair passes prune by blocks of 128 consecutive rows, which only pays when rows that are close in the
// embedding are close in the index.  A scan may come in any order, so the rows are worked on in the order of a
// 32-bit code -- the signs of 32 fixed +-1 projections about their means, one bit each: rows of one instance share most bits and
// land in a few runs.  The order changes nothing in the result (neighbourhood sizes, the sets and the border rule
// are properties of the point set; cluster numbers are re-derived from the caller's indices, cl_rank_assign_kernel),
// only which blocks ever meet.
// The projections are taken about their MEANS over the scene: embeddings that fill a small part of the sphere are
// rarely cut by hyperplanes through the origin (nearly every bit would be the same for every row).
// Step 1: proj[i][k] = (x_i / |x_i|) . r_k, and their sums over the rows (LDS per workgroup, then 32 atomics).
// the +-1 of dimension d in projection k: bit k of a hash of d, as a float; D <= 256
struct ClSigns { float v[256 * 32]; };
constexpr ClSigns cl_make_signs() {
    ClSigns s{};
    for (int d = 0; d < 256; ++d) {
        unsigned hsh = (unsigned)d * 0x9E3779B1u + 0x7F4A7C15u;
        hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12; hsh *= 0x297A2D39u; hsh ^= hsh >> 15;
        for (int k = 0; k < 32; ++k) s.v[d * 32 + k] = (hsh >> k & 1u) ? -1.f : 1.f;
    }
    return s;
}
__constant__ ClSigns cl_signs = cl_make_signs();

constexpr bool cl_sign_neg(int d, int k) {   // the table's entry as a compile-time fact
    unsigned hsh = (unsigned)d * 0x9E3779B1u + 0x7F4A7C15u;
    hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12; hsh *= 0x297A2D39u; hsh ^= hsh >> 15;
    return (hsh >> k & 1u) != 0u;
}
template <int D0, int K0>
__device__ __forceinline__ void cl_project_dim_c(float (&acc)[32], float x) {
    if constexpr (K0 < 32) {
        acc[K0] = cl_sign_neg(D0, K0) ? acc[K0] - x : acc[K0] + x;
        cl_project_dim_c<D0, K0 + 1>(acc, x);
    }
}
template <int D0, int DC>
__device__ __forceinline__ void cl_project_dims_c(float (&acc)[32], float &ss, const float *mine) {
    if constexpr (D0 < DC) {
        const float x = mine[D0];
        ss = fmaf(x, x, ss);
        cl_project_dim_c<D0, 0>(acc, x);
        cl_project_dims_c<D0 + 1, DC>(acc, ss, mine);
    }
}

template <bool STAGED, int DC = 0>   // DC: the row length as a compile-time constant (0: read from D): the signs are then
                                     // part of the instruction stream (an add or a subtract), no table loads at all
__global__ __launch_bounds__(256) void cl_project_kernel(const float *__restrict__ emb, long N, int D, float *__restrict__ proj,
                                                         float *__restrict__ psum) {
    __shared__ float part[4][32];
    extern __shared__ float tile[];   // STAGED: [blockDim.x][D | 1] (cl_stage_rows)
    const long i0 = (long)blockIdx.x * blockDim.x;
    const long i = i0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int S = cl_stage_stride(D);
    if constexpr (STAGED) cl_stage_rows(tile, S, emb, nullptr, i0, N, D);
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    if (i < N) {
        const float *e = emb + i * D;
        const float *mine = tile + threadIdx.x * S;
        float ss = 0.f;
        // (d is the same in every lane: the 32 signs of dimension d come from the table as scalar loads and meet x as
        // the +-1.0 operand of one fma per projection -- x * +-1 is exact, so the sum is the same as adding +-x.
        // Made in the kernel, per dimension, they were ~100 scalar or vector instructions, most of this kernel's time)
        auto dim = [&](int d, float x) __attribute__((always_inline)) {
            const float *sg = cl_signs.v + __builtin_amdgcn_readfirstlane(d) * 32;
            ss = fmaf(x, x, ss);
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] = fmaf(x, sg[k], acc[k]);
        };
        if constexpr (STAGED && DC > 0) {
            cl_project_dims_c<0, DC>(acc, ss, mine);
        } else if constexpr (STAGED) {
            int d = 0;
            for (; d + 1 < D; d += 2) {   // (two dimensions' sign loads and LDS reads in flight)
                const float x0 = mine[d], x1 = mine[d + 1];
                dim(d, x0);
                dim(d + 1, x1);
            }
            if (d < D) dim(d, mine[d]);
        } else if ((D & 3) == 0 && (reinterpret_cast<size_t>(emb) & 15) == 0) {   // rows of whole 16-byte pieces: a quarter of the loads
            for (int d = 0; d < D; d += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(e + d);
#pragma unroll
                for (int j = 0; j < 4; ++j) dim(d + j, v[j]);
            }
        } else {
            for (int d = 0; d < D; ++d) dim(d, e[d]);
        }
        const float inv = ss > 0.f ? 1.f / sqrtf(ss) : 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            acc[k] *= inv;
            if (!(acc[k] == acc[k])) acc[k] = 0.f;   // a row with a NaN or an infinity: any code will do, but keep it out of the means
        }
        if constexpr (!STAGED) {
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                *reinterpret_cast<f32x4 *>(proj + i * 32 + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
        }
    }
    if constexpr (STAGED) {
        // the 32 results go over the first 32 floats of the thread's own row (D >= 32 here), and leave as whole lines:
        // the workgroup's rows of proj are one stretch of memory; the column sums are taken from the tile as well
        float *mine = tile + threadIdx.x * S;
#pragma unroll
        for (int k = 0; k < 32; ++k) mine[k] = acc[k];   // (zeros beyond row N)
        __syncthreads();
        const long left = N - i0;
        const int nrows = left < (long)blockDim.x ? (int)left : (int)blockDim.x;
        for (int idx = threadIdx.x; idx < nrows * 32; idx += blockDim.x) proj[i0 * 32 + idx] = tile[(idx >> 5) * S + (idx & 31)];
        const int c = threadIdx.x & 31, q = threadIdx.x >> 5, per = blockDim.x >> 2;   // CL_STAGE_ROWS = 128 threads: four quarters
        float v = 0.f;
        for (int r = q * per; r < (q + 1) * per; ++r) v += tile[r * S + c];
        part[q][c] = v;
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) part[wave][k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = 0.f;
        for (int w = 0; w < (STAGED ? 4 : nw); ++w) v += part[w][threadIdx.x];
        // CL_SUM_COPIES copies of the 32 sums, a workgroup adds to copy blockIdx % CL_SUM_COPIES: atomics on one line
        // take their turns in the L2
        atomicAdd(psum + (blockIdx.x % CL_SUM_COPIES) * 32 + threadIdx.x, v);
    }
}

// mean of projection k: the copies of its sum in a fixed order (every kernel that wants the mean gets the same value)
__device__ __forceinline__ float cl_proj_mean(const float *__restrict__ psum, int k, long N) {
    float t = 0.f;
    for (int c = 0; c < CL_SUM_COPIES; ++c) t += psum[c * 32 + k];
    return t * (1.f / (float)N);
}

// Step 2: how many rows sit within `delta` (= eps) of each projection's mean.  A hyperplane that passes through an
// instance cuts it into two runs -- far apart in the order if the bit is a significant one, neighbours if it is not.
// So the projections that cut through the least mass become the most significant bits (step 3); with the bits in
// fixed order the count and link passes took 10-25 % longer, depending on the luck of the sign matrix
// (tools/exp_cluster_presorted.py).
__global__ __launch_bounds__(256) void cl_near_kernel(const float *__restrict__ proj, const float *__restrict__ psum, long N,
                                                      float delta, int *__restrict__ near) {
    __shared__ int part[4][32];
    __shared__ float mean[32];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 32) mean[threadIdx.x] = cl_proj_mean(psum, threadIdx.x, N);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < N) v = *reinterpret_cast<const f32x4 *>(proj + i * 32 + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(i < N && fabsf(v[j] - mean[k + j]) < delta);
            if (lane == 0) part[wave][k + j] = (int)__builtin_popcountll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < 32)
        atomicAdd(near + (blockIdx.x % CL_SUM_COPIES) * 32 + threadIdx.x,
                  part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// Step 3: bit = projection above its mean, at position 31 - (rank of the projection by mass near its mean, ascending).
__global__ __launch_bounds__(256) void cl_code_kernel(const float *__restrict__ proj, const float *__restrict__ psum,
                                                      const int *__restrict__ near, long N, unsigned *__restrict__ keys,
                                                      int *__restrict__ vals, unsigned long long *__restrict__ sorted_flag) {
    __shared__ int pos[32], mass[32];
    if (blockIdx.x == 0 && threadIdx.x == 0) *sorted_flag = 1ull;   // stats[7]: this workspace holds a working order
    __shared__ float mean[32];
    if (threadIdx.x < 32) {
        int m = 0;
        for (int c = 0; c < CL_SUM_COPIES; ++c) m += near[c * 32 + threadIdx.x];
        mass[threadIdx.x] = m;
        mean[threadIdx.x] = cl_proj_mean(psum, threadIdx.x, N);
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int mine = mass[threadIdx.x];
        int rank = 0;
        for (int j = 0; j < 32; ++j) {
            const int o = mass[j];
            rank += (o < mine) || (o == mine && j < (int)threadIdx.x);
        }
        pos[threadIdx.x] = 31 - rank;
    }
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    unsigned key = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(proj + i * 32 + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) key |= (unsigned)(v[j] > mean[k + j]) << pos[k + j];
    }
    keys[i] = key;
    vals[i] = (int)i;
}

// ------------------------------------------------------------------ exact pair test
#define CL_EX_BATCH 4
// The 2 * G 16-byte pieces of both rows are fetched CL_EX_BATCH steps at a time (loads in flight, then
// that stretch of the sum): fetched and consumed one step at a time the test was a chain of
// 2 G dependent memory round trips, and a wave in it holds up its whole workgroup at the next barrier.
__device__ __forceinline__ bool cl_exact_within(const float *__restrict__ Xp, int G, long i, long j, double eps2) {
    const float *pa = Xp + ((i >> 5) * G * 64 + (i & 31)) * 4;
    const float *pb = Xp + ((j >> 5) * G * 64 + (j & 31)) * 4;
    double s = 0.0;
    for (int st0 = 0; st0 < 2 * G; st0 += CL_EX_BATCH) {   // step st = 2 g + h: dims 4 st .. 4 st + 3, ascending
        f32x4 a[CL_EX_BATCH], b[CL_EX_BATCH];
#pragma unroll
        for (int u = 0; u < CL_EX_BATCH; ++u) {
            const int st = st0 + u, g = st >> 1, h = st & 1;
            a[u] = *reinterpret_cast<const f32x4 *>(pa + g * 256 + h * 128);
            b[u] = *reinterpret_cast<const f32x4 *>(pb + g * 256 + h * 128);
        }
#pragma unroll
        for (int u = 0; u < CL_EX_BATCH; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double d = (double)a[u][q] - (double)b[u][q];
                s += d * d;   // -ffp-contract=off: separate multiply and add, as the oracle
            }
    }
    return s <= eps2;
}

// ------------------------------------------------------------------ union-find (roots = lowest index)
__device__ __forceinline__ int cl_find(int *parent, int x) {
    int p = CL_LD(parent + x);
    while (p != x) {
        const int gp = CL_LD(parent + p);
        if (gp != p) CL_ST(parent + x, gp);   // path halving; any ancestor is a valid parent
        x = p;
        p = gp;
    }
    return x;
}

__device__ __forceinline__ int cl_unite(int *parent, int i, int j) {
    int a = cl_find(parent, i), b = cl_find(parent, j);
    while (a != b) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int old = atomicCAS(parent + hi, hi, lo);   // hook the larger root under the smaller index
        if (old == hi) return lo;
        a = cl_find(parent, old);
        b = cl_find(parent, lo);
    }
    return a;
}

// ------------------------------------------------------------------ the pair pass
struct ClArgs {
    const float *Xp;
    const float *Xb;        // split-bf16 fragment image (the MFMA operands)
    const float *nrm_row;   // nh = -|x|^2/2 of the wave-owned points (-inf on pad rows)
    const float *nrm_col;   // nh of the streamed points (-inf: pad; pass 2: also non-core)
    long Npad;
    float eps2f, tau;
    double eps2;
    int *cnt;               // pass 1: neighbourhood sizes (atomicAdd across column splits)
    int *parent;            // pass 2
    const unsigned char *core;
    const int *colblk_has_core;   // pass 2: per block of 128 columns
    int *nbr_cnt, *nbr;     // pass 2: core neighbours of non-core rows, `cap` slots per row
    int cap;
    // the pass's work list (cl_plan_kernel): chunk k = entries [work_e0[k], + work_meta[1]) of row block work_rb[k]'s list;
    // work_meta[0] chunks in all, one workgroup each
    const int *work_rb, *work_e0, *work_meta;
    int nsplit;   // pass 2: workgroups per row block
    // pass 2 notes (D <= 64, -1 = none yet): blk_uni[b] = a parent that EVERY core point of block b named at some moment,
    // rb_uni[rb] = a parent every row of row block rb named when one of its workgroups finished -- and all its real rows
    // are core points.  Sets only grow, so "same root now" proves for good that a cell can join nothing and holds no row
    // that needs a neighbour list: later workgroups of the row block strike such cells before anything is staged.
    int *blk_uni, *rb_uni;
    const int *act_cnt;     // per 256-point row block (one workgroup): number of active 128-point blocks
    const int *act_list;    // [row block][nb128]: their indices, ascending
    int nb128;
    unsigned char *hit;     // pass 1 writes [row block][nb128]: 1 where some pair of the two blocks came within the "maybe" threshold
    unsigned long long *stats;   // [0] exact tests pass 1, [1] exact tests pass 2, [2] unite calls, [3] border appends
    // pass 1: undecided pairs are queued per lane ([wave][slot][lane] = partner index) and decided by
    // cl_resolve_kernel afterwards; a lane whose queue is full decides on the spot
    int *defer_q, *defer_cnt;
    int defer_cap;
};

// Orientation: D = X_J . X_I^T -- the streamed points j are the ROWS of the MFMA result (one per
// accumulator register), the wave's own 32 points i are the COLUMNS (one per lane; the two lane
// halves hold different rows j of the same i).  Everything that belongs to i -- threshold,
// neighbour count, parent -- is therefore per-lane state, and counting is two VALU operations
// per pair with no cross-lane or scalar work.
// The accumulators start at nh[j] = -|x_j|^2/2 (staged through LDS with the block), so
//   d2 <= eps^2  <=>  acc = x_i.x_j - |x_j|^2/2  >=  (|x_i|^2 - eps^2)/2 =: h_i;
// hd = h + tau/2 (definitely inside), hm = h - tau/2 (maybe inside).
// Software pipeline over "units" of UT streamed tiles: the MFMAs of one unit are issued
// interleaved with the compares of the previous unit (two small accumulator sets).
#define CL_BLK_FLOATS (8192 + 128)   // 32 fragments + nh[128] per LDS buffer
#define CL_RING 3                          // streamed blocks resident in LDS: one in use, two in flight
#define CL_PAR_BASE (CL_RING * CL_BLK_FLOATS)   // then three rotating parent[128] slots (pass 2)
#define CL_PU_BASE (CL_PAR_BASE + 3 * 128)       // then per parent slot: the common parent of each fetching wave's columns
#define CL_LDS_FLOATS (CL_PU_BASE + 8)
#define CL_CHUNK_MIN 8      // shortest chunk of a list a workgroup is started for (cl_plan_kernel)
#define CL_SLIST_MAX 2048   // active-list entries one workgroup may hold in LDS (after the ring)

#define CL_DENSE_BITS 8   // candidate pairs per lane from which a unit of the link pass counts as dense (slow_unit)
#define CL_NW 8   // waves per workgroup, one workgroup per CU: 256 wave-owned points share one streamed block
#define CL_ROWBLK (CL_NW * 32)
#define CL_NPAD_UNIT 256   // lcm(CL_ROWBLK, 128): rows pad to whole workgroups and whole streamed blocks

template <int G, int CT, int MODE>
__global__ __launch_bounds__(CL_NW * 64, 2) void cl_pair_kernel(ClArgs A) {   // 512 threads, one workgroup per CU, 2 waves per SIMD
    extern __shared__ __attribute__((aligned(16))) float lds[];   // CL_LDS_FLOATS
    static_assert(G * CT == 32, "a streamed block is 32 fragments");
    constexpr int JB = CT * 32;
    constexpr int UT = CT >= 2 ? 2 : 1;   // tiles per unit (independent MFMA chains)
    constexpr int UPS = CT / UT;          // units per streamed block
    constexpr int EL = UT * 16;           // pairs per lane per unit
    // MODE 0: pass 1; MODE 2: pass 1 visiting every block pair once (the Gram tile of (A, B) is the transpose of
    // (B, A)'s): lists hold only blocks at or after the row block's own, and beyond those a cell counts its definite
    // pairs for the streamed points too (column counts by ballot, below); MODE 1: pass 2.
    constexpr bool P1 = MODE != 1, SYM = MODE == 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    // Pass 1 takes chunk blockIdx.x of the work list (the grid is the host's bound on the number of chunks).  Pass 2
    // keeps one workgroup per (row block, nsplit-th of its list), the grid row-block-fastest: workgroups of one
    // row block then run one after the other rather than side by side, each starting from the parents the last left
    // (side by side, every one of them unites the same rows again: 29 M -> 84 M union attempts, 1.3 -> 3.1 ms).
    const int chunk = blockIdx.x;
    if (P1 && chunk >= A.work_meta[0]) return;
    const int rb = P1 ? A.work_rb[chunk] : (int)blockIdx.x;   // this workgroup's 256 rows
    const long i0 = ((long)rb * CL_NW + wave) * 32;
    const long my = i0 + c;               // the point this lane owns

    // The streamed blocks of this row block come from its active list: 128-point blocks whose bounding
    // sphere can reach ours within eps (cl_active_kernel); everything else is skipped without a load.
    constexpr int SUB = 128 / JB;         // streamed blocks per 128-point block
    const int total = A.act_cnt[rb];
    int e0, e1;
    if (P1) {
        e0 = A.work_e0[chunk];
        e1 = e0 + A.work_meta[1] < total ? e0 + A.work_meta[1] : total;
    } else {
        e0 = (int)((long)total * blockIdx.y / A.nsplit);
        e1 = (int)((long)total * (blockIdx.y + 1) / A.nsplit);
    }
    int nlist = e1 - e0;
    if (nlist <= 0) return;
    int *slist = reinterpret_cast<int *>(lds + CL_LDS_FLOATS);   // this workgroup's slice of the list
    for (int i = threadIdx.x; i < nlist; i += CL_NW * 64) slist[i] = A.act_list[(long)rb * A.nb128 + e0 + i];
    if (MODE == 1 && SUB == 1 && A.rb_uni) {
        int *scr = reinterpret_cast<int *>(lds);   // (the ring is not in use yet)
        if (threadIdx.x == 0) scr[15] = CL_LD(A.rb_uni + rb);   // one reader: the note may be written while we look
        __syncthreads();
        const int rr = scr[15];
        if (rr >= 0) {   // an earlier workgroup of this row block left its rows in one set: strike what is in that set already
            const int r0 = cl_find(A.parent, rr);
            int kept = 0;
            for (int c0 = 0; c0 < nlist; c0 += CL_NW * 64) {   // compaction in place, 512 entries at a time
                const int e = c0 + (int)threadIdx.x;
                int B = 0;
                bool keep = false;
                if (e < nlist) {
                    B = slist[e];
                    const int bu = CL_LD(A.blk_uni + B);
                    keep = !(bu >= 0 && cl_find(A.parent, bu) == r0);
                }
                const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                if (lane == 0) scr[wave] = __builtin_popcountll(km);
                __syncthreads();
                int off = kept, tot = 0;
                for (int w2 = 0; w2 < CL_NW; ++w2) {
                    if (w2 < wave) off += scr[w2];
                    tot += scr[w2];
                }
                __syncthreads();   // every thread holds its entry and has read the wave counts
                if (keep) slist[off + __builtin_popcountll(km & ((1ull << lane) - 1ull))] = B;
                kept += tot;
            }
            nlist = kept;
            __syncthreads();
            if (nlist == 0) return;
        }
    }
    const long len = (long)nlist * SUB;
    if (MODE == 1 && threadIdx.x < 8) lds[CL_PU_BASE + threadIdx.x] = __int_as_float(-1);   // pass 2, a wave that stages no parents: "no core point"
    // MODE 2: the three parent slots of pass 2 serve as rotating column counters (block t uses slot t % 3: its last
    // unit is compared during block t+1's first step, and it is flushed to cnt[] at the entry of block t+2)
    int *colslots = reinterpret_cast<int *>(lds + CL_PAR_BASE);
    if (SYM && threadIdx.x < 3 * 128) colslots[threadIdx.x] = 0;
    __syncthreads();
    auto jbof = [&](long pos) -> long { return (long)slist[pos / SUB] * SUB + pos % SUB; };
    // Pass 1: every workgroup walks the blocks in the same order (a block is fetched from HBM once per
    // XCD, then served from its L2).  Pass 2: staggered starts spread the union-find merges instead of
    // piling every workgroup onto one block's roots at the same time.
    long cur = P1 ? 0 : ((long)rb * (CL_NW * 32 / JB)) % len;

    auto stage = [&](long jb, int buf) {
        float *dst = lds + buf * CL_BLK_FLOATS;
#pragma unroll
        for (int q = 0; q < 32 / CL_NW; ++q) {
            const int f = wave * (32 / CL_NW) + q;
            __builtin_amdgcn_global_load_lds((cl_gbl_void *)(A.Xb + (jb * 32 + f) * 256 + lane * 4),
                                             (cl_lds_void *)(dst + f * 256), 16, 0, 0);
        }
        // nh and (pass 2) parent of the block's JB points: waves 0/1 (2/3) move 64 dwords each
        if (wave * 64 < JB)
            __builtin_amdgcn_global_load_lds((cl_gbl_void *)(A.nrm_col + jb * JB + wave * 64 + lane),
                                             (cl_lds_void *)(dst + 8192 + wave * 64), 4, 0, 0);
    };
    // Pass 2 also needs the block's parent[] values.  They must be read coherently (agent scope:
    // other XCDs' L2s are not snooped by a plain or DMA load), so waves 2/3 fetch them one block
    // ahead into a register and park them in LDS before the next barrier.  Block t uses slot t % 3:
    // it is written during step t-1 and still read during step t+1 (the pipelined compares of
    // the block's last unit), so three slots keep writer and readers apart.
    const bool pfetch = MODE == 1 && wave >= 2 && (wave - 2) * 64 < JB;
    const int pslot = CL_PAR_BASE + (wave - 2) * 64 + lane;
    // What the columns one fetching wave staged have in common: -1 none of them is a core point (non-core points carry
    // parent -1), p >= 0 every core point among them names p, -2 they name different parents.
    auto common_parent = [&](int pv) -> int {
        const bool in = (wave - 2) * 64 + lane < JB && pv != -1;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
        if (!m) return -1;
        const int first = __builtin_amdgcn_readlane(pv, __builtin_ctzll(m));
        return __builtin_amdgcn_ballot_w64(in && pv != first) ? -2 : first;
    };
    // block numbers are looked up one step ahead of their use: the LDS read is off the critical path
    long jb_now = jbof(cur);
    long jb_nxt = jbof(cur + 1 == len ? 0 : cur + 1);
    stage(jb_now, 0);
    if (len > 1) stage(jb_nxt, 1);
    // LDS-DMA instructions one stage() issues from this wave: what may stay in flight at a block entry
    const bool dma5 = wave * 64 < JB;
    if (pfetch) {
        const int pv = CL_LD(A.parent + jb_now * JB + (wave - 2) * 64 + lane);
        lds[pslot] = __int_as_float(pv);
        const int cp = common_parent(pv);
        if (lane == 0) lds[CL_PU_BASE + wave - 2] = __int_as_float(cp);
    }
    int pnext = 0;
    int pi_nxt = MODE == 1 ? CL_LD(A.parent + (((long)rb * CL_NW + wave) * 32 + (lane & 31))) : 0;

    constexpr int KS = G / 2;             // k-steps of 16
    constexpr int NSL = 3 * KS;           // MFMA slots per tile: hi.hi, hi.lo, lo.hi per k-step
    cl_bf16x8 ah[KS], al[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        ah[ks] = *reinterpret_cast<const cl_bf16x8 *>(A.Xb + ((i0 >> 5) * G + 2 * ks) * 256 + lane * 4);
        al[ks] = *reinterpret_cast<const cl_bf16x8 *>(A.Xb + ((i0 >> 5) * G + 2 * ks + 1) * 256 + lane * 4);
    }
    const float hh = -A.nrm_row[my] - 0.5f * A.eps2f;   // nrm_* hold -|x|^2/2 (-inf on pad rows: h = +inf, nothing reaches it)
    const float hm = hh - 0.5f * A.tau, hd = hh + 0.5f * A.tau;
    const bool mycore = MODE == 1 && A.core[my];
    int dq_n = 0;   // pass 1: queued undecided pairs of this lane
    const long dq_wave = (long)chunk * CL_NW + wave;
    int cntd = 0, cntm = 0, expdiff = 0;   // pass 1: definite(+resolved) / maybe counts; pass 2: cntm = maybe-edges seen
    int pi = 0;
    // pass 2: diagnostics counters, flushed once per wave at the end (an atomic per event on one shared word was a
    // large part of the slow path's time)
    unsigned st_exact = 0, st_union = 0, st_border = 0;

    // ---- per-unit pieces --------------------------------------------------
    auto init_acc = [&](f32x16(&acc)[UT], const float *blk, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 nh = *reinterpret_cast<const f32x4 *>(blk + 8192 + (u * UT + ut) * 32 + 8 * gq + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[ut][4 * gq + q] = nh[q];
            }
    };
    auto mfma_slot = [&](f32x16(&acc)[UT], const float *ubuf, int sl) __attribute__((always_inline)) {
        const int ks = sl / 3, term = sl % 3;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            // rows = streamed points (hi fragment for the first two terms: CSE'd), columns = this wave's points
            const cl_bf16x8 b = *reinterpret_cast<const cl_bf16x8 *>(ubuf + (ut * G + 2 * ks + (term == 2)) * 256 + lane * 4);
            acc[ut] = MFMAB(b, term == 1 ? al[ks] : ah[ks], acc[ut]);
        }
    };
    // colv (MODE 2): lane L collects the number of this wave's rows that have column L of the unit definitely inside
    // -- the compare's lane mask is there anyway; its two halves (the lane halves hold different columns) are counted
    // on the scalar unit and dropped into the column's lane.
    auto fast_pair = [&](const f32x16(&acc)[UT], int e, int &colv) __attribute__((always_inline)) {
        const float v = acc[e >> 4][e & 15];
        if (P1) {
            const bool d = v >= hd;
            cntd += d;
            if (SYM) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(d);
                const int r = e & 15, idx = (e >> 4) * 32 + 8 * (r >> 2) + (r & 3);
                const int n0 = __builtin_popcount((unsigned)m), n1 = __builtin_popcount((unsigned)(m >> 32));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(n0), "n"(idx));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(n1), "n"(idx + 4));
            }
        }
        cntm += (v >= hm);
    };
    // a finished unit in which some lane saw an undecided pair (pass 1) or any candidate edge (pass 2)
    auto slow_unit = [&](const f32x16(&acc)[UT], const float *par, int u, long col0) __attribute__((always_inline)) {
        unsigned need = 0, sure = 0;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 pjv;
                if (MODE == 1) pjv = *reinterpret_cast<const f32x4 *>(par + (u * UT + ut) * 32 + 8 * gq + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * gq + q, e = ut * 16 + r;
                    const float v = acc[ut][r];
                    if (P1) {
                        need |= (unsigned)(v >= hm && !(v >= hd)) << e;
                    } else {
                        const bool nd = v >= hm && __float_as_int(pjv[q]) != pi;
                        need |= (unsigned)nd << e;
                        sure |= (unsigned)(nd && v >= hd) << e;
                    }
                }
            }
        if (MODE == 1 && __builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_popcount(need) >= CL_DENSE_BITS)) >= 16) {
            // A dense unit: the wave's rows and the unit's columns are one instance's.  Lane by lane every row
            // would unite with every column -- chains of dependent coherent loads, nearly all of them redundant.
            // Every call is several memory round trips and a wave makes its calls in lockstep, so what counts is
            // the number of ROUNDS of calls, not the number of calls:
            //  1. hub (only while the rows still name different parents): each lane half takes the first column
            //     its first certain lane is certain of; every core lane certain of it too unites with it (one
            //     round, mostly each lane hooking its own root), then looks its root up again (one more round):
            //     now the rows name one root;
            //  2. without touching memory: column by column (the same column for all lanes of a lane half at bit
            //     e) one core lane whose edge is certain leads -- the lead rotates with e -- and every core lane
            //     that names the leader's parent is in the leader's set already and strikes the column;
            //  3. each lane works through what is left to it: about one column per lane, all lanes side by side.
            {
                const unsigned long long cm = __builtin_amdgcn_ballot_w64(mycore);
                const int p0 = cm ? __builtin_amdgcn_readlane(pi, __builtin_ctzll(cm)) : 0;
                if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(mycore && pi != p0)) >= CL_DENSE_BITS) {
                    const unsigned long long sm = __builtin_amdgcn_ballot_w64(mycore && sure != 0u);
                    const unsigned slo = (unsigned)sm, shi = (unsigned)(sm >> 32);
                    int hub0 = -1, hub1 = -1;
                    if (slo) hub0 = __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)sure, __builtin_ctz(slo)));
                    if (shi) hub1 = __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)sure, 32 + __builtin_ctz(shi)));
                    const int hub = h ? hub1 : hub0;
                    if (hub >= 0 && mycore && (sure >> hub & 1)) {
                        const int ut = hub >> 4, r = hub & 15;
                        pi = cl_unite(A.parent, (int)my, (int)(col0 + ut * 32 + 8 * (r >> 2) + 4 * h + (r & 3)));
                        ++st_union;
                        need &= ~(1u << hub);
                    }
                    if (mycore) pi = cl_find(A.parent, pi);
                }
            }
            unsigned todo = 0u;
#pragma nounroll
            for (int e = 0; e < EL; ++e) {
                const bool mine = need >> e & 1;
                if (!__builtin_amdgcn_ballot_w64(mine)) continue;
                const bool certain = mine && (sure >> e & 1);
                const unsigned long long ms = __builtin_amdgcn_ballot_w64(certain && mycore);
                const unsigned lo32 = (unsigned)ms, hi32 = (unsigned)(ms >> 32);
                int l0 = -1, l1 = -1, q0 = 0, q1 = 0;
                if (lo32) {
                    l0 = (__builtin_ctz((lo32 >> e) | (lo32 << ((32 - e) & 31))) + e) & 31;   // first certain lane at or after lane e
                    q0 = __builtin_amdgcn_readlane(pi, l0);
                }
                if (hi32) {
                    l1 = 32 + ((__builtin_ctz((hi32 >> e) | (hi32 << ((32 - e) & 31))) + e) & 31);
                    q1 = __builtin_amdgcn_readlane(pi, l1);
                }
                const int lead = h ? l1 : l0;
                if (mine && !(mycore && lead >= 0 && lane != lead && pi == (h ? q1 : q0))) todo |= 1u << e;
            }
            while (todo) {
                const int e = __builtin_ctz(todo);
                todo &= todo - 1;
                const int ut = e >> 4, r = e & 15;
                const long other = col0 + ut * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (mycore && __float_as_int(par[(u * UT + ut) * 32 + 8 * (r >> 2) + 4 * h + (r & 3)]) == pi) continue;
                bool ok = sure >> e & 1;
                if (!ok) {
                    ok = cl_exact_within(A.Xp, G, my, other, A.eps2);
                    ++st_exact;
                }
                if (ok) {
                    if (mycore) {
                        pi = cl_unite(A.parent, (int)my, (int)other);
                        ++st_union;
                    } else {
                        ++st_border;
                        const int slot = atomicAdd(A.nbr_cnt + my, 1);
                        if (slot < A.cap) A.nbr[my * A.cap + slot] = (int)other;
                    }
                }
            }
            return;
        }
        while (need) {
            const int e = __builtin_ctz(need);
            need &= need - 1;
            const int ut = e >> 4, r = e & 15;
            const long other = col0 + ut * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (P1) {
                const bool both = SYM && (col0 >> 7) > 2L * rb + 1;   // the pair counts for the streamed point too
                if (dq_n < A.defer_cap) {
                    A.defer_q[(dq_wave * A.defer_cap + dq_n) * 64 + lane] = (int)other | (both ? (int)0x80000000 : 0);
                    ++dq_n;
                } else {
                    atomicAdd(A.stats + 0, 1ull);
                    const bool in = cl_exact_within(A.Xp, G, my, other, A.eps2);
                    cntd += in;
                    if (both && in) atomicAdd(A.cnt + other, 1);
                }
            } else {
                // the staged parent against the CURRENT own parent: after a call the row usually names the root that
                // the rest of the block's columns name too, and those edges need no call of their own
                if (mycore && __float_as_int(par[(u * UT + ut) * 32 + 8 * (r >> 2) + 4 * h + (r & 3)]) == pi) continue;
                bool ok = sure >> e & 1;
                if (!ok) {
                    ok = cl_exact_within(A.Xp, G, my, other, A.eps2);
                    ++st_exact;
                }
                if (ok) {
                    if (mycore) {
                        pi = cl_unite(A.parent, (int)my, (int)other);
                        ++st_union;
                    } else {
                        ++st_border;
                        const int slot = atomicAdd(A.nbr_cnt + my, 1);
                        if (slot < A.cap) A.nbr[my * A.cap + slot] = (int)other;
                    }
                }
            }
        }
    };

    // one pipeline stage: GEMM of unit w into `accn` while unit w-1 (in `accp`) is compared
    const long W = len * UPS;
    long col0_prev = 0, jb_cur = 0;
    long cb1 = 0, cb2 = 0;   // MODE 2: first column of the blocks one and two steps back
    bool prev_active = false, blk_active = false;
    const float *par_prev = lds;
    int u_prev = 0;
    auto unit_step = [&](f32x16(&accn)[UT], f32x16(&accp)[UT], long w, auto uconst) __attribute__((always_inline)) {
        bool active = false;
        const long t = w / UPS;
        constexpr int u = decltype(uconst)::value;   // w % UPS, known at compile time (w advances in pairs)
        if (w < W) {
            if (u == 0) {
                jb_cur = jb_now;
                const long nxt = (cur + 1 == len) ? 0 : cur + 1;
                const long nn = (nxt + 1 == len) ? 0 : nxt + 1;
                const long jb_after = jbof(nn);
                // Block t has landed once at most the DMA instructions of block t+1 (this wave's share, the
                // youngest loads it has issued) are still in flight: vector-memory loads retire in issue
                // order, and anything else the wave has queued only makes the wait stricter.  The barrier
                // then publishes every wave's share and proves that all waves have left block t-1, whose
                // ring slot block t+2 takes.
                if (t + 1 < len) {
                    if (dma5) CL_WAIT_VMCNT(32 / CL_NW + 1);
                    else CL_WAIT_VMCNT(32 / CL_NW);
                } else {
                    CL_WAIT_VMCNT(0);
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (SYM) {   // every wave has dropped its last counts for block t-2: out they go (zeros stay for block t+1)
                    if (t >= 2 && threadIdx.x < JB) {
                        int *sl = colslots + (int)((t + 1) % 3) * 128;
                        const int cv = sl[threadIdx.x];
                        if (cv) {
                            atomicAdd(A.cnt + cb2 + threadIdx.x, cv);
                            sl[threadIdx.x] = 0;
                        }
                    }
                    cb2 = cb1;
                    cb1 = jb_cur * JB;
                }
                if (MODE == 1) {   // own parent, one block stale: the filter below only needs SOME past ancestor of `my`
                    pi = pi_nxt;
                    pi_nxt = CL_LD(A.parent + my);
                }
                if (pfetch && t + 1 < len) pnext = CL_LD(A.parent + jb_nxt * JB + (wave - 2) * 64 + lane);
                if (t + 2 < len) stage(jb_after, (int)((t + 2) % CL_RING));
                cur = nxt;
                jb_now = jb_nxt;
                jb_nxt = jb_after;
                blk_active = P1 || A.colblk_has_core[(jb_cur * JB) >> 7] != 0;
                if (MODE == 1) {
                    // A cell whose columns all name one parent that every row of this wave names too joins nothing
                    // (both values are past ancestors, and sets only grow): the wave sits the block out.  Rows that
                    // are not core points still need their neighbour lists, so a wave that holds one never skips.
                    const int c0 = __float_as_int(lds[CL_PU_BASE + (int)(t % 3) * 2]), c1 = __float_as_int(lds[CL_PU_BASE + (int)(t % 3) * 2 + 1]);
                    const int cp = (c0 == -2 || c1 == -2 || (c0 >= 0 && c1 >= 0 && c0 != c1)) ? -2 : (c0 > c1 ? c0 : c1);
                    const bool settled = hh == __builtin_inff() || (mycore && pi == cp);
                    if (cp == -1 || (cp >= 0 && !__builtin_amdgcn_ballot_w64(!settled))) blk_active = false;
                    if (SUB == 1 && A.blk_uni && cp >= 0 && threadIdx.x == 0) CL_ST(A.blk_uni + jb_cur, cp);
                }
            }
            active = blk_active;
        }
        const float *blk = lds + (int)(t % CL_RING) * CL_BLK_FLOATS;
        const float *ubuf = blk + u * UT * G * 256;
        const long col0 = jb_cur * JB + u * UT * 32;
        const int before = cntm;
        int colv = 0;
        // Most finished units hold no candidate pair at all (the lists are 2.4 x the cells that interact, and inside
        // those most 32 x 64 units are empty too): the largest of a lane's values against the lane's "maybe"
        // threshold, one vote, and the per-value compares, counts and column votes are left out.
        bool pairs = prev_active, full = false;
        if (CL_UNIT_VOTE && prev_active) {
            float mx = accp[0][0];
#pragma unroll
            for (int e = 1; e < EL; ++e) mx = fmaxf(mx, accp[e >> 4][e & 15]);
            pairs = __builtin_amdgcn_ballot_w64(mx >= hm) != 0ull;
            if (P1 && pairs) {   // and the other end: every value of every lane definitely inside (a unit in the middle of an instance)
                float mn = accp[0][0];
#pragma unroll
                for (int e = 1; e < EL; ++e) mn = fminf(mn, accp[e >> 4][e & 15]);
                if (__builtin_amdgcn_ballot_w64(mn >= hd) == ~0ull) {
                    cntd += EL;
                    cntm += EL;
                    colv = 32;   // each of the unit's 64 streamed points has all 32 points of this wave inside
                    pairs = false;
                    full = true;
                }
            }
        }
        if (active && pairs) {   // the steady state: one straight-line block, compares between the MFMAs
            init_acc(accn, blk, u);
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                mfma_slot(accn, ubuf, sl);
#pragma unroll
                for (int e = sl * EL / NSL; e < (sl + 1) * EL / NSL; ++e) fast_pair(accp, e, colv);
            }
        } else {
            if (active) {
                init_acc(accn, blk, u);
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) mfma_slot(accn, ubuf, sl);
            }
            if (pairs) {
#pragma unroll
                for (int e = 0; e < EL; ++e) fast_pair(accp, e, colv);
            }
        }
        if (u == 0 && pfetch && w < W && t + 1 < len)   // parents of the next block -> its LDS buffer (read after the next barrier)
        {
            lds[(int)((t + 1) % 3) * 128 + pslot] = __int_as_float(pnext);
            const int cp = common_parent(pnext);
            if (lane == 0) lds[CL_PU_BASE + (int)((t + 1) % 3) * 2 + wave - 2] = __int_as_float(cp);
        }
        if (SYM && (pairs || full) && (col0_prev >> 7) > 2L * rb + 1 && lane < UT * 32)   // (the row block's own two blocks are visited both ways round)
            atomicAdd(reinterpret_cast<int *>(const_cast<float *>(par_prev)) + u_prev * UT * 32 + lane, colv);
        if (prev_active) {
            // Pass 1 notes which streamed blocks held any candidate pair at all for this row block: pass 2 tests the
            // same accumulator values against the same threshold, so it can leave every other block out
            // (cl_relist_kernel).  Several waves may write the same byte; they all write 1.  An undecided pair is
            // a candidate pair, so the common case of a unit without any costs one vote.
            if (P1) {
                if (__builtin_amdgcn_ballot_w64(cntm != before)) {
                    if (lane == 0) A.hit[(long)rb * A.nb128 + (col0_prev >> 7)] = 1;
                    if (__builtin_amdgcn_ballot_w64(cntm - cntd != expdiff)) {
                        slow_unit(accp, par_prev, u_prev, col0_prev);
                        expdiff = cntm - cntd;
                    }
                }
            } else if (__builtin_amdgcn_ballot_w64(cntm != before)) {
                slow_unit(accp, par_prev, u_prev, col0_prev);
            }
        }
        prev_active = active;
        col0_prev = col0;
        par_prev = lds + CL_PAR_BASE + (int)(t % 3) * 128;
        u_prev = u;
    };

    f32x16 accA[UT], accB[UT];
    for (long w = 0; w <= W; w += 2) {
        unit_step(accA, accB, w, std::integral_constant<int, 0>{});
        if (w + 1 <= W) unit_step(accB, accA, w + 1, std::integral_constant<int, (UPS > 1 ? 1 : 0)>{});
    }
    if (MODE == 1 && SUB == 1 && A.rb_uni) {
        // do all rows of this row block name one root now, and is every real row a core point?
        int *scr = reinterpret_cast<int *>(lds);
        __syncthreads();   // (everybody is done with the ring)
        const bool pad = hh == __builtin_inff();
        int pr = -1;
        if (mycore) {
            pr = cl_find(A.parent, pi);
            if (pr != (int)my && h == 0) CL_ST(A.parent + my, pr);   // a non-root's parent, only ever shortened: the rows leave flat,
        }                                                          // so whoever streams them as columns finds one parent
        const unsigned long long cm = __builtin_amdgcn_ballot_w64(mycore);
        const int first = cm ? __builtin_amdgcn_readlane(pr, __builtin_ctzll(cm)) : -1;
        const bool split = __builtin_amdgcn_ballot_w64(mycore && pr != first) != 0;    // core rows under different roots
        const bool stray = __builtin_amdgcn_ballot_w64(!pad && !mycore) != 0;          // a real row that is not a core point
        if (lane == 0) {
            scr[wave] = split ? -2 : first;
            scr[8 + wave] = stray;
        }
        __syncthreads();
        if (threadIdx.x < 3) {   // 0, 1: the row block's two blocks of 128 (core rows only matter); 2: the row block
            const int w0 = threadIdx.x == 1 ? 4 : 0, w1 = threadIdx.x == 0 ? 4 : CL_NW;
            int r = -1, bad = 0;
            for (int w2 = w0; w2 < w1; ++w2) {
                const int v = scr[w2];
                r = (r == -2 || v == -2) ? -2 : (v == -1 ? r : (r == -1 || r == v) ? v : -2);
                bad |= scr[8 + w2];
            }
            if (threadIdx.x < 2) {
                if (r >= 0) CL_ST(A.blk_uni + 2 * rb + (int)threadIdx.x, r);
            } else if (r >= 0 && !bad) {
                CL_ST(A.rb_uni + rb, r);
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            st_exact += __shfl_xor(st_exact, o);
            st_union += __shfl_xor(st_union, o);
            st_border += __shfl_xor(st_border, o);
        }
        if (lane == 0) {
            if (st_exact) atomicAdd(A.stats + 1, (unsigned long long)st_exact);
            if (st_union) atomicAdd(A.stats + 2, (unsigned long long)st_union);
            if (st_border) atomicAdd(A.stats + 3, (unsigned long long)st_border);
        }
    }
    if (SYM) {   // the column counts of the last two blocks
        __syncthreads();
        if (threadIdx.x < JB) {
            const int c1 = colslots[(int)((len - 1) % 3) * 128 + threadIdx.x];
            if (c1) atomicAdd(A.cnt + cb1 + threadIdx.x, c1);
            if (len >= 2) {
                const int c2 = colslots[(int)((len - 2) % 3) * 128 + threadIdx.x];
                if (c2) atomicAdd(A.cnt + cb2 + threadIdx.x, c2);
            }
        }
    }
    if (P1) {
        A.defer_cnt[dq_wave * 64 + lane] = dq_n;
        {
            int qs = dq_n;   // queued tests of this wave -> stats[0]
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) qs += __shfl_xor(qs, o);
            if (lane == 0 && qs) atomicAdd(A.stats + 0, (unsigned long long)qs);
        }
        cntd += __shfl_xor(cntd, 32);   // the two lane halves saw different rows of the same point
        if (h == 0 && cntd) atomicAdd(A.cnt + my, cntd);
    }
}

// ------------------------------------------------------------------ block pruning
// Bounding sphere of every 128-point block (centre = mean, radius = max distance to it, real rows only).
// Embeddings of one instance sit close together and scans are index-coherent, so most block pairs of a
// scene are farther apart than eps and never need their 128 x 128 distance tile.
#define CL_SPH_MARGIN 1e-4f   // fp32 slack of centre / radius / centre-distance arithmetic (vectors are unit length)
// Centre i as one more row of the split-bf16 fragment layout of Xb (cl_normalize_kernel), 32 centres per tile, for
// cl_reach_kernel, and its accumulator start value (R^2 - |c|^2) / 2, R = radius + eps + margin (-inf: empty block
// or padding, cen == nullptr).  Called by a whole workgroup: thread k writes dimension k's two halves.
__device__ __forceinline__ void cl_pack_centre(float *__restrict__ Cb, float *__restrict__ nhc, int G, int i, const float *cen,
                                               float rad, float eps, int t, int nthreads) {
    const int Dp = 8 * G;
    for (int k = t; k < Dp; k += nthreads) {
        const float v = cen ? cen[k] : 0.f;
        const int ks = k >> 4, hb = (k >> 3) & 1, e = k & 7;
        unsigned short *fh = reinterpret_cast<unsigned short *>(Cb + ((((long)i >> 5) * G + 2 * ks) * 64 + hb * 32 + (i & 31)) * 4);
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        fh[e] = __builtin_bit_cast(unsigned short, hi);
        fh[512 + e] = __builtin_bit_cast(unsigned short, lo);   // the lo fragment follows the hi fragment
    }
    if (t == 0) {
        float start = -__builtin_inff();
        if (cen) {
            float sq = 0.f;
            for (int k = 0; k < Dp; ++k) sq = fmaf(cen[k], cen[k], sq);
            const float R = rad + eps + CL_SPH_MARGIN;
            start = 0.5f * (R * R - sq);
        }
        nhc[i] = start;
    }
}

// The same sphere as cl_sphere_kernel's, from the block's normalised rows while the staged cl_normalize_kernel still
// has them in LDS (row r at tile + r * S, the real rows are the first nreal): the separate kernel read the 43 MB of
// Xp again (32 us at 166k x 64).  Called by the whole workgroup of 128 threads.
__device__ void cl_block_sphere_from_tile(const float *tile, int S, int D, int G, int nreal, int b, const ClSphereOut &sp) {
    __shared__ float cen[CL_STAGE_DMAX];
    __shared__ float red[2];
    const int t = threadIdx.x, Dp = 8 * G, lane = t & 63, wave = t >> 6;
    if (sp.Cb && b == sp.nb - 1)
        for (int i2 = sp.nb; i2 < sp.nbpad; ++i2) cl_pack_centre(sp.Cb, sp.nhc, G, i2, nullptr, 0.f, sp.eps, t, 128);
    if (nreal == 0) {
        if (t == 0) sp.radius[b] = -1.f;
        for (int k = t; k < Dp; k += 128) sp.centre[(long)b * Dp + k] = 0.f;
        if (sp.Cb) cl_pack_centre(sp.Cb, sp.nhc, G, b, nullptr, 0.f, sp.eps, t, 128);
        return;
    }
    if (t < Dp) {   // (Dp <= 128: a thread per column, conflict-free across the columns)
        float sum = 0.f;
        if (t < D)
            for (int r = 0; r < nreal; ++r) sum += tile[r * S + t];
        cen[t] = sum / (float)nreal;
    }
    __syncthreads();
    float d2 = 0.f;
    if (t < nreal)
        for (int k = 0; k < D; ++k) {
            const float d = tile[t * S + k] - cen[k];
            d2 = fmaf(d, d, d2);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o));
    if (lane == 0) red[wave] = d2;
    __syncthreads();
    const float rad = sqrtf(fmaxf(red[0], red[1])) + CL_SPH_MARGIN;
    if (t == 0) sp.radius[b] = rad;
    for (int k = t; k < Dp; k += 128) sp.centre[(long)b * Dp + k] = cen[k];
    if (sp.Cb) cl_pack_centre(sp.Cb, sp.nhc, G, b, cen, rad, sp.eps, t, 128);
}

// (Cb != nullptr: the block's centre also goes into the reach stage's packed set, and the last block pads the set to
// whole tiles of 32 -- a kernel of its own took 17 us for this)
__global__ __launch_bounds__(128) void cl_sphere_kernel(const float *__restrict__ Xp, const float *__restrict__ nh, int G,
                                                        float *__restrict__ centre, float *__restrict__ radius,
                                                        float *__restrict__ Cb, float *__restrict__ nhc, int nb, int nbpad, float eps) {
    extern __shared__ float sph[];   // [8G] centre, [2][8G] the two waves' column sums, [4] scalars
    float *cen = sph, *part = sph + 8 * G, *sc = sph + 3 * 8 * G;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, Dp = 8 * G;
    const long i = (long)b * 128 + t;
    const bool real = nh[i] > -__builtin_inff();
    const float *px = Xp + ((i >> 5) * G * 64 + (i & 31)) * 4;
    // column sums: a shuffle reduction per column inside each wave (no barrier per column), the two waves' sums through LDS
    const unsigned long long rm = __builtin_amdgcn_ballot_w64(real);
    if (lane == 0) sc[wave] = (float)__builtin_popcountll(rm);
    for (int k = 0; k < Dp; ++k) {
        float v = real ? px[(k >> 3) * 256 + ((k >> 2) & 1) * 128 + (k & 3)] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) part[wave * Dp + k] = v;
    }
    __syncthreads();
    const float cntf = sc[0] + sc[1];
    if (Cb && b == nb - 1)
        for (int i2 = nb; i2 < nbpad; ++i2) cl_pack_centre(Cb, nhc, G, i2, nullptr, 0.f, eps, t, 128);
    if (cntf == 0.f) {
        if (t == 0) radius[b] = -1.f;
        for (int k = t; k < Dp; k += 128) centre[(long)b * Dp + k] = 0.f;
        if (Cb) cl_pack_centre(Cb, nhc, G, b, nullptr, 0.f, eps, t, 128);
        return;
    }
    for (int k = t; k < Dp; k += 128) cen[k] = (part[k] + part[Dp + k]) / cntf;
    __syncthreads();
    float d2 = 0.f;
    if (real)
        for (int k = 0; k < Dp; ++k) {
            const float d = px[(k >> 3) * 256 + ((k >> 2) & 1) * 128 + (k & 3)] - cen[k];
            d2 = fmaf(d, d, d2);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o));
    if (lane == 0) sc[2 + wave] = d2;
    __syncthreads();
    const float rad = sqrtf(fmaxf(sc[2], sc[3])) + CL_SPH_MARGIN;
    if (t == 0) radius[b] = rad;
    for (int k = t; k < Dp; k += 128) centre[(long)b * Dp + k] = cen[k];   // (up to 256 dimensions, 128 threads)
    if (Cb) cl_pack_centre(Cb, nhc, G, b, cen, rad, eps, t, 128);
}

// grid = row groups of 256 points.  list[ib][0..cnt) = column blocks jb (ascending) that can reach either of the
// group's two 128-point blocks: |c - c_jb| <= r + r_jb + eps.
__global__ __launch_bounds__(256) void cl_active_kernel(const float *__restrict__ centre, const float *__restrict__ radius,
                                                        int nb, int Dp, float eps, int *__restrict__ act_cnt,
                                                        int *__restrict__ act_list) {
    extern __shared__ float ci[];   // [2][Dp] the centres of the row group's two 128-point blocks, then [256] ints scan scratch
    int *scan = reinterpret_cast<int *>(ci + 2 * Dp);
    __shared__ int base;
    const int ib = blockIdx.x, t = threadIdx.x;   // ib: a workgroup's 256 rows = 128-point blocks 2 ib and 2 ib + 1
    for (int k = t; k < 2 * Dp; k += 256) ci[k] = centre[(long)2 * ib * Dp + k];
    if (t == 0) base = 0;
    __syncthreads();
    const float ri0 = radius[2 * ib], ri1 = radius[2 * ib + 1];
    for (int j0 = 0; j0 < nb; j0 += 256) {
        const int jb = j0 + t;
        int act = 0;
        if (jb < nb) {
            const float rj = radius[jb];
            if (rj >= 0.f) {
                float d20 = 0.f, d21 = 0.f;
                const float *cj = centre + (long)jb * Dp;
                for (int k = 0; k < Dp; ++k) {
                    const float c = cj[k], d0 = ci[k] - c, d1 = ci[Dp + k] - c;
                    d20 = fmaf(d0, d0, d20);
                    d21 = fmaf(d1, d1, d21);
                }
                const float reach0 = ri0 + rj + eps + CL_SPH_MARGIN, reach1 = ri1 + rj + eps + CL_SPH_MARGIN;
                act = (ri0 >= 0.f && d20 <= reach0 * reach0) || (ri1 >= 0.f && d21 <= reach1 * reach1);
            }
        }
        scan[t] = act;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {   // inclusive scan
            const int v = t >= o ? scan[t - o] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        if (act) act_list[(long)ib * nb + base + scan[t] - 1] = jb;
        __syncthreads();
        if (t == 255) base += scan[255];
        __syncthreads();
    }
    if (t == 0) act_cnt[ib] = base;
}

// The bounding spheres alone prune poorly once a block holds an outlier: its sphere spans everything.  A pair
// (i in A, j in B) within eps needs i within eps of B's sphere AND j within eps of A's sphere, so per pair of blocks
// cl_reach_kernel notes whether ANY point of A reaches B's sphere; cl_relist_kernel keeps a listed block only if
// the note holds both ways round.  An outlier block against a tight instance block fails the second half: none
// of the outliers is near the instance.
#define CL_REACH_SLACK 4e-5f   // on x.c - |c|^2/2: three-term bf16 split (<= 1.5e-5 for unit-length x, |c| <= 1) + fp32 accumulation + roundings of the terms
// One workgroup per block A (wave = 32 of its points, operands in registers as in the pair kernel), every tile of 32
// centres in turn: D = C . X^T on the matrix pipe (hi.hi + hi.lo + lo.hi), accumulators started at (R^2 - |c|^2) / 2, so
//   |x - c|^2 <= R^2   <=>   acc >= |x|^2 / 2;
// a ballot per accumulator register tells whether any of the wave's points reaches that centre's sphere.  Dense
// (166k points x 1,301 centres x 64 = 28 GFLOP on bf16 MFMAs) it is cheaper than the listed pairs were on the vector ALU.
template <int G, int NS>   // NS: sets of 32 points per wave (2 where the registers allow: a tile's centre fragments are then
                           // fetched once for 64 points -- the kernel is bound by that traffic through the L1, not by the matrix pipe)
__global__ __launch_bounds__(256 / NS) void cl_reach_kernel(const float *__restrict__ Xb, const float *__restrict__ nh,
                                                            const float *__restrict__ Cb, const float *__restrict__ nhc, int nb,
                                                            int ntile, unsigned char *__restrict__ reach) {
    constexpr int KS = G / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const long A = blockIdx.x;
    cl_bf16x8 ah[NS][KS], al[NS][KS];
    // D = X . C^T: a lane holds one centre's column, its 16 registers 16 of the set's points (register r of lane half
    // h: point 8 (r / 4) + 4 h + r % 4).  The accumulators start at (R^2 - |c|^2) / 2 - (|x|^2 / 2 - slack), so a
    // point reaches the centre's sphere where its register ends >= 0, and "any of the lane's points" is a maximum
    // inside the lane: no ballot per register and no mask arithmetic on the scalar unit (the earlier form, centres as
    // rows and a ballot per register, spent ~100 scalar instructions per tile on the 32 votes).
    float mh[NS][16];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const long i0 = (A * 4 + wave * NS + s) * 32;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ah[s][ks] = *reinterpret_cast<const cl_bf16x8 *>(Xb + ((i0 >> 5) * G + 2 * ks) * 256 + lane * 4);
            al[s][ks] = *reinterpret_cast<const cl_bf16x8 *>(Xb + ((i0 >> 5) * G + 2 * ks + 1) * 256 + lane * 4);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mh[s][r] = nh[i0 + 8 * (r >> 2) + 4 * h + (r & 3)] + CL_REACH_SLACK;   // (-inf on pad rows: reach nothing)
    }
    // (fetching a tile's centre fragments one tile ahead was tried: 139 -> 183 us, the other waves of the SIMD hide the
    // loads better than the extra registers allow)
    for (int jt = blockIdx.y; jt < ntile; jt += gridDim.y) {
        const float start = nhc[jt * 32 + (lane & 31)];
        f32x16 acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = start + mh[s][r];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const cl_bf16x8 bh = *reinterpret_cast<const cl_bf16x8 *>(Cb + ((long)jt * G + 2 * ks) * 256 + lane * 4);
            const cl_bf16x8 bl = *reinterpret_cast<const cl_bf16x8 *>(Cb + ((long)jt * G + 2 * ks + 1) * 256 + lane * 4);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                acc[s] = MFMAB(ah[s][ks], bh, acc[s]);
                acc[s] = MFMAB(al[s][ks], bh, acc[s]);
                acc[s] = MFMAB(ah[s][ks], bl, acc[s]);
            }
        }
        float best = -__builtin_inff();
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 16; r += 2) best = fmaxf(best, fmaxf(acc[s][r], acc[s][r + 1]));
        const int col = jt * 32 + (lane & 31);
        if (best >= 0.f && col < nb) reach[A * nb + col] = 1;   // both lane halves and the block's waves may write the same 1
    }
}

// Between the passes: a row block's list keeps the blocks in which pass 1 saw a candidate pair and that hold a
// core point -- the only ones in which pass 2 can find an edge.  In place: an entry moves to a position at or
// before its own, and a chunk of 256 entries is read (barriers of the scan) before any of it is written.
// With `reach` (before pass 1) the rule is the point-to-sphere test of cl_reach_kernel instead, both ways round.
__global__ __launch_bounds__(256) void cl_relist_kernel(const unsigned char *__restrict__ hit, const int *__restrict__ blk_has_core,
                                                        const unsigned char *__restrict__ reach, int nb, int sym,
                                                        int *__restrict__ act_cnt, int *__restrict__ act_list) {
    __shared__ int scan[256];
    __shared__ int base;
    const int ib = blockIdx.x, t = threadIdx.x;
    // sym (pass 1 visits every block pair once, MODE 2 of the pair kernel): before pass 1 the list keeps only blocks
    // at or after the row block's own; after it pass 1's list is that half, so every block is a candidate again, and
    // for the blocks before its own the row block finds its hits filed under the OTHER side (the workgroup that held
    // block jb as rows against this row block's two blocks as columns: a superset at that side's 256-row granularity).
    // With the reach notes every block is a candidate as well (a note both ways round implies that the spheres reach, so
    // the sphere lists of cl_active_kernel are only built when the reach stage is switched off).
    const bool all = sym || reach;
    const int total = all ? nb : act_cnt[ib];
    int *list = act_list + (long)ib * nb;
    if (t == 0) base = 0;
    __syncthreads();
    for (int e0 = (reach && sym) ? (2 * ib) & ~255 : 0; e0 < total; e0 += 256) {   // (reach && sym: nothing before the row block's own blocks)
        const int e = e0 + t;
        int jb = 0, keep = 0;
        if (e < total) {
            jb = all ? e : list[e];
            if (reach) {
                const long a0 = 2L * ib, a1 = a0 + 1;
                keep = (reach[a0 * nb + jb] && reach[(long)jb * nb + a0]) || (reach[a1 * nb + jb] && reach[(long)jb * nb + a1]);
                if (sym && jb < 2 * ib) keep = 0;
            } else {
                const bool h = (!sym || jb >= 2 * ib) ? hit[(long)ib * nb + jb] != 0
                                                      : (hit[(long)(jb >> 1) * nb + 2 * ib] | hit[(long)(jb >> 1) * nb + 2 * ib + 1]) != 0;
                keep = h && blk_has_core[jb] != 0;
            }
        }
        scan[t] = keep;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {   // inclusive scan
            const int v = t >= o ? scan[t - o] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        if (keep) list[base + scan[t] - 1] = jb;
        __syncthreads();
        if (t == 255) base += scan[255];
        __syncthreads();
    }
    if (t == 0) act_cnt[ib] = base;
}

// The work list of a pair pass.  Lists differ wildly in length (an outlier block meets hundreds of blocks, an instance
// block a few dozen), so a workgroup per row block -- or per fixed fraction of one -- leaves the chip waiting for the
// longest.  Lists are cut into chunks of CH entries, CH chosen so that the chunks fill at most `kslots` + one per row
// block of the grid the host launches; a workgroup takes one chunk.
__global__ __launch_bounds__(1024) void cl_plan_kernel(const int *__restrict__ act_cnt, int nrowblk, int kslots,
                                                       int *__restrict__ work_rb, int *__restrict__ work_e0,
                                                       int *__restrict__ work_meta) {
    __shared__ long long tsum[16];
    __shared__ int wsum[16];
    __shared__ int base, ch_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    long long tot = 0;
    for (int i = t; i < nrowblk; i += 1024) tot += act_cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) tsum[wave] = tot;
    __syncthreads();
    if (t == 0) {
        long long all = 0;
        for (int w = 0; w < 16; ++w) all += tsum[w];
        long long ch = (all + kslots - 1) / kslots;
        ch = ch < CL_CHUNK_MIN ? CL_CHUNK_MIN : (ch > CL_SLIST_MAX ? CL_SLIST_MAX : ch);
        ch_s = (int)ch;
        base = 0;
    }
    __syncthreads();
    const int CH = ch_s;
    for (int s0 = 0; s0 < nrowblk; s0 += 1024) {
        const int i = s0 + t;
        const int n = i < nrowblk ? (act_cnt[i] + CH - 1) / CH : 0;
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = base + incl - n;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        for (int c = 0; c < n; ++c) {
            work_rb[off + c] = i;
            work_e0[off + c] = c * CH;
        }
        __syncthreads();
        if (t == 0) {
            int all = 0;
            for (int w = 0; w < 16; ++w) all += wsum[w];
            base += all;
        }
        __syncthreads();
    }
    if (t == 0) {
        work_meta[0] = base;
        work_meta[1] = CH;
    }
}

#define CL_DEFER_CAP 16   // queued undecided pairs per lane (pass 1); beyond that a lane decides on the spot
#define CL_RESOLVE_SG 4   // queue slots one thread of cl_resolve_kernel works through
// ------------------------------------------------------------------ deferred exact tests of pass 1
// One thread per queue slot ([wave][slot][lane]).  The pair passes run their 8 waves in lockstep (a
// barrier per streamed block), so a wave that stops for a float64 test -- two dependent memory round
// trips -- stalls seven others; 94 % of the blocks had such a wave.  Queued, the tests run here, all
// in flight at once.
__global__ __launch_bounds__(256) void cl_resolve_kernel(const float *__restrict__ Xp, int G, double eps2,
                                                         const int *__restrict__ q, const int *__restrict__ qcnt, int cap,
                                                         long nwaves, const int *__restrict__ work_rb,
                                                         const int *__restrict__ work_meta, int *__restrict__ cnt) {
    // thread = (wave, slot group, lane): CL_RESOLVE_SG slots each, so that a quarter of the workgroups is started
    // than with a thread per slot (most queues are empty, and at 152,000 workgroups the launch rate was the cost)
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // ((wave * groups) + group) * 64 + lane
    const int lane = (int)(e & 63);
    const long ws = e >> 6;
    const int groups = (cap + CL_RESOLVE_SG - 1) / CL_RESOLVE_SG;
    const long wave = ws / groups;
    const int s0 = (int)(ws % groups) * CL_RESOLVE_SG;
    if (wave >= nwaves || wave / CL_NW >= work_meta[0]) return;
    const int n = qcnt[wave * 64 + lane];
    if (s0 >= n) return;
    const long my = ((long)work_rb[wave / CL_NW] * CL_NW + wave % CL_NW) * 32 + (lane & 31);   // wave / CL_NW: the pair pass's chunk
    int add = 0;   // (the tests are counted in stats[0] by the pair pass: one atomic per wave there)
    for (int s = s0; s < n && s < s0 + CL_RESOLVE_SG; ++s) {
        const int qv = q[(wave * cap + s) * 64 + lane];   // sign bit: the pair counts for the other point too (MODE 2 of the pair pass)
        const long other = qv & 0x7fffffff;
        const bool in = cl_exact_within(Xp, G, my, other, eps2);
        add += in;
        if (in && qv < 0) atomicAdd(cnt + other, 1);
    }
    if (add) atomicAdd(cnt + my, add);
}

// ------------------------------------------------------------------ glue kernels
// after pass 1: core flags, column norms with non-core masked out, union-find initial state
__global__ __launch_bounds__(256) void cl_prepare_link_kernel(const int *__restrict__ cnt, const float *__restrict__ nrm, long N,
                                                              long Npad, int min_samples, unsigned char *__restrict__ core,
                                                              float *__restrict__ nrm_core, int *__restrict__ parent,
                                                              int *__restrict__ nbr_cnt, int *__restrict__ blk_has_core) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    const bool k = i < N && cnt[i] >= min_samples;
    core[i] = k;
    nrm_core[i] = k ? nrm[i] : -__builtin_inff();
    parent[i] = k ? (int)i : -1;   // -1: not a core point (never a member of a set; the link pass reads it as "no parent")
    nbr_cnt[i] = 0;
    if (k) blk_has_core[i >> 7] = 1;   // benign race: everybody writes 1
}

// every core point names its root (before the sets are numbered by the caller's indices)
__global__ __launch_bounds__(256) void cl_flatten_kernel(int *__restrict__ parent, const unsigned char *__restrict__ core, long N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N || !core[i]) return;
    const int r = cl_find(parent, (int)i);
    if (r != (int)i) CL_ST(parent + i, r);   // a non-root's parent: only ever shortened
}

// roots -> consecutive cluster numbers 1.. in ascending root order.  Two launches over chunks of 8,192 indices:
// cl_rank_count_kernel counts each chunk's roots, cl_rank_assign_kernel sums the counts of the chunks before its own
// (a few dozen values) and numbers its roots; a single workgroup walking all N took 97 us at 166,499 points.
// With a working order (owner != null) the scan runs over the CALLER's indices: owner[o] = root of the set whose
// lowest caller index among core points is o (-1 elsewhere), so sets are numbered as if the rows had never moved.
// Without one, parent[] has been flattened (cl_flatten_kernel) and a root is a core point that names itself.
#define CL_RANK_PT 8   // consecutive indices per thread, 1,024 threads per chunk
__device__ __forceinline__ int cl_rank_slot(const int *parent, const unsigned char *core, const int *owner, long N, long i) {
    if (i >= N) return -1;
    if (owner) return owner[i];
    return core[i] && parent[i] == (int)i ? (int)i : -1;
}
__global__ __launch_bounds__(1024) void cl_rank_count_kernel(const int *__restrict__ parent, const unsigned char *__restrict__ core,
                                                             const int *__restrict__ owner, long N, int *__restrict__ chunk_cnt) {
    __shared__ int wsum[16];
    const long i0 = ((long)blockIdx.x * 1024 + threadIdx.x) * CL_RANK_PT;
    int n = 0;
#pragma unroll
    for (int j = 0; j < CL_RANK_PT; ++j) n += cl_rank_slot(parent, core, owner, N, i0 + j) >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        chunk_cnt[blockIdx.x] = tot;
    }
}
__global__ __launch_bounds__(1024) void cl_rank_assign_kernel(const int *__restrict__ parent, const unsigned char *__restrict__ core,
                                                              const int *__restrict__ owner, long N,
                                                              const int *__restrict__ chunk_cnt, int *__restrict__ rank,
                                                              int *__restrict__ num_clusters) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // roots in the chunks before this one (every workgroup sums them itself: at most N / 8192 values)
        int b = 0;
        for (int c = threadIdx.x; c < (int)blockIdx.x; c += 1024) b += chunk_cnt[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o);
        if (lane == 0) wsum[wave] = b;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            base_s = tot;
        }
        __syncthreads();
    }
    const long i0 = ((long)blockIdx.x * 1024 + threadIdx.x) * CL_RANK_PT;
    int slot[CL_RANK_PT], mine = 0;
#pragma unroll
    for (int j = 0; j < CL_RANK_PT; ++j) {
        slot[j] = cl_rank_slot(parent, core, owner, N, i0 + j);
        mine += slot[j] >= 0;
    }
    int incl = mine;                             // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    __syncthreads();                             // (wsum is reused)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = base_s + incl - mine;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int j = 0; j < CL_RANK_PT; ++j)
        if (slot[j] >= 0) rank[slot[j]] = ++off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023 && num_clusters) *num_clusters = off;   // the last thread of the last chunk ends on the total
}

__global__ __launch_bounds__(256) void cl_label_kernel(const int *__restrict__ parent, const unsigned char *__restrict__ core,
                                                       const int *__restrict__ rank, const int *__restrict__ nbr_cnt,
                                                       const int *__restrict__ nbr, int cap, long N, int32_t *__restrict__ labels,
                                                       const int *__restrict__ perm) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const long o = perm ? (long)perm[i] : i;   // the caller's index of working row i
    if (core[i]) {
        labels[o] = rank[parent[i]];
        return;
    }
    int n = nbr_cnt[i];
    n = n < cap ? n : cap;
    int best = 0x7fffffff;   // the lowest-numbered cluster among the core neighbours
    for (int q = 0; q < n; ++q) {
        const int r = rank[parent[nbr[i * cap + q]]];
        best = r < best ? r : best;
    }
    labels[o] = n > 0 ? best : 0;
}

// working order only: the lowest caller index among the core points of every set, then the set's root filed under it
// (it flattens on its way: every core point finds its root and names it -- parents only ever move towards the root,
// so a reader that meets a half-flattened path still ends at the root)
__global__ __launch_bounds__(256) void cl_minorig_kernel(int *__restrict__ parent, const unsigned char *__restrict__ core,
                                                         const int *__restrict__ perm, long N, int *__restrict__ minorig) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < N && core[i];
    const int r = in ? cl_find(parent, (int)i) : -1;
    if (in && r != (int)i) CL_ST(parent + i, r);
    int v = in ? perm[i] : 0x7fffffff;
    // neighbours in the working order mostly share a root: the wave's lanes that name the first lane's root send one
    // atomic between them, the others their own
    const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
    if (!m) return;
    const int r0 = __builtin_amdgcn_readlane(r, __builtin_ctzll(m));
    const bool same = in && r == r0;
    int w = same ? v : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int u = __shfl_xor(w, o);
        w = u < w ? u : w;
    }
    if (same ? (threadIdx.x & 63) == __builtin_ctzll(m) : in) atomicMin(minorig + r, same ? w : v);
}
__global__ __launch_bounds__(256) void cl_owner_kernel(const int *__restrict__ parent, const unsigned char *__restrict__ core, long N,
                                                       const int *__restrict__ minorig, int *__restrict__ owner) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < N && core[i] && parent[i] == (int)i) owner[minorig[i]] = (int)i;
}
__global__ __launch_bounds__(256) void cl_unpermute_kernel(const int *__restrict__ src, const int *__restrict__ perm,
                                                           const unsigned long long *__restrict__ sorted_flag, long N,
                                                           int32_t *__restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < N) dst[*sorted_flag ? (long)perm[i] : i] = src[i];
}

// ------------------------------------------------------------------ host side
struct ClWs {
    float *Xp, *Xb, *nrm, *nrm_core;
    int *cnt, *parent, *nbr_cnt, *rank, *blk_has_core, *nbr;
    unsigned char *core;
    unsigned long long *stats;
    float *centre, *radius;
    float *Cb, *nhc;   // centres in the fragment layout of Xb, and their accumulator start values (cl_pack_centre, from cl_sphere_kernel)
    int *act_cnt, *act_list;
    unsigned char *hit, *reach;
    unsigned *key_a, *key_b;   // working order: code of every row, double-buffered for the sort
    int *perm, *val_b;         // working row -> caller's index (the sort's values)
    int *minorig, *owner, *chunk_cnt;
    int *blk_uni, *rb_uni;
    float *psum;               // CL_SUM_COPIES x (32 projection sums, 32 near-mean counts)
    char *zero_begin, *ones_begin;   // two stretches cleared by one memset each at the start of a call (cl_carve)
    size_t zero_bytes, ones_bytes;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int *defer_q, *defer_cnt;
    long defer_waves;
    int *work_rb, *work_e0, *work_meta;
    int kslots, kbound;   // chunks the plan may fill beyond one per row block; the pair passes' grid
};


// Grid of the pair passes: `kslots` chunks the plan aims at (enough to fill the chip a dozen times over, and enough
// that a chunk of the longest possible list fits the LDS slice) plus one partial chunk per row block.
// pass 2: workgroups per row block -- enough to fill the chip several times, and a slice of the longest list fits in LDS
static int cl_nsplit(long Npad) {
    const long nrowblk = Npad / CL_ROWBLK, nb128 = Npad / 128;
    // (the first workgroups of a row block do the merging, the later ones mostly strike what is merged already:
    // eight or so per row block, 0.64 -> 0.55 ms at 166,499 points against four)
    long ns = (4096 + nrowblk - 1) / nrowblk;
    ns = ns > 16 ? 16 : ns;
    ns = ns < 1 ? 1 : (ns > nb128 ? nb128 : ns);
    if (ns * CL_SLIST_MAX < nb128) ns = (nb128 + CL_SLIST_MAX - 1) / CL_SLIST_MAX;
    return (int)ns;
}

static void cl_chunks(long Npad, int &kslots, int &kbound) {
    const long nrowblk = Npad / CL_ROWBLK, nb128 = Npad / 128;
    long k = 8 * nrowblk;
    k = k < 256 ? 256 : (k > 4096 ? 4096 : k);
    const long kmin = nrowblk * ((nb128 + CL_SLIST_MAX - 1) / CL_SLIST_MAX);
    if (k < kmin) k = kmin;
    kslots = (int)k;
    kbound = (int)(k + nrowblk);
}

static size_t cl_carve(ClWs &w, char *base, long Npad, int G, long N, int cap) {
    size_t o = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + o : nullptr;
        o += (bytes + 255) / 256 * 256;
        return p;
    };
    w.Xp = (float *)take(sizeof(float) * (size_t)Npad * G * 8);
    w.Xb = (float *)take(sizeof(float) * (size_t)Npad * G * 8);   // hi + lo bf16 per value: the same size
    w.nrm = (float *)take(sizeof(float) * (Npad + 64));        // + 64: the 64-dword block prefetch may overrun
    w.nrm_core = (float *)take(sizeof(float) * (Npad + 64));
    w.parent = (int *)take(sizeof(int) * (Npad + 64));
    w.nbr_cnt = (int *)take(sizeof(int) * Npad);
    w.rank = (int *)take(sizeof(int) * Npad);
    w.core = (unsigned char *)take((size_t)Npad);
    const size_t nb = (size_t)Npad / 128;
    cl_chunks(Npad, w.kslots, w.kbound);
    w.defer_waves = (long)w.kbound * CL_NW;
    // everything that must start a call at zero sits in one stretch, everything that starts at -1 in another: a
    // memset each at the start of the call instead of a dozen between the kernels (each one a launch on the stream)
    const size_t z0 = o;
    w.stats = (unsigned long long *)take(sizeof(unsigned long long) * 8);
    w.psum = (float *)take(sizeof(float) * 64 * CL_SUM_COPIES);
    w.cnt = (int *)take(sizeof(int) * Npad);
    w.blk_has_core = (int *)take(sizeof(int) * (Npad / 128 + 1));
    w.hit = (unsigned char *)take((nb / 2 + 1) * nb);
    w.reach = (unsigned char *)take(nb * nb);   // [block A][block B]: some point of A reaches B's sphere
    w.defer_cnt = (int *)take(sizeof(int) * (size_t)w.defer_waves * 64);   // (workgroups without work write nothing)
    w.zero_begin = base ? base + z0 : nullptr;
    w.zero_bytes = o - z0;
    const size_t f0 = o;
    w.blk_uni = (int *)take(sizeof(int) * (Npad / 128 + 2));
    w.rb_uni = (int *)take(sizeof(int) * (Npad / 256 + 2));
    w.owner = (int *)take(sizeof(int) * Npad);
    w.ones_begin = base ? base + f0 : nullptr;
    w.ones_bytes = o - f0;
    w.centre = (float *)take(sizeof(float) * nb * G * 8);
    w.radius = (float *)take(sizeof(float) * nb);
    w.Cb = (float *)take(sizeof(float) * ((nb + 31) / 32 * 32) * G * 8);
    w.nhc = (float *)take(sizeof(float) * ((nb + 31) / 32 * 32 + 64));
    w.act_cnt = (int *)take(sizeof(int) * nb);
    w.act_list = (int *)take(sizeof(int) * (nb / 2 + 1) * nb);   // one row per workgroup of 256 points
    w.nbr = (int *)take(sizeof(int) * (size_t)(N > 0 ? N : 1) * (cap > 0 ? cap : 1));
    w.key_a = (unsigned *)take(sizeof(int) * Npad);
    w.key_b = (unsigned *)take(sizeof(int) * Npad);
    w.perm = (int *)take(sizeof(int) * Npad);
    w.val_b = (int *)take(sizeof(int) * Npad);
    w.minorig = (int *)take(sizeof(int) * Npad);
    w.chunk_cnt = (int *)take(sizeof(int) * (Npad / (1024 * CL_RANK_PT) + 2));
    w.sort_tmp_bytes = (size_t)Npad * 8 + (4u << 20);   // the double-buffered radix sort needs its digit counts only; checked at the call
    w.sort_tmp = take(w.sort_tmp_bytes);
    w.work_rb = (int *)take(sizeof(int) * w.kbound);
    w.work_e0 = (int *)take(sizeof(int) * w.kbound);
    w.work_meta = (int *)take(sizeof(int) * 4);
    w.defer_q = (int *)take(sizeof(int) * (size_t)w.defer_waves * CL_DEFER_CAP * 64);
    return o;
}

static int cl_geometry(int D, int &G, int &CT) {
    if (D <= 64) { G = 8; CT = 4; }
    else if (D <= 128) { G = 16; CT = 2; }
    else { G = 32; CT = 1; }
    return 0;
}

static int cl_check(const char *fn, long N, int D, double eps, int min_samples) {
    P2N_REQUIRE(N >= 0 && N <= 0x7fffff00L, "%s: N=%ld out of range", fn, N);
    P2N_REQUIRE(D >= 1 && D <= 256, "%s: D=%d not in 1..256", fn, D);
    P2N_REQUIRE(eps > 0.0 && eps < 1e3, "%s: eps=%g must be positive", fn, eps);
    P2N_REQUIRE(min_samples >= 1 && min_samples <= 65536, "%s: min_samples=%d not in 1..65536", fn, min_samples);
    return P2N_OK;
}

extern "C" size_t p2n_cluster_workspace_bytes(long N, int D, int min_samples) {
    if (N < 0 || D < 1 || D > 256 || min_samples < 1) return 0;
    int G, CT;
    cl_geometry(D, G, CT);
    ClWs w;
    return cl_carve(w, nullptr, (N + CL_NPAD_UNIT - 1) / CL_NPAD_UNIT * CL_NPAD_UNIT, G, N, min_samples - 1);
}

static size_t cl_stage_lds_bytes(int D) { return sizeof(float) * CL_STAGE_ROWS * (size_t)(D | 1); }

// rows staged through LDS up to CL_STAGE_DMAX dimensions, read in place above that
static int cl_launch_normalize(const float *emb, long N, long Npad, int D, int G, float *out_rm, float *Xp, float *nrm, float *Xb,
                               const int *perm, const ClSphereOut &sp, hipStream_t stream) {
    if (D <= CL_STAGE_DMAX) {
        const size_t ldsb = cl_stage_lds_bytes(D);
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cl_normalize_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        hipLaunchKernelGGL(cl_normalize_kernel<true>, dim3((unsigned)((Npad + CL_STAGE_ROWS - 1) / CL_STAGE_ROWS)), dim3(CL_STAGE_ROWS), ldsb,
                           stream, emb, N, Npad, D, G, out_rm, Xp, nrm, Xb, perm, sp);
    } else {
        hipLaunchKernelGGL(cl_normalize_kernel<false>, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, stream, emb, N, Npad, D, G,
                           out_rm, Xp, nrm, Xb, perm, ClSphereOut{});
    }
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_normalize_rows(const float *emb, long N, int D, float *out, void *stream_) {
    P2N_REQUIRE(N >= 0 && D >= 1 && D <= 256, "p2n_normalize_rows: N=%ld D=%d (D in 1..256)", N, D);
    if (N == 0) return P2N_OK;
    P2N_REQUIRE(emb && out, "p2n_normalize_rows: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    int rc = cl_launch_normalize(emb, N, N, D, (D + 7) / 8, out, nullptr, nullptr, nullptr, nullptr, ClSphereOut{}, stream);
    if (rc) return rc;
    return P2N_OK;
}

// rows are worked on in code order unless P2N_CL_SORT=0 (read per call: tests cover both orders); tiny inputs are one block anyway
static bool p2n_cluster_sorts(long N) {
    const char *e = getenv("P2N_CL_SORT");
    return N > 256 && !(e && atoi(e) == 0);
}

static bool p2n_cluster_symmetric() {
    const char *e = getenv("P2N_CL_SYM");
    return !(e && atoi(e) == 0);
}

static bool p2n_cluster_link_notes() {
    const char *e = getenv("P2N_CL_NOTES");
    return !(e && atoi(e) == 0);
}

static bool p2n_cluster_refines() {
    const char *e = getenv("P2N_CL_REACH");
    return !(e && atoi(e) == 0);
}

template <int G, int CT>
// kbound: pass 1 the bound on the work list's chunks, pass 2 the number of row blocks
static int cl_launch_pairs(int mode, const ClArgs &a, int kbound, hipStream_t stream) {
    const size_t ldsb = (CL_LDS_FLOATS + CL_SLIST_MAX) * sizeof(float);   // 107 KiB: above the default dynamic-LDS limit
    P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cl_pair_kernel<G, CT, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cl_pair_kernel<G, CT, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cl_pair_kernel<G, CT, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    if (mode == 0)
        hipLaunchKernelGGL((cl_pair_kernel<G, CT, 0>), dim3((unsigned)kbound), dim3(CL_NW * 64), ldsb, stream, a);
    else if (mode == 2)
        hipLaunchKernelGGL((cl_pair_kernel<G, CT, 2>), dim3((unsigned)kbound), dim3(CL_NW * 64), ldsb, stream, a);
    else
        hipLaunchKernelGGL((cl_pair_kernel<G, CT, 1>), dim3((unsigned)kbound, a.nsplit), dim3(CL_NW * 64), ldsb, stream, a);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_cluster_embeddings(const float *emb, long N, int D, double eps, int min_samples, int32_t *labels,
                                      int32_t *num_clusters, void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = cl_check("p2n_cluster_embeddings", N, D, eps, min_samples);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) {
        if (num_clusters) P2N_CHECK_HIP(hipMemsetAsync(num_clusters, 0, sizeof(int32_t), stream));
        return P2N_OK;
    }
    P2N_REQUIRE(emb && labels && workspace, "p2n_cluster_embeddings: null pointer");
    int G, CT;
    cl_geometry(D, G, CT);
    const long Npad = (N + CL_NPAD_UNIT - 1) / CL_NPAD_UNIT * CL_NPAD_UNIT;
    const int cap = min_samples - 1;
    ClWs w;
    const size_t need = cl_carve(w, (char *)workspace, Npad, G, N, cap);
    if (workspace_bytes < need) {
        p2n_set_error("p2n_cluster_embeddings: workspace %zu < %zu bytes", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    const unsigned nb256 = (unsigned)((Npad + 255) / 256);
    int ps = p2n_prof_begin(P2N_K_CLUSTER_PREP, stream);
    P2N_CHECK_HIP(hipMemsetAsync(w.zero_begin, 0, w.zero_bytes, stream));      // stats, sums, counts, flags, queue lengths (cl_carve)
    P2N_CHECK_HIP(hipMemsetAsync(w.ones_begin, 0xff, w.ones_bytes, stream));   // the link pass's notes, owner: -1
    if (p2n_cluster_sorts(N)) P2N_CHECK_HIP(hipMemsetAsync(w.minorig, 0x7f, sizeof(int) * N, stream));   // (cl_minorig_kernel takes minima)
    const int *perm = nullptr;
    if (p2n_cluster_sorts(N)) {   // the working order (cl_code_kernel)
        // (the projections borrow Xp, which cl_normalize_kernel fills afterwards: 32 of its >= 64 floats per row; their
        // sums and near-threshold counts, CL_SUM_COPIES copies of each, start at zero with the rest of the cleared stretch)
        float *psum = w.psum;
        int *near = reinterpret_cast<int *>(psum + 32 * CL_SUM_COPIES);
        if (D >= 32 && D <= CL_STAGE_DMAX) {   // (the staged kernel hands its 32 results out through the row's own LDS)
            const size_t ldsb = cl_stage_lds_bytes(D);
            P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cl_project_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            if (D == 64) {
                P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cl_project_kernel<true, 64>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
                hipLaunchKernelGGL((cl_project_kernel<true, 64>), dim3((unsigned)((N + CL_STAGE_ROWS - 1) / CL_STAGE_ROWS)), dim3(CL_STAGE_ROWS), ldsb,
                                   stream, emb, N, D, w.Xp, psum);
            } else
            hipLaunchKernelGGL(cl_project_kernel<true>, dim3((unsigned)((N + CL_STAGE_ROWS - 1) / CL_STAGE_ROWS)), dim3(CL_STAGE_ROWS), ldsb,
                               stream, emb, N, D, w.Xp, psum);
        } else {
            hipLaunchKernelGGL(cl_project_kernel<false>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, emb, N, D, w.Xp, psum);
        }
        hipLaunchKernelGGL(cl_near_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, w.Xp, psum, N, (float)eps, near);
        hipLaunchKernelGGL(cl_code_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, w.Xp, psum, near, N, w.key_a, w.perm, w.stats + 7);
        rocprim::double_buffer<unsigned> kb(w.key_a, w.key_b);
        rocprim::double_buffer<int> vb(w.perm, w.val_b);
        size_t tmp = 0;
        P2N_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, tmp, kb, vb, (size_t)N, 0, 32, stream));
        P2N_REQUIRE(tmp <= w.sort_tmp_bytes, "p2n_cluster_embeddings: sort scratch %zu > %zu bytes", tmp, w.sort_tmp_bytes);
        P2N_CHECK_HIP(rocprim::radix_sort_pairs(w.sort_tmp, tmp, kb, vb, (size_t)N, 0, 32, stream));
        if (vb.current() != w.perm)
            P2N_CHECK_HIP(hipMemcpyAsync(w.perm, vb.current(), sizeof(int) * N, hipMemcpyDeviceToDevice, stream));
        perm = w.perm;
    }
    const int nb128 = (int)(Npad / 128);
    const bool staged = D <= CL_STAGE_DMAX;   // the staged normalise kernel leaves the block spheres as well
    ClSphereOut sp{w.centre, w.radius, p2n_cluster_refines() ? w.Cb : nullptr, w.nhc, nb128, (nb128 + 31) / 32 * 32, (float)eps};
    rc = cl_launch_normalize(emb, N, Npad, D, G, nullptr, w.Xp, w.nrm, w.Xb, perm, staged ? sp : ClSphereOut{}, stream);
    if (rc) return rc;
    if (!staged)
        hipLaunchKernelGGL(cl_sphere_kernel, dim3(nb128), dim3(128), sizeof(float) * (3 * 8 * G + 4), stream, w.Xp, w.nrm, G, w.centre,
                           w.radius, sp.Cb, w.nhc, nb128, sp.nbpad, (float)eps);
    if (!p2n_cluster_refines())
        hipLaunchKernelGGL(cl_active_kernel, dim3(nb128 / 2), dim3(256), sizeof(float) * (16 * G + 256), stream, w.centre, w.radius, nb128,
                           8 * G, (float)eps, w.act_cnt, w.act_list);
    const int sym = p2n_cluster_refines() && p2n_cluster_symmetric();   // (the halved lists are made by the reach stage's relist)
    if (p2n_cluster_refines()) {
        const int ntile = (nb128 + 31) / 32;
        // the tiles of a block shared out over gy workgroups: the kernel waits on its centre fragments, and wants all the
        // waves a SIMD holds (8 x 1,024 SIMDs) -- at 166k points one workgroup per block is 5 per SIMD (107 -> 98 us with two)
        int gy = (8192 + 4 * nb128 - 1) / (4 * nb128);
        gy = gy > 8 ? 8 : gy;
        gy = gy > ntile ? ntile : gy;
        if (G == 8)
            hipLaunchKernelGGL((cl_reach_kernel<8, 1>), dim3(nb128, gy), dim3(256), 0, stream, w.Xb, w.nrm, w.Cb, w.nhc, nb128, ntile, w.reach);
        else if (G == 16)
            hipLaunchKernelGGL((cl_reach_kernel<16, 1>), dim3(nb128, gy), dim3(256), 0, stream, w.Xb, w.nrm, w.Cb, w.nhc, nb128, ntile, w.reach);
        else
            hipLaunchKernelGGL((cl_reach_kernel<32, 1>), dim3(nb128, gy), dim3(256), 0, stream, w.Xb, w.nrm, w.Cb, w.nhc, nb128, ntile, w.reach);
        hipLaunchKernelGGL(cl_relist_kernel, dim3(nb128 / 2), dim3(256), 0, stream, (const unsigned char *)nullptr, (const int *)nullptr,
                           w.reach, nb128, sym, w.act_cnt, w.act_list);
    }
    p2n_prof_end(ps, stream);

    ClArgs a;
    a.Xp = w.Xp; a.Xb = w.Xb; a.nrm_row = w.nrm; a.nrm_col = w.nrm; a.Npad = Npad;
    a.eps2 = eps * eps;
    a.eps2f = (float)a.eps2;
    // > 2 * (8G) * 2^-24 (fp32 accumulation of a dot of unit vectors) + norm and threshold roundings,
    // + 2 * 1.2e-5 for the products the bf16 split leaves out (tau is on the d2 scale: twice the dot's)
    a.tau = 2.5e-5f * (float)(G / 8) + 2.5e-5f;
    a.cnt = w.cnt; a.parent = w.parent; a.core = w.core; a.colblk_has_core = w.blk_has_core;
    a.nbr_cnt = w.nbr_cnt; a.nbr = w.nbr; a.cap = cap; a.stats = w.stats;
    const long nrowblk = Npad / CL_ROWBLK;
    a.work_rb = w.work_rb; a.work_e0 = w.work_e0; a.work_meta = w.work_meta;
    a.defer_q = w.defer_q; a.defer_cnt = w.defer_cnt; a.defer_cap = CL_DEFER_CAP;
    a.act_cnt = w.act_cnt; a.act_list = w.act_list; a.nb128 = nb128; a.hit = w.hit;

    ps = p2n_prof_begin(P2N_K_CLUSTER_COUNT, stream);
    hipLaunchKernelGGL(cl_plan_kernel, dim3(1), dim3(1024), 0, stream, w.act_cnt, (int)nrowblk, w.kslots, w.work_rb, w.work_e0, w.work_meta);
    if (G == 8) rc = cl_launch_pairs<8, 4>(sym ? 2 : 0, a, w.kbound, stream);
    else if (G == 16) rc = cl_launch_pairs<16, 2>(sym ? 2 : 0, a, w.kbound, stream);
    else rc = cl_launch_pairs<32, 1>(sym ? 2 : 0, a, w.kbound, stream);
    p2n_prof_end(ps, stream);
    if (rc) return rc;

    {   // the queued float64 decisions of pass 1 (every lane of the pass wrote its queue length)
        const long slots = w.defer_waves * ((CL_DEFER_CAP + CL_RESOLVE_SG - 1) / CL_RESOLVE_SG) * 64;
        ps = p2n_prof_begin(P2N_K_CLUSTER_COUNT, stream);
        hipLaunchKernelGGL(cl_resolve_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, w.Xp, G, a.eps2,
                           w.defer_q, w.defer_cnt, CL_DEFER_CAP, w.defer_waves, w.work_rb, w.work_meta, w.cnt);
        p2n_prof_end(ps, stream);
    }
    hipLaunchKernelGGL(cl_prepare_link_kernel, dim3(nb256), dim3(256), 0, stream, w.cnt, w.nrm, N, Npad, min_samples, w.core,
                       w.nrm_core, w.parent, w.nbr_cnt, w.blk_has_core);
    a.nrm_col = w.nrm_core;
    ps = p2n_prof_begin(P2N_K_CLUSTER_LINK, stream);
    hipLaunchKernelGGL(cl_relist_kernel, dim3((unsigned)nrowblk), dim3(256), 0, stream, w.hit, w.blk_has_core, (const unsigned char *)nullptr, nb128, sym, w.act_cnt,
                       w.act_list);
    a.nsplit = cl_nsplit(Npad);
    a.blk_uni = nullptr; a.rb_uni = nullptr;
    if (p2n_cluster_link_notes()) {
        a.blk_uni = w.blk_uni; a.rb_uni = w.rb_uni;   // (-1 since the start of the call)
    }
    if (G == 8) rc = cl_launch_pairs<8, 4>(1, a, (int)nrowblk, stream);
    else if (G == 16) rc = cl_launch_pairs<16, 2>(1, a, (int)nrowblk, stream);
    else rc = cl_launch_pairs<32, 1>(1, a, (int)nrowblk, stream);
    p2n_prof_end(ps, stream);
    if (rc) return rc;

    ps = p2n_prof_begin(P2N_K_CLUSTER_PREP, stream);
    if (perm) {   // number the sets by the caller's indices
        const unsigned nbn = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(cl_minorig_kernel, dim3(nbn), dim3(256), 0, stream, w.parent, w.core, perm, N, w.minorig);
        hipLaunchKernelGGL(cl_owner_kernel, dim3(nbn), dim3(256), 0, stream, w.parent, w.core, N, w.minorig, w.owner);
    }
    else   // (the numbering reads parent[] flat either way)
        hipLaunchKernelGGL(cl_flatten_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, w.parent, w.core, N);
    {
        const int *owner = perm ? w.owner : (const int *)nullptr;
        const unsigned nchunk = (unsigned)((N + 1024 * CL_RANK_PT - 1) / (1024 * CL_RANK_PT));
        hipLaunchKernelGGL(cl_rank_count_kernel, dim3(nchunk), dim3(1024), 0, stream, w.parent, w.core, owner, N, w.chunk_cnt);
        hipLaunchKernelGGL(cl_rank_assign_kernel, dim3(nchunk), dim3(1024), 0, stream, w.parent, w.core, owner, N, w.chunk_cnt, w.rank,
                           num_clusters);
    }
    hipLaunchKernelGGL(cl_label_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, w.parent, w.core, w.rank,
                       w.nbr_cnt, w.nbr, cap, N, labels, perm);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// neighbourhood sizes of the last p2n_cluster_embeddings call on this workspace (test / diagnostics aid)
extern "C" int p2n_cluster_counts(const void *workspace, long N, int D, int min_samples, int32_t *counts, void *stream_) {
    P2N_REQUIRE(N >= 0 && D >= 1 && D <= 256 && min_samples >= 1, "p2n_cluster_counts: bad arguments");
    if (N == 0) return P2N_OK;
    int G, CT;
    cl_geometry(D, G, CT);
    ClWs w;
    cl_carve(w, (char *)workspace, (N + CL_NPAD_UNIT - 1) / CL_NPAD_UNIT * CL_NPAD_UNIT, G, N, min_samples - 1);
    hipLaunchKernelGGL(cl_unpermute_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w.cnt, w.perm,
                       w.stats + 7, N, counts);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// slow-path counters of the last call: stats[0] pairs decided in double in pass 1, [1] in pass 2,
// [2] union attempts, [3] border-neighbour records (device pointer to 4 x uint64)
extern "C" int p2n_cluster_stats(const void *workspace, long N, int D, int min_samples, uint64_t *stats, void *stream_) {
    P2N_REQUIRE(N >= 0 && D >= 1 && D <= 256 && min_samples >= 1 && stats, "p2n_cluster_stats: bad arguments");
    if (N == 0) {
        P2N_CHECK_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(uint64_t), (hipStream_t)stream_));
        return P2N_OK;
    }
    int G, CT;
    cl_geometry(D, G, CT);
    ClWs w;
    cl_carve(w, (char *)workspace, (N + CL_NPAD_UNIT - 1) / CL_NPAD_UNIT * CL_NPAD_UNIT, G, N, min_samples - 1);
    P2N_CHECK_HIP(hipMemcpyAsync(stats, w.stats, 4 * sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream_));
    return P2N_OK;
}
