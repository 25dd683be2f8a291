// InstanceSegLoss for gfx950 (MI355X): forward values and the gradient w.r.t. the
// embeddings, in one call.  Replaces reference losses.py:332-373 (InstanceSegLoss),
// :375-452 (DiscriminativeLoss: pull / push) and :454-489 (SmoothnessLoss: the
// reference's only kNN + neighbour gather) including what autograd derives from them.
//
// The reference loops in Python over clouds and instances (~2*I masked selects per
// cloud) and materialises an N x N distance matrix per cloud (268 MB at N = 8192).
// Here: instances arrive as dense ids (one integer per point), per-instance sums are
// segmented atomics, the per-cloud I x I push term runs in one workgroup, and the
// neighbour search is p2n_knn_self (never N x N).  All float arithmetic is fp32.
//
// Inputs prepared by the host (integer plumbing only, see losses.py): inst[P] = dense id
// of (cloud, label) in ascending (cloud, label) order, cnt[T] points per id, bg[T] = 1
// for label 0, cstart[B+1] = first id of each cloud.
#include "p2n_common.h"

#define NORM_EPS 1e-12f

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The wave's number inside its workgroup as a SCALAR: threadIdx.x >> 6 is wave-uniform by construction, but the
// compiler keeps it (and every row base, loop bound and pointer derived from it) in vector registers unless told.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float block_sum(float v, float *sh) {   // 256 threads
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// en = e / max(|e|, eps); rn = 1 / max(|e|, eps).  One wave per point, lanes over channels.
__global__ __launch_bounds__(256) void p2n_normalize_fwd_kernel(const float *__restrict__ e, long P, int E,
                                                               float *__restrict__ en, float *__restrict__ rn) {
    const long p = (long)blockIdx.x * 4 + wave_id();
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) { const float v = e[p * E + c]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    const float r = 1.0f / fmaxf(sqrtf(s), NORM_EPS);
    for (int c = lane; c < E; c += 64) en[p * E + c] = e[p * E + c] * r;
    if (lane == 0) rn[p] = r;
}

// de = (den - en*(en.den)) * rn
__global__ __launch_bounds__(256) void p2n_normalize_bwd_kernel(const float *den,   // may alias out
                                                               const float *__restrict__ en,
                                                               const float *__restrict__ rn, long P, int E,
                                                               float *out) {
    const long p = (long)blockIdx.x * 4 + wave_id();
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s = fmaf(en[p * E + c], den[p * E + c], s);
    s = wave_sum(s);
    const float r = rn[p];
    for (int c = lane; c < E; c += 64) out[p * E + c] = (den[p * E + c] - en[p * E + c] * s) * r;
}

// The two per-instance accumulations (instance sums, pull pass A), workgroup-private:
// a workgroup walks SEG_CHUNK consecutive points of ONE cloud and adds into an LDS copy of that
// cloud's instance rows (ids of a cloud are contiguous, cstart), then flushes the rows it touched
// with one global atomic per element.  The direct form sends 64 global atomics per point to a few
// dozen rows: 262,144 points hammer ~1,000 addresses-rows and the memory-side atomic units serialise.
// MODE 0: sums[i] += en[p].  MODE 1: pullsum[i] += max(|en[p]-mu[i]| - delta_v, 0), G[i] += unit vector.
#define SEG_CHUNK 512
#define SEG_THREADS 1024   // 16 waves x 4 points per step: a wave walks 32 points (each step is a chain of dependent loads)
#define SEG_MAX_LDS (60 * 1024)
template <int MODE>
__global__ __launch_bounds__(SEG_THREADS) void p2n_seg_lds_kernel(const float *__restrict__ en, const int *__restrict__ inst,
                                                         const int *__restrict__ bg, const int *__restrict__ cstart,
                                                         const float *__restrict__ mu, int N, int E, float delta_v,
                                                         int want_grad, float *__restrict__ rows_out,
                                                         float *__restrict__ pullsum, int cap) {
    extern __shared__ float lacc[];   // [Tb][E] (+ [Tb] hinge sums for MODE 1)
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = wave_id();
    const int c0 = cstart[b], Tb = cstart[b + 1] - c0;
    // a cloud with more instance rows than the LDS copy holds (cap) adds straight into global memory
    const bool priv = Tb <= cap;
    float *gacc = rows_out + (long)c0 * E;   // the same rows in global memory
    float *lhs = lacc + (long)Tb * E;
    // separate LDS / global atomics (one generic pointer would turn both into flat atomics)
    auto add_row = [&](int off, float val) __attribute__((always_inline)) {
        if (priv) atomicAdd(&lacc[off], val);
        else atomicAdd(&gacc[off], val);
    };
    if (priv) {
        const int nacc = Tb * E + (MODE == 1 ? Tb : 0);
        for (int i = threadIdx.x; i < nacc; i += SEG_THREADS) lacc[i] = 0.f;
    }
    __syncthreads();
    const int n0 = blockIdx.x * SEG_CHUNK;
    const int n1 = min(N, n0 + SEG_CHUNK);
    for (int nb = n0 + wave * 4; nb < n1; nb += SEG_THREADS / 16) {   // four points per wave step, loads first
        int li[4];
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = nb + u;
            li[u] = -1;
            if (n < n1) {
                const int i = inst[(long)b * N + n];
                if (!bg[i]) li[u] = i - c0;
            }
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int c = lane + 64 * c4;
                v[u][c4] = (n < n1 && c < E) ? en[((long)b * N + n) * E + c] : 0.f;   // not behind the inst -> bg chain
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (li[u] < 0) continue;   // wave-uniform
            if (MODE == 0) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int c = lane + 64 * c4;
                    if (c < E) add_row(li[u] * E + c, v[u][c4]);
                }
            } else {
                float d[4], s = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int c = lane + 64 * c4;
                    d[c4] = c < E ? v[u][c4] - mu[(long)(c0 + li[u]) * E + c] : 0.f;
                    s = fmaf(d[c4], d[c4], s);
                }
                s = wave_sum(s);
                const float dist = sqrtf(s);
                const float hv = dist - delta_v;
                if (hv > 0.f) {
                    if (lane == 0) { if (priv) atomicAdd(&lhs[li[u]], hv); else atomicAdd(&pullsum[c0 + li[u]], hv); }
                    if (want_grad && dist > 0.f) {
                        const float r = 1.0f / dist;
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const int c = lane + 64 * c4;
                            if (c < E) add_row(li[u] * E + c, d[c4] * r);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!priv) return;
    for (int i = threadIdx.x; i < Tb * E; i += SEG_THREADS)
        if (lacc[i] != 0.f) atomicAdd(&rows_out[(long)c0 * E + i], lacc[i]);
    if (MODE == 1)
        for (int i = threadIdx.x; i < Tb; i += SEG_THREADS)
            if (lhs[i] != 0.f) atomicAdd(&pullsum[c0 + i], lhs[i]);
}

// One workgroup per cloud: means, normalised means, push loss and its gradient w.r.t. the means.
//   mu[i] = sums[i]/cnt[i]; nmu = mu/max(|mu|,eps)
//   push_b = sum_{i<j} max(2*delta_d - |nmu_i - nmu_j|, 0) / (pairs + 1e-6)      (losses.py:431-441)
//   dmu[i] = gradient of (push_b / B) w.r.t. mu[i]
// The pair loop is a chain of short dependent steps (row difference, wave sum, sqrt): 16 waves per cloud and the
// cloud's normalised means in LDS (when they fit DISC_LDS_FLOATS; from global memory otherwise) keep a step at
// LDS latency instead of an L2 round trip (167 -> ~20 us at 32 clouds x 32 instances).
#define DISC_THREADS 1024
#define DISC_LDS_FLOATS 12288
__global__ __launch_bounds__(DISC_THREADS) void p2n_disc_cloud_kernel(const float *__restrict__ sums,
                                                                      const int *__restrict__ cnt, const int *__restrict__ bg,
                                                                      const int *__restrict__ cstart, int E, int B,
                                                                      float delta_d, float *__restrict__ mu,
                                                                      float *__restrict__ nmu, float *__restrict__ dmu,
                                                                      float *__restrict__ push_b, int want_grad) {
    __shared__ float lrow[DISC_LDS_FLOATS];
    __shared__ float sh[DISC_THREADS / 64];
    const int b = blockIdx.x;
    const int i0 = cstart[b], i1 = cstart[b + 1];
    const int lane = threadIdx.x & 63, wave = wave_id();
    const bool priv = (long)(i1 - i0) * E <= DISC_LDS_FLOATS;
    int ninst = 0;
    for (int i = i0; i < i1; ++i) ninst += bg[i] ? 0 : 1;
    // means and normalised means (wave per instance row)
    for (int i = i0 + wave; i < i1; i += DISC_THREADS / 64) {
        float v[4], s = 0.f;   // E <= 256: up to 4 channels per lane
        const bool isbg = bg[i] != 0;
        int q = 0;
        for (int c = lane; c < E; c += 64, ++q) {
            v[q] = isbg ? 0.f : sums[(long)i * E + c] / (float)cnt[i];
            mu[(long)i * E + c] = v[q];
            s = fmaf(v[q], v[q], s);
        }
        s = wave_sum(s);
        const float r = 1.0f / fmaxf(sqrtf(s), NORM_EPS);
        q = 0;
        for (int c = lane; c < E; c += 64, ++q) {
            const float n = v[q] * r;
            nmu[(long)i * E + c] = n;
            if (priv) lrow[(i - i0) * E + c] = n;
        }
    }
    __syncthreads();
    auto row = [&](int i, int c) { return priv ? lrow[(i - i0) * E + c] : nmu[(long)i * E + c]; };
    float loss = 0.f;
    const float pairs = 0.5f * (float)ninst * (float)(ninst - 1);
    const float w = 1.0f / (pairs + 1e-6f);
    if (ninst > 1) {
        // wave per instance i, lanes over channels; full row j != i so no atomics are needed
        for (int i = i0 + wave; i < i1; i += DISC_THREADS / 64) {
            if (bg[i]) continue;
            float gacc[4] = {0.f, 0.f, 0.f, 0.f};
            float ri[4] = {0.f, 0.f, 0.f, 0.f};
            {
                int q = 0;
                for (int c = lane; c < E; c += 64, ++q) ri[q] = row(i, c);
            }
            for (int j = i0; j < i1; ++j) {
                if (j == i || bg[j]) continue;
                float dv[4] = {0.f, 0.f, 0.f, 0.f};
                float s = 0.f;
                int q = 0;
                for (int c = lane; c < E; c += 64, ++q) {
                    dv[q] = ri[q] - row(j, c);
                    s = fmaf(dv[q], dv[q], s);
                }
                s = wave_sum(s);
                const float d = sqrtf(s);
                const float term = 2.0f * delta_d - d;
                if (term > 0.f) {
                    if (j > i && lane == 0) loss += term;
                    if (want_grad && d > 0.f) {
                        const float coef = -w / d / (float)B;
#pragma unroll
                        for (int u = 0; u < 4; ++u) gacc[u] = fmaf(coef, dv[u], gacc[u]);
                    }
                }
            }
            if (want_grad) {
                // back through nmu = mu / |mu|
                float dot = 0.f, n2 = 0.f;
                int q = 0;
                for (int c = lane; c < E; c += 64, ++q) {
                    dot = fmaf(ri[q], gacc[q], dot);
                    const float v = mu[(long)i * E + c];
                    n2 = fmaf(v, v, n2);
                }
                dot = wave_sum(dot);
                n2 = wave_sum(n2);
                const float r = 1.0f / fmaxf(sqrtf(n2), NORM_EPS);
                q = 0;
                for (int c = lane; c < E; c += 64, ++q)
                    dmu[(long)i * E + c] = (gacc[q] - ri[q] * dot) * r;
            }
        }
    } else if (want_grad) {
        for (int idx = threadIdx.x; idx < (i1 - i0) * E; idx += DISC_THREADS) dmu[(long)i0 * E + idx] = 0.f;
    }
    loss = wave_sum(loss);
    if (lane == 0) sh[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int q = 0; q < DISC_THREADS / 64; ++q) t += sh[q];
        push_b[b] = ninst > 1 ? t * w : 0.f;
    }
}

// per cloud: pull_b = sum_i (pullsum[i]/cnt[i]) / (I_b + 1e-6)          (losses.py:413-428)
__global__ void p2n_pull_cloud_kernel(const float *__restrict__ pullsum, const int *__restrict__ cnt,
                                      const int *__restrict__ bg, const int *__restrict__ cstart, int B,
                                      float *__restrict__ pull_b, float *__restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float acc = 0.f;
    int ninst = 0;
    for (int i = cstart[b]; i < cstart[b + 1]; ++i)
        if (!bg[i]) { acc += pullsum[i] / (float)cnt[i]; ++ninst; }
    const float denom = (float)ninst + 1e-6f;
    pull_b[b] = ninst ? acc / denom : 0.f;
    coef[b] = ninst ? 1.0f / denom / (float)B : 0.f;   // d(pull)/d(instance mean distance)
}

// pull pass B: gradients w.r.t. the normalised embeddings
//   dpull[p] = coef_b/n_i * (u_p - G_i/n_i);  dpush[p] = dmu_i / n_i
__global__ __launch_bounds__(256) void p2n_pull_b_kernel(const float *__restrict__ en, const int *__restrict__ inst,
                                                        const int *__restrict__ bg, const int *__restrict__ cnt,
                                                        const float *__restrict__ mu, const float *__restrict__ G,
                                                        const float *__restrict__ dmu, const float *__restrict__ coef,
                                                        const float *__restrict__ rn, long P, int N, int E, float delta_v,
                                                        float *__restrict__ dpull, float *__restrict__ dpush) {
    // gradients w.r.t. the normalised embedding, taken straight back through en = e / |e| (what
    // p2n_normalize_bwd_kernel does for a stored gradient): out = (g - en * <en, g>) * rn
    const long p = (long)blockIdx.x * 4 + wave_id();
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    const int i = inst[p];
    if (bg[i]) {
        for (int c = lane; c < E; c += 64) { dpull[p * E + c] = 0.f; dpush[p * E + c] = 0.f; }
        return;
    }
    float ev[4], dv[4];   // E <= 256: up to four channels per lane
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        ev[q] = dv[q] = 0.f;
        if (c < E) {
            ev[q] = en[p * E + c];
            dv[q] = ev[q] - mu[(long)i * E + c];
            s = fmaf(dv[q], dv[q], s);
        }
    }
    s = wave_sum(s);
    const float dist = sqrtf(s);
    const bool act = (dist - delta_v > 0.f) && dist > 0.f;
    const float r = act ? 1.0f / dist : 0.f;
    const float rn_i = 1.0f / (float)cnt[i];
    const float cb = coef[p / N] * rn_i;
    float gp[4], gs[4], sp = 0.f, ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        gp[q] = gs[q] = 0.f;
        if (c < E) {
            const float u = dv[q] * r;
            gp[q] = cb * (u - G[(long)i * E + c] * rn_i);
            gs[q] = dmu[(long)i * E + c] * rn_i;
            sp = fmaf(ev[q], gp[q], sp);
            ss = fmaf(ev[q], gs[q], ss);
        }
    }
    sp = wave_sum(sp);
    ss = wave_sum(ss);
    const float rp = rn[p];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        if (c < E) {
            dpull[p * E + c] = (gp[q] - ev[q] * sp) * rp;
            dpush[p * E + c] = (gs[q] - ev[q] * ss) * rp;
        }
    }
}

// ---- smoothness, row-parallel form (E <= 256) ------------------------------------------------
// A wave owns one point; its four 16-lane rows each take one neighbour at a time, a lane holds
// 4 * V channels of that neighbour's row.  The k neighbour indices are
// fetched with one load, and the row loads of SM_U neighbour groups (4 * SM_U neighbours) are all in
// flight before the first distance is reduced: the first version walked the neighbours one by one,
// index load -> row load -> 64-lane shuffle reduction, each step waiting for the one before.
#define SM_U 5   // neighbour groups in flight (20 neighbours: the reference's k)

__device__ __forceinline__ float row16_sum(float v) {   // sum over each 16-lane row, in every lane of the row
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

template <int V, bool BWD>
__global__ __launch_bounds__(256) void p2n_smooth_vec_kernel(const float *__restrict__ en, const int *__restrict__ idx,
                                                            long P, int N, int E, int k1, float w,
                                                            float *__restrict__ part, float *__restrict__ den) {
    __shared__ float sh[4];
    const long p = (long)blockIdx.x * 4 + wave_id();
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    const bool live = p < P;
    const long pp = live ? p : 0;
    const long base = (pp / N) * (long)N;
    // all neighbour indices of the point with one load (k1 <= 64): lane j holds column j
    int myidx = (live && lane < k1) ? idx[pp * k1 + lane] : 0;
    // an unfilled neighbour slot (-1: fewer than k1 finite distances, i.e. non-finite coordinates) becomes the point
    // itself -- distance 0, no contribution, and no read outside the cloud
    if ((unsigned)myidx >= (unsigned)N) myidx = (int)(pp - base);
    // element (v, e) of a lane is channel 64 v + 16 e + l: every load / atomic instruction touches 16
    // consecutive floats (one 64-byte segment) per row
    float ep[V][4], self[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 64 * v + 16 * e + l;
            ep[v][e] = c < E ? en[pp * E + c] : 0.f;
            self[v][e] = 0.f;
        }
    float acc = 0.f;
    const int T = (k1 - 1 + 3) / 4;   // neighbour groups of four
    for (int t0 = 0; t0 < T; t0 += SM_U) {
        long q[SM_U];
        bool ok[SM_U];
        float eq[SM_U][V][4];
#pragma unroll
        for (int u = 0; u < SM_U; ++u) {
            const int j = 1 + 4 * (t0 + u) + g;   // column 0 is the point itself (reference losses.py:479)
            ok[u] = live && j < k1;
            q[u] = base + __shfl(myidx, ok[u] ? j : 0);
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 64 * v + 16 * e + l;
                    eq[u][v][e] = (ok[u] && c < E) ? en[q[u] * E + c] : ep[v][e];
                }
        }
#pragma unroll
        for (int u = 0; u < SM_U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    eq[u][v][e] -= ep[v][e];
                    s = fmaf(eq[u][v][e], eq[u][v][e], s);
                }
            s = row16_sum(s);
            const float d = sqrtf(s);
            if (!BWD) {
                acc += (ok[u] && l == 0) ? d : 0.f;
            } else if (ok[u] && d > 0.f) {
                const float r = w / d;
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 64 * v + 16 * e + l;
                        if (c < E) {
                            const float gq = eq[u][v][e] * r;
                            atomicAdd(&den[q[u] * E + c], gq);
                            self[v][e] -= gq;
                        }
                    }
            }
        }
    }
    if (!BWD) {
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) part[blockIdx.x] = acc;
    } else {
        // the four rows' shares of den[p]: add them up inside the wave, one atomic per element
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = self[v][e];
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                const int c = 64 * v + 16 * e + l;
                if (live && g == 0 && c < E) atomicAdd(&den[p * E + c], t);
            }
    }
}

// ---- smoothness backward without float atomics ---------------------------------------------------
// d/d en[q] of sum_p sum_{x in N(p)} |en[x] - en[p]| collects one unit vector per edge that touches q:
//   den[q] = w * sum over x in N(q) and over p with q in N(p) of (en[q] - en[x or p]) / |en[q] - en[..]|.
// The scatter form above sends 64 float atomics per edge (335 M at B=32 x 8192 x 20: the memory-side
// atomic rate bounds it at ~1.1 ms).  Here the reverse neighbour lists are built first (in-degree
// count, exclusive scan, fill: integer atomics, one per edge) and every point then PULLS both of its
// lists, 16 lanes per row, accumulating in registers and storing den[q] once -- no memset, no float atomic.
__global__ __launch_bounds__(256) void p2n_rev_count_kernel(const int *__restrict__ idx, long P, int N, int k1,
                                                           int *__restrict__ deg) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long p = e / k1;
    const int j = (int)(e % k1);
    if (p >= P || j == 0) return;
    int q = idx[e];
    if ((unsigned)q >= (unsigned)N) q = (int)(p % N);   // unfilled slot: a self edge (weight 0), as everywhere else
    atomicAdd(&deg[(p / N) * (long)N + q], 1);
}

// exclusive scan of deg[0..P) -> off[0..P], three launches: per-block scan of 1024 elements, scan of the
// block totals by one workgroup, offset add
__global__ __launch_bounds__(256) void p2n_scan_block_kernel(const int *__restrict__ in, long n, int *__restrict__ out,
                                                            int *__restrict__ totals) {
    __shared__ int sh[256];
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    int v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = i0 + q < n ? in[i0 + q] : 0; s += v[q]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - s;   // exclusive prefix of this thread inside the block
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (i0 + q < n) out[i0 + q] = run; run += v[q]; }
    if (threadIdx.x == 255) totals[blockIdx.x] = sh[255];
}

__global__ __launch_bounds__(256) void p2n_scan_totals_kernel(int *__restrict__ totals, long nb, int *__restrict__ grand) {
    __shared__ int sh[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long b0 = 0; b0 < nb; b0 += 256) {
        const long i = b0 + threadIdx.x;
        const int v = i < nb ? totals[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) totals[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand = carry;
}

__global__ __launch_bounds__(256) void p2n_scan_add_kernel(int *__restrict__ out, long n, const int *__restrict__ totals) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int t = totals[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < n) out[i0 + q] += t;
}

// rev[off[q] + slot] = p for every edge p -> q; deg[] counts down to zero (the slot order is arbitrary)
__global__ __launch_bounds__(256) void p2n_rev_fill_kernel(const int *__restrict__ idx, long P, int N, int k1,
                                                          const int *__restrict__ off, int *__restrict__ deg,
                                                          int *__restrict__ rev) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long p = e / k1;
    const int j = (int)(e % k1);
    if (p >= P || j == 0) return;
    int ql = idx[e];
    if ((unsigned)ql >= (unsigned)N) ql = (int)(p % N);
    const long q = (p / N) * (long)N + ql;
    const int slot = atomicSub(&deg[q], 1) - 1;
    rev[off[q] + slot] = (int)(p % N);
}

// The same lists built with the histogram in LDS (clouds of up to REV_MAX_N points).  Every cloud has exactly
// N*(k1-1) edges, so its lists start at b*N*(k1-1) and no cross-cloud scan is needed.  A cloud's targets are cut
// into H slices, one workgroup each: the workgroup reads ALL edges of the cloud (coalesced, L2-resident), counts
// those that point into its slice in LDS and, in passing, how many point below it (= where its lists start),
// scans, writes the offsets, and uses the histogram words as fill cursors.  The scattered global atomics of the
// three-kernel form above (2 per edge, ~25 G/s) become LDS atomics; what is left scattered is one 4-byte store
// per edge, spread over B*H compute units (one CU retires only ~1 scattered store per 2 clocks).
#define REV_THREADS 1024
#define REV_MAX_N 32768
#define REV_U 8
#define REV_WORDS (REV_MAX_N + REV_MAX_N / 32)
__global__ __launch_bounds__(REV_THREADS) void p2n_rev_build_kernel(const int *__restrict__ idx, int B, int N, int k1,
                                                                    int slice, int *__restrict__ off,
                                                                    int *__restrict__ rev) {
    extern __shared__ int rv_bin[];
    __shared__ int wtot[REV_THREADS / 64], wbelow[REV_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int lo = blockIdx.x * slice, hi = min(N, lo + slice);   // this workgroup's targets
    const int k = k1 - 1;
    const long gbase = (long)b * N * k;               // first list slot of this cloud
    const int *I = idx + (long)b * N * k1;
    const int ne = N * k1;                            // < 2^31 (checked by the caller)
    const int words = slice + (slice >> 5);
    for (int i = tid; i < words; i += REV_THREADS) rv_bin[i] = 0;
    __syncthreads();
    // edge e = p*k1 + j; column j == 0 (the point itself or a duplicate of it) is not an edge.
    // tid = tp*k1 + tj once; the workgroup's base edge advances by REV_THREADS = sp*k1 + sj per round.
    const int tp = tid / k1, tj = tid - tp * k1;
    const int sp = REV_THREADS / k1, sj = REV_THREADS - sp * k1;
    int below = 0;
    {
        // REV_U rounds of loads in flight per step: one load per round would leave the pass latency-bound
        int j0 = 0, p0 = 0;                           // round base = p0*k1 + j0
        for (int e0 = 0; e0 < ne; e0 += REV_U * REV_THREADS) {
            int qv[REV_U];
#pragma unroll
            for (int u = 0; u < REV_U; ++u) {
                const int e = e0 + u * REV_THREADS + tid;
                qv[u] = e < ne ? I[e] : 0;
            }
#pragma unroll
            for (int u = 0; u < REV_U; ++u) {
                int j = j0 + tj, p = p0 + tp;
                if (j >= k1) { j -= k1; ++p; }
                if (e0 + u * REV_THREADS + tid < ne && j != 0) {
                    int q = qv[u];
                    if ((unsigned)q >= (unsigned)N) q = p;   // unfilled slot (-1, non-finite input): a self edge, weight 0
                    if (q < lo) ++below;
                    else if (q < hi) atomicAdd(&rv_bin[(q - lo) + ((q - lo) >> 5)], 1);
                }
                j0 += sj; p0 += sp;
                if (j0 >= k1) { j0 -= k1; ++p0; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    if (lane == 0) wbelow[wave] = below;
    __syncthreads();
    // exclusive scan of the slice's histogram: `per` consecutive bins per thread (per <= 32), wave scan, wave totals
    const int per = slice / REV_THREADS;              // slice is a multiple of REV_THREADS (see the launch)
    int mine = 0;
    const int i0 = tid * per;
    for (int j = 0; j < per; ++j) mine += rv_bin[(i0 + j) + ((i0 + j) >> 5)];
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < REV_THREADS / 64; ++w) {
        base += wbelow[w];
        if (w < wave) base += wtot[w];
    }
    for (int j = 0; j < per; ++j) {
        const int w = (i0 + j) + ((i0 + j) >> 5);
        const int c = rv_bin[w];
        rv_bin[w] = base;
        base += c;
    }
    __syncthreads();
    for (int i = lo + tid; i < hi; i += REV_THREADS) off[(long)b * N + i] = (int)(gbase + rv_bin[(i - lo) + ((i - lo) >> 5)]);
    if (b == B - 1 && blockIdx.x == 0 && tid == 0) off[(long)B * N] = (int)((long)B * N * k);
    __syncthreads();
    {
        int j0 = 0, p0 = 0;
        for (int e0 = 0; e0 < ne; e0 += REV_U * REV_THREADS) {
            int qv[REV_U];
#pragma unroll
            for (int u = 0; u < REV_U; ++u) {
                const int e = e0 + u * REV_THREADS + tid;
                qv[u] = e < ne ? I[e] : 0;
            }
#pragma unroll
            for (int u = 0; u < REV_U; ++u) {
                int j = j0 + tj, p = p0 + tp;
                if (j >= k1) { j -= k1; ++p; }
                if (e0 + u * REV_THREADS + tid < ne && j != 0) {
                    int q = qv[u];
                    if ((unsigned)q >= (unsigned)N) q = p;
                    if (q >= lo && q < hi) {
                        const int slot = atomicAdd(&rv_bin[(q - lo) + ((q - lo) >> 5)], 1);
                        rev[gbase + slot] = p;
                    }
                }
                j0 += sj; p0 += sp;
                if (j0 >= k1) { j0 -= k1; ++p0; }
            }
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void p2n_smooth_pull_kernel(const float *__restrict__ en, const int *__restrict__ idx,
                                                             const int *__restrict__ off, const int *__restrict__ rev,
                                                             const float *__restrict__ rn, long P, int N, int E, int k1,
                                                             float w, float *__restrict__ out, float *__restrict__ part) {
    // part != NULL: the forward value rides along -- the distances of the point's own k edges are summed per
    // workgroup into part[blockIdx.x], exactly what p2n_smooth_vec_kernel<V,false> would have written
    __shared__ float sh[4];
    const long p_ = (long)blockIdx.x * 4 + wave_id();
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    const bool live = p_ < P;
    const long p = live ? p_ : 0;
    const long base = (p / N) * (long)N;
    int myidx = lane < k1 ? idx[p * k1 + lane] : 0;
    if ((unsigned)myidx >= (unsigned)N) myidx = (int)(p - base);   // unfilled slot (-1): the point itself, see p2n_smooth_vec_kernel
    const int r0 = off[p], nrev = live ? off[p + 1] - r0 : 0;
    const int k = k1 - 1, total = live ? k + nrev : 0;
    float dsum = 0.f;
    float ep[V][4], acc[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 64 * v + 16 * e + l;
            ep[v][e] = c < E ? en[p * E + c] : 0.f;
            acc[v][e] = 0.f;
        }
    for (int t0 = 0; t0 < total; t0 += 64) {
        // lane t holds list position t0 + t: own neighbours first, then the points that list p
        const int pos = t0 + lane;
        int cand = __shfl(myidx, pos < k ? pos + 1 : 0);
        if (pos >= k && pos < total) cand = rev[r0 + pos - k];
        const int chunk = min(64, total - t0);
        for (int u0 = 0; u0 < chunk; u0 += 4 * SM_U) {
            bool ok[SM_U];
            float eq[SM_U][V][4];
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                const int t = u0 + 4 * u + g;
                ok[u] = t < chunk;
                const long q = base + __shfl(cand, ok[u] ? t : 0);
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 64 * v + 16 * e + l;
                        eq[u][v][e] = (ok[u] && c < E) ? en[q * E + c] : ep[v][e];
                    }
            }
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        eq[u][v][e] = ep[v][e] - eq[u][v][e];
                        s = fmaf(eq[u][v][e], eq[u][v][e], s);
                    }
                s = row16_sum(s);
                const float d = sqrtf(s);
                dsum += (ok[u] && l == 0 && t0 + u0 + 4 * u + g < k) ? d : 0.f;   // own edges only: each counted once
                const float r = (ok[u] && d > 0.f) ? w / d : 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[v][e] = fmaf(eq[u][v][e], r, acc[v][e]);
            }
        }
    }
    // den[p] (sum of the four rows' shares), taken straight back through en = e / |e|:
    // out = (den - en * <en, den>) * rn, the point's own row is still in registers
    float dot = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = acc[v][e];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            acc[v][e] = t;
            dot = fmaf(ep[v][e], t, dot);   // ep is zero beyond channel E
        }
    dot = row16_sum(dot);
    const float rp = rn[p];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 64 * v + 16 * e + l;
            if (live && g == 0 && c < E) out[p * E + c] = (acc[v][e] - ep[v][e] * dot) * rp;
        }
    if (part) {
        dsum = block_sum(dsum, sh);
        if (threadIdx.x == 0) part[blockIdx.x] = dsum;
    }
}

static void smooth_launch(bool bwd, const float *en, const int *idx, long P, int N, int E, int k1, float w, float *part,
                          float *den, unsigned blocks, hipStream_t s);

// out[0..2] = {pull, push, smooth}: deterministic final sums (fixed summation tree).  One workgroup of 1024
// threads, eight independent 16-byte loads in flight per thread: the 65,536 partial sums of a 32 x 8192 batch
// took 62 us as a chain of dependent 4-byte loads on 256 threads.
#define FIN_THREADS 1024
__global__ __launch_bounds__(FIN_THREADS) void p2n_loss_finalize_kernel(const float *__restrict__ pull_b,
                                                                       const float *__restrict__ push_b, int B,
                                                                       const float *__restrict__ spart, long nspart,
                                                                       float smooth_scale, float *__restrict__ out) {
    __shared__ float sh[3][FIN_THREADS / 64];
    const int t = threadIdx.x;
    float a = 0.f, b = 0.f;
    for (int i = t; i < B; i += FIN_THREADS) { a += pull_b[i]; b += push_b[i]; }
    // spart is 256-byte aligned (workspace carve): float4 body, scalar tail
    float c4[4] = {0.f, 0.f, 0.f, 0.f};
    const long nv = nspart >> 2;
    const float4 *sp4 = reinterpret_cast<const float4 *>(spart);
    long i = t;
    for (; i + 7L * FIN_THREADS < nv; i += 8L * FIN_THREADS) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = sp4[i + (long)u * FIN_THREADS];
#pragma unroll
        for (int u = 0; u < 8; ++u) { c4[0] += q[u].x; c4[1] += q[u].y; c4[2] += q[u].z; c4[3] += q[u].w; }
    }
    for (; i < nv; i += FIN_THREADS) { const float4 q = sp4[i]; c4[0] += q.x; c4[1] += q.y; c4[2] += q.z; c4[3] += q.w; }
    float c = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    for (long j = (nv << 2) + t; j < nspart; j += FIN_THREADS) c += spart[j];
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if ((t & 63) == 0) { sh[0][t >> 6] = a; sh[1][t >> 6] = b; sh[2][t >> 6] = c; }
    __syncthreads();
    if (t == 0) {
        float ta = 0.f, tb = 0.f, tc = 0.f;
        for (int w = 0; w < FIN_THREADS / 64; ++w) { ta += sh[0][w]; tb += sh[1][w]; tc += sh[2][w]; }
        out[0] = ta / (float)B; out[1] = tb / (float)B; out[2] = tc * smooth_scale;
    }
}

static void smooth_launch(bool bwd, const float *en, const int *idx, long P, int N, int E, int k1, float w, float *part,
                          float *den, unsigned blocks, hipStream_t s) {
    const int V = (E + 63) / 64;   // E <= 256, k1 <= 64 (checked by the entry point)
#define SM_GO(VV)                                                                                                         \
    do {                                                                                                                  \
        if (bwd) hipLaunchKernelGGL((p2n_smooth_vec_kernel<VV, true>), dim3(blocks), dim3(256), 0, s, en, idx, P, N, E, k1, w, part, den); \
        else hipLaunchKernelGGL((p2n_smooth_vec_kernel<VV, false>), dim3(blocks), dim3(256), 0, s, en, idx, P, N, E, k1, w, part, den);   \
    } while (0)
    if (V == 1) SM_GO(1);
    else if (V == 2) SM_GO(2);
    else if (V == 3) SM_GO(3);
    else SM_GO(4);
#undef SM_GO
}

// ------------------------------------------------------------------ C ABI
namespace {
struct LossWs {
    float *en, *rn, *sums, *mu, *nmu, *dmu, *G, *pullsum, *pull_b, *push_b, *coef, *spart, *den;
    int *idx, *deg, *off, *rev, *totals;
    char *knn_ws;
    size_t knn_bytes, total;
};
size_t loss_carve(LossWs &w, char *base, long P, int E, int T, int B, int k1) {
    const int N = (int)(P / B);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    w.en = (float *)take(sizeof(float) * P * E);
    w.rn = (float *)take(sizeof(float) * P);
    w.sums = (float *)take(sizeof(float) * (size_t)T * E);     // sums, G, pullsum are zeroed together
    w.G = (float *)take(sizeof(float) * (size_t)T * E);
    w.pullsum = (float *)take(sizeof(float) * (size_t)T);
    w.mu = (float *)take(sizeof(float) * (size_t)T * E);
    w.nmu = (float *)take(sizeof(float) * (size_t)T * E);
    w.dmu = (float *)take(sizeof(float) * (size_t)T * E);
    w.pull_b = (float *)take(sizeof(float) * B);
    w.push_b = (float *)take(sizeof(float) * B);
    w.coef = (float *)take(sizeof(float) * B);
    w.spart = (float *)take(sizeof(float) * ((P + 3) / 4));
    w.den = (float *)take(sizeof(float) * P * E);
    w.idx = (int *)take(sizeof(int) * P * k1);
    w.deg = (int *)take(sizeof(int) * (P + 4));
    w.off = (int *)take(sizeof(int) * (P + 4));
    w.rev = (int *)take(sizeof(int) * P * k1);
    w.totals = (int *)take(sizeof(int) * ((P + 1023) / 1024 + 4));
    w.knn_bytes = p2n_knn_self_workspace_bytes(B, N);   // Morton-ordered copy + tile boxes of the neighbour search
    w.knn_ws = take(w.knn_bytes);
    return off;
}
}  // namespace

extern "C" size_t p2n_instance_seg_loss_workspace_bytes(int B, int N, int E, int T, int k) {
    if (B < 1 || N < 1 || E < 1 || T < 1 || k < 1) return 0;
    LossWs w;
    return loss_carve(w, nullptr, (long)B * N, E, T, B, k + 1);
}

// InstanceSegLoss forward (+ gradient).  points (B,N,3), emb (B,N,E); inst (B*N) dense ids,
// cnt/bg (T), cstart (B+1).  losses[3] = {pull, push, smoothness} (device).  When the three
// demb_* (B,N,E) are non-NULL they receive d pull / d emb, d push / d emb, d smooth / d emb.
static int loss_run(const float *points, const float *emb, const int32_t *inst, const int32_t *cnt,
                    const int32_t *bg, const int32_t *cstart, int B, int N, int E, int T,
                    float delta_v, float delta_d, int k, const int32_t *knn_idx, float *losses, float *demb_pull,
                    float *demb_push, float *demb_smooth, void *workspace, size_t workspace_bytes,
                    void *stream_) {
    P2N_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && T >= 1, "p2n_instance_seg_loss: bad shape B=%d N=%d T=%d", B, N, T);
    P2N_REQUIRE(E >= 1 && E <= 256, "p2n_instance_seg_loss: embedding_dim %d not in [1,256]", E);
    P2N_REQUIRE(k >= 1 && k + 1 <= 64, "p2n_instance_seg_loss: k=%d not in [1,63]", k);
    P2N_REQUIRE(k + 1 <= N, "p2n_instance_seg_loss: k+1=%d neighbours requested from clouds of %d points", k + 1, N);
    P2N_REQUIRE(points && emb && inst && cnt && bg && cstart && losses, "p2n_instance_seg_loss: null pointer");
    const bool want = demb_pull && demb_push && demb_smooth;
    P2N_REQUIRE(want || (!demb_pull && !demb_push && !demb_smooth), "p2n_instance_seg_loss: pass all three gradient buffers or none");
    const long P = (long)B * N;
    LossWs w;
    const size_t need = loss_carve(w, nullptr, P, E, T, B, k + 1);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_instance_seg_loss: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    loss_carve(w, (char *)workspace, P, E, T, B, k + 1);
    hipStream_t s = (hipStream_t)stream_;
    const unsigned pb = (unsigned)((P + 3) / 4);
    const int k1 = k + 1;
    hipLaunchKernelGGL(p2n_normalize_fwd_kernel, dim3(pb), dim3(256), 0, s, emb, P, E, w.en, w.rn);
    P2N_CHECK_HIP(hipGetLastError());
    // sums, G and pullsum are adjacent in the carve: zero them with one memset
    P2N_CHECK_HIP(hipMemsetAsync(w.sums, 0, (size_t)((char *)w.mu - (char *)w.sums), s));
    // workgroup-private accumulation: LDS holds up to seg_cap instance rows of one cloud (a cloud with
    // more falls back to global atomics inside the kernel); E <= 256: four channels per lane
    const int seg_cap = (int)(SEG_MAX_LDS / ((E + 1) * sizeof(float)));
    const size_t seg_lds = (size_t)seg_cap * (E + 1) * sizeof(float);
    const dim3 seg_grid((N + SEG_CHUNK - 1) / SEG_CHUNK, B);
    if (seg_lds > 48 * 1024) {
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_seg_lds_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds));
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_seg_lds_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds));
    }
    hipLaunchKernelGGL(p2n_seg_lds_kernel<0>, seg_grid, dim3(SEG_THREADS), seg_lds, s, w.en, inst, bg, cstart, (const float *)nullptr,
                       N, E, 0.f, 0, w.sums, (float *)nullptr, seg_cap);
    P2N_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(p2n_disc_cloud_kernel, dim3(B), dim3(DISC_THREADS), 0, s, w.sums, cnt, bg, cstart, E, B, delta_d, w.mu, w.nmu,
                       w.dmu, w.push_b, want ? 1 : 0);
    P2N_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(p2n_seg_lds_kernel<1>, seg_grid, dim3(SEG_THREADS), seg_lds, s, w.en, inst, bg, cstart, w.mu, N, E, delta_v,
                       want ? 1 : 0, w.G, w.pullsum, seg_cap);
    P2N_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(p2n_pull_cloud_kernel, dim3((B + 63) / 64), dim3(64), 0, s, w.pullsum, cnt, bg, cstart, B, w.pull_b,
                       w.coef);
    P2N_CHECK_HIP(hipGetLastError());
    // smoothness: exact kNN of every point within its own cloud (k+1 incl. the point itself)
    const int *nbr = knn_idx;   // the caller's search result (p2n_instance_seg_loss_knn), or searched here
    if (!nbr) {
        int rc = p2n_knn_self(points, B, N, k1, w.idx, nullptr, w.knn_ws, w.knn_bytes, stream_);
        if (rc) return rc;
        nbr = w.idx;
    }
    const float smooth_scale = 1.0f / ((float)N * (float)k) / (float)B;
    const bool fused = P * (long)k1 < 0x7fffffffL;   // reverse-list backward; it also produces the forward partial sums
    if (!want || !fused) {
        smooth_launch(false, w.en, nbr, P, N, E, k1, 0.f, w.spart, nullptr, (unsigned)pb, s);
        P2N_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(p2n_loss_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, s, w.pull_b, w.push_b, B, w.spart, (long)pb,
                           smooth_scale, losses);
        P2N_CHECK_HIP(hipGetLastError());
    }
    if (!want) return P2N_OK;
    // gradients w.r.t. the normalised embeddings, then back through the normalisation
    hipLaunchKernelGGL(p2n_pull_b_kernel, dim3(pb), dim3(256), 0, s, w.en, inst, bg, cnt, w.mu, w.G, w.dmu, w.coef, w.rn, P, N,
                       E, delta_v, demb_pull, demb_push);
    P2N_CHECK_HIP(hipGetLastError());
    if (fused) {
        // reverse neighbour lists, then a pull per point (no float atomics)
        const long ne = P * (long)k1, nb = (P + 1 + 1023) / 1024;
        if (N <= REV_MAX_N && (long)N * k1 < 0x7fffffffL) {
            // H slices of the cloud's targets, one workgroup each: enough workgroups for the chip, slices a
            // multiple of the workgroup size (the scan gives every thread the same number of bins)
            int H = 256 / B;
            H = H < 1 ? 1 : (H > 16 ? 16 : H);
            int slice = ((N + H - 1) / H + REV_THREADS - 1) / REV_THREADS * REV_THREADS;
            H = (N + slice - 1) / slice;
            const size_t lds = sizeof(int) * (size_t)(slice + slice / 32);
            if (lds > 48 * 1024)
                P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_rev_build_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(p2n_rev_build_kernel, dim3(H, B), dim3(REV_THREADS), lds, s, nbr, B, N, k1, slice, w.off, w.rev);
        } else {
        P2N_CHECK_HIP(hipMemsetAsync(w.deg, 0, sizeof(int) * (P + 4), s));
        hipLaunchKernelGGL(p2n_rev_count_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, nbr, P, N, k1, w.deg);
        hipLaunchKernelGGL(p2n_scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, s, w.deg, P + 1, w.off, w.totals);
        hipLaunchKernelGGL(p2n_scan_totals_kernel, dim3(1), dim3(256), 0, s, w.totals, nb, w.totals + nb);
        hipLaunchKernelGGL(p2n_scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, s, w.off, P + 1, w.totals);
        hipLaunchKernelGGL(p2n_rev_fill_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, nbr, P, N, k1, w.off, w.deg,
                           w.rev);
        }
        const int V = (E + 63) / 64;
        if (V == 1) hipLaunchKernelGGL(p2n_smooth_pull_kernel<1>, dim3((unsigned)pb), dim3(256), 0, s, w.en, nbr, w.off, w.rev, w.rn, P, N, E, k1, smooth_scale, demb_smooth, w.spart);
        else if (V == 2) hipLaunchKernelGGL(p2n_smooth_pull_kernel<2>, dim3((unsigned)pb), dim3(256), 0, s, w.en, nbr, w.off, w.rev, w.rn, P, N, E, k1, smooth_scale, demb_smooth, w.spart);
        else if (V == 3) hipLaunchKernelGGL(p2n_smooth_pull_kernel<3>, dim3((unsigned)pb), dim3(256), 0, s, w.en, nbr, w.off, w.rev, w.rn, P, N, E, k1, smooth_scale, demb_smooth, w.spart);
        else hipLaunchKernelGGL(p2n_smooth_pull_kernel<4>, dim3((unsigned)pb), dim3(256), 0, s, w.en, nbr, w.off, w.rev, w.rn, P, N, E, k1, smooth_scale, demb_smooth, w.spart);
        hipLaunchKernelGGL(p2n_loss_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, s, w.pull_b, w.push_b, B, w.spart, (long)pb,
                           smooth_scale, losses);
    } else {
        P2N_CHECK_HIP(hipMemsetAsync(w.den, 0, sizeof(float) * P * E, s));
        smooth_launch(true, w.en, nbr, P, N, E, k1, smooth_scale, nullptr, w.den, (unsigned)pb, s);
        hipLaunchKernelGGL(p2n_normalize_bwd_kernel, dim3(pb), dim3(256), 0, s, w.den, w.en, w.rn, P, E, demb_smooth);
    }
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}


extern "C" int p2n_instance_seg_loss(const float *points, const float *emb, const int32_t *inst, const int32_t *cnt,
                                     const int32_t *bg, const int32_t *cstart, int B, int N, int E, int T,
                                     float delta_v, float delta_d, int k, float *losses, float *demb_pull,
                                     float *demb_push, float *demb_smooth, void *workspace, size_t workspace_bytes,
                                     void *stream_) {
    return loss_run(points, emb, inst, cnt, bg, cstart, B, N, E, T, delta_v, delta_d, k, nullptr, losses, demb_pull,
                    demb_push, demb_smooth, workspace, workspace_bytes, stream_);
}

// The same with the neighbour search done by the caller: knn_idx (B,N,k+1) = p2n_knn_self(points, k+1).  The search
// depends on the coordinates only, so a training loop can run it on a second stream beside the model's forward
// (pointcloudsegmentation_amd/losses.py: InstanceSegLoss.prepare) instead of between the forward and the backward.
extern "C" int p2n_instance_seg_loss_knn(const float *points, const float *emb, const int32_t *inst, const int32_t *cnt,
                                         const int32_t *bg, const int32_t *cstart, const int32_t *knn_idx, int B, int N,
                                         int E, int T, float delta_v, float delta_d, int k, float *losses,
                                         float *demb_pull, float *demb_push, float *demb_smooth, void *workspace,
                                         size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(knn_idx, "p2n_instance_seg_loss_knn: null knn_idx");
    return loss_run(points, emb, inst, cnt, bg, cstart, B, N, E, T, delta_v, delta_d, k, knn_idx, losses, demb_pull,
                    demb_push, demb_smooth, workspace, workspace_bytes, stream_);
}

// =========================================================================== label densification
// The loss kernels want every point's (cloud, label) pair as a dense id, ids ascending in (cloud, label) -- what
// reference losses.py:389-395 gets from torch.unique per cloud.  Through torch.unique on the device that is a radix
// sort + merge passes + scans: ~55 launches, 0.56 ms per step at 32 x 8192, all launch latency.  A cloud has tens of
// instances, so here one workgroup per cloud keeps its labels in registers and peels off the distinct values in
// ascending order: pass i counts the points equal to the i-th smallest label and finds the next larger one (one
// 64-bit min + one add reduction per pass).  Clouds with more than `max_per_cloud` distinct labels set the overflow
// flag and the caller takes another route.
#define DI_THREADS 1024
#define DI_PER_THREAD 16   // labels held in registers: N <= 16384; larger clouds re-read them from memory
#define DI_SPAN 8192        // label span (max - min + 1) the LDS histogram route covers

__device__ __forceinline__ long long di_block_min(long long v, long long *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long w = __shfl_xor(v, o); v = w < v ? w : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    long long m = sh[0];
#pragma unroll
    for (int w = 1; w < DI_THREADS / 64; ++w) m = sh[w] < m ? sh[w] : m;
    return m;
}
__device__ __forceinline__ int di_block_sum(int v, int *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int m = 0;
#pragma unroll
    for (int w = 0; w < DI_THREADS / 64; ++w) m += sh[w];
    return m;
}

// grid B: uniq[b][0..tcount[b]) ascending, ucnt[b][i] points per value
__global__ __launch_bounds__(DI_THREADS) void p2n_di_peel_kernel(const long long *__restrict__ labels, int N, int maxi,
                                                                long long *__restrict__ uniq, int *__restrict__ ucnt,
                                                                int *__restrict__ tcount, int *__restrict__ overflow) {
    __shared__ long long shm[DI_THREADS / 64];
    __shared__ int shs[DI_THREADS / 64];
    const int b = blockIdx.x, t = threadIdx.x;
    const long long *L = labels + (long)b * N;
    const long long NONE = 0x7fffffffffffffffLL;
    const bool inreg = N <= DI_THREADS * DI_PER_THREAD;
    long long r[DI_PER_THREAD];
    if (inreg) {
#pragma unroll
        for (int i = 0; i < DI_PER_THREAD; ++i) r[i] = t + i * DI_THREADS < N ? L[t + i * DI_THREADS] : NONE;
    }
    // first value: the minimum (a label equal to INT64_MAX would be taken for "none": rejected below as overflow)
    long long cur, top;
    {
        long long m = NONE, mx = -NONE;   // (a thread without labels keeps -INT64_MAX: its negation below is representable)
        if (inreg) {
#pragma unroll
            for (int i = 0; i < DI_PER_THREAD; ++i) {
                m = r[i] < m ? r[i] : m;
                mx = (r[i] != NONE && r[i] > mx) ? r[i] : mx;
            }
        } else {
            for (int i = t; i < N; i += DI_THREADS) { const long long v = L[i]; m = v < m ? v : m; mx = (v != NONE && v > mx) ? v : mx; }
        }
        cur = di_block_min(m, shm);
        top = -di_block_min(-mx, shm);   // an upper bound of the labels, exact unless every label is INT64_MIN
    }
    // Instance ids are small integers in practice (ScanNet: a few dozen per scene): when the cloud's labels span fewer
    // than DI_SPAN values, one LDS histogram and one block scan give the ascending distinct values and their counts --
    // the peeling loop below costs two block reductions per DISTINCT label (33 labels: ~120 us for 32 clouds) and stays
    // as the general route (any int64 values).
    if (cur != NONE && (unsigned long long)top - (unsigned long long)cur < (unsigned long long)DI_SPAN) {
        __shared__ int hist[DI_SPAN];
        __shared__ int wtot[DI_THREADS / 64];
        for (int i = t; i < DI_SPAN; i += DI_THREADS) hist[i] = 0;
        __syncthreads();
        int skipped = 0;   // labels equal to INT64_MAX (the sentinel) are not counted: reported as overflow like below
        if (inreg) {
#pragma unroll
            for (int i = 0; i < DI_PER_THREAD; ++i)
                if (t + i * DI_THREADS < N) { if (r[i] != NONE) atomicAdd(&hist[(int)(r[i] - cur)], 1); else ++skipped; }
        } else {
            for (int i = t; i < N; i += DI_THREADS) { const long long v = L[i]; if (v != NONE) atomicAdd(&hist[(int)(v - cur)], 1); else ++skipped; }
        }
        __syncthreads();
        constexpr int PER = DI_SPAN / DI_THREADS;   // consecutive bins per thread
        int nz = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) nz += hist[t * PER + i] != 0 ? 1 : 0;
        int incl = nz;   // inclusive scan inside the wave, then over the waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if ((t & 63) >= o) incl += v; }
        if ((t & 63) == 63) wtot[t >> 6] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < DI_THREADS / 64; ++w2) { before += w2 < (t >> 6) ? wtot[w2] : 0; total += wtot[w2]; }
        int pos = before + incl - nz;
        const int bad = __syncthreads_or(skipped != 0);
        if (total > maxi || bad) {
            if (t == 0) { *overflow = 1; tcount[b] = total < maxi ? total : maxi; }
            return;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = hist[t * PER + i];
            if (c) { uniq[(long)b * maxi + pos] = cur + t * PER + i; ucnt[(long)b * maxi + pos] = c; ++pos; }
        }
        if (t == 0) tcount[b] = total;
        return;
    }
    int n = 0;
    while (cur != NONE) {
        if (n >= maxi) { if (t == 0) *overflow = 1; break; }
        long long nx = NONE;
        int c = 0;
        if (inreg) {
#pragma unroll
            for (int i = 0; i < DI_PER_THREAD; ++i) {
                c += r[i] == cur ? 1 : 0;
                nx = (r[i] > cur && r[i] < nx) ? r[i] : nx;
            }
        } else {
            for (int i = t; i < N; i += DI_THREADS) {
                const long long v = L[i];
                c += v == cur ? 1 : 0;
                nx = (v > cur && v < nx) ? v : nx;
            }
        }
        c = di_block_sum(c, shs);
        nx = di_block_min(nx, shm);
        if (t == 0) { uniq[(long)b * maxi + n] = cur; ucnt[(long)b * maxi + n] = c; }
        ++n;
        cur = nx;
    }
    if (t == 0) {
        tcount[b] = n;
        int seen = 0;   // a label equal to INT64_MAX hides behind the sentinel: the counts then do not add up
        for (int i = 0; i < n; ++i) seen += ucnt[(long)b * maxi + i];
        if (seen != N && n < maxi) *overflow = 1;
    }
}

// one block: cstart = exclusive prefix of tcount; then cnt / bg compacted in (cloud, label) order
__global__ __launch_bounds__(256) void p2n_di_compact_kernel(const long long *__restrict__ uniq, const int *__restrict__ ucnt,
                                                            const int *__restrict__ tcount, int B, int maxi,
                                                            int *__restrict__ cstart, int *__restrict__ cnt, int *__restrict__ bg) {
    __shared__ int base_sh;
    if (threadIdx.x == 0) base_sh = 0;
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        const int base = base_sh, n = tcount[b];
        for (int i = threadIdx.x; i < n; i += 256) {
            cnt[base + i] = ucnt[(long)b * maxi + i];
            bg[base + i] = uniq[(long)b * maxi + i] == 0 ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) { cstart[b] = base; base_sh = base + n; }
        __syncthreads();
    }
    if (threadIdx.x == 0) cstart[B] = base_sh;
}

// per point: position of its label among its cloud's distinct values
__global__ __launch_bounds__(256) void p2n_di_rank_kernel(const long long *__restrict__ labels, int N, long total, int maxi,
                                                         const long long *__restrict__ uniq, const int *__restrict__ tcount,
                                                         const int *__restrict__ cstart, int *__restrict__ inst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    const long long v = labels[i];
    const long long *u = uniq + (long)b * maxi;
    int lo = 0, hi = tcount[b] - 1;
    while (lo < hi) {   // first position with u[pos] >= v (v is present unless the cloud overflowed)
        const int mid = (lo + hi) >> 1;
        if (u[mid] < v) lo = mid + 1; else hi = mid;
    }
    inst[i] = cstart[b] + lo;
}

extern "C" size_t p2n_dense_instances_workspace_bytes(int B, int max_per_cloud) {
    if (B < 1 || max_per_cloud < 1) return 0;
    return (size_t)B * max_per_cloud * (sizeof(long long) + sizeof(int)) + sizeof(int) * (size_t)B + 256;
}

// labels (B,N) int64 -> inst (B*N), cnt / bg (capacity B*max_per_cloud each; the first cstart[B] entries are written),
// cstart (B+1), overflow (1 int, set to 1 when a cloud has more than max_per_cloud distinct labels or a label equal
// to INT64_MAX: the outputs are then unusable).  All outputs on the device; the caller reads cstart[B] and overflow.
extern "C" int p2n_dense_instances(const long long *labels, int B, int N, int max_per_cloud, int32_t *inst, int32_t *cnt,
                                   int32_t *bg, int32_t *cstart, int32_t *overflow, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(B >= 1 && N >= 1 && max_per_cloud >= 1, "p2n_dense_instances: bad shape B=%d N=%d max=%d", B, N, max_per_cloud);
    P2N_REQUIRE(labels && inst && cnt && bg && cstart && overflow, "p2n_dense_instances: null pointer");
    const size_t need = p2n_dense_instances_workspace_bytes(B, max_per_cloud);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_dense_instances: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream_;
    long long *uniq = (long long *)workspace;
    int *ucnt = (int *)(uniq + (size_t)B * max_per_cloud);
    int *tcount = ucnt + (size_t)B * max_per_cloud;
    P2N_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int), s));
    hipLaunchKernelGGL(p2n_di_peel_kernel, dim3(B), dim3(DI_THREADS), 0, s, labels, N, max_per_cloud, uniq, ucnt, tcount, overflow);
    hipLaunchKernelGGL(p2n_di_compact_kernel, dim3(1), dim3(256), 0, s, uniq, ucnt, tcount, B, max_per_cloud, cstart, cnt, bg);
    const long total = (long)B * N;
    hipLaunchKernelGGL(p2n_di_rank_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, labels, N, total, max_per_cloud,
                       uniq, tcount, cstart, inst);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// out = g[0]*a + g[1]*b + g[2]*c, the three weights read on the device (the backward of the loss: upstream gradients
// of pull / push / smoothness times their stored embedding gradients, one pass instead of five)
__global__ __launch_bounds__(256) void p2n_combine3_kernel(const float *__restrict__ g, const float *__restrict__ a,
                                                          const float *__restrict__ b, const float *__restrict__ c,
                                                          long n4, float *__restrict__ out) {
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4 *>(a)[i], y = reinterpret_cast<const f32x4 *>(b)[i],
                    z = reinterpret_cast<const f32x4 *>(c)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (g0 * x[j] + g1 * y[j]) + g2 * z[j];   // torch's order: (g0*a + g1*b) + g2*c, no fma
        reinterpret_cast<f32x4 *>(out)[i] = o;
    }
}

extern "C" int p2n_combine3(const float *g3, const float *a, const float *b, const float *c, long n, float *out, void *stream_) {
    P2N_REQUIRE(g3 && a && b && c && out && n >= 0, "p2n_combine3: null pointer");
    P2N_REQUIRE(n % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0,
                "p2n_combine3: n must be a multiple of 4 and the arrays 16-byte aligned");
    if (n == 0) return P2N_OK;
    long nb = (n / 4 + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(p2n_combine3_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream_, g3, a, b, c, n / 4, out);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
