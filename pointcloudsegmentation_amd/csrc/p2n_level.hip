// Fused hierarchical levels for gfx950 (MI355X): one kernel per set-abstraction level and one per
// feature-propagation level.  BASELINE.json's north star names these levels; the reference model has
// none, so the formulas are this build's specification (parity unpinned):
//
//   set abstraction   out[b][m][co] = max_{s<32} MLP( [ feat[b][idx[b][m][s]] ; xyz[b][idx[b][m][s]] - new_xyz[b][m] ] )[co]
//   propagation       out[b][q][co] = MLP( [ sum_j w[b][q][j] * feat2[b][idx3[b][q][j]] ; skip[b][q] ] )[co]
//
// with MLP = [Conv(k=1) + BatchNorm(eval) + ReLU] x n (the last layer of the propagation chain may be a
// plain Conv: the embedding head rides on the last level).  The module's own channel order puts the
// coordinates / the skip features FIRST (canonical PointNet++); the packed weight image rotates the
// columns so that the gathered feature channels start at k = 0 and every lane fetches them as 16-byte
// pieces of point-major rows.
//
// Design (same MFMA orientation as p2n_fwd.hip: D = W.X, lane = column, exact-fp32
// v_mfma_f32_32x32x2_f32): a wave owns 32 columns -- the 32 neighbours of one centroid, or 32 points of a
// propagation level -- and carries them through the whole chain IN REGISTERS: the accumulator layout of a
// layer is the B-operand layout of the next.  The grouped tensor (B,3+C,M,ns), every intermediate layer
// output and the interpolated features never exist in memory: the layer-0 operand is gathered straight
// into registers (prefetched one k-segment ahead of the MFMAs that consume it), and a set-abstraction
// level leaves through a max over the wave's 32 lanes (DPP) as one point-major row (B,M,Cout).
// Weights: one linear stream of 1-KiB lane-linear fragments in consumption order.  Small chains keep the
// whole stream resident in LDS (loaded once per persistent workgroup, no barrier inside the loop); large
// ones stream it through two 32-KiB chunks by LDS-DMA exactly like the eval forward.
#include "p2n_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define CHUNK_FLOATS (P2N_CHUNK * P2N_FRAG_FLOATS)   // 8192 floats = 32 KiB
#define LV_MAX_LAYERS 4
#ifndef FP_SEG1
#define FP_SEG1 4   // operand groups fetched ahead per step, one wave per SIMD
#endif
#ifndef FP_SEG2
#define FP_SEG2 2   // ... two waves per SIMD (256 registers)
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// ------------------------------------------------------------------ geometry of a fused chain
// Padded geometry of one kernel instantiation.  Layer l consumes G[l] k-groups of 8 channels (G[0] = G0,
// G[l] = 4*CT[l-1]) and produces CT[l] tiles of 32 channels, P[l] tiles per pass; its fragments are
// ordered [pass][g][tile in pass].
struct LvShape {
    int nl, G0;
    int CT[LV_MAX_LAYERS], P[LV_MAX_LAYERS];
    int streamed;
};

struct LvLayout {
    int G[LV_MAX_LAYERS];
    long frag0[LV_MAX_LAYERS + 1];   // first fragment of each layer
    long stream_floats;              // fragments (streamed: padded to an even number of chunks)
    long scale[LV_MAX_LAYERS], shift[LV_MAX_LAYERS];   // float offsets from the image base
    long tab, tab_floats, total;
};

static inline LvLayout lv_layout(const LvShape &S) {
    LvLayout L;
    long f = 0;
    for (int l = 0; l < S.nl; ++l) {
        L.G[l] = l == 0 ? S.G0 : 4 * S.CT[l - 1];
        L.frag0[l] = f;
        f += (long)L.G[l] * S.CT[l];
    }
    L.frag0[S.nl] = f;
    if (S.streamed) {
        long nchunk = (f + P2N_CHUNK - 1) / P2N_CHUNK;
        nchunk = (nchunk + 1) / 2 * 2;
        L.stream_floats = nchunk * CHUNK_FLOATS;
    } else {
        L.stream_floats = f * P2N_FRAG_FLOATS;
    }
    long o = L.stream_floats;
    L.tab = o;
    for (int l = 0; l < S.nl; ++l) { L.scale[l] = o; o += 32 * S.CT[l]; L.shift[l] = o; o += 32 * S.CT[l]; }
    L.tab_floats = o - L.tab;
    L.total = o;
    return L;
}

// ------------------------------------------------------------------ pack
struct LvPackArgs {
    LvShape S;
    LvLayout L;
    int cin[LV_MAX_LAYERS], cout[LV_MAX_LAYERS], bn[LV_MAX_LAYERS];
    int rot;                  // layer 0: packed column k reads module column (k + rot) % cin[0]
    double eps;
    const float *W[LV_MAX_LAYERS], *b[LV_MAX_LAYERS], *gamma[LV_MAX_LAYERS], *beta[LV_MAX_LAYERS],
        *mean[LV_MAX_LAYERS], *var[LV_MAX_LAYERS];
};

__global__ void p2n_level_pack_kernel(LvPackArgs A, float *__restrict__ packed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.L.total) return;
    float v = 0.f;
    if (i >= A.L.tab) {
        long r = i - A.L.tab;
        for (int l = 0; l < A.S.nl; ++l) {
            const int cp = 32 * A.S.CT[l];
            if (r < 2L * cp) {
                const int c = (int)(r % cp);
                const bool is_shift = r >= cp;
                if (c < A.cout[l]) {
                    const double bias = A.b[l][c];
                    if (A.bn[l]) {   // BatchNorm(eval) folded in double, as the eval forward's pack does
                        const double s = (double)A.gamma[l][c] / sqrt((double)A.var[l][c] + A.eps);
                        const double t = (bias - (double)A.mean[l][c]) * s + (double)A.beta[l][c];
                        v = (float)(is_shift ? t : s);
                    } else {
                        v = is_shift ? (float)bias : 1.f;
                    }
                }
                break;
            }
            r -= 2L * cp;
        }
    } else {
        const long f = i / P2N_FRAG_FLOATS;
        const int e = (int)(i % P2N_FRAG_FLOATS);
        const int lane = e >> 2, j = e & 3;
        if (f < A.L.frag0[A.S.nl]) {
            int l = 0;
            while (f >= A.L.frag0[l + 1]) ++l;
            const long r = f - A.L.frag0[l];
            const int P = A.S.P[l], per = A.L.G[l] * P;
            const int pass = (int)(r / per), r2 = (int)(r % per);
            const int g = r2 / P, ct = pass * P + r2 % P;
            const int row = 32 * ct + (lane & 31), k = 8 * g + 4 * (lane >> 5) + j;
            if (row < A.cout[l] && k < A.cin[l]) {
                const int col = (l == 0) ? (k + A.rot) % A.cin[0] : k;
                v = A.W[l][(long)row * A.cin[l] + col];
            }
        }
    }
    packed[i] = v;
}

// Channel-split images of a two-layer propagation chain, appended to its ordinary image.  A level with few
// columns (fewer 128-point blocks than CUs) runs layer by layer instead: grid = (column blocks, pairs of output
// tiles), so that every CU gets a workgroup; the layer-0 operand is then fetched once per tile pair.  Sub-image
// (layer l, pair y) = the fragments of output tiles 2y and 2y+1 in [g][tile] order, padded to an even number of
// chunks, followed by 64 scale and 64 shift values.
struct LvSplit {
    long base[2];      // float offset of layer l's first sub-image from the image base
    long stride[2];    // floats per sub-image
    int pairs[2];      // sub-images of layer l
    long total;        // floats of the whole image including the split part
};

static inline LvSplit lv_split_layout(const LvShape &S, const LvLayout &L) {
    LvSplit X;
    long o = L.total;
    for (int l = 0; l < 2; ++l) {
        long nchunk = ((long)L.G[l] * 2 + P2N_CHUNK - 1) / P2N_CHUNK;
        nchunk = (nchunk + 1) / 2 * 2;
        X.base[l] = o;
        X.stride[l] = nchunk * CHUNK_FLOATS + 128;
        X.pairs[l] = S.CT[l] / 2;
        o += X.stride[l] * X.pairs[l];
    }
    X.total = o;
    return X;
}

__global__ void p2n_level_pack_split_kernel(LvPackArgs A, LvSplit X, float *__restrict__ packed) {
    const long i = A.L.total + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= X.total) return;
    const int l = i >= X.base[1] ? 1 : 0;
    const long r = i - X.base[l];
    const int y = (int)(r / X.stride[l]);
    const long q = r % X.stride[l];
    const long stream_floats = X.stride[l] - 128;
    float v = 0.f;
    if (q >= stream_floats) {
        const int t = (int)(q - stream_floats);
        const int c = 64 * y + (t & 63);
        const bool is_shift = t >= 64;
        if (c < A.cout[l]) {
            const double bias = A.b[l][c];
            const double s = (double)A.gamma[l][c] / sqrt((double)A.var[l][c] + A.eps);
            const double sh = (bias - (double)A.mean[l][c]) * s + (double)A.beta[l][c];
            v = (float)(is_shift ? sh : s);
        }
    } else {
        const long f = q / P2N_FRAG_FLOATS;
        const int e = (int)(q % P2N_FRAG_FLOATS);
        const int lane = e >> 2, j = e & 3;
        if (f < (long)A.L.G[l] * 2) {
            const int g = (int)(f >> 1), ct = 2 * y + (int)(f & 1);
            const int row = 32 * ct + (lane & 31), k = 8 * g + 4 * (lane >> 5) + j;
            if (row < A.cout[l] && k < A.cin[l]) {
                const int col = (l == 0) ? (k + A.rot) % A.cin[0] : k;
                v = A.W[l][(long)row * A.cin[l] + col];
            }
        }
    }
    packed[i] = v;
}

// ------------------------------------------------------------------ device helpers
// The weight stream as a wave sees it.  STREAMED: two 32-KiB chunks in LDS; entering the first fragment
// of chunk c every wave waits for it (vmcnt(0) + barrier, which also proves that the other buffer is no
// longer read) and starts the LDS-DMA of chunk c+1 -- chunk 0 again after the last one: the stream has an
// even number of chunks, so the next pass over it finds chunk 0 already resident in buffer 0.
// RESIDENT: the whole stream sits in LDS.
template <bool STREAMED, int NCHUNK>
struct WStream {
    const float *stream;
    float *wbuf;
    int wave, lane;
    __device__ __forceinline__ void stage(int c) const {
        // opaque chunk number: otherwise the 8 x NCHUNK global addresses of a stream are hoisted out of the
        // block loop as loop invariants and spill
        asm volatile("" : "+s"(c));
        // buffer LDS-DMA: stream offset on the scalar unit, one constant lane offset -- no vector-ALU address
        // arithmetic, whose time would add to the MFMAs' (DESIGN_HISTORY section 7)
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(stream), 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < P2N_CHUNK / P2N_WAVES_PER_WG; ++i) {
            const int f = wv * (P2N_CHUNK / P2N_WAVES_PER_WG) + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(wbuf + (c & 1) * CHUNK_FLOATS + f * P2N_FRAG_FLOATS), 16,
                                                     lane * 16, (c * P2N_CHUNK + f) * (P2N_FRAG_FLOATS * 4), 0, 0);
        }
    }
    __device__ __forceinline__ void load_resident(int nfrag) const {
        for (int f = wave; f < nfrag; f += P2N_WAVES_PER_WG)
            __builtin_amdgcn_global_load_lds((gbl_void *)(stream + (long)f * P2N_FRAG_FLOATS + lane * 4),
                                             (lds_void *)(wbuf + f * P2N_FRAG_FLOATS), 16, 0, 0);
    }
    __device__ __forceinline__ f32x4 frag(int f) const {
        if (STREAMED) {
            if (f % P2N_CHUNK == 0) {
                p2n_dma_barrier();
                stage((f / P2N_CHUNK + 1) % NCHUNK);
            }
            return *reinterpret_cast<const f32x4 *>(wbuf + ((f / P2N_CHUNK) & 1) * CHUNK_FLOATS +
                                                    (f % P2N_CHUNK) * P2N_FRAG_FLOATS + lane * 4);
        }
        return *reinterpret_cast<const f32x4 *>(wbuf + f * P2N_FRAG_FLOATS + lane * 4);
    }
};

template <bool RELU>
__device__ __forceinline__ f32x16 lv_epilogue(const f32x16 acc, const float *scale, const float *shift, int ct, int h) {
    f32x16 out;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 32 * ct + 8 * g + 4 * h);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 32 * ct + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = fmaf(acc[4 * g + j], sc[j], sh[j]);
            out[4 * g + j] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
    return out;
}

template <int CT>
__device__ __forceinline__ void lv_zero(f32x16 (&t)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) t[c][i] = 0.f;
}

// One pass of a layer whose input lives in tile registers: P output tiles over all KT input tiles.
// Fragments [g][tile in pass] from stream index f0.
template <int KT, int P, class WS>
__device__ __forceinline__ void lv_pass(const WS &ws, int f0, const f32x16 (&in)[KT], f32x16 (&acc)[P]) {
    // the fragment of step i+1 is read from LDS before the MFMAs of step i issue: a wave alone on its SIMD
    // otherwise waits out the LDS latency in front of every group of four MFMAs
    f32x4 a = ws.frag(f0);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const int i = (kt * 4 + g) * P + c;
                f32x4 an = a;
                if (i + 1 < KT * 4 * P) an = ws.frag(f0 + i + 1);
                __builtin_amdgcn_sched_barrier(0);   // or the scheduler sinks the read back to its first use
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[c] = MFMA32(a[j], in[kt][4 * g + j], acc[c]);
                a = an;
            }
}

// Layer 0: the operand comes from a loader in segments of LV_SEG k-groups (fetched ahead per step); segment s+1 is issued before
// the MFMAs of segment s and finished (arithmetic on the loaded values) after them.  Segment 0 arrives
// already issued in `raw` (lv_issue0): the set-abstraction kernel issues it one block ahead.
template <int G0, int LV_SEG, class Loader>
__device__ __forceinline__ void lv_issue0(const Loader &ld, typename Loader::Raw (&raw)[LV_SEG]) {
#pragma unroll
    for (int q = 0; q < LV_SEG; ++q)
        if (q < G0) ld.issue(q, raw[q]);
}

template <int G0, int CT, int LV_SEG, class WS, class Loader>
__device__ __forceinline__ void lv_layer0(const WS &ws, const Loader &ld, typename Loader::Raw (&raw)[LV_SEG],
                                          f32x16 (&acc)[CT]) {
    constexpr int NSEG = (G0 + LV_SEG - 1) / LV_SEG;
    f32x4 cur[LV_SEG];
#pragma unroll
    for (int q = 0; q < LV_SEG; ++q)
        if (q < G0) cur[q] = ld.finish(q, raw[q]);
    f32x4 a = ws.frag(0);
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        if (s + 1 < NSEG) {
#pragma unroll
            for (int q = 0; q < LV_SEG; ++q)
                if ((s + 1) * LV_SEG + q < G0) ld.issue((s + 1) * LV_SEG + q, raw[q]);
        }
#pragma unroll
        for (int q = 0; q < LV_SEG; ++q) {
            const int g = s * LV_SEG + q;
            if (g < G0) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    f32x4 an = a;          // next step's fragment, read ahead of this step's MFMAs
                    if (g * CT + c + 1 < G0 * CT) an = ws.frag(g * CT + c + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[c] = MFMA32(a[j], cur[q][j], acc[c]);
                    a = an;
                }
            }
        }
        if (s + 1 < NSEG) {
#pragma unroll
            for (int q = 0; q < LV_SEG; ++q)
                if ((s + 1) * LV_SEG + q < G0) cur[q] = ld.finish((s + 1) * LV_SEG + q, raw[q]);
        }
    }
}

// (p2n_tile_half_max: p2n_common.h)
__device__ __forceinline__ float lv_tile_half_max(const f32x16 v, int lane) { return p2n_tile_half_max(v, lane); }

__device__ __forceinline__ f32x4 lv_ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// ------------------------------------------------------------------ set-abstraction level
struct SaArgs {
    const float *xyz;      // rows of ldx floats, first three = coordinates: point (b,n) at (b*N+n)*ldx
    const float *new_xyz;  // (B,M,3)
    const float *feat;     // rows of ldf floats, first C = features (point-major), or NULL with C = 0
    const int32_t *idx;    // (B,M,32)
    const float *packed;
    float *out;            // (B,M,Cout) point-major
    long total;            // B*M centroids
    long tab;
    int ldx, ldf, C, N, M, Cout, vec;
};

// Layer-0 operand of a set-abstraction level: k < C the neighbour's features, C <= k < C+3 its
// coordinates relative to the centroid, zero beyond.  issue() only loads; the centroid is subtracted in
// finish(), so a block's loads can be issued a whole block ahead without anything waiting on them.
// Exact fit (C == 8*GF, 16-byte aligned rows): groups 0..GF-1 are one 16-byte load per lane, group GF holds
// the coordinates in the lower wave half.  Any other C takes the per-element path.
template <int GF>
struct SaLoader {
    typedef f32x4 Raw;
    const float *row;      // this lane's neighbour's feature row (exact: advanced by the lane's half, 4*h floats)
    const float *xr;       // this lane's neighbour's coordinates
    f32x4 cen;             // the centroid (x,y,z,0)
    int C, h, exact;
    __device__ __forceinline__ float pick(int k) const {
        if (k < C) return row[k];
        const int r = k - C;
        return r < 3 ? xr[r] : 0.f;
    }
    __device__ __forceinline__ void issue(int g, f32x4 &raw) const {
        if (exact) {
            if (g < GF) raw = lv_ld4(row + 8 * g);
            else if (g == GF) raw = f32x4{xr[0], xr[1], xr[2], 0.f};
            else raw = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const int k0 = 8 * g + 4 * h;
            raw = f32x4{pick(k0), pick(k0 + 1), pick(k0 + 2), pick(k0 + 3)};
        }
    }
    __device__ __forceinline__ f32x4 finish(int g, const f32x4 &raw) const {
        if (exact) {
            if (g != GF) return raw;
            const f32x4 d = raw - cen;
            return h == 0 ? d : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int r0 = 8 * g + 4 * h - C;          // position of the quad's first element in the coordinate triple
        f32x4 o = raw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + j;
            const float cv = r == 0 ? cen[0] : r == 1 ? cen[1] : cen[2];
            if (r >= 0 && r < 3) o[j] = raw[j] - cv;
        }
        return o;
    }
};

template <class Loader>
__device__ __forceinline__ void sa_make_loader(const SaArgs &A, int C, long c, int nb, int h, Loader &ld) {
    const long b = c / A.M;
    nb = nb < 0 ? 0 : (nb >= A.N ? A.N - 1 : nb);                // indices are clamped, not trusted
    const float *cr = A.new_xyz + c * 3;
    ld.xr = A.xyz + (b * A.N + nb) * A.ldx;
    ld.row = A.feat + (b * A.N + nb) * A.ldf + (A.vec ? 4 * h : 0);
    ld.cen = f32x4{cr[0], cr[1], cr[2], 0.f};
    ld.C = C; ld.h = h; ld.exact = A.vec;
}

template <int G0, int GF, int CT1, int CT2, int CT3, int P3, bool STREAMED, bool PIPE>
__global__ __launch_bounds__(256, 2) void p2n_sa_level_kernel(SaArgs A) {
    constexpr int F2 = G0 * CT1, F3 = F2 + 4 * CT1 * CT2, NFRAG = F3 + 4 * CT2 * CT3;
    constexpr int NCHUNK = ((NFRAG + P2N_CHUNK - 1) / P2N_CHUNK + 1) / 2 * 2;
    constexpr int WFLOATS = STREAMED ? 2 * CHUNK_FLOATS : NFRAG * P2N_FRAG_FLOATS;
    constexpr int S1 = 0, H1 = 32 * CT1, S2 = 64 * CT1, H2 = S2 + 32 * CT2, S3 = 64 * (CT1 + CT2), H3 = S3 + 32 * CT3;
    constexpr int TABN = 64 * (CT1 + CT2 + CT3);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tab = lds + WFLOATS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const WStream<STREAMED, NCHUNK> ws{A.packed, lds, wave, lane};
    if (STREAMED) ws.stage(0);
    else ws.load_resident(NFRAG);
    for (int i = threadIdx.x; i < TABN; i += 256) tab[i] = A.packed[A.tab + i];

    // PIPE: software pipeline over the workgroup's blocks (4 centroids each): the neighbour index is fetched
    // two blocks ahead, the first operand segment and the centroid one block ahead (the small levels, whose
    // block is shorter than two dependent memory round trips).  Otherwise a block loads its own operands.
    const long nblk = (A.total + P2N_WAVES_PER_WG - 1) / P2N_WAVES_PER_WG;
    const long last = A.total - 1;
    long blk = blockIdx.x;
    long cent = blk * P2N_WAVES_PER_WG + wave;
    bool valid = cent < A.total;
    long c = valid ? cent : last;
    constexpr int SEG = 4;
    SaLoader<GF> ld;
    f32x4 raw[SEG];
    int C = A.C;
    int nb_next = 0;
    if (PIPE && blk < nblk) {
        sa_make_loader(A, C, c, A.idx[c * 32 + p], h, ld);
        lv_issue0<G0, SEG>(ld, raw);
        if (blk + gridDim.x < nblk) {
            const long c2 = (blk + gridDim.x) * P2N_WAVES_PER_WG + wave;
            nb_next = A.idx[(c2 < A.total ? c2 : last) * 32 + p];
        }
    }
    if (!STREAMED) p2n_dma_barrier();      // (streamed: the first chunk gate is the barrier)

    for (; blk < nblk; blk += gridDim.x) {
        asm volatile("" ::: "memory");   // keeps the LDS weight reads inside the iteration (hoisted, they spill)
        asm volatile("" : "+s"(C));      // ... and the per-element path's channel predicates
        if (!PIPE) {
            cent = blk * P2N_WAVES_PER_WG + wave;
            valid = cent < A.total;
            c = valid ? cent : last;
            sa_make_loader(A, C, c, A.idx[c * 32 + p], h, ld);
            lv_issue0<G0, SEG>(ld, raw);
        }
        f32x16 a1[CT1];
        lv_zero(a1);
        lv_layer0<G0, CT1, SEG>(ws, ld, raw, a1);
        // the next block's loads go out here, behind this block's layer-0 MFMAs
        const bool valid_cur = valid;
        const long c_cur = c;
        const long nxt = blk + gridDim.x;
        if (PIPE && nxt < nblk) {
            cent = nxt * P2N_WAVES_PER_WG + wave;
            valid = cent < A.total;
            c = valid ? cent : last;
            sa_make_loader(A, C, c, nb_next, h, ld);
            lv_issue0<G0, SEG>(ld, raw);
            if (nxt + gridDim.x < nblk) {
                const long c2 = (nxt + gridDim.x) * P2N_WAVES_PER_WG + wave;
                nb_next = A.idx[(c2 < A.total ? c2 : last) * 32 + p];
            }
        }
#pragma unroll
        for (int ct = 0; ct < CT1; ++ct) a1[ct] = lv_epilogue<true>(a1[ct], tab + S1, tab + H1, ct, h);
        f32x16 a2[CT2];
        lv_zero(a2);
        lv_pass<CT1, CT2>(ws, F2, a1, a2);
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct) a2[ct] = lv_epilogue<true>(a2[ct], tab + S2, tab + H2, ct, h);
        float *orow = A.out + c_cur * A.Cout;
#pragma unroll
        for (int q = 0; q < CT3 / P3; ++q) {
            f32x16 a3[P3];
            lv_zero(a3);
            lv_pass<CT2, P3>(ws, F3 + q * (4 * CT2 * P3), a2, a3);
#pragma unroll
            for (int cc = 0; cc < P3; ++cc) {
                const int ct = q * P3 + cc;
                const f32x16 v = lv_epilogue<true>(a3[cc], tab + S3, tab + H3, ct, h);
                const float mx = lv_tile_half_max(v, lane);
                // lanes 0-15 of each half hold 16 of the tile's 32 channels: together one 128-byte line
                const int ch = 32 * ct + 8 * ((lane & 15) >> 2) + 4 * h + (lane & 3);
                if (valid_cur && (lane & 16) == 0 && ch < A.Cout) orow[ch] = mx;
            }
        }
    }
    if (STREAMED) p2n_dma_barrier();   // drain the chunk staged ahead before the workgroup's LDS is released
}

// ------------------------------------------------------------------ feature-propagation level
struct FpArgs {
    const int32_t *idx3;   // (B,n,3) three nearest coarse points
    const float *w3;       // (B,n,3) inverse-distance weights
    const float *feat2;    // (B,m,C2) coarse features, point-major, C2 % 4 == 0
    const float *skip;     // rows of lds floats, first C1 = skip features of the fine points, or NULL with C1 = 0
    const float *packed;
    float *out;            // (B,n,Cout) point-major
    long total;            // B*n points
    long tab;
    int lds, C1, C2, n, m, Cout, vec;
};

struct FpRaw { f32x4 a, b, c; };

// k-groups 0..GI-1 are the interpolated coarse channels (exactly C2 = 8*GI of them), the groups after
// them the skip channels (C1 <= 8*(G0-GI), zero weights beyond C1)
template <int GI>
struct FpLoader {
    typedef FpRaw Raw;
    const float *r0, *r1, *r2;   // the three coarse rows, already advanced by this lane's half (4*h floats)
    const float *srow;           // skip row
    float w0, w1, w2;
    int C1, h, vec;
    __device__ __forceinline__ float pick(int k) const { return k < C1 ? srow[k] : 0.f; }
    __device__ __forceinline__ void issue(int g, FpRaw &raw) const {
        if (g < GI) {
            raw.a = lv_ld4(r0 + 8 * g); raw.b = lv_ld4(r1 + 8 * g); raw.c = lv_ld4(r2 + 8 * g);
        } else {
            const int s0 = 8 * (g - GI) + 4 * h;
            if (vec) raw.a = lv_ld4(srow + s0);
            else raw.a = f32x4{pick(s0), pick(s0 + 1), pick(s0 + 2), pick(s0 + 3)};
        }
    }
    __device__ __forceinline__ f32x4 finish(int g, const FpRaw &raw) const {
        if (g >= GI) return raw.a;
        f32x4 o;   // (w0*f0 + w1*f1) + w2*f2, rounded like p2n_three_interpolate
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = __fadd_rn(__fadd_rn(__fmul_rn(w0, raw.a[j]), __fmul_rn(w1, raw.b[j])), __fmul_rn(w2, raw.c[j]));
        return o;
    }
};

template <class Args, class Loader>
__device__ __forceinline__ void fp_make_loader(const Args &A, long col, int h, Loader &ld) {
    const long b = col / A.n;
    const int32_t *ip = A.idx3 + col * 3;
    const float *wp = A.w3 + col * 3;
    int i0 = ip[0], i1 = ip[1], i2 = ip[2];
    i0 = i0 < 0 ? 0 : (i0 >= A.m ? A.m - 1 : i0);
    i1 = i1 < 0 ? 0 : (i1 >= A.m ? A.m - 1 : i1);
    i2 = i2 < 0 ? 0 : (i2 >= A.m ? A.m - 1 : i2);
    ld.r0 = A.feat2 + (b * A.m + i0) * A.C2 + 4 * h;
    ld.r1 = A.feat2 + (b * A.m + i1) * A.C2 + 4 * h;
    ld.r2 = A.feat2 + (b * A.m + i2) * A.C2 + 4 * h;
    ld.w0 = wp[0]; ld.w1 = wp[1]; ld.w2 = wp[2];
    ld.srow = A.skip + col * A.lds;
    ld.C1 = A.C1; ld.h = h; ld.vec = A.vec;
}

template <bool RELU>
__device__ __forceinline__ void fp_store_tile(float *orow, int Cout, int ct, int h, const f32x16 &v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = 32 * ct + 8 * g + 4 * h;
        if ((Cout & 3) == 0) {
            if (ch < Cout) *reinterpret_cast<f32x4 *>(orow + ch) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ch + j < Cout) orow[ch + j] = v[4 * g + j];
        }
    }
}

// two layers (both Conv+BN+ReLU); the second in passes of P2 tiles, stored as they finish
template <int G0, int GI, int CT1, int CT2, int P2, int MINW, int SEG>
__global__ __launch_bounds__(256, MINW) void p2n_fp2_level_kernel(FpArgs A) {
    constexpr int F2 = G0 * CT1, NFRAG = F2 + 4 * CT1 * CT2;
    constexpr int NCHUNK = ((NFRAG + P2N_CHUNK - 1) / P2N_CHUNK + 1) / 2 * 2;
    constexpr int S1 = 0, H1 = 32 * CT1, S2 = 64 * CT1, H2 = S2 + 32 * CT2, TABN = 64 * (CT1 + CT2);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tab = lds + 2 * CHUNK_FLOATS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const WStream<true, NCHUNK> ws{A.packed, lds, wave, lane};
    ws.stage(0);
    for (int i = threadIdx.x; i < TABN; i += 256) tab[i] = A.packed[A.tab + i];
    const long nblk = (A.total + P2N_PTS_PER_WG - 1) / P2N_PTS_PER_WG;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long col0 = (blk * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE + p;
        const bool valid = col0 < A.total;
        const long col = valid ? col0 : A.total - 1;
        FpLoader<GI> ld;
        fp_make_loader(A, col, h, ld);
        FpRaw raw[SEG];
        lv_issue0<G0, SEG>(ld, raw);
        f32x16 a1[CT1];
        lv_zero(a1);
        lv_layer0<G0, CT1, SEG>(ws, ld, raw, a1);
#pragma unroll
        for (int ct = 0; ct < CT1; ++ct) a1[ct] = lv_epilogue<true>(a1[ct], tab + S1, tab + H1, ct, h);
        float *orow = A.out + col * A.Cout;
#pragma unroll
        for (int q = 0; q < CT2 / P2; ++q) {
            f32x16 a2[P2];
            lv_zero(a2);
            lv_pass<CT1, P2>(ws, F2 + q * (4 * CT1 * P2), a1, a2);
#pragma unroll
            for (int cc = 0; cc < P2; ++cc) {
                const f32x16 v = lv_epilogue<true>(a2[cc], tab + S2, tab + H2, q * P2 + cc, h);
                if (valid) fp_store_tile<true>(orow, A.Cout, q * P2 + cc, h, v);
            }
        }
    }
    p2n_dma_barrier();
}

// one layer of a channel-split level: grid.y = pair of output tiles (64 channels), own sub-image
template <int G0, int GI, int SEG>
__global__ __launch_bounds__(256, 2) void p2n_fp1_level_kernel(FpArgs A, long img_stride) {
    constexpr int NFRAG = G0 * 2;
    constexpr int NCHUNK = ((NFRAG + P2N_CHUNK - 1) / P2N_CHUNK + 1) / 2 * 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tab = lds + 2 * CHUNK_FLOATS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const float *img = A.packed + (long)blockIdx.y * img_stride;
    const WStream<true, NCHUNK> ws{img, lds, wave, lane};
    ws.stage(0);
    if (threadIdx.x < 128) tab[threadIdx.x] = img[(long)NCHUNK * CHUNK_FLOATS + threadIdx.x];
    const long nblk = (A.total + P2N_PTS_PER_WG - 1) / P2N_PTS_PER_WG;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long col0 = (blk * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE + p;
        const bool valid = col0 < A.total;
        const long col = valid ? col0 : A.total - 1;
        FpLoader<GI> ld;
        fp_make_loader(A, col, h, ld);
        FpRaw raw[SEG];
        lv_issue0<G0, SEG>(ld, raw);
        f32x16 a1[2];
        lv_zero(a1);
        lv_layer0<G0, 2, SEG>(ws, ld, raw, a1);
        float *orow = A.out + col * A.Cout + 64 * blockIdx.y;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x16 v = lv_epilogue<true>(a1[ct], tab, tab + 64, ct, h);
            if (valid) fp_store_tile<true>(orow, A.Cout - 64 * (int)blockIdx.y, ct, h, v);
        }
    }
    p2n_dma_barrier();
}

// four layers: the last propagation level with the embedding head on top (Conv+BN+ReLU x3, plain Conv)
template <int G0, int GI, int CT1, int CT2, int CT3, int CT4, int SEG>
__global__ __launch_bounds__(256, 2) void p2n_fp4_level_kernel(FpArgs A) {
    constexpr int F2 = G0 * CT1, F3 = F2 + 4 * CT1 * CT2, F4 = F3 + 4 * CT2 * CT3, NFRAG = F4 + 4 * CT3 * CT4;
    constexpr int NCHUNK = ((NFRAG + P2N_CHUNK - 1) / P2N_CHUNK + 1) / 2 * 2;
    constexpr int S1 = 0, H1 = 32 * CT1, S2 = 64 * CT1, H2 = S2 + 32 * CT2, S3 = 64 * (CT1 + CT2), H3 = S3 + 32 * CT3,
                  S4 = 64 * (CT1 + CT2 + CT3), H4 = S4 + 32 * CT4, TABN = 64 * (CT1 + CT2 + CT3 + CT4);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tab = lds + 2 * CHUNK_FLOATS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const WStream<true, NCHUNK> ws{A.packed, lds, wave, lane};
    ws.stage(0);
    for (int i = threadIdx.x; i < TABN; i += 256) tab[i] = A.packed[A.tab + i];
    const long nblk = (A.total + P2N_PTS_PER_WG - 1) / P2N_PTS_PER_WG;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long col0 = (blk * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE + p;
        const bool valid = col0 < A.total;
        const long col = valid ? col0 : A.total - 1;
        FpLoader<GI> ld;
        fp_make_loader(A, col, h, ld);
        FpRaw raw[SEG];
        lv_issue0<G0, SEG>(ld, raw);
        f32x16 a1[CT1];
        lv_zero(a1);
        lv_layer0<G0, CT1, SEG>(ws, ld, raw, a1);
#pragma unroll
        for (int ct = 0; ct < CT1; ++ct) a1[ct] = lv_epilogue<true>(a1[ct], tab + S1, tab + H1, ct, h);
        f32x16 a2[CT2];
        lv_zero(a2);
        lv_pass<CT1, CT2>(ws, F2, a1, a2);
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct) a2[ct] = lv_epilogue<true>(a2[ct], tab + S2, tab + H2, ct, h);
        f32x16 a3[CT3];
        lv_zero(a3);
        lv_pass<CT2, CT3>(ws, F3, a2, a3);
#pragma unroll
        for (int ct = 0; ct < CT3; ++ct) a3[ct] = lv_epilogue<true>(a3[ct], tab + S3, tab + H3, ct, h);
        f32x16 a4[CT4];
        lv_zero(a4);
        lv_pass<CT3, CT4>(ws, F4, a3, a4);
        float *orow = A.out + col * A.Cout;
#pragma unroll
        for (int ct = 0; ct < CT4; ++ct) {
            const f32x16 v = lv_epilogue<false>(a4[ct], tab + S4, tab + H4, ct, h);
            if (valid) fp_store_tile<false>(orow, A.Cout, ct, h, v);
        }
    }
    p2n_dma_barrier();
}

// ------------------------------------------------------------------ centroid gather
// new_xyz[b][m] = xyz[b][idx[b][m]] (the centroids FPS picked)
__global__ __launch_bounds__(256) void p2n_gather_xyz_kernel(const float *__restrict__ xyz, int ldx,
                                                            const int32_t *__restrict__ idx, long total, int N, int M,
                                                            float *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / M;
    int n = idx[i];
    n = n < 0 ? 0 : (n >= N ? N - 1 : n);
    const float *s = xyz + (b * N + n) * ldx;
    out[i * 3] = s[0]; out[i * 3 + 1] = s[1]; out[i * 3 + 2] = s[2];
}

extern "C" int p2n_gather_xyz(const float *xyz, int ldx, const int32_t *idx, int B, int N, int M, float *out,
                              void *stream) {
    P2N_REQUIRE(B >= 0 && N >= 1 && M >= 0 && ldx >= 3, "p2n_gather_xyz: bad shape");
    const long total = (long)B * M;
    if (total == 0) return P2N_OK;
    P2N_REQUIRE(xyz && idx && out, "p2n_gather_xyz: null pointer");
    const int ps = p2n_prof_begin(P2N_K_GATHER_XYZ, (hipStream_t)stream);
    hipLaunchKernelGGL(p2n_gather_xyz_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       xyz, ldx, idx, total, N, M, out);
    p2n_prof_end(ps, (hipStream_t)stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// ------------------------------------------------------------------ host: instantiations and dispatch
// kind 0 = set abstraction (3 layers), 1 = propagation (2 layers), 2 = propagation + head (4 layers).
// A chain runs on the smallest instantiation its channel counts fit into (zero padding is exact).
struct LvInst {
    int kind;
    int gi;        // propagation: k-groups of interpolated channels (C2 must equal 8*gi); 0 for set abstraction
    LvShape S;
};

static const LvInst g_inst[] = {
    {0, 0, {3, 1, {1, 1, 2, 0}, {1, 1, 2, 0}, 0}},     // SA: K <= 8,   mlp <= (32, 32, 64)    resident (13 KiB)
    {0, 0, {3, 9, {2, 2, 4, 0}, {2, 2, 4, 0}, 0}},     // SA: K <= 72,  mlp <= (64, 64, 128)   resident (66 KiB)
    {0, 0, {3, 17, {4, 4, 8, 0}, {4, 4, 2, 0}, 1}},    // SA: K <= 136, mlp <= (128, 128, 256) streamed
    {1, 32, {2, 40, {8, 4, 0, 0}, {8, 4, 0, 0}, 1}},   // FP: C2 = 256, C1 <= 64,  mlp <= (256, 128)
    {1, 32, {2, 48, {8, 8, 0, 0}, {8, 2, 0, 0}, 1}},   // FP: C2 = 256, C1 <= 128, mlp <= (256, 256)
    {2, 16, {4, 17, {4, 4, 4, 2}, {4, 4, 4, 2}, 1}},   // FP + head: C2 = 128, C1 <= 8, mlp <= (128, 128, 128, 64)
};
#define LV_NINST ((int)(sizeof(g_inst) / sizeof(g_inst[0])))

// smallest instantiation the chain fits into; c2 = interpolated channels of a propagation chain (-1: SA)
static int lv_find(int kind, int nl, const int *dims, int c2) {
    for (int i = 0; i < LV_NINST; ++i) {
        const LvShape &S = g_inst[i].S;
        if (g_inst[i].kind != kind || S.nl != nl) continue;
        bool ok = dims[0] >= 1 && dims[0] <= 8 * S.G0;
        if (kind != 0) ok = ok && c2 == 8 * g_inst[i].gi && dims[0] - c2 >= 0 && dims[0] - c2 <= 8 * (S.G0 - g_inst[i].gi);
        for (int l = 0; l < nl && ok; ++l) ok = dims[l + 1] >= 1 && dims[l + 1] <= 32 * S.CT[l];
        if (ok) return i;
    }
    return -1;
}

static int lv_kind_layers(int kind) { return kind == 0 ? 3 : kind == 1 ? 2 : kind == 2 ? 4 : -1; }

extern "C" long p2n_level_packed_floats(int kind, int nl, const int *dims, int rot) {
    if (!dims || nl != lv_kind_layers(kind)) return -1;
    const int i = lv_find(kind, nl, dims, kind == 0 ? -1 : dims[0] - rot);
    if (i < 0) return -1;
    const LvLayout L = lv_layout(g_inst[i].S);
    return kind == 1 ? lv_split_layout(g_inst[i].S, L).total : L.total;
}

extern "C" int p2n_level_pack(int kind, int nl, const int *dims, const void *const *params_host, int rot,
                              double eps, float *packed, void *stream) {
    P2N_REQUIRE(dims && params_host && packed, "p2n_level_pack: null pointer");
    P2N_REQUIRE(nl == lv_kind_layers(kind), "p2n_level_pack: kind %d takes %d layers, got %d", kind, lv_kind_layers(kind), nl);
    P2N_REQUIRE(rot >= 0 && rot <= dims[0], "p2n_level_pack: rot %d out of range", rot);
    P2N_REQUIRE(kind != 0 || rot == 3, "p2n_level_pack: a set-abstraction chain has its 3 coordinate columns first (rot = 3)");
    const int i = lv_find(kind, nl, dims, kind == 0 ? -1 : dims[0] - rot);
    P2N_REQUIRE(i >= 0, "p2n_level_pack: no fused instantiation for this chain (kind %d, input %d channels)", kind, dims[0]);
    LvPackArgs a;
    a.S = g_inst[i].S;
    a.L = lv_layout(a.S);
    a.rot = rot;
    a.eps = eps;
    for (int l = 0; l < LV_MAX_LAYERS; ++l) {
        a.cin[l] = a.cout[l] = a.bn[l] = 0;
        a.W[l] = a.b[l] = a.gamma[l] = a.beta[l] = a.mean[l] = a.var[l] = nullptr;
    }
    for (int l = 0; l < nl; ++l) {
        a.cin[l] = dims[l]; a.cout[l] = dims[l + 1];
        const void *const *q = params_host + 6 * l;
        a.W[l] = (const float *)q[0]; a.b[l] = (const float *)q[1];
        a.gamma[l] = (const float *)q[2]; a.beta[l] = (const float *)q[3];
        a.mean[l] = (const float *)q[4]; a.var[l] = (const float *)q[5];
        a.bn[l] = q[2] != nullptr;
        P2N_REQUIRE(a.W[l] && a.b[l], "p2n_level_pack: layer %d has no weight/bias", l);
        P2N_REQUIRE(!a.bn[l] || (q[3] && q[4] && q[5]), "p2n_level_pack: layer %d has an incomplete BatchNorm", l);
        const bool want_bn = !(kind == 2 && l == nl - 1);
        P2N_REQUIRE(a.bn[l] == (int)want_bn, "p2n_level_pack: layer %d: the fused kernels take Conv+BatchNorm+ReLU "
                    "layers (and a plain last layer for kind 2)", l);
    }
    hipStream_t s = (hipStream_t)stream;
    const int ps = p2n_prof_begin(P2N_K_LEVEL_PACK, s);
    hipLaunchKernelGGL(p2n_level_pack_kernel, dim3((unsigned)((a.L.total + 255) / 256)), dim3(256), 0, s, a, packed);
    if (kind == 1) {
        const LvSplit X = lv_split_layout(a.S, a.L);
        hipLaunchKernelGGL(p2n_level_pack_split_kernel, dim3((unsigned)((X.total - a.L.total + 255) / 256)), dim3(256), 0,
                           s, a, X, packed);
    }
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

template <typename K>
static int lv_allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return P2N_OK;
}

static size_t lv_lds_bytes(const LvShape &S, const LvLayout &L) {
    return sizeof(float) * ((S.streamed ? 2 * (size_t)CHUNK_FLOATS : (size_t)L.stream_floats) + L.tab_floats);
}

#define LV_LAUNCH(kernel, args, nblk, maxwg)                                                        \
    do {                                                                                            \
        int rc_ = lv_allow_lds(kernel, lds);                                                        \
        if (rc_) return rc_;                                                                        \
        const long grid_ = (nblk) < (maxwg) ? (nblk) : (maxwg);                                      \
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid_), dim3(256), lds, s, args);                 \
    } while (0)

extern "C" int p2n_sa_level(const float *xyz, int ldx, const float *new_xyz, const float *feat, int ldf,
                            const int32_t *idx, int B, int C, int N, int M, int ns, const int *mlp,
                            const float *packed, float *out, void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && mlp, "p2n_sa_level: bad shape");
    P2N_REQUIRE(ns == 32, "p2n_sa_level: the fused level takes nsample = 32 (one wave per centroid), got %d", ns);
    const int dims[4] = {C + 3, mlp[0], mlp[1], mlp[2]};
    const int i = lv_find(0, 3, dims, -1);
    P2N_REQUIRE(i >= 0, "p2n_sa_level: no fused instantiation for C=%d mlp=(%d,%d,%d)", C, mlp[0], mlp[1], mlp[2]);
    const long total = (long)B * M;
    if (total == 0) return P2N_OK;
    P2N_REQUIRE(xyz && new_xyz && idx && packed && out && (C == 0 || feat), "p2n_sa_level: null pointer");
    P2N_REQUIRE(ldx >= 3 && (C == 0 || ldf >= C), "p2n_sa_level: row strides too small");
    P2N_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)out & 15) == 0, "p2n_sa_level: packed/out must be 16-byte aligned");
    const LvShape &S = g_inst[i].S;
    const LvLayout L = lv_layout(S);
    SaArgs a;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feat = C ? feat : xyz; a.idx = idx; a.packed = packed; a.out = out;
    a.total = total; a.tab = L.tab; a.ldx = ldx; a.ldf = C ? ldf : ldx; a.C = C; a.N = N; a.M = M; a.Cout = mlp[2];
    a.vec = C == 8 * (S.G0 - 1) && (C == 0 || ((ldf & 3) == 0 && ((uintptr_t)feat & 15) == 0));   // exact fit
    const size_t lds = lv_lds_bytes(S, L);
    const long nblk = (total + P2N_WAVES_PER_WG - 1) / P2N_WAVES_PER_WG;
    hipStream_t s = (hipStream_t)stream;
    const int ps = p2n_prof_begin(P2N_K_SA_LEVEL, s);
    switch (i) {
        case 0: LV_LAUNCH((p2n_sa_level_kernel<1, 0, 1, 1, 2, 2, false, true>), a, nblk, 2048L); break;
        case 1: LV_LAUNCH((p2n_sa_level_kernel<9, 8, 2, 2, 4, 4, false, true>), a, nblk, 512L); break;
        default: LV_LAUNCH((p2n_sa_level_kernel<17, 16, 4, 4, 8, 2, true, false>), a, nblk, 512L); break;
    }
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// A two-layer level runs channel-split (layer by layer, see LvSplit) when it has fewer 128-point blocks than
// half the CUs and its middle width fills its tiles exactly; the layer-0 output then passes through `workspace`.
#define LV_SPLIT_MAX_BLOCKS 128
static bool lv_split_applies(int inst, long total, const int *mlp) {
    const LvShape &S = g_inst[inst].S;
    return g_inst[inst].kind == 1 && (total + P2N_PTS_PER_WG - 1) / P2N_PTS_PER_WG <= LV_SPLIT_MAX_BLOCKS &&
           mlp[0] == 32 * S.CT[0];
}

extern "C" size_t p2n_fp_level_workspace_bytes(int B, int n, int C1, int C2, int nl, const int *mlp) {
    if (B < 1 || n < 1 || !mlp || nl != 2) return 0;
    const int dims[3] = {C1 + C2, mlp[0], mlp[1]};
    const int i = lv_find(1, 2, dims, C2);
    if (i < 0 || !lv_split_applies(i, (long)B * n, mlp)) return 0;
    return sizeof(float) * (size_t)B * n * mlp[0];
}

extern "C" int p2n_fp_level(const int32_t *idx3, const float *w3, const float *feat2, const float *skip, int lds_skip,
                            int B, int n, int m, int C1, int C2, int nl, const int *mlp, const float *packed,
                            float *out, void *workspace, size_t workspace_bytes, void *stream) {
    P2N_REQUIRE(B >= 0 && n >= 0 && m >= 1 && C1 >= 0 && C2 >= 4 && mlp && (nl == 2 || nl == 4), "p2n_fp_level: bad shape");
    P2N_REQUIRE((C2 & 3) == 0, "p2n_fp_level: the coarse channel count must be a multiple of 4, got %d", C2);
    int dims[LV_MAX_LAYERS + 1] = {C1 + C2, 0, 0, 0, 0};
    for (int l = 0; l < nl; ++l) dims[l + 1] = mlp[l];
    const int kind = nl == 2 ? 1 : 2;
    const int i = lv_find(kind, nl, dims, C2);
    P2N_REQUIRE(i >= 0, "p2n_fp_level: no fused instantiation for C1=%d C2=%d with %d layers", C1, C2, nl);
    const long total = (long)B * n;
    if (total == 0) return P2N_OK;
    P2N_REQUIRE(idx3 && w3 && feat2 && packed && out && (C1 == 0 || skip), "p2n_fp_level: null pointer");
    P2N_REQUIRE(C1 == 0 || lds_skip >= C1, "p2n_fp_level: skip row stride too small");
    P2N_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)feat2 & 15) == 0,
                "p2n_fp_level: packed/out/feat2 must be 16-byte aligned");
    const LvShape &S = g_inst[i].S;
    const LvLayout L = lv_layout(S);
    FpArgs a;
    a.idx3 = idx3; a.w3 = w3; a.feat2 = feat2; a.skip = C1 ? skip : feat2; a.packed = packed; a.out = out;
    a.total = total; a.tab = L.tab; a.lds = C1 ? lds_skip : 0; a.C1 = C1; a.C2 = C2; a.n = n; a.m = m; a.Cout = mlp[nl - 1];
    a.vec = C1 == 8 * (S.G0 - g_inst[i].gi) && (lds_skip & 3) == 0 && ((uintptr_t)skip & 15) == 0;
    const size_t lds = lv_lds_bytes(S, L);
    const long nblk = (total + P2N_PTS_PER_WG - 1) / P2N_PTS_PER_WG;
    hipStream_t s = (hipStream_t)stream;
    const int ps = p2n_prof_begin(P2N_K_FP_LEVEL, s);
    if (lv_split_applies(i, total, mlp) && workspace && ((uintptr_t)workspace & 15) == 0 &&
        workspace_bytes >= sizeof(float) * (size_t)total * mlp[0]) {
        const LvSplit X = lv_split_layout(S, L);
        const size_t lds = sizeof(float) * (2 * (size_t)CHUNK_FLOATS + 128);
        const long gx = nblk < 512 ? nblk : 512;
        FpArgs a0 = a;                       // layer 0: interpolate + skip -> workspace rows of mlp[0] floats
        a0.packed = packed + X.base[0]; a0.out = (float *)workspace; a0.Cout = mlp[0];
        FpArgs a1 = a;                       // layer 1: workspace rows -> out
        a1.packed = packed + X.base[1]; a1.skip = (const float *)workspace; a1.lds = mlp[0]; a1.C1 = mlp[0]; a1.vec = 1;
        int rc;
        if (i == 3) {
            if ((rc = lv_allow_lds(p2n_fp1_level_kernel<40, 32, 4>, lds))) return rc;
            hipLaunchKernelGGL((p2n_fp1_level_kernel<40, 32, 4>), dim3((unsigned)gx, X.pairs[0]), dim3(256), lds, s, a0, X.stride[0]);
        } else {
            if ((rc = lv_allow_lds(p2n_fp1_level_kernel<48, 32, 4>, lds))) return rc;
            hipLaunchKernelGGL((p2n_fp1_level_kernel<48, 32, 4>), dim3((unsigned)gx, X.pairs[0]), dim3(256), lds, s, a0, X.stride[0]);
        }
        if ((rc = lv_allow_lds(p2n_fp1_level_kernel<32, 0, 4>, lds))) return rc;
        hipLaunchKernelGGL((p2n_fp1_level_kernel<32, 0, 4>), dim3((unsigned)gx, X.pairs[1]), dim3(256), lds, s, a1, X.stride[1]);
        p2n_prof_end(ps, s);
        P2N_CHECK_HIP(hipGetLastError());
        return P2N_OK;
    }
    // few blocks (at most one workgroup per CU anyway): the variant compiled for one wave per SIMD, which
    // may use the whole register file to run further ahead of its MFMAs
    const bool lone = nblk <= 384;
    switch (i) {
        case 3:
            if (lone) LV_LAUNCH((p2n_fp2_level_kernel<40, 32, 8, 4, 4, 1, FP_SEG1>), a, nblk, 512L);
            else LV_LAUNCH((p2n_fp2_level_kernel<40, 32, 8, 4, 4, 2, FP_SEG2>), a, nblk, 512L);
            break;
        case 4:
            if (lone) LV_LAUNCH((p2n_fp2_level_kernel<48, 32, 8, 8, 2, 1, FP_SEG1>), a, nblk, 512L);
            else LV_LAUNCH((p2n_fp2_level_kernel<48, 32, 8, 8, 2, 2, FP_SEG2>), a, nblk, 512L);
            break;
        default: LV_LAUNCH((p2n_fp4_level_kernel<17, 16, 4, 4, 4, 2, 4>), a, nblk, 512L); break;
    }
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
