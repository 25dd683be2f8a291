// Point2Net eval forward for gfx950 (MI355X): three launches.
//
//   p2n_encoder_kernel  layers 0..5 (reference point2net.py:9-34,36-43), f3 spill,
//                       per-workgroup channel max (reference point2net.py:46)
//   p2n_mid_kernel      per-cloud max of the partials -> global_f, and the global
//                       half of head layer 6 folded into a per-cloud shift
//                       (W6[:,256:] . g replaces the concat of reference
//                       point2net.py:78-79)
//   p2n_head_kernel     head layers 6..8 (reference point2net.py:57-65,82) and
//                       the (B,E,N)->(B,N,E) transpose of :85 folded into the store
//
// Design: one wave owns 32 points (one 32-column block of v_mfma_f32_32x32x2_f32)
// and carries their activations through every layer IN REGISTERS.  With D = W.X
// (A = weights, rows = output channels; B = activations, columns = points) the
// accumulator layout of layer l (lane = point, registers = channels) is exactly the
// B-operand layout of layer l+1 once the reduction index is permuted
// (k = 32*kt + 8*g + 4*h + j  <->  register 4g+j of tile kt, lane half h), and the
// permutation is absorbed into the weight image at pack time.  So activations never
// touch LDS or HBM between layers; the only HBM traffic is x in, the f3 spill across
// the global-max dependency, and emb out.
//
// Weights: the packed image is a linear stream of 1-KiB lane-linear fragments in the
// exact order the kernel consumes them, cut into 32-fragment chunks.  A workgroup
// (4 waves, 128 points) keeps two chunks in LDS: while every wave computes on chunk c
// (one ds_read_b128 per 4 MFMAs), the chunk after it lands by LDS-DMA
// (global_load_lds_dwordx4, 8 fragments per wave, no staging registers); one
// __syncthreads() per chunk (8192 MFMA cycles per wave) retires the DMA and frees the
// other buffer.  fp32 in, fp32 accumulate: exact-f32 MFMA.
#include <mutex>
#include "p2n_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define CHUNK_FLOATS (P2N_CHUNK * P2N_FRAG_FLOATS)   // 8192 floats = 32 KiB

// ------------------------------------------------------------------ pack
__global__ void p2n_pack_kernel(const float *__restrict__ raw, float *__restrict__ packed, P2nLayout L) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.total) return;
    float v = 0.f;
    if (i >= L.w6g) {
        long r = i - L.w6g;
        int co = (int)(r / 256), k = (int)(r % 256);
        v = raw[L.raw_w[6] + (long)co * 512 + 256 + k];
    } else if (i >= L.scale[0]) {
        long r = i - L.scale[0];
        for (int l = 0; l < P2N_NLAYER; ++l) {
            int cp = L.cpad[l];
            if (r < 2L * cp) {
                int c = (int)(r % cp);
                bool is_shift = r >= cp;
                if (c < L.cout[l]) {
                    double b = raw[L.raw_b[l] + c];
                    if (l < 8) {
                        double s = (double)raw[L.raw_g[l] + c] / sqrt((double)raw[L.raw_var[l] + c] + 1e-5);
                        double t = (b - (double)raw[L.raw_mean[l] + c]) * s + (double)raw[L.raw_beta[l] + c];
                        v = (float)(is_shift ? t : s);
                    } else {
                        v = is_shift ? (float)b : 1.f;
                    }
                }
                break;
            }
            r -= 2L * cp;
        }
    } else {
        // weight fragment streams
        bool head = i >= L.head_stream;
        long r = i - (head ? L.head_stream : L.enc_stream);
        long f = r / P2N_FRAG_FLOATS;
        int e = (int)(r % P2N_FRAG_FLOATS);
        int lane = e >> 2, j = e & 3, row = lane & 31, h = lane >> 5;
        int l = -1, ct = 0, G = 0;
        if (!head) {
            if (f < L.enc_frags) {
                for (int ll = 0; ll < 6; ++ll) {
                    long nf = (long)(L.cpad[ll] / 32) * (L.kpad[ll] / 8);
                    if (f < nf) {
                        l = ll;
                        const int CT = L.cpad[ll] / 32;
                        if (ll == 0) { ct = (int)f; G = 0; }                      // K = 8: one fragment per tile
                        else if (ll < 5) {                                       // [kt][ct][g]
                            ct = (int)((f >> 2) % CT); G = (int)(4 * (f / (4 * CT)) + (f & 3));
                        } else {                                                 // [quarter][kt][ct%2][g]
                            const long r2 = f & 63;
                            ct = (int)(2 * (f >> 6) + ((r2 >> 2) & 1)); G = (int)(4 * (r2 >> 3) + (r2 & 3));
                        }
                        break;
                    }
                    f -= nf;
                }
            }
        } else if (f < L.head_frags) {
            if (f < 256) {            // layer 6: [kt][ct][g]
                l = 6; ct = (int)((f >> 2) & 7); G = (int)(4 * (f >> 5) + (f & 3));
            } else if (f < 384) {     // layer 7: [kt][ct][g]
                long q = f - 256;
                l = 7; ct = (int)((q >> 2) & 3); G = (int)(4 * (q >> 4) + (q & 3));
            } else {                  // layer 8: [ct][kt][g]
                long q = f - 384;
                l = 8; ct = (int)(q >> 4); G = (int)(q & 15);
            }
        }
        if (l >= 0) {
            int co = 32 * ct + row, k = 8 * G + 4 * h + j;
            int klim = (l == 6) ? 256 : L.cin[l];
            if (co < L.cout[l] && k < klim) {
                // BatchNorm's scale is folded into the weight (product in double, one rounding): the kernels start
                // their accumulators at the shift and the epilogue is the ReLU alone -- one vector-ALU instruction
                // per output element instead of two, and vector-ALU time adds to MFMA time (DESIGN_HISTORY section 7)
                double w = raw[L.raw_w[l] + (long)co * L.cin[l] + k];
                if (l < 8) w *= (double)raw[L.raw_g[l] + co] / sqrt((double)raw[L.raw_var[l] + co] + 1e-5);
                v = (float)w;
            }
        }
    }
    packed[i] = v;
}

// ------------------------------------------------------------------ helpers
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// This wave's quarter (8 fragments) of chunk c of a weight stream -> LDS buffer c&1.
// LDS-DMA: the destination is wave-uniform base + lane*16, i.e. the lane-linear fragment image.
// As a buffer instruction: the stream base sits in a scalar resource, the fragment's offset is scalar arithmetic and the
// lane offset one constant register, so staging costs no vector-ALU instruction (whose time adds to the MFMAs':
// DESIGN_HISTORY section 7; written as a pointer it cost one 64-bit vector add and a readfirstlane per fragment).
__device__ __forceinline__ void stage_chunk(const float *__restrict__ stream, float *wbuf, int c, int wave, int lane) {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(stream), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < P2N_CHUNK / P2N_WAVES_PER_WG; ++i) {
        const int f = wv * (P2N_CHUNK / P2N_WAVES_PER_WG) + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(wbuf + (c & 1) * CHUNK_FLOATS + f * P2N_FRAG_FLOATS), 16,
                                                 lane * 16, (c * P2N_CHUNK + f) * (P2N_FRAG_FLOATS * 4), 0, 0);
    }
}

// Entering fragment f of the stream: at a chunk boundary wait for the chunk (vmcnt(0) +
// barrier, p2n_dma_barrier), which also proves every wave left the other buffer,
// then start the DMA of the next chunk into that other buffer.
__device__ __forceinline__ void chunk_gate(int f, const float *__restrict__ stream, float *wbuf, int wave, int lane) {
    if (f % P2N_CHUNK == 0) {
        p2n_dma_barrier();
        stage_chunk(stream, wbuf, f / P2N_CHUNK + 1, wave, lane);
    }
}

__device__ __forceinline__ f32x4 lds_frag(const float *wbuf, int f, int lane) {
    return *reinterpret_cast<const f32x4 *>(wbuf + ((f / P2N_CHUNK) & 1) * CHUNK_FLOATS +
                                            (f % P2N_CHUNK) * P2N_FRAG_FLOATS + lane * 4);
}

// The layer's shift (BatchNorm shift with the conv bias folded in, or the per-cloud layer-6 shift) is where the
// accumulators START; the scale sits in the packed weights.  Channel of register 4g+j in lane half h: 32*ct + 8g + 4h + j.
__device__ __forceinline__ f32x16 shift_tile(const float *shift, int ct, int h) {
    f32x16 out;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 32 * ct + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * g + j] = sh[j];
    }
    return out;
}
template <int CT>
__device__ __forceinline__ void shift_tiles(f32x16 (&t)[CT], const float *shift, int h) {
#pragma unroll
    for (int c = 0; c < CT; ++c) t[c] = shift_tile(shift, c, h);
}
__device__ __forceinline__ f32x16 relu_tile(const f32x16 acc) {
    f32x16 out;
#pragma unroll
    // ReLU on the bit pattern: positive floats are positive integers, negative ones (and -0) negative, so
    // max_i32(bits, 0) is relu(x) in ONE instruction.  fmaxf(x, 0) on an MFMA result costs two: the compiler first
    // canonicalises a value it cannot prove quiet (v_max x, x) -- the fused multiply-add used to do that on the side.
    for (int i = 0; i < 16; ++i) {
        const int b = __float_as_int(acc[i]);
        out[i] = __int_as_float(b > 0 ? b : 0);
    }
    return out;
}

// One pointwise layer on a wave's 32 points, k-tile outer: CT independent accumulators
// take turns on the matrix pipe (no MFMA ever waits on its predecessor) and every epilogue
// of the layer happens once, after the last k-tile.  Consumes KT*CT*4 consecutive weight
// fragments, ordered [kt][ct][g], starting at stream index F0 (compile-time, so LDS
// offsets are immediates).  acc comes in holding the layer's shift (shift_tiles).
template <int KT, int CT, int F0>
__device__ __forceinline__ void layer_kt_outer(const float *__restrict__ stream, float *wbuf, int wave, int lane,
                                               const f32x16 (&in)[KT], f32x16 (&acc)[CT]) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = F0 + (kt * CT + ct) * 4 + g;
                chunk_gate(f, stream, wbuf, wave, lane);
                const f32x4 a = lds_frag(wbuf, f, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ct] = MFMA32(a[j], in[kt][4 * g + j], acc[ct]);
            }
}

template <int CT>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) t[c][i] = 0.f;
}

// max over the 16 lanes of a DPP row, on the bit patterns of non-negative floats
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    v = v > t ? v : t;
    return v;
}

struct EncArgs {
    const float *x;
    const float *packed;
    float *spill;     // [B*tiles*4 groups][8 ct][4 g][64 lanes][4]
    float *partial;   // [B*tiles][256]
    float *f1, *f2, *f3;
    int N, F, tiles;
    long stream;
    long tab;            // start of the scale/shift tables in the packed image
    int tab_n;           // floats to stage into LDS
    int scale[6], shift[6];   // offsets inside the table
    // RAW (p2n_forward_eval_raw, one scene): layer 0 reads the scene as the caller holds it -- float64 xyz / colours --
    // and forms the fp32 features in its loader with numpy's own float64 arithmetic (reference inference.py:88-103)
    const double *raw_pts, *raw_col, *raw_cs;   // (N,3), (N,3) | NULL, {cx, cy, cz, scale, colours-above-1 flag}
};

// feature store in the reference's (B,C,N) layout
template <int C>
__device__ __forceinline__ void store_feat(float *__restrict__ f, const f32x16 v, int b, int ct, int h, long n, int N) {
    if (n >= N) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[((long)b * C + 32 * ct + 8 * g + 4 * h + j) * N + n] = v[4 * g + j];
}

#define ENC_LDS_FLOATS (2 * CHUNK_FLOATS + 256 + P2N_TAB_MAX)

template <bool FEATS, bool RAW = false>
__global__ __launch_bounds__(256, 2) void p2n_encoder_kernel(EncArgs A) {
    // one LDS array: two weight chunks, [256] channel maxima, scale/shift tables
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wbuf = lds;
    unsigned *smax = reinterpret_cast<unsigned *>(lds + 2 * CHUNK_FLOATS);
    float *tab = lds + 2 * CHUNK_FLOATS + 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / A.tiles, tile = blockIdx.x % A.tiles;
    const long n = ((long)tile * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE + p;
    const float *pk = A.packed;
    const float *stream = pk + A.stream;
    stage_chunk(stream, wbuf, 0, wave, lane);
    for (int i = threadIdx.x; i < A.tab_n; i += 256) tab[i] = pk[A.tab + i];
    smax[threadIdx.x] = 0u;
    const bool valid = n < A.N;
    const bool wave_full = n - p + P2N_PTS_PER_WAVE <= A.N;   // the wave's 32 points all lie inside the cloud
    const long grp = (long)blockIdx.x * P2N_WAVES_PER_WG + wave;

    // layer 0 operand: x[n][4h+j], zero beyond F (K padded to 8) and beyond the cloud
    f32x4 xin;
    if (RAW) {
        // the input normalisation's affine, fused (csrc/p2n_prep.hip: prep_write_kernel's arithmetic, bit for bit):
        // float((p - centroid) / scale) in float64 with one rounding to float; colours / 255 when any exceeds 1, 0.5 without colours
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * h + j;
            float v = 0.f;
            if (valid && k < 3) v = (float)(__dsub_rn(A.raw_pts[n * 3 + k], A.raw_cs[k]) / A.raw_cs[3]);
            else if (valid && k < 6) v = A.raw_col ? (float)(A.raw_cs[4] > 0.0 ? A.raw_col[n * 3 + (k - 3)] / 255.0 : A.raw_col[n * 3 + (k - 3)]) : 0.5f;
            xin[j] = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * h + j;
            xin[j] = (valid && k < A.F) ? A.x[((long)b * A.N + n) * A.F + k] : 0.f;
        }
    }

    f32x16 a0[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        chunk_gate(ct, stream, wbuf, wave, lane);   // ct == 0: first barrier (chunk 0, tables, smax)
        f32x16 acc = shift_tile(tab + A.shift[0], ct, h);
        const f32x4 a = lds_frag(wbuf, ct, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], xin[j], acc);
        a0[ct] = relu_tile(acc);
    }
    // stream positions: L0 2 frags, L1 16, L2 32, L3 64, L4 128, L5 4 x 64
    f32x16 a1[2];
    shift_tiles(a1, tab + A.shift[1], h);
    layer_kt_outer<2, 2, 2>(stream, wbuf, wave, lane, a0, a1);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        a1[ct] = relu_tile(a1[ct]);
        if (FEATS) if (A.f1) store_feat<64>(A.f1, a1[ct], b, ct, h, n, A.N);
    }
    f32x16 a2[4];
    shift_tiles(a2, tab + A.shift[2], h);
    layer_kt_outer<2, 4, 18>(stream, wbuf, wave, lane, a1, a2);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) a2[ct] = relu_tile(a2[ct]);
    f32x16 a3[4];
    shift_tiles(a3, tab + A.shift[3], h);
    layer_kt_outer<4, 4, 50>(stream, wbuf, wave, lane, a2, a3);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        a3[ct] = relu_tile(a3[ct]);
        if (FEATS) if (A.f2) store_feat<128>(A.f2, a3[ct], b, ct, h, n, A.N);
    }
    f32x16 a4[8];
    shift_tiles(a4, tab + A.shift[4], h);
    layer_kt_outer<4, 8, 114>(stream, wbuf, wave, lane, a3, a4);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) a4[ct] = relu_tile(a4[ct]);
    // layer 5 in four passes of two output tiles: with its 128 input registers live, two
    // independent accumulators (32 registers) leave room for the epilogue without spilling;
    // results leave through the spill and the channel max
    float *sp = A.spill + grp * (256L * P2N_PTS_PER_WAVE) + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x16 a5[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) a5[c2] = shift_tile(tab + A.shift[5], 2 * q + c2, h);
        if (q == 0) layer_kt_outer<8, 2, 242>(stream, wbuf, wave, lane, a4, a5);
        else if (q == 1) layer_kt_outer<8, 2, 306>(stream, wbuf, wave, lane, a4, a5);
        else if (q == 2) layer_kt_outer<8, 2, 370>(stream, wbuf, wave, lane, a4, a5);
        else layer_kt_outer<8, 2, 434>(stream, wbuf, wave, lane, a4, a5);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int ct = 2 * q + c2;
            const f32x16 v = relu_tile(a5[c2]);
            if (FEATS) if (A.f3) store_feat<256>(A.f3, v, b, ct, h, n, A.N);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                *reinterpret_cast<f32x4 *>(sp + (ct * 4 + g) * P2N_FRAG_FLOATS) = o;
            }
            // channel max over the workgroup's points: post-ReLU values are >= 0, so their bit patterns order as
            // integers and 0 is the neutral element for masked points.  Halving transpose-reduce over the 32 lanes
            // of each wave half (15 DPP maxima + one swizzle per tile, p2n_common.h; 16 x four DPP steps before),
            // after which lane i of a row holds the maximum of register i: one LDS atomic per tile.
            f32x16 vm = v;
            if (!wave_full) {   // wave-uniform: only a cloud's last wave has points past its end
#pragma unroll
                for (int i = 0; i < 16; ++i) vm[i] = valid ? v[i] : 0.f;
            }
            const int mx = __float_as_int(p2n_tile_half_max(vm, lane));
            if ((lane & 16) == 0)
                atomicMax(&smax[32 * ct + 8 * ((lane & 15) >> 2) + 4 * h + (lane & 3)], (unsigned)(mx > 0 ? mx : 0));   // (-0.0 -> 0)
        }
    }
    p2n_dma_barrier();   // also drains the spare chunk's DMA before the workgroup's LDS is released
    A.partial[(long)blockIdx.x * 256 + threadIdx.x] = __uint_as_float(smax[threadIdx.x]);
}

// ------------------------------------------------------------------ mid
// grid (B, 8): every block reduces the cloud's partial maxima (cheap, redundant),
// block y computes 32 rows of c = W6[:,256:] . g and the per-cloud layer-6 shift.
__global__ __launch_bounds__(256) void p2n_mid_kernel(const float *__restrict__ partial, int tiles,
                                                      const float *__restrict__ packed, long w6g, long scale6,
                                                      long shift6, float *__restrict__ shift6c,
                                                      float *__restrict__ global_f) {
    __shared__ __attribute__((aligned(16))) float g[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    const float *pp = partial + (long)b * tiles * 256 + t;
    int i = 0;
    for (; i + 3 < tiles; i += 4) {
        m0 = fmaxf(m0, pp[(long)i * 256]);
        m1 = fmaxf(m1, pp[(long)(i + 1) * 256]);
        m2 = fmaxf(m2, pp[(long)(i + 2) * 256]);
        m3 = fmaxf(m3, pp[(long)(i + 3) * 256]);
    }
    for (; i < tiles; ++i) m0 = fmaxf(m0, pp[(long)i * 256]);
    const float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    g[t] = m;
    if (blockIdx.y == 0 && global_f) global_f[(long)b * 256 + t] = m;
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(&g[4 * lane]);
    float s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int co = blockIdx.y * 32 + wave * 8 + q;
        const f32x4 w = *reinterpret_cast<const f32x4 *>(packed + w6g + (long)co * 256 + 4 * lane);
        s[q] = (w[0] * gv[0] + w[1] * gv[1]) + (w[2] * gv[2] + w[3] * gv[3]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o);
        const int co = blockIdx.y * 32 + wave * 8 + q;
        if (lane == 0) shift6c[(long)b * 256 + co] = fmaf(s[q], packed[scale6 + co], packed[shift6 + co]);
    }
}

// ------------------------------------------------------------------ head
struct HeadArgs {
    const float *packed;
    const float *spill;
    const float *shift6c;   // [B][256]
    float *emb;             // (B,N,E)
    int N, E, tiles, ct8;   // ct8 = cpad[8]/32
    long stream;
    long tab;
    int tab_n;
    int scale6, scale7, shift7, scale8, shift8;   // offsets inside the table
};

__global__ __launch_bounds__(256, 2) void p2n_head_kernel(HeadArgs A) {
    // LDS carve: two weight chunks, [tab_n] scale/shift tables, [256] this cloud's layer-6 shift
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf = smem;
    float *tab = smem + 2 * CHUNK_FLOATS;
    float *sh6 = tab + A.tab_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / A.tiles, tile = blockIdx.x % A.tiles;
    const long n0 = ((long)tile * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE;
    const long grp = (long)blockIdx.x * P2N_WAVES_PER_WG + wave;
    const float *pk = A.packed;
    const float *stream = pk + A.stream;
    stage_chunk(stream, wbuf, 0, wave, lane);
    for (int i = threadIdx.x; i < A.tab_n; i += 256) tab[i] = pk[A.tab + i];
    sh6[threadIdx.x] = A.shift6c[(long)b * 256 + threadIdx.x];

    // ---- layer 6, k-tile outer: the f3 tile of k-slice kt is loaded once (next one
    // prefetched), used by all eight output tiles, dropped.  Chunk kt of the stream holds
    // exactly the 32 fragments of k-slice kt.
    const f32x4 *sp = reinterpret_cast<const f32x4 *>(A.spill + grp * (256L * P2N_PTS_PER_WAVE)) + lane;
    __syncthreads();   // tables and this cloud's layer-6 shift are in LDS: the accumulators start from them
    f32x16 acc6[8];
    shift_tiles(acc6, sh6, h);
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[g] = sp[g * P2N_WAVE];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        if (kt < 7) {
#pragma unroll
            for (int g = 0; g < 4; ++g) nxt[g] = sp[((kt + 1) * 4 + g) * P2N_WAVE];
        }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = kt * 32 + ct * 4 + g;
                chunk_gate(f, stream, wbuf, wave, lane);
                const f32x4 a = lds_frag(wbuf, f, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc6[ct] = MFMA32(a[j], cur[g][j], acc6[ct]);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) cur[g] = nxt[g];
    }
    f32x16 a6[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a6[c] = relu_tile(acc6[c]);

    // ---- layer 7, k-tile outer as well: a6[kt] dies as soon as its slice is consumed
    f32x16 a7[4];
    shift_tiles(a7, tab + A.shift7, h);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 256 + kt * 16 + ct * 4 + g;
                chunk_gate(f, stream, wbuf, wave, lane);
                const f32x4 a = lds_frag(wbuf, f, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) a7[ct] = MFMA32(a[j], a6[kt][4 * g + j], a7[ct]);
            }
#pragma unroll
    for (int c = 0; c < 4; ++c) a7[c] = relu_tile(a7[c]);

    // ---- layer 8: runtime number of 32-channel tiles, two per chunk (16 fragments each).
    // Output: lane (p,h) holds 4 consecutive channels per register group, so every point's
    // row is written as 16-byte pieces that together fill whole 128-byte lines of emb (B,N,E).
    const int E = A.E;
    const long n = n0 + p;
    const bool valid = n < A.N;
    float *dst = A.emb + ((long)b * A.N + n) * E;
    const bool vec = (E & 3) == 0;
    const int nch8 = (A.ct8 + 1) / 2;
#pragma unroll 1
    for (int c8 = 0; c8 < nch8; ++c8) {
        const int c = 12 + c8;
        p2n_dma_barrier();
        stage_chunk(stream, wbuf, c + 1, wave, lane);
        const float *wb = wbuf + (c & 1) * CHUNK_FLOATS + lane * 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int ct8 = 2 * c8 + half;
            if (ct8 < A.ct8) {
                f32x16 acc = shift_tile(tab + A.shift8, ct8, h);   // the plain last layer: its bias
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(wb + (half * 16 + kt * 4 + g) * P2N_FRAG_FLOATS);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], a7[kt][4 * g + j], acc);
                    }
                const f32x16 v = acc;
                if (valid) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int e0 = 32 * ct8 + 8 * g + 4 * h;
                        if (vec) {
                            if (e0 < E) {
                                f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                                *reinterpret_cast<f32x4 *>(dst + e0) = o;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (e0 + j < E) dst[e0 + j] = v[4 * g + j];
                        }
                    }
                }
            }
        }
    }
    // drain the spare chunk's DMA before the workgroup's LDS is released
    p2n_dma_barrier();
}

// ------------------------------------------------------------------ C ABI
extern "C" long p2n_param_count(int F, int E) {
    if (F < 1 || E < 1) return -1;
    return p2n_make_layout(F, E).raw_total;
}

extern "C" long p2n_packed_param_count(int F, int E) {
    if (F < 1 || F > 8 || E < 1 || E > 256) return -1;
    return p2n_make_layout(F, E).total;
}

extern "C" int p2n_pack_eval_params(const float *raw, int F, int E, float *packed, void *stream) {
    P2N_REQUIRE(raw && packed, "p2n_pack_eval_params: null pointer");
    P2N_REQUIRE(F >= 1 && F <= 8, "p2n_pack_eval_params: feature_dim %d not in [1,8]", F);
    P2N_REQUIRE(E >= 1 && E <= 256, "p2n_pack_eval_params: embedding_dim %d not in [1,256]", E);
    const P2nLayout L = p2n_make_layout(F, E);
    const int threads = 256;
    const long blocks = (L.total + threads - 1) / threads;
    hipLaunchKernelGGL(p2n_pack_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, raw, packed, L);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

static inline int tiles_per_cloud(int N) { return (N + P2N_PTS_PER_WG - 1) / P2N_PTS_PER_WG; }

extern "C" size_t p2n_forward_eval_workspace_bytes(int B, int N, int F, int E) {
    (void)F; (void)E;
    if (B < 1 || N < 1) return 0;
    const size_t tiles = (size_t)B * tiles_per_cloud(N);
    return sizeof(float) * (tiles * P2N_WAVES_PER_WG * 256 * P2N_PTS_PER_WAVE + tiles * 256 + (size_t)B * 256);
}

template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return P2N_OK;
}

static int forward_eval_impl(const float *x, int B, int N, int F, int E, const float *packed, float *emb,
                                float *global_f, float *f1, float *f2, float *f3, void *workspace,
                                size_t workspace_bytes, void *stream_, const double *raw_pts = nullptr,
                                const double *raw_col = nullptr, const double *raw_cs = nullptr) {
    // shared body of p2n_forward_eval and p2n_forward_eval_noalloc: launches only, no allocation
    P2N_REQUIRE(B >= 0 && N >= 0, "p2n_forward_eval: negative shape B=%d N=%d", B, N);
    P2N_REQUIRE(F >= 1 && F <= 8, "p2n_forward_eval: feature_dim %d not in [1,8]", F);
    P2N_REQUIRE(E >= 1 && E <= 256, "p2n_forward_eval: embedding_dim %d not in [1,256]", E);
    if (B == 0 || N == 0) return P2N_OK;
    P2N_REQUIRE((x || raw_pts) && packed, "p2n_forward_eval: null x/packed");
    P2N_REQUIRE(emb || global_f || f1 || f2 || f3, "p2n_forward_eval: no output requested");
    P2N_REQUIRE((long)B * tiles_per_cloud(N) < (1L << 31), "p2n_forward_eval: B*ceil(N/128) exceeds the grid limit");
    const size_t need = p2n_forward_eval_workspace_bytes(B, N, F, E);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_forward_eval: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    P2N_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)packed & 15) == 0,
                "p2n_forward_eval: workspace and packed must be 16-byte aligned");
    if (emb && (E & 3) == 0) P2N_REQUIRE(((uintptr_t)emb & 15) == 0, "p2n_forward_eval: emb must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const P2nLayout L = p2n_make_layout(F, E);
    const int tiles = tiles_per_cloud(N);
    const size_t ntiles = (size_t)B * tiles;
    float *spill = (float *)workspace;
    float *partial = spill + ntiles * P2N_WAVES_PER_WG * 256 * P2N_PTS_PER_WAVE;
    float *shift6c = partial + ntiles * 256;

    EncArgs ea;
    ea.x = x; ea.packed = packed; ea.spill = spill; ea.partial = partial;
    ea.f1 = f1; ea.f2 = f2; ea.f3 = f3;
    ea.N = N; ea.F = F; ea.tiles = tiles; ea.stream = L.enc_stream;
    ea.tab = L.scale[0]; ea.tab_n = (int)(L.scale[6] - L.scale[0]);
    for (int l = 0; l < 6; ++l) { ea.scale[l] = (int)(L.scale[l] - L.scale[0]); ea.shift[l] = (int)(L.shift[l] - L.scale[0]); }
    ea.raw_pts = raw_pts; ea.raw_col = raw_col; ea.raw_cs = raw_cs;
    const size_t enc_lds = sizeof(float) * ENC_LDS_FLOATS;
    int ps = p2n_prof_begin(P2N_K_ENCODER, stream);
    if (raw_pts) {
        int rc = allow_big_lds(p2n_encoder_kernel<false, true>, enc_lds);
        if (rc) return rc;
        hipLaunchKernelGGL((p2n_encoder_kernel<false, true>), dim3((unsigned)ntiles), dim3(256), enc_lds, stream, ea);
    } else if (f1 || f2 || f3) {
        int rc = allow_big_lds(p2n_encoder_kernel<true>, enc_lds);
        if (rc) return rc;
        hipLaunchKernelGGL(p2n_encoder_kernel<true>, dim3((unsigned)ntiles), dim3(256), enc_lds, stream, ea);
    } else {
        int rc = allow_big_lds(p2n_encoder_kernel<false>, enc_lds);
        if (rc) return rc;
        hipLaunchKernelGGL(p2n_encoder_kernel<false>, dim3((unsigned)ntiles), dim3(256), enc_lds, stream, ea);
    }
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    ps = p2n_prof_begin(P2N_K_MID, stream);

    hipLaunchKernelGGL(p2n_mid_kernel, dim3(B, 8), dim3(256), 0, stream, partial, tiles, packed, L.w6g, L.scale[6],
                       L.shift[6], shift6c, global_f);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());

    if (!emb) return P2N_OK;   // encoder-only call (reference point2net.py:36-48)
    HeadArgs ha;
    ha.packed = packed; ha.spill = spill; ha.shift6c = shift6c; ha.emb = emb;
    ha.N = N; ha.E = E; ha.tiles = tiles; ha.ct8 = L.cpad[8] / 32; ha.stream = L.head_stream;
    ha.tab = L.scale[6]; ha.tab_n = (int)(L.w6g - L.scale[6]);
    ha.scale6 = 0; ha.scale7 = (int)(L.scale[7] - L.scale[6]); ha.shift7 = (int)(L.shift[7] - L.scale[6]);
    ha.scale8 = (int)(L.scale[8] - L.scale[6]); ha.shift8 = (int)(L.shift[8] - L.scale[6]);
    const size_t lds = sizeof(float) * (2 * (size_t)CHUNK_FLOATS + ha.tab_n + 256);
    {
        int rc = allow_big_lds(p2n_head_kernel, lds);
        if (rc) return rc;
    }
    ps = p2n_prof_begin(P2N_K_HEAD, stream);
    hipLaunchKernelGGL(p2n_head_kernel, dim3((unsigned)ntiles), dim3(256), lds, stream, ha);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_forward_eval(const float *x, int B, int N, int F, int E, const float *packed, float *emb,
                                float *global_f, float *f1, float *f2, float *f3, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return forward_eval_impl(x, B, N, F, E, packed, emb, global_f, f1, f2, f3, workspace, workspace_bytes, stream);
}

// One scene straight from the caller's float64 arrays: reference inference.py:88-103 (subtract the centroid, divide by
// max |.|, colours / 255 if any exceeds 1, concat, .float()) happens in layer 0's loader instead of a pass that writes an
// (N,6) tensor for the encoder to re-read.  prep_cs = the 8-double block p2n_preprocess_points(features = NULL) filled on
// the same stream: {cx, cy, cz, scale, colour flag, ...}.  Embeddings are bit-identical to the two-step path.
extern "C" int p2n_forward_eval_raw(const double *points, const double *colors, int N, int E, const float *packed,
                                    const double *prep_cs, float *emb, float *global_f, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    P2N_REQUIRE(points && prep_cs, "p2n_forward_eval_raw: null points / prep_cs");
    P2N_REQUIRE((((uintptr_t)points | (uintptr_t)colors | (uintptr_t)prep_cs) & 7) == 0, "p2n_forward_eval_raw: float64 arrays must be 8-byte aligned");
    return forward_eval_impl(nullptr, 1, N, 6, E, packed, emb, global_f, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream,
                             points, colors, prep_cs);
}

// ------------------------------------------------------------------ allocation-free variant
// For callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the f3 spill, the per-workgroup channel maxima and the per-cloud layer-6 shift -- the
// workspace of p2n_forward_eval -- live in a file-scope device array.  Every GPU the library runs on
// holds its own copy (34 MB) at its own address until the process ends; the address is looked up once
// per device, with that device current, and only that pointer is ever handed to kernels (the array's
// name in host code is a host-side stand-in, not a device address).
#define P2N_NOALLOC_TILE_FLOATS (P2N_WAVES_PER_WG * 256 * P2N_PTS_PER_WAVE + 256)
#define P2N_NOALLOC_FLOATS ((size_t)P2N_NOALLOC_MAX_TILES * P2N_NOALLOC_TILE_FLOATS + (size_t)P2N_NOALLOC_MAX_TILES * 256)
static __device__ __align__(256) float scratch[P2N_NOALLOC_FLOATS];

#define P2N_MAX_DEVICES 64
static float *g_scratch_ptr[P2N_MAX_DEVICES];
static std::mutex g_scratch_mutex;

extern "C" int p2n_forward_eval_noalloc(const float *x, int B, int N, int F, int E, const float *packed,
                                        float *emb, float *global_f, float *f1, float *f2, float *f3,
                                        void *stream) {
    P2N_REQUIRE(B >= 0 && N >= 0, "p2n_forward_eval_noalloc: negative shape B=%d N=%d", B, N);
    if (B == 0 || N == 0) return P2N_OK;
    const long ntiles = (long)B * tiles_per_cloud(N);
    P2N_REQUIRE(ntiles <= P2N_NOALLOC_MAX_TILES,
                "p2n_forward_eval_noalloc: B*ceil(N/128) = %ld exceeds the built-in scratch of %d tiles "
                "(%d points); use p2n_forward_eval with a caller workspace", ntiles, P2N_NOALLOC_MAX_TILES,
                P2N_NOALLOC_MAX_TILES * P2N_PTS_PER_WG);
    // The scratch that is used must be the one of the GPU the tensors live on, which is not necessarily the
    // current device: take it from x itself and refuse a mismatch (the kernels are launched on the current
    // device; another GPU's scratch address would fault there).
    P2N_REQUIRE(x, "p2n_forward_eval_noalloc: null x");
    int dev = -1;
    P2N_CHECK_HIP(hipGetDevice(&dev));
    hipPointerAttribute_t at;
    P2N_CHECK_HIP(hipPointerGetAttributes(&at, x));
    P2N_REQUIRE(at.type == hipMemoryTypeDevice, "p2n_forward_eval_noalloc: x is not device memory");
    P2N_REQUIRE(at.device == dev,
                "p2n_forward_eval_noalloc: x lives on device %d but the current device is %d: make the tensors' device "
                "current (hipSetDevice) before the call", at.device, dev);
    P2N_REQUIRE(dev >= 0 && dev < P2N_MAX_DEVICES, "p2n_forward_eval_noalloc: device ordinal %d out of range", dev);
    float *ws;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            float *p = nullptr;
            P2N_CHECK_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
            P2N_REQUIRE(p != nullptr, "p2n_forward_eval_noalloc: hipGetSymbolAddress returned a null address");
            g_scratch_ptr[dev] = p;
        }
        ws = g_scratch_ptr[dev];
    }
    // nothing in the scratch has to start at zero: the encoder writes every spill group, partial row and
    // (through the mid kernel) shift row that this call's head kernel reads
    return forward_eval_impl(x, B, N, F, E, packed, emb, global_f, f1, f2, f3, ws, sizeof(float) * P2N_NOALLOC_FLOATS,
                             stream);
}
