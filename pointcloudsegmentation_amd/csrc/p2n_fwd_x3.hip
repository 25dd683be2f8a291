// Point2Net eval forward, "fp32x3": fp32-accurate products on the bf16 matrix pipe (opt-in,
// model.precision = "fp32x3"; the default path is the exact-fp32 MFMA of p2n_fwd.hip).
// Reference: point2net.py:36-48 (encoder), :57-65,:67-87 (head), same three launches and the same
// register-resident dataflow as p2n_fwd.hip / p2n_fwd_bf16.hip.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  Every fp32 value is the EXACT sum of three
// bf16 values, x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 3 x 8 significand bits,
// round-to-nearest parts carry their own signs, both subtractions are exact), and a bf16 x bf16 product is
// exact in fp32.  Of the nine part products of w.x the six with weight >= 2^-16 are kept,
//     wh.xh + (wh.xm + wm.xh) + (wm.xm + wh.xl + wl.xh),
// the three dropped ones (wm.xl, wl.xm, wl.xl) are <= 2^-24 |w.x| each, i.e. below the rounding of the fp32
// accumulation itself.  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight fp32 MFMAs (64 cycles
// each) per 32x32x16 block: 2.67x less matrix-pipe time at fp32 accuracy.  Inputs, outputs, the f3 spill,
// the global feature and all tables stay fp32; the weights are the SAME fp32 values the fp32 path packs
// (BatchNorm scale folded in double, one rounding), split at pack time.
//
// Dataflow: one wave = 32 points; activations stay in registers as fp32 accumulator tiles and are split
// into bf16 parts one k-step (16 channels = registers 8s..8s+7 of a tile) at a time, right where they are
// consumed.  All layers run k-step-outer with at most four live output tiles (64 accumulator registers):
// 128->256 and 256->256 take two passes over their input (which is split twice: ~25 % more vector-ALU work
// for those layers, in exchange for 256 registers per wave, i.e. two waves per SIMD).  Weight stream: 1-KiB
// fragments (64 lanes x 8 bf16), three per (k-step, output tile) -- h, m, l -- in consumption order through
// the four-deep LDS ring of p2n_ring_bf16.h (8 waves = 256 points per workgroup).
#include "p2n_ring_bf16.h"

// ------------------------------------------------------------------ image layout
// encoder stream (fragments): L0 [ct][part] 6 | L1 [ks][ct][part] 24 | L2 48 | L3 96 | L4 192 |
// L5 four passes x 96  = 750.   head stream: L6 [ks][ct][part] 384 | L7 192 | L8: one chunk per output tile
// ([ks][part], 24 of the chunk's 32 fragments used), from chunk 18.
#define X3_ENC_FRAGS 750
#define X3_L1 6
#define X3_L2 30
#define X3_L3 78
#define X3_L4 174
#define X3_L5 366
#define X3_L7 384
#define X3_L8_CHUNK 18
#define X3_HEAD_FRAGS (X3_L8_CHUNK * P2N_CHUNK)   // layers 6 and 7: 576 fragments, linear

struct P2nLayoutX {
    int F, E, ct8;
    long raw_w[P2N_NLAYER], raw_b[P2N_NLAYER], raw_g[P2N_NLAYER], raw_beta[P2N_NLAYER], raw_mean[P2N_NLAYER],
        raw_var[P2N_NLAYER];
    int cin[P2N_NLAYER], cout[P2N_NLAYER], cpad[P2N_NLAYER];
    long enc_stream, head_stream, head_frags;
    long scale[P2N_NLAYER], shift[P2N_NLAYER];
    long w6g;
    long total;   // floats
};

static inline P2nLayoutX p2n_make_layout_x(int F, int E) {
    const P2nLayout A = p2n_make_layout(F, E);
    P2nLayoutX L;
    L.F = F; L.E = E; L.ct8 = A.cpad[8] / 32;
    for (int l = 0; l < P2N_NLAYER; ++l) {
        L.raw_w[l] = A.raw_w[l]; L.raw_b[l] = A.raw_b[l]; L.raw_g[l] = A.raw_g[l]; L.raw_beta[l] = A.raw_beta[l];
        L.raw_mean[l] = A.raw_mean[l]; L.raw_var[l] = A.raw_var[l];
        L.cin[l] = A.cin[l]; L.cout[l] = A.cout[l]; L.cpad[l] = A.cpad[l];
    }
    long o = 0;
    L.enc_stream = o;
    o += ((X3_ENC_FRAGS + P2N_CHUNK - 1) / P2N_CHUNK + 3) * (long)P2N_CHUNK * P2N_FRAG_FLOATS;   // + the ring's look-ahead
    L.head_stream = o; L.head_frags = (long)(X3_L8_CHUNK + L.ct8) * P2N_CHUNK;
    o += (X3_L8_CHUNK + L.ct8 + 3) * (long)P2N_CHUNK * P2N_FRAG_FLOATS;
    for (int l = 0; l < P2N_NLAYER; ++l) { L.scale[l] = o; o += L.cpad[l]; L.shift[l] = o; o += L.cpad[l]; }
    L.w6g = o; o += 256L * 256;
    L.total = o;
    return L;
}

// fragment f of a stream -> (layer, output tile, k-step, part); l = -1: padding
__device__ __forceinline__ void x3_decode(bool head, long f, int ct8n, int &l, int &ct, int &ks, int &part) {
    l = -1; ct = ks = part = 0;
    auto kouter = [&](long q, int CT) { part = (int)(q % 3); q /= 3; ct = (int)(q % CT); ks = (int)(q / CT); };
    if (!head) {
        if (f < X3_L1) { l = 0; part = (int)(f % 3); ct = (int)(f / 3); }
        else if (f < X3_L2) { l = 1; kouter(f - X3_L1, 2); }
        else if (f < X3_L3) { l = 2; kouter(f - X3_L2, 4); }
        else if (f < X3_L4) { l = 3; kouter(f - X3_L3, 4); }
        else if (f < X3_L5) { l = 4; kouter(f - X3_L4, 8); }
        else if (f < X3_ENC_FRAGS) { l = 5; const long q = f - X3_L5; kouter(q % 96, 2); ct += 2 * (int)(q / 96); }
    } else {
        if (f < X3_L7) { l = 6; kouter(f, 8); }
        else if (f < X3_L8_CHUNK * P2N_CHUNK) { l = 7; kouter(f - X3_L7, 4); }
        else {
            const long q = f - X3_L8_CHUNK * P2N_CHUNK;
            const int t = (int)(q / P2N_CHUNK), r = (int)(q % P2N_CHUNK);
            if (t < ct8n && r < 24) { l = 8; ct = t; ks = r / 3; part = r % 3; }
        }
    }
}

// one float slot of the image = two bf16 of a weight fragment, or one fp32 table / w6g entry
__global__ void p2n_pack_x3_kernel(const float *__restrict__ raw, float *__restrict__ packed, P2nLayoutX L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.total) return;
    float v = 0.f;
    if (i >= L.w6g) {
        const long r = i - L.w6g;
        v = raw[L.raw_w[6] + (r / 256) * 512 + 256 + (r % 256)];
    } else if (i >= L.scale[0]) {
        long r = i - L.scale[0];
        for (int l = 0; l < P2N_NLAYER; ++l) {
            const int cp = L.cpad[l];
            if (r < 2L * cp) {
                const int c = (int)(r % cp);
                const bool is_shift = r >= cp;
                if (c < L.cout[l]) {
                    const double b = raw[L.raw_b[l] + c];
                    if (l < 8) {
                        const double s = (double)raw[L.raw_g[l] + c] / sqrt((double)raw[L.raw_var[l] + c] + 1e-5);
                        const double t = (b - (double)raw[L.raw_mean[l] + c]) * s + (double)raw[L.raw_beta[l] + c];
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
        const bool head = i >= L.head_stream;
        const long r = i - (head ? L.head_stream : L.enc_stream);
        const long f = r / P2N_FRAG_FLOATS;
        const int e = (int)(r % P2N_FRAG_FLOATS);        // float slot inside the fragment: lane*4 + pair
        const int lane = e >> 2, pr = e & 3, row = lane & 31, h = lane >> 5;
        int l, ct, ks, part;
        x3_decode(head, f, L.ct8, l, ct, ks, part);
        if (l >= 0) {
            unsigned bits[2];
            for (int q = 0; q < 2; ++q) {
                const int j = 2 * pr + q;
                // layer 0 reads x in natural order (k = 8h + j); every other layer reads an accumulator tile
                const int k = (l == 0) ? 8 * h + j : 32 * (ks >> 1) + (j & 3) + 8 * (2 * (ks & 1) + (j >> 2)) + 4 * h;
                const int co = 32 * ct + row;
                const int klim = (l == 6) ? 256 : L.cin[l];
                float w = 0.f;
                if (co < L.cout[l] && k < klim) {
                    // the fp32 path's packed weight (BatchNorm scale folded in double, ONE rounding to float) ...
                    double wd = raw[L.raw_w[l] + (long)co * L.cin[l] + k];
                    if (l < 8) wd *= (double)raw[L.raw_g[l] + co] / sqrt((double)raw[L.raw_var[l] + co] + 1e-5);
                    w = (float)wd;
                }
                // ... as the exact sum of three bf16 parts
                const float hi = bf16_round(w);
                const float r1 = w - hi;
                const float mi = bf16_round(r1);
                const float r2 = r1 - mi;
                bits[q] = bf16_bits(part == 0 ? hi : part == 1 ? mi : r2);
            }
            v = __uint_as_float(bits[0] | (bits[1] << 16));
        }
    }
    packed[i] = v;
}

// ------------------------------------------------------------------ split and multiply
// registers r0..r0+7 of an accumulator tile = the fragment of one k-step -> its three bf16 parts
__device__ __forceinline__ void split8(const f32x16 &v, int r0, bf16x8 &xh, bf16x8 &xm, bf16x8 &xl) {
#if defined(X3_EXP) && (X3_EXP & 1)   // timing experiment (wrong results): no split arithmetic
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 a = {v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3]}, b = {v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7]};
    xh = __builtin_bit_cast(bf16x8, a); xm = __builtin_bit_cast(bf16x8, b); xl = xh;
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = v[r0 + j];
        xh[j] = (__bf16)a;
        const float r1 = a - (float)xh[j];     // exact
        xm[j] = (__bf16)r1;
        const float r2 = r1 - (float)xm[j];    // exact
        xl[j] = (__bf16)r2;
    }
}

// the six products of one (k-step, output tile), operands already in registers (layer 8's runtime loop)
__device__ __forceinline__ f32x16 mma6(f32x16 acc, const bf16x8 wh, const bf16x8 wm, const bf16x8 wl, const bf16x8 xh,
                                       const bf16x8 xm, const bf16x8 xl) {
    acc = MFMAB(wh, xl, acc);
    acc = MFMAB(wh, xm, acc);
    acc = MFMAB(wh, xh, acc);
    acc = MFMAB(wm, xm, acc);
    acc = MFMAB(wm, xh, acc);
    acc = MFMAB(wl, xh, acc);
    return acc;
}

// What a wave carries along the weight stream: the ring, two LDS base addresses (a ds_read's immediate offset
// reaches 64 KiB, the ring is 128 KiB: without the second base every other read costs a vector add) and the three
// fragment registers of the CURRENT step, which are refilled in place for the next step while this step multiplies.
typedef const __attribute__((address_space(3))) float x3_lds_cf;
typedef const __attribute__((address_space(3))) bf16x8 x3_lds_frag;
struct X3Ctx {
    const float *stream;
    float *wbuf;
    int wave, lane;
    x3_lds_cf *wlo, *whi;     // wbuf + lane*4 floats, and the same + 64 KiB (explicit LDS pointers: ds_read, never flat)
    bf16x8 wh, wm, wl;
};

__device__ __forceinline__ void x3_ctx_init(X3Ctx &c, const float *stream, float *wbuf, int wave, int lane) {
    c.stream = stream; c.wbuf = wbuf; c.wave = wave; c.lane = lane;
    c.wlo = (x3_lds_cf *)wbuf + lane * 4;
    x3_lds_cf *hi = c.wlo + 2 * BCHUNK_FLOATS;
    asm volatile("" : "+v"(hi));   // opaque: kept in a register instead of being re-derived per read
    c.whi = hi;
}

// chunk gate at fragment f; chunk 0 is entered explicitly at the top of each kernel (its barrier is also what makes
// the tables visible, so it has to precede the first accumulator start)
__device__ __forceinline__ void gate_x(int f, X3Ctx &c) {
#if defined(X3_EXP) && (X3_EXP & 2)   // timing experiment (wrong results): no chunk hand-over (no barrier, no DMA)
    return;
#endif
    if (f % P2N_CHUNK == 0 && f != 0) enter_chunk_b(f / P2N_CHUNK, c.stream, c.wbuf, c.wave, c.lane);
}
__device__ __forceinline__ bf16x8 frag_x(const X3Ctx &c, int f) {
    const int slot = (f / P2N_CHUNK) % RING;
    x3_lds_cf *base = slot >= 2 ? c.whi : c.wlo;
    return *(x3_lds_frag *)(base + (slot & 1) * BCHUNK_FLOATS + (f % P2N_CHUNK) * P2N_FRAG_FLOATS);
}
// the fragments of the stream's first step (behind the kernel's explicit enter_chunk_b(0))
__device__ __forceinline__ void x3_prime(X3Ctx &c) {
    c.wh = frag_x(c, 0); c.wm = frag_x(c, 1); c.wl = frag_x(c, 2);
}

// One step = fragments F, F+1, F+2 (h, m, l of one (k-step, output tile)), already in c.wh/wm/wl, and their six
// MFMAs.  Each fragment register is refilled with the NEXT step's fragment (F+3..F+5: the stream is linear) as soon as
// its last MFMA has issued, so every LDS read has three to five MFMAs (96-160 cycles) to land and no second set of
// fragment registers is needed.  Reads ascend along the stream, so a chunk gate between them never leaves a read of
// the chunk that the gate hands back to the DMA.
#define X3_PIN() __builtin_amdgcn_sched_barrier(0x6)
template <int F, bool LAST>
__device__ __forceinline__ f32x16 step_x3(X3Ctx &c, f32x16 acc, const bf16x8 xh, const bf16x8 xm, const bf16x8 xl) {
    // X3_PIN: vector- and scalar-ALU instructions may cross, MFMAs and LDS reads keep this order (left alone, the
    // scheduler gathers the reads in front of their first use and the latency is exposed again)
    acc = MFMAB(c.wh, xl, acc);
    acc = MFMAB(c.wh, xm, acc);
    acc = MFMAB(c.wh, xh, acc);
    if (!LAST) { gate_x(F + 3, c); c.wh = frag_x(c, F + 3); }
    X3_PIN();
    acc = MFMAB(c.wm, xm, acc);
    acc = MFMAB(c.wm, xh, acc);
    if (!LAST) { gate_x(F + 4, c); c.wm = frag_x(c, F + 4); }
    X3_PIN();
    acc = MFMAB(c.wl, xh, acc);
    if (!LAST) { gate_x(F + 5, c); c.wl = frag_x(c, F + 5); }
    X3_PIN();
    return acc;
}

template <int F0, int CT, int KS_I, int FEND, int CT_I, int CT_END>
struct TileLoop {   // steps of output tiles CT_I .. CT_END-1 of k-step KS_I
    static __device__ __forceinline__ void run(X3Ctx &c, f32x16 (&acc)[CT], const bf16x8 xh, const bf16x8 xm, const bf16x8 xl) {
        if constexpr (CT_I < CT_END) {
            constexpr int F = F0 + (KS_I * CT + CT_I) * 3;
            acc[CT_I] = step_x3<F, F + 3 >= FEND>(c, acc[CT_I], xh, xm, xl);
            TileLoop<F0, CT, KS_I, FEND, CT_I + 1, CT_END>::run(c, acc, xh, xm, xl);
        }
    }
};

// One pass of a layer on a wave's 32 points, k-step outer: CT live output tiles (acc comes in holding the
// layer's shift), KS k-steps of the fp32 input tiles `in`, fragments [ks][ct][part] from stream index F0; FEND = the
// end of the whole stream (the last step prefetches nothing).  The parts of k-step ks+1 are formed behind the first
// step of k-step ks, so their ~45 vector-ALU instructions sit in the MFMA gaps of the steps that follow; a scheduling
// barrier per k-step keeps the compiler from hoisting later splits (12 live registers each) further ahead.
template <int KS, int CT, int F0, int FEND, int KS_I = 0>
struct LayerX3 {
    static __device__ __forceinline__ void steps(X3Ctx &c, const f32x16 (&in)[KS / 2], f32x16 (&acc)[CT], const bf16x8 xh,
                                                 const bf16x8 xm, const bf16x8 xl) {
        TileLoop<F0, CT, KS_I, FEND, 0, 1>::run(c, acc, xh, xm, xl);
        bf16x8 nh = xh, nm = xm, nl = xl;
        if constexpr (KS_I + 1 < KS) split8(in[(KS_I + 1) >> 1], 8 * ((KS_I + 1) & 1), nh, nm, nl);
        TileLoop<F0, CT, KS_I, FEND, 1, CT>::run(c, acc, xh, xm, xl);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KS_I + 1 < KS) LayerX3<KS, CT, F0, FEND, KS_I + 1>::steps(c, in, acc, nh, nm, nl);
    }
    static __device__ __forceinline__ void run(X3Ctx &c, const f32x16 (&in)[KS / 2], f32x16 (&acc)[CT]) {
        bf16x8 xh, xm, xl;
        split8(in[0], 0, xh, xm, xl);
        steps(c, in, acc, xh, xm, xl);
    }
};

template <int CT>
__device__ __forceinline__ void shift_tiles_x(f32x16 (&t)[CT], const float *shift, int ct0, int h) {
#pragma unroll
    for (int c = 0; c < CT; ++c) t[c] = shift_tile_b(shift, ct0 + c, h);
}

// ------------------------------------------------------------------ encoder
struct EncArgsX {
    const float *x;
    const float *packed;
    float *spill;     // fp32, register-image order [group][8 ct][4 g][64 lanes][4] = 32 KiB per 32 points (as p2n_fwd.hip)
    float *partial;   // [B*tiles][256]
    float *f1, *f2, *f3;
    int N, F, tiles;
    long stream, tab;
    int tab_n;
    int shift[6];
};

#define ENCX_LDS_FLOATS (RING * BCHUNK_FLOATS)

template <bool FEATS>
__global__ __launch_bounds__(NWB * 64, 2) void p2n_encoder_x3_kernel(EncArgsX A) {
    // the weight ring is the only dynamic LDS (see p2n_encoder_bf16_kernel: separate static arrays keep the compiler
    // from draining the DMA look-ahead in front of every table read)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ unsigned smax[256];
    __shared__ __attribute__((aligned(16))) float tab[P2N_TAB_MAX];
    float *wbuf = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / A.tiles, tile = blockIdx.x % A.tiles;
    const long n = ((long)tile * NWB + wave) * P2N_PTS_PER_WAVE + p;
    const float *stream = A.packed + A.stream;
    for (int i = threadIdx.x; i < A.tab_n; i += NWB * 64) tab[i] = A.packed[A.tab + i];   // queued before the DMA
    if (threadIdx.x < 256) smax[threadIdx.x] = 0u;
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) stage_chunk_b(stream, wbuf, c, wave, lane);
    const bool valid = n < A.N;
    const long grp = (long)blockIdx.x * NWB + wave;

    // layer 0 operand (natural k order): lane half 0 holds features 0..7, half 1 the zero padding of K = 16
    bf16x8 xh, xm, xl;
    {
        f32x16 xv;
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = (valid && h == 0 && j < A.F) ? A.x[((long)b * A.N + n) * A.F + j] : 0.f;
        split8(xv, 0, xh, xm, xl);
    }
    enter_chunk_b(0, stream, wbuf, wave, lane);   // chunk 0 landed; tables and smax visible
    X3Ctx c;
    x3_ctx_init(c, stream, wbuf, wave, lane);
    x3_prime(c);
    // ---- layer 0: F -> 64                fragments 0..5
    f32x16 a0[2];
    shift_tiles_x(a0, tab + A.shift[0], 0, h);
    a0[0] = step_x3<0, false>(c, a0[0], xh, xm, xl);
    a0[1] = step_x3<3, false>(c, a0[1], xh, xm, xl);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) a0[ct] = relu_tile_b(a0[ct]);
    // ---- layer 1: 64 -> 64
    f32x16 a1[2];
    shift_tiles_x(a1, tab + A.shift[1], 0, h);
    LayerX3<4, 2, X3_L1, X3_ENC_FRAGS>::run(c, a0, a1);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        a1[ct] = relu_tile_b(a1[ct]);
        if (FEATS) if (A.f1) store_feat_b<64>(A.f1, a1[ct], b, ct, h, n, A.N);
    }
    // ---- layer 2: 64 -> 128
    f32x16 a2[4];
    shift_tiles_x(a2, tab + A.shift[2], 0, h);
    LayerX3<4, 4, X3_L2, X3_ENC_FRAGS>::run(c, a1, a2);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) a2[ct] = relu_tile_b(a2[ct]);
    // ---- layer 3: 128 -> 128
    f32x16 a3[4];
    shift_tiles_x(a3, tab + A.shift[3], 0, h);
    LayerX3<8, 4, X3_L3, X3_ENC_FRAGS>::run(c, a2, a3);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        a3[ct] = relu_tile_b(a3[ct]);
        if (FEATS) if (A.f2) store_feat_b<128>(A.f2, a3[ct], b, ct, h, n, A.N);
    }
    // ---- layer 4: 128 -> 256, all eight output tiles live (64 input + 128 accumulator registers)
    f32x16 a4[8];
    shift_tiles_x(a4, tab + A.shift[4], 0, h);
    LayerX3<8, 8, X3_L4, X3_ENC_FRAGS>::run(c, a3, a4);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) a4[ct] = relu_tile_b(a4[ct]);
    // ---- layer 5: 256 -> 256, four passes of two output tiles (its 128 input registers stay live: 32 accumulator
    // registers leave room for the fragment rotation without spilling); results leave through the fp32 spill and the
    // channel max
    float *sp = A.spill + grp * (256L * P2N_PTS_PER_WAVE) + lane * 4;
    unsigned wm[8];   // this lane's row maxima, one per output tile (they reach LDS once, behind the stream)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x16 a5[2];
        shift_tiles_x(a5, tab + A.shift[5], 2 * q, h);
        if (q == 0) LayerX3<16, 2, X3_L5, X3_ENC_FRAGS>::run(c, a4, a5);
        else if (q == 1) LayerX3<16, 2, X3_L5 + 96, X3_ENC_FRAGS>::run(c, a4, a5);
        else if (q == 2) LayerX3<16, 2, X3_L5 + 192, X3_ENC_FRAGS>::run(c, a4, a5);
        else LayerX3<16, 2, X3_L5 + 288, X3_ENC_FRAGS>::run(c, a4, a5);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int ct = 2 * q + c2;
            const f32x16 v = relu_tile_b(a5[c2]);
            if (FEATS) if (A.f3) store_feat_b<256>(A.f3, v, b, ct, h, n, A.N);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                *reinterpret_cast<f32x4 *>(sp + (ct * 4 + g) * P2N_FRAG_FLOATS) = o;
            }
            wm[ct] = tile_rowmax(v, valid, lane);
        }
    }
    __syncthreads();   // drains the look-ahead DMAs (before the workgroup's LDS is released and before the atomics)
    {
        const int i = lane & 15;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) atomicMax(&smax[32 * ct + 8 * (i >> 2) + 4 * h + (i & 3)], wm[ct]);
    }
    __syncthreads();
    if (threadIdx.x < 256) A.partial[(long)blockIdx.x * 256 + threadIdx.x] = __uint_as_float(smax[threadIdx.x]);
}

// ------------------------------------------------------------------ head
struct HeadArgsX {
    const float *packed;
    const float *spill;
    const float *shift6c;
    float *emb;
    int N, E, tiles, ct8;
    long stream, tab;
    int tab_n;
    int shift7, shift8;
};

// k-step KS_I of layer 6: its fp32 operand registers come from the spill (two 16-byte loads per lane, fetched two
// k-steps ahead and split one k-step ahead, behind the first step) and meet all eight output tiles
template <int KS_I>
struct Layer6X3 {
    static __device__ __forceinline__ void run(X3Ctx &c, const f32x4 *sp, f32x4 n0, f32x4 n1, f32x16 (&acc)[8],
                                               const bf16x8 xh, const bf16x8 xm, const bf16x8 xl) {
        // n0, n1: the raw registers of k-step KS_I + 1 (slots 2ks, 2ks+1 of the spill: registers 8s..8s+7 of tile ct
        // are its groups g = 2s, 2s+1)
        f32x4 f0 = n0, f1 = n1;
        if constexpr (KS_I + 2 < 16) {
            f0 = sp[(2 * (KS_I + 2)) * P2N_WAVE];
            f1 = sp[(2 * (KS_I + 2) + 1) * P2N_WAVE];
        }
        TileLoop<0, 8, KS_I, X3_HEAD_FRAGS, 0, 1>::run(c, acc, xh, xm, xl);
        bf16x8 nh = xh, nm = xm, nl = xl;
        if constexpr (KS_I + 1 < 16) {
            f32x16 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = n0[j]; v[4 + j] = n1[j]; }
            split8(v, 0, nh, nm, nl);
        }
        TileLoop<0, 8, KS_I, X3_HEAD_FRAGS, 1, 8>::run(c, acc, xh, xm, xl);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KS_I + 1 < 16) Layer6X3<KS_I + 1>::run(c, sp, f0, f1, acc, nh, nm, nl);
    }
};

__global__ __launch_bounds__(NWB * 64, 2) void p2n_head_x3_kernel(HeadArgsX A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // the weight ring only (see the encoder)
    __shared__ __attribute__((aligned(16))) float tab[2 * (256 + 128 + 256)];
    __shared__ __attribute__((aligned(16))) float sh6[256];
    float *wbuf = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / A.tiles, tile = blockIdx.x % A.tiles;
    const long n0 = ((long)tile * NWB + wave) * P2N_PTS_PER_WAVE;
    const long grp = (long)blockIdx.x * NWB + wave;
    const float *stream = A.packed + A.stream;
    for (int i = threadIdx.x; i < A.tab_n; i += NWB * 64) tab[i] = A.packed[A.tab + i];
    if (threadIdx.x < 256) sh6[threadIdx.x] = A.shift6c[(long)b * 256 + threadIdx.x];

    // f3 of this wave's 32 points, exactly as the encoder's registers held it: slot (4ct + g) is 1 KiB
    const f32x4 *sp = reinterpret_cast<const f32x4 *>(A.spill + grp * (256L * P2N_PTS_PER_WAVE)) + lane;
    const f32x4 c0 = sp[0], c1 = sp[P2N_WAVE], d0 = sp[2 * P2N_WAVE], d1 = sp[3 * P2N_WAVE];
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) stage_chunk_b(stream, wbuf, c, wave, lane);   // behind the first f3 loads in the queue
    enter_chunk_b(0, stream, wbuf, wave, lane);   // chunk 0 landed; tables and this cloud's layer-6 shift visible: the accumulators start from them
    X3Ctx c;
    x3_ctx_init(c, stream, wbuf, wave, lane);
    x3_prime(c);

    // ---- layer 6: 256 (+ folded global half) -> 256     fragments 0..383
    f32x16 a6[8];
    shift_tiles_x(a6, sh6, 0, h);
    {
        f32x16 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = c0[j]; v[4 + j] = c1[j]; }
        bf16x8 xh, xm, xl;
        split8(v, 0, xh, xm, xl);
        Layer6X3<0>::run(c, sp, d0, d1, a6, xh, xm, xl);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) a6[c] = relu_tile_b(a6[c]);
    // ---- layer 7: 256 -> 128                             fragments 384..575
    f32x16 a7[4];
    shift_tiles_x(a7, tab + A.shift7, 0, h);
    LayerX3<16, 4, X3_L7, X3_HEAD_FRAGS>::run(c, a6, a7);
    bf16x8 x7h[8], x7m[8], x7l[8];   // layer 8 runs a runtime number of output tiles over the same input: split once
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x16 v = relu_tile_b(a7[c]);
        split8(v, 0, x7h[2 * c], x7m[2 * c], x7l[2 * c]);
        split8(v, 8, x7h[2 * c + 1], x7m[2 * c + 1], x7l[2 * c + 1]);
    }
    // ---- layer 8: 128 -> E, one chunk per output tile (24 fragments [ks][part]), chunks 18..
    const int E = A.E;
    const long n = n0 + p;
    const bool valid = n < A.N;
    float *dst = A.emb + ((long)b * A.N + n) * E;
    const bool vec = (E & 3) == 0;
#pragma unroll 1
    for (int ct8 = 0; ct8 < A.ct8; ++ct8) {
        const int ch = X3_L8_CHUNK + ct8;
        enter_chunk_b(ch, stream, wbuf, wave, lane);
        const float *wb = wbuf + (ch % RING) * BCHUNK_FLOATS + lane * 4;
        f32x16 acc = shift_tile_b(tab + A.shift8, ct8, h);   // the plain last layer: its bias
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(wb + (ks * 3) * P2N_FRAG_FLOATS);
            const bf16x8 wm = *reinterpret_cast<const bf16x8 *>(wb + (ks * 3 + 1) * P2N_FRAG_FLOATS);
            const bf16x8 wl = *reinterpret_cast<const bf16x8 *>(wb + (ks * 3 + 2) * P2N_FRAG_FLOATS);
            acc = mma6(acc, wh, wm, wl, x7h[ks], x7m[ks], x7l[ks]);
        }
        const f32x16 v = acc;
        if (valid) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = 32 * ct8 + 8 * g + 4 * h;
                if (vec) {
                    if (e0 < E) {
                        const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
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
    __syncthreads();   // drain the look-ahead DMAs before the workgroup's LDS is released
}

// ------------------------------------------------------------------ C ABI
extern "C" long p2n_packed_param_count_x3(int F, int E) {
    if (F < 1 || F > 8 || E < 1 || E > 256) return -1;
    return p2n_make_layout_x(F, E).total;
}

extern "C" int p2n_pack_eval_params_x3(const float *raw, int F, int E, float *packed, void *stream) {
    P2N_REQUIRE(raw && packed, "p2n_pack_eval_params_x3: null pointer");
    P2N_REQUIRE(F >= 1 && F <= 8, "p2n_pack_eval_params_x3: feature_dim %d not in [1,8]", F);
    P2N_REQUIRE(E >= 1 && E <= 256, "p2n_pack_eval_params_x3: embedding_dim %d not in [1,256]", E);
    const P2nLayoutX L = p2n_make_layout_x(F, E);
    hipLaunchKernelGGL(p2n_pack_x3_kernel, dim3((unsigned)((L.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw,
                       packed, L);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

static inline int tiles_per_cloud_x(int N) { return (N + NWB * P2N_PTS_PER_WAVE - 1) / (NWB * P2N_PTS_PER_WAVE); }

extern "C" size_t p2n_forward_eval_x3_workspace_bytes(int B, int N, int F, int E) {
    (void)F; (void)E;
    if (B < 1 || N < 1) return 0;
    const size_t tiles = (size_t)B * tiles_per_cloud_x(N);
    return sizeof(float) * (tiles * NWB * 256 * P2N_PTS_PER_WAVE + tiles * 256 + (size_t)B * 256);
}

template <typename K>
static int allow_big_lds_x(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)bytes));
    return P2N_OK;
}

extern "C" int p2n_forward_eval_x3(const float *x, int B, int N, int F, int E, const float *packed, float *emb,
                                   float *global_f, float *f1, float *f2, float *f3, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(B >= 0 && N >= 0, "p2n_forward_eval_x3: negative shape B=%d N=%d", B, N);
    P2N_REQUIRE(F >= 1 && F <= 8, "p2n_forward_eval_x3: feature_dim %d not in [1,8]", F);
    P2N_REQUIRE(E >= 1 && E <= 256, "p2n_forward_eval_x3: embedding_dim %d not in [1,256]", E);
    if (B == 0 || N == 0) return P2N_OK;
    P2N_REQUIRE(x && packed, "p2n_forward_eval_x3: null x/packed");
    P2N_REQUIRE(emb || global_f || f1 || f2 || f3, "p2n_forward_eval_x3: no output requested");
    P2N_REQUIRE((long)B * tiles_per_cloud_x(N) < (1L << 31), "p2n_forward_eval_x3: B*ceil(N/256) exceeds the grid limit");
    const size_t need = p2n_forward_eval_x3_workspace_bytes(B, N, F, E);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_forward_eval_x3: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    P2N_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)packed & 15) == 0,
                "p2n_forward_eval_x3: workspace and packed must be 16-byte aligned");
    if (emb && (E & 3) == 0) P2N_REQUIRE(((uintptr_t)emb & 15) == 0, "p2n_forward_eval_x3: emb must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const P2nLayoutX L = p2n_make_layout_x(F, E);
    const int tiles = tiles_per_cloud_x(N);
    const size_t ntiles = (size_t)B * tiles;
    float *spill = (float *)workspace;
    float *partial = spill + ntiles * NWB * 256 * P2N_PTS_PER_WAVE;
    float *shift6c = partial + ntiles * 256;

    EncArgsX ea;
    ea.x = x; ea.packed = packed; ea.spill = spill; ea.partial = partial;
    ea.f1 = f1; ea.f2 = f2; ea.f3 = f3;
    ea.N = N; ea.F = F; ea.tiles = tiles; ea.stream = L.enc_stream;
    ea.tab = L.scale[0]; ea.tab_n = (int)(L.scale[6] - L.scale[0]);
    for (int l = 0; l < 6; ++l) ea.shift[l] = (int)(L.shift[l] - L.scale[0]);
    const size_t enc_lds = sizeof(float) * ENCX_LDS_FLOATS;
    int ps = p2n_prof_begin(P2N_K_ENCODER, stream);
    if (f1 || f2 || f3) {
        int rc = allow_big_lds_x(p2n_encoder_x3_kernel<true>, enc_lds);
        if (rc) return rc;
        hipLaunchKernelGGL(p2n_encoder_x3_kernel<true>, dim3((unsigned)ntiles), dim3(NWB * 64), enc_lds, stream, ea);
    } else {
        int rc = allow_big_lds_x(p2n_encoder_x3_kernel<false>, enc_lds);
        if (rc) return rc;
        hipLaunchKernelGGL(p2n_encoder_x3_kernel<false>, dim3((unsigned)ntiles), dim3(NWB * 64), enc_lds, stream, ea);
    }
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    ps = p2n_prof_begin(P2N_K_MID, stream);
    hipLaunchKernelGGL(p2n_mid_b_kernel, dim3(B, 8), dim3(256), 0, stream, partial, tiles, packed, L.w6g, L.scale[6],
                       L.shift[6], shift6c, global_f);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    if (!emb) return P2N_OK;
    HeadArgsX ha;
    ha.packed = packed; ha.spill = spill; ha.shift6c = shift6c; ha.emb = emb;
    ha.N = N; ha.E = E; ha.tiles = tiles; ha.ct8 = L.ct8; ha.stream = L.head_stream;
    ha.tab = L.scale[6]; ha.tab_n = (int)(L.w6g - L.scale[6]);
    ha.shift7 = (int)(L.shift[7] - L.scale[6]); ha.shift8 = (int)(L.shift[8] - L.scale[6]);
    const size_t lds = sizeof(float) * RING * (size_t)BCHUNK_FLOATS;
    {
        int rc = allow_big_lds_x(p2n_head_x3_kernel, lds);
        if (rc) return rc;
    }
    ps = p2n_prof_begin(P2N_K_HEAD, stream);
    hipLaunchKernelGGL(p2n_head_x3_kernel, dim3((unsigned)ntiles), dim3(NWB * 64), lds, stream, ha);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
