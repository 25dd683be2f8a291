// Point2Net eval forward for gfx950 (MI355X): three launches.
//
//   p2n_encoder_kernel  layers 0..5 (reference point2net.py:9-34,36-43), f3 spill,
//                       per-workgroup channel max (reference point2net.py:46)
//   p2n_mid_kernel      per-cloud max of the partials -> global_f, and the global
//                       half of head layer 6 folded into a per-cloud shift
//                       (W6[:,256:] . g replaces the concat of reference
//                       point2net.py:78-79)
//   p2n_head_kernel     head layers 6..8 (reference point2net.py:57-65,82) and
//                       the (B,E,N)->(B,N,E) transpose of :85 as a coalesced store
//
// Design: one wave owns 32 points (one 32-column block of v_mfma_f32_32x32x2_f32)
// and carries their activations through every layer IN REGISTERS.  With D = W.X
// (A = weights, rows = output channels; B = activations, columns = points) the
// accumulator layout of layer l (lane = point, registers = channels) is exactly the
// B-operand layout of layer l+1 once the reduction index is permuted
// (k = 32*kt + 8*g + 4*h + j  <->  register 4g+j of tile kt, lane half h), and the
// permutation is absorbed into the weight image at pack time.  So activations never
// touch LDS or HBM between layers; the only HBM traffic is x in, the f3 spill across
// the global-max dependency, and emb out.  Weights stream from L2 as 1-KiB
// lane-linear fragments (one global_load_dwordx4 per 4 MFMAs per wave) through a
// register ring P2N_RING fragments deep.  fp32 in, fp32 accumulate: exact-f32 MFMA.
#include "p2n_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

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
                    if (f < nf) { l = ll; ct = (int)(f / (L.kpad[ll] / 8)); G = (int)(f % (L.kpad[ll] / 8)); break; }
                    f -= nf;
                }
            }
        } else if (f < L.head_frags) {
            if (f < 8 * 48) {
                int ct6 = (int)(f / 48), q = (int)(f % 48);
                if (q < 32) { l = 6; ct = ct6; G = q; }
                else { l = 7; ct = (q - 32) / 4; G = 4 * ct6 + ((q - 32) & 3); }
            } else {
                long q = f - 8 * 48;
                l = 8; ct = (int)(q / 16); G = (int)(q % 16);
            }
        }
        if (l >= 0) {
            int co = 32 * ct + row, k = 8 * G + 4 * h + j;
            int klim = (l == 6) ? 256 : L.cin[l];
            if (co < L.cout[l] && k < klim) v = raw[L.raw_w[l] + (long)co * L.cin[l] + k];
        }
    }
    packed[i] = v;
}

// ------------------------------------------------------------------ helpers
// y = relu(acc*scale + shift) on one 32x32 accumulator tile; channel of register
// 4g+j in lane half h is 32*ct + 8g + 4h + j.
template <bool RELU>
__device__ __forceinline__ f32x16 epilogue(const f32x16 acc, const float *__restrict__ scale,
                                           const float *__restrict__ shift, int ct, int h) {
    f32x16 out;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 32 * ct + 8 * g + 4 * h);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 32 * ct + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = fmaf(acc[4 * g + j], sc[j], sh[j]);
            out[4 * g + j] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
    return out;
}

// One pointwise layer on a wave's 32 points.  Consumes CT*KT*4 consecutive weight
// fragments starting at stream index F0 (compile-time, so ring slots are static).
template <int KT, int CT, int F0, typename Sink>
__device__ __forceinline__ void layer_stream(const f32x4 *__restrict__ wl, f32x4 (&ring)[P2N_RING],
                                             const f32x16 (&in)[KT], Sink &&sink) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = F0 + (ct * KT + kt) * 4 + g;
                const f32x4 a = ring[f % P2N_RING];
                ring[f % P2N_RING] = wl[(long)(f + P2N_RING) * P2N_WAVE];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], in[kt][4 * g + j], acc);
            }
        }
        sink(ct, acc);
    }
}

// max over the 32 lanes that share a lane half; result valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half_wave_max(float v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true);  // row_half_mirror
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true);  // row_mirror
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1,3
    v = fmaxf(v, __int_as_float(t));
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

template <bool FEATS>
__global__ __launch_bounds__(256) void p2n_encoder_kernel(EncArgs A) {
    // one LDS array: [4][256] wave maxima, then the scale/shift tables (epilogues read
    // them with ds_read so they never wait on the global weight stream's vmcnt)
    __shared__ __attribute__((aligned(16))) float lds[P2N_WAVES_PER_WG * 256 + P2N_TAB_MAX];
    float (*smax)[256] = reinterpret_cast<float (*)[256]>(lds);
    float *tab = lds + P2N_WAVES_PER_WG * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / A.tiles, tile = blockIdx.x % A.tiles;
    const long n = ((long)tile * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE + p;
    for (int i = threadIdx.x; i < A.tab_n; i += 256) tab[i] = A.packed[A.tab + i];
    __syncthreads();
    const bool valid = n < A.N;
    const long grp = (long)blockIdx.x * P2N_WAVES_PER_WG + wave;

    const float *pk = A.packed;
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(pk + A.stream) + lane;
    f32x4 ring[P2N_RING];
#pragma unroll
    for (int i = 0; i < P2N_RING; ++i) ring[i] = wl[(long)i * P2N_WAVE];

    // layer 0 operand: x[n][4h+j], zero beyond F (K padded to 8) and beyond the cloud
    f32x4 xin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 4 * h + j;
        xin[j] = (valid && k < A.F) ? A.x[((long)b * A.N + n) * A.F + k] : 0.f;
    }

    f32x16 a0[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const f32x4 a = ring[ct % P2N_RING];
        ring[ct % P2N_RING] = wl[(long)(ct + P2N_RING) * P2N_WAVE];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], xin[j], acc);
        a0[ct] = epilogue<true>(acc, tab + A.scale[0], tab + A.shift[0], ct, h);
    }
    // stream positions: L0 2 frags, L1 16, L2 32, L3 64, L4 128, L5 256
    f32x16 a1[2];
    layer_stream<2, 2, 2>(wl, ring, a0, [&](int ct, const f32x16 acc) {
        a1[ct] = epilogue<true>(acc, tab + A.scale[1], tab + A.shift[1], ct, h);
        if (FEATS) if (A.f1) store_feat<64>(A.f1, a1[ct], b, ct, h, n, A.N);
    });
    f32x16 a2[4];
    layer_stream<2, 4, 18>(wl, ring, a1, [&](int ct, const f32x16 acc) {
        a2[ct] = epilogue<true>(acc, tab + A.scale[2], tab + A.shift[2], ct, h);
    });
    f32x16 a3[4];
    layer_stream<4, 4, 50>(wl, ring, a2, [&](int ct, const f32x16 acc) {
        a3[ct] = epilogue<true>(acc, tab + A.scale[3], tab + A.shift[3], ct, h);
        if (FEATS) if (A.f2) store_feat<128>(A.f2, a3[ct], b, ct, h, n, A.N);
    });
    f32x16 a4[8];
    layer_stream<4, 8, 114>(wl, ring, a3, [&](int ct, const f32x16 acc) {
        a4[ct] = epilogue<true>(acc, tab + A.scale[4], tab + A.shift[4], ct, h);
    });
    float *sp = A.spill + grp * (256L * P2N_PTS_PER_WAVE) + lane * 4;
    layer_stream<8, 8, 242>(wl, ring, a4, [&](int ct, const f32x16 acc) {
        f32x16 v = epilogue<true>(acc, tab + A.scale[5], tab + A.shift[5], ct, h);
        if (FEATS) if (A.f3) store_feat<256>(A.f3, v, b, ct, h, n, A.N);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            *reinterpret_cast<f32x4 *>(sp + (ct * 4 + g) * P2N_FRAG_FLOATS) = o;
        }
        // post-ReLU values are >= 0, so 0 is the neutral element for masked points
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = half_wave_max(valid ? v[i] : 0.f);
        if (p == 16) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                *reinterpret_cast<f32x4 *>(&smax[wave][32 * ct + 8 * g + 4 * h]) = o;
            }
        }
    });
    __syncthreads();
    {
        const int c = threadIdx.x;
        float m = fmaxf(fmaxf(smax[0][c], smax[1][c]), fmaxf(smax[2][c], smax[3][c]));
        A.partial[(long)blockIdx.x * 256 + c] = m;
    }
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
    float m = 0.f;
    for (int i = 0; i < tiles; ++i) m = fmaxf(m, partial[((long)b * tiles + i) * 256 + t]);
    g[t] = m;
    if (blockIdx.y == 0 && global_f) global_f[(long)b * 256 + t] = m;
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(&g[4 * lane]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int co = blockIdx.y * 32 + wave * 8 + i;
        const f32x4 w = *reinterpret_cast<const f32x4 *>(packed + w6g + (long)co * 256 + 4 * lane);
        float s = (w[0] * gv[0] + w[1] * gv[1]) + (w[2] * gv[2] + w[3] * gv[3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) shift6c[(long)b * 256 + co] = fmaf(s, packed[scale6 + co], packed[shift6 + co]);
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

__global__ __launch_bounds__(256) void p2n_head_kernel(HeadArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / A.tiles, tile = blockIdx.x % A.tiles;
    const long n0 = ((long)tile * P2N_WAVES_PER_WG + wave) * P2N_PTS_PER_WAVE;
    const long grp = (long)blockIdx.x * P2N_WAVES_PER_WG + wave;
    const float *pk = A.packed;
    // LDS carve: [tab_n] scale/shift tables, [256] this cloud's layer-6 shift, then the
    // four per-wave output tiles
    float *tab = smem;
    float *sh6 = smem + A.tab_n;
    for (int i = threadIdx.x; i < A.tab_n; i += 256) tab[i] = pk[A.tab + i];
    sh6[threadIdx.x] = A.shift6c[(long)b * 256 + threadIdx.x];
    __syncthreads();
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(pk + A.stream) + lane;
    f32x4 ring[P2N_RING];
#pragma unroll
    for (int i = 0; i < P2N_RING; ++i) ring[i] = wl[(long)i * P2N_WAVE];

    f32x16 f3in[8];
    {
        const f32x4 *sp = reinterpret_cast<const f32x4 *>(A.spill + grp * (256L * P2N_PTS_PER_WAVE)) + lane;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = sp[(ct * 4 + g) * P2N_WAVE];
#pragma unroll
                for (int j = 0; j < 4; ++j) f3in[ct][4 * g + j] = v[j];
            }
    }
    f32x16 acc7[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc7[c][i] = 0.f;

    // layer 6 tile ct6 (32 frags) feeds, as k-slice ct6, the four layer-7 tiles (16 frags)
#pragma unroll
    for (int ct6 = 0; ct6 < 8; ++ct6) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = ct6 * 48 + kt * 4 + g;
                const f32x4 a = ring[f % P2N_RING];
                ring[f % P2N_RING] = wl[(long)(f + P2N_RING) * P2N_WAVE];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], f3in[kt][4 * g + j], acc);
            }
        const f32x16 a6 = epilogue<true>(acc, tab + A.scale6, sh6, ct6, h);
#pragma unroll
        for (int ct7 = 0; ct7 < 4; ++ct7)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = ct6 * 48 + 32 + ct7 * 4 + g;
                const f32x4 a = ring[f % P2N_RING];
                ring[f % P2N_RING] = wl[(long)(f + P2N_RING) * P2N_WAVE];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc7[ct7] = MFMA32(a[j], a6[4 * g + j], acc7[ct7]);
            }
    }
    f32x16 a7[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a7[c] = epilogue<true>(acc7[c], tab + A.scale7, tab + A.shift7, c, h);

    // layer 8: runtime number of 32-channel tiles, results staged in LDS as [point][channel]
    const int rs = A.ct8 * 32 + 4;   // padded row stride (floats): conflict-free b128 writes
    float *tilebuf = smem + A.tab_n + 256 + (long)wave * 32 * rs;
    const f32x4 *wl8 = wl + 384L * P2N_WAVE;
#pragma unroll 1
    for (int ct8 = 0; ct8 < A.ct8; ++ct8) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = kt * 4 + g;   // 384 % RING == 0 and 16 % RING == 0: slots stay static
                const f32x4 a = ring[f % P2N_RING];
                ring[f % P2N_RING] = wl8[(long)(f + P2N_RING) * P2N_WAVE];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], a7[kt][4 * g + j], acc);
            }
        const f32x16 v = epilogue<false>(acc, tab + A.scale8, tab + A.shift8, ct8, h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            *reinterpret_cast<f32x4 *>(&tilebuf[p * rs + 32 * ct8 + 8 * g + 4 * h]) = o;
        }
        wl8 += 16L * P2N_WAVE;
    }
    __syncthreads();
    // the wave's 32 points x E floats are one contiguous block of emb
    const int E = A.E;
    long rows = (long)A.N - n0;
    if (rows > 32) rows = 32;
    if (rows <= 0) return;
    float *dst = A.emb + ((long)b * A.N + n0) * E;
    if ((E & 3) == 0) {
        const int cpr = E >> 2;
        const int total = (int)rows * cpr;
        for (int c = lane; c < total; c += 64) {
            const int row = c / cpr, col = c - row * cpr;
            *reinterpret_cast<f32x4 *>(dst + (long)c * 4) = *reinterpret_cast<const f32x4 *>(&tilebuf[row * rs + col * 4]);
        }
    } else {
        const int total = (int)rows * E;
        for (int c = lane; c < total; c += 64) {
            const int row = c / E, col = c - row * E;
            dst[c] = tilebuf[row * rs + col];
        }
    }
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

extern "C" int p2n_forward_eval(const float *x, int B, int N, int F, int E, const float *packed, float *emb,
                                float *global_f, float *f1, float *f2, float *f3, void *workspace,
                                size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(B >= 0 && N >= 0, "p2n_forward_eval: negative shape B=%d N=%d", B, N);
    P2N_REQUIRE(F >= 1 && F <= 8, "p2n_forward_eval: feature_dim %d not in [1,8]", F);
    P2N_REQUIRE(E >= 1 && E <= 256, "p2n_forward_eval: embedding_dim %d not in [1,256]", E);
    if (B == 0 || N == 0) return P2N_OK;
    P2N_REQUIRE(x && packed, "p2n_forward_eval: null x/packed");
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
    int ps = p2n_prof_begin(P2N_K_ENCODER, stream);
    if (f1 || f2 || f3)
        hipLaunchKernelGGL(p2n_encoder_kernel<true>, dim3((unsigned)ntiles), dim3(256), 0, stream, ea);
    else
        hipLaunchKernelGGL(p2n_encoder_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, stream, ea);
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
    const size_t lds = sizeof(float) * ((size_t)ha.tab_n + 256 + P2N_WAVES_PER_WG * 32 * (size_t)(L.cpad[8] + 4));
    if (lds > 64 * 1024) {
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(p2n_head_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    ps = p2n_prof_begin(P2N_K_HEAD, stream);
    hipLaunchKernelGGL(p2n_head_kernel, dim3((unsigned)ntiles), dim3(256), lds, stream, ha);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
