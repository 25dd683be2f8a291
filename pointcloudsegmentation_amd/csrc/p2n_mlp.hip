// Generic shared-MLP layer and group max-pool for set-abstraction / feature-propagation
// levels (gfx950).  The reference has no hierarchical levels, so this
// operator is specified by its formula only (parity unpinned):
//     out[b][co][l] = act( scale[co] * sum_k W[co][k] * in[b][k][l] + shift[co] )
// i.e. Conv(k=1) + folded BatchNorm(eval) + optional ReLU over any flattened trailing
// extent L (points, or centroid x nsample of a grouped tensor).  Exact-fp32 MFMA, same
// orientation as p2n_fwd.hip (D = W.X, lane = point).  K is consumed in passes of <= 256.
#include "p2n_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__global__ void p2n_mlp_pack_w_kernel(const float *__restrict__ W, int ldw, int col0, int rows_real, int k_real,
                                      float *__restrict__ img, int CT, int G) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)CT * G * 256) return;
    const int e = (int)(i & 255), lane = e >> 2, j = e & 3;
    const long f = i >> 8;
    const int ct = (int)(f / G), g = (int)(f % G);
    const int row = 32 * ct + (lane & 31), k = 8 * g + 4 * (lane >> 5) + j;
    img[i] = (row < rows_real && k < k_real) ? W[(long)row * ldw + col0 + k] : 0.f;
}

struct PwArgs {
    const float *in;     // (B, Cin, L)
    const float *wimg;   // [CT][KT*4][64][4] for this K pass
    const float *scale, *shift;
    float *out;          // (B, Cout, L)
    long L;
    int Cin, k0, K, Cout, CT, relu, add_prev, final_pass;
};

template <int KT>
__global__ __launch_bounds__(256) void p2n_pointwise_kernel(PwArgs A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const long l = ((long)blockIdx.x * 4 + wave) * 32 + p;
    const bool valid = l < A.L;
    const int b = blockIdx.y;
    const float *in = A.in + ((long)b * A.Cin + A.k0) * A.L;
    float *out = A.out + (long)b * A.Cout * A.L;
    f32x16 x[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 32 * kt + 8 * g + 4 * h + j;
                x[kt][4 * g + j] = (valid && k < A.K) ? in[(long)k * A.L + l] : 0.f;
            }
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(A.wimg) + lane;
#pragma unroll 1
    for (int ct = 0; ct < A.CT; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = wl[(long)((ct * KT + kt) * 4 + g) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], x[kt][4 * g + j], acc);
            }
        if (valid)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 32 * ct + 8 * g + 4 * h + j;
                    if (c < A.Cout) {
                        float v = acc[4 * g + j];
                        float *o = out + (long)c * A.L + l;
                        if (A.add_prev) v += *o;
                        if (A.final_pass) {
                            v = fmaf(v, A.scale[c], A.shift[c]);
                            if (A.relu) v = fmaxf(v, 0.f);
                        }
                        *o = v;
                    }
                }
    }
}

// out[b][c][m] = max_s in[b][c][m][s]
__global__ __launch_bounds__(256) void p2n_group_max_kernel(const float *__restrict__ in, long rows, int ns,
                                                           float *__restrict__ out) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *q = in + r * ns;
    float m = q[0];
    for (int s = 1; s < ns; ++s) m = fmaxf(m, q[s]);
    out[r] = m;
}

extern "C" size_t p2n_pointwise_layer_workspace_bytes(int Cin, int Cout) {
    (void)Cin;
    if (Cout < 1) return 0;
    return sizeof(float) * (size_t)((Cout + 31) / 32) * 32 * 256;   // one K pass of <= 256
}

extern "C" int p2n_pointwise_layer(const float *in, int B, int Cin, long L, const float *W, const float *scale,
                                   const float *shift, int relu, int Cout, float *out, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && L >= 0, "p2n_pointwise_layer: bad shape B=%d Cin=%d Cout=%d L=%ld", B, Cin, Cout, L);
    if (B == 0 || L == 0) return P2N_OK;
    P2N_REQUIRE(in && W && scale && shift && out, "p2n_pointwise_layer: null pointer");
    P2N_REQUIRE(B <= 65535 && (L + 127) / 128 < (1L << 31), "p2n_pointwise_layer: B or L too large");
    const size_t need = p2n_pointwise_layer_workspace_bytes(Cin, Cout);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_pointwise_layer: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream_;
    const int CT = (Cout + 31) / 32;
    const int ps = p2n_prof_begin(P2N_K_MLP_LAYER, s);
    for (int k0 = 0; k0 < Cin; k0 += 256) {
        const int K = Cin - k0 < 256 ? Cin - k0 : 256;
        const int KT = K <= 32 ? 1 : K <= 64 ? 2 : K <= 128 ? 4 : 8;
        const long n = (long)CT * KT * 4 * 256;
        hipLaunchKernelGGL(p2n_mlp_pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, Cin, k0, Cout, K,
                           (float *)workspace, CT, KT * 4);
        PwArgs a;
        a.in = in; a.wimg = (const float *)workspace; a.scale = scale; a.shift = shift; a.out = out;
        a.L = L; a.Cin = Cin; a.k0 = k0; a.K = K; a.Cout = Cout; a.CT = CT; a.relu = relu;
        a.add_prev = k0 > 0; a.final_pass = k0 + 256 >= Cin;
        const dim3 grid((unsigned)((L + 127) / 128), B), blk(256);
        switch (KT) {
            case 1: hipLaunchKernelGGL(p2n_pointwise_kernel<1>, grid, blk, 0, s, a); break;
            case 2: hipLaunchKernelGGL(p2n_pointwise_kernel<2>, grid, blk, 0, s, a); break;
            case 4: hipLaunchKernelGGL(p2n_pointwise_kernel<4>, grid, blk, 0, s, a); break;
            default: hipLaunchKernelGGL(p2n_pointwise_kernel<8>, grid, blk, 0, s, a); break;
        }
        P2N_CHECK_HIP(hipGetLastError());
    }
    p2n_prof_end(ps, s);
    return P2N_OK;
}

// ---- group maximum on the channel-major output of a train-mode shared MLP (set-abstraction levels in training) ------
// in (C, ld): columns g*ns .. g*ns+ns-1 of a row are group g (g < G).  out (C, G) = the group's maximum, arg (C, G) =
// the first position that attains it (what the backward routes the gradient to).  ns in {8, 16, 32, 64}: ns/4 lanes
// read one group as consecutive 16-byte pieces (a wave reads 1 KiB contiguous per step) and reduce by xor-shuffles;
// other ns: one thread per group.  Keys order as (value, lower position first); a NaN with its sign bit clear is the
// largest value, as in torch.max.
__device__ __forceinline__ unsigned p2n_ordered_bits(float v) {
    unsigned b = __float_as_uint(v);
    b = b == 0x80000000u ? 0u : b;                    // -0 == +0: the first of them is the first maximum
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

template <int NS>
__global__ __launch_bounds__(256) void p2n_group_max_cm_kernel(const float *__restrict__ in, long ld, long G, float *__restrict__ out,
                                                               unsigned char *__restrict__ arg) {
    constexpr int LPG = NS / 4;                       // lanes per group
    const int c = blockIdx.y;
    const long piece = (long)blockIdx.x * 256 + threadIdx.x;      // 16-byte piece of the row
    const long g = piece / LPG;
    const int sub = (int)(piece % LPG);
    const bool ok = g < G;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4 *>(in + (long)c * ld + piece * 4);
    unsigned long long k = 0ull;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long kj = ((unsigned long long)p2n_ordered_bits(e[j]) << 32) | (unsigned)(255 - (sub * 4 + j));
        k = kj > k ? kj : k;
    }
#pragma unroll
    for (int m = 1; m < LPG; m <<= 1) {
        const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(k >> 32), m) << 32) | (unsigned)__shfl_xor((int)k, m);
        k = o > k ? o : k;
    }
    if (ok && sub == 0) {
        const unsigned ob = (unsigned)(k >> 32);
        out[(long)c * G + g] = __uint_as_float(ob ^ ((ob >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        arg[(long)c * G + g] = (unsigned char)(255 - (unsigned)(k & 0xFFu));
    }
}

__global__ __launch_bounds__(256) void p2n_group_max_cm_any_kernel(const float *__restrict__ in, long ld, long G, int ns,
                                                                   float *__restrict__ out, unsigned char *__restrict__ arg) {
    const int c = blockIdx.y;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float *q = in + (long)c * ld + g * ns;
    unsigned best = p2n_ordered_bits(q[0]);
    int pos = 0;
    for (int s = 1; s < ns; ++s) {
        const unsigned b = p2n_ordered_bits(q[s]);
        if (b > best) { best = b; pos = s; }
    }
    out[(long)c * G + g] = q[pos];   // (-0 where the group's first maximum is -0; the cooperative kernels return +0: equal)
    arg[(long)c * G + g] = (unsigned char)pos;
}

// backward: din (C, ld) = gout[c][g] at column g*ns + arg[c][g], zero elsewhere (pad columns G*ns .. ld-1 included)
__global__ __launch_bounds__(256) void p2n_group_max_cm_bwd_kernel(const float *__restrict__ gout, const unsigned char *__restrict__ arg,
                                                                   long G, int ns, long ld, float *__restrict__ din) {
    const int c = blockIdx.y;
    const long col = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= ld) return;
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long cj = col + j, g = cj / ns;
        e[j] = (g < G && (int)(cj - g * ns) == (int)arg[(long)c * G + g]) ? gout[(long)c * G + g] : 0.f;
    }
    *reinterpret_cast<float4 *>(din + (long)c * ld + col) = make_float4(e[0], e[1], e[2], e[3]);
}

extern "C" int p2n_group_max_cm(const float *in, int C, long ld, long G, int ns, float *out, unsigned char *arg, void *stream_) {
    P2N_REQUIRE(C >= 0 && G >= 0 && ns >= 1 && ns <= 255 && ld >= G * ns && ld % 4 == 0, "p2n_group_max_cm: bad shape C=%d ld=%ld G=%ld ns=%d", C, ld, G, ns);
    if (C == 0 || G == 0) return P2N_OK;
    P2N_REQUIRE(in && out && arg, "p2n_group_max_cm: null pointer");
    P2N_REQUIRE(C <= 65535, "p2n_group_max_cm: more than 65,535 channels");
    hipStream_t s = (hipStream_t)stream_;
    const int ps = p2n_prof_begin(P2N_K_GROUP_MAX, s);
    const unsigned gx = (unsigned)((G * (ns / 4) + 255) / 256);
    switch (ns) {
        case 8: hipLaunchKernelGGL(p2n_group_max_cm_kernel<8>, dim3(gx, C), dim3(256), 0, s, in, ld, G, out, arg); break;
        case 16: hipLaunchKernelGGL(p2n_group_max_cm_kernel<16>, dim3(gx, C), dim3(256), 0, s, in, ld, G, out, arg); break;
        case 32: hipLaunchKernelGGL(p2n_group_max_cm_kernel<32>, dim3(gx, C), dim3(256), 0, s, in, ld, G, out, arg); break;
        case 64: hipLaunchKernelGGL(p2n_group_max_cm_kernel<64>, dim3(gx, C), dim3(256), 0, s, in, ld, G, out, arg); break;
        default: hipLaunchKernelGGL(p2n_group_max_cm_any_kernel, dim3((unsigned)((G + 255) / 256), C), dim3(256), 0, s, in, ld, G, ns, out, arg);
    }
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_group_max_cm_bwd(const float *gout, const unsigned char *arg, int C, long G, int ns, long ld, float *din, void *stream_) {
    P2N_REQUIRE(C >= 0 && G >= 0 && ns >= 1 && ns <= 255 && ld >= G * ns && ld % 4 == 0, "p2n_group_max_cm_bwd: bad shape C=%d ld=%ld G=%ld ns=%d", C, ld, G, ns);
    if (C == 0 || ld == 0) return P2N_OK;
    P2N_REQUIRE(gout && arg && din, "p2n_group_max_cm_bwd: null pointer");
    P2N_REQUIRE(C <= 65535, "p2n_group_max_cm_bwd: more than 65,535 channels");
    hipStream_t s = (hipStream_t)stream_;
    const int ps = p2n_prof_begin(P2N_K_GROUP_MAX, s);
    hipLaunchKernelGGL(p2n_group_max_cm_bwd_kernel, dim3((unsigned)((ld / 4 + 255) / 256), C), dim3(256), 0, s, gout, arg, G, ns, ld, din);
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_group_max(const float *in, int B, int C, int M, int ns, float *out, void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && M >= 0 && ns >= 1, "p2n_group_max: bad shape");
    const long rows = (long)B * C * M;
    if (rows == 0) return P2N_OK;
    P2N_REQUIRE(in && out, "p2n_group_max: null pointer");
    const int ps = p2n_prof_begin(P2N_K_GROUP_MAX, (hipStream_t)stream);
    hipLaunchKernelGGL(p2n_group_max_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       rows, ns, out);
    p2n_prof_end(ps, (hipStream_t)stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
