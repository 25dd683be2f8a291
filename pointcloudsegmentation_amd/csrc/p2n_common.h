// Shared host/device definitions of libp2n_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/p2n.h"

#define P2N_NLAYER 9
#define P2N_WAVE 64
#define P2N_PTS_PER_WAVE 32          // one 32-column MFMA block of points per wave
#define P2N_WAVES_PER_WG 4
#define P2N_PTS_PER_WG (P2N_PTS_PER_WAVE * P2N_WAVES_PER_WG)
#define P2N_CHUNK 32                 // weight fragments per LDS chunk (32 KiB); two chunks are resident
#define P2N_FRAG_FLOATS 256          // one fragment = 64 lanes x 4 floats = 1 KiB
#define P2N_TAB_MAX 1792             // encoder scale/shift tables: 2*(64+64+128+128+256+256)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Packed eval-parameter image.  All offsets in floats from the image base.
struct P2nLayout {
    int F, E;
    int cin[P2N_NLAYER], cout[P2N_NLAYER];
    int kpad[P2N_NLAYER];   // reduction length the kernels run (multiple of 8; layer 6: f3 half only)
    int cpad[P2N_NLAYER];   // output channels rounded up to 32
    long raw_w[P2N_NLAYER], raw_b[P2N_NLAYER], raw_g[P2N_NLAYER], raw_beta[P2N_NLAYER],
        raw_mean[P2N_NLAYER], raw_var[P2N_NLAYER];
    long raw_total;
    long enc_stream;        // layers 0..5 in consumption order [layer][ct][k-group]
    long enc_frags;
    long head_stream;       // layers 6 and 7 as [kt][ct][g], layer 8 as [ct][kt][g]
    long head_frags;        // both streams are zero-padded to a chunk multiple plus one spare chunk
    long scale[P2N_NLAYER], shift[P2N_NLAYER];
    long w6g;               // (256,256) row-major: the global-feature half of layer 6
    long total;
};

static inline P2nLayout p2n_make_layout(int F, int E) {
    P2nLayout L;
    const int ci[P2N_NLAYER] = {F, 64, 64, 128, 128, 256, 512, 256, 128};
    const int co[P2N_NLAYER] = {64, 64, 128, 128, 256, 256, 256, 128, E};
    L.F = F; L.E = E;
    long r = 0;
    for (int l = 0; l < P2N_NLAYER; ++l) {
        L.cin[l] = ci[l]; L.cout[l] = co[l];
        L.kpad[l] = (l == 0) ? 8 : (l == 6 ? 256 : ci[l]);
        L.cpad[l] = (co[l] + 31) / 32 * 32;
        L.raw_w[l] = r; r += (long)ci[l] * co[l];
        L.raw_b[l] = r; r += co[l];
        if (l < 8) {
            L.raw_g[l] = r; r += co[l];
            L.raw_beta[l] = r; r += co[l];
            L.raw_mean[l] = r; r += co[l];
            L.raw_var[l] = r; r += co[l];
        } else {
            L.raw_g[l] = L.raw_beta[l] = L.raw_mean[l] = L.raw_var[l] = -1;
        }
    }
    L.raw_total = r;
    long o = 0;
    L.enc_stream = o;
    L.enc_frags = 0;
    for (int l = 0; l < 6; ++l) L.enc_frags += (long)(L.cpad[l] / 32) * (L.kpad[l] / 8);
    o += ((L.enc_frags + P2N_CHUNK - 1) / P2N_CHUNK + 1) * P2N_CHUNK * P2N_FRAG_FLOATS;
    L.head_stream = o;
    L.head_frags = 256 + 128 + (long)(L.cpad[8] / 32) * 16;
    o += ((L.head_frags + P2N_CHUNK - 1) / P2N_CHUNK + 1) * P2N_CHUNK * P2N_FRAG_FLOATS;
    for (int l = 0; l < P2N_NLAYER; ++l) { L.scale[l] = o; o += L.cpad[l]; L.shift[l] = o; o += L.cpad[l]; }
    L.w6g = o; o += 256L * 256;
    L.total = o;
    return L;
}

// Barrier that publishes LDS-DMA data (global_load ... lds).  The compiler does not always count an
// LDS-DMA as something a workgroup barrier has to wait for (seen in the clustering pair kernel: only
// s_waitcnt lgkmcnt(0) in front of s_barrier), so the wait is spelled out: every wave drains its own
// vector-memory queue (vmcnt(0): the DMA has written LDS) before it signals the barrier.
// s_waitcnt encoding on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14].
#ifdef __HIPCC__
__device__ __forceinline__ void p2n_dma_barrier() {
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8) | (0 << 14));   // vmcnt(0) lgkmcnt(0)
    __syncthreads();
}

// Channel maxima of one 32x32 tile over the 32 lanes of each wave half (one centroid's 32 neighbours), as a
// halving transpose-reduce: at every step lanes pair up inside their 16-lane DPP row (xor 1, xor 2, rotate
// 4, rotate 8), each lane keeps one register of a pair and takes the partner's copy of it, so 16 registers
// shrink to 8, 4, 2, 1 with 15 exchanges instead of 64; one ds_swizzle (lane ^ 16, no memory access) then
// joins the two rows of the half.  Lane i of a row ends up with the maximum of register i, i.e. of channel
// 32ct + 8(i>>2) + 4h + (i&3).  Values are post-ReLU (>= 0, or -0.0): their bit patterns order as signed
// integers, and integer max folds the DPP operand into the instruction.
template <int N, int CTRL>
__device__ __forceinline__ void p2n_halve_max(int (&m)[16], bool upper) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int keep = upper ? m[2 * q + 1] : m[2 * q];
        const int give = upper ? m[2 * q] : m[2 * q + 1];
        const int t = __builtin_amdgcn_update_dpp(0, give, CTRL, 0xF, 0xF, true);
        m[q] = keep > t ? keep : t;
    }
}

__device__ __forceinline__ float p2n_tile_half_max(const f32x16 v, int lane) {
    int m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = __float_as_int(v[i]);
    p2n_halve_max<8, 0xB1>(m, lane & 1);    // quad_perm [1,0,3,2]
    p2n_halve_max<4, 0x4E>(m, lane & 2);    // quad_perm [2,3,0,1]
    p2n_halve_max<2, 0x124>(m, lane & 4);   // row_ror:4
    p2n_halve_max<1, 0x128>(m, lane & 8);   // row_ror:8
    const int t = __builtin_amdgcn_ds_swizzle(m[0], 0x401F);   // bit mode: and 0x1F, or 0, xor 0x10
    return __int_as_float(m[0] > t ? m[0] : t);
}

#endif

// ---- optional per-kernel timing (p2n_api.hip) ----
#define P2N_K_ENCODER 0
#define P2N_K_MID 1
#define P2N_K_HEAD 2
#define P2N_K_FPS 3
#define P2N_K_BALL 4
#define P2N_K_KNN 5
#define P2N_K_GATHER 6
#define P2N_K_INTERP 7
#define P2N_K_TRAIN_GEMM 8
#define P2N_K_TRAIN_WGRAD 9
#define P2N_K_CLUSTER_PREP 10
#define P2N_K_CLUSTER_COUNT 11
#define P2N_K_CLUSTER_LINK 12
#define P2N_K_PREP 13
#define P2N_K_MLP_LAYER 14
#define P2N_K_GROUP_MAX 15
#define P2N_K_SA_LEVEL 16
#define P2N_K_FP_LEVEL 17
#define P2N_K_LEVEL_PACK 18
#define P2N_K_GATHER_XYZ 19
#define P2N_K_COUNT 32
int p2n_prof_begin(int kernel_id, hipStream_t s);
void p2n_prof_end(int slot, hipStream_t s);

// ---- error plumbing (p2n_api.hip) ----
void p2n_set_error(const char *fmt, ...);
#define P2N_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            p2n_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return P2N_EHIP;                                                         \
        }                                                                            \
    } while (0)
#define P2N_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            p2n_set_error(__VA_ARGS__);         \
            return P2N_EINVAL;                  \
        }                                       \
    } while (0)
