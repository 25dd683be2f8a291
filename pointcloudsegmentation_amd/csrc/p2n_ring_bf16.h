// Shared by the bf16 eval path (p2n_fwd_bf16.hip) and the split-bf16 "fp32x3" eval path (p2n_fwd_x3.hip): the
// four-deep LDS weight ring filled by buffer LDS-DMA with counted vmcnt waits, accumulator start / ReLU / channel-max
// helpers on 32x32 tiles, and the fp32 mid kernel (per-cloud maxima -> global_f and the folded layer-6 shift).
#pragma once
#include "p2n_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define BCHUNK_FLOATS (P2N_CHUNK * P2N_FRAG_FLOATS)   // 32 KiB

__device__ __forceinline__ unsigned bf16_bits(float v) {
    const __bf16 b = (__bf16)v;   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

// ------------------------------------------------------------------ helpers
typedef __attribute__((address_space(3))) void lds_void_b;
typedef const __attribute__((address_space(1))) void gbl_void_b;

// A bf16 chunk is consumed in ~1 us, less than an LDS-DMA takes to land, so the ring is four chunks
// deep (128 KiB of LDS, one 8-wave workgroup per CU): entering chunk c the workgroup queues chunk
// c+3 and waits only for chunk c with a COUNTED vmcnt that leaves the two younger chunks in flight.
#define NWB 8                      // waves per workgroup (256 points)
#define RING 4
#define DMA_PER_WAVE (P2N_CHUNK / NWB)

__device__ __forceinline__ void stage_chunk_b(const float *__restrict__ stream, float *wbuf, int c, int wave, int lane) {
    // buffer LDS-DMA: stream offset on the scalar unit, one constant lane offset (no vector-ALU address arithmetic)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(stream), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int f = wv * DMA_PER_WAVE + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_b *)(wbuf + (c % RING) * BCHUNK_FLOATS + f * P2N_FRAG_FLOATS), 16,
                                                 lane * 16, (c * P2N_CHUNK + f) * (P2N_FRAG_FLOATS * 4), 0, 0);
    }
}

// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]);
// lgkmcnt(0): this wave's own LDS writes (tables) and reads are done before it signals the barrier
#define WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14))

// Entering chunk c (every wave, uniformly): this wave's share of chunk c has landed once at most the
// 2*DMA_PER_WAVE younger DMA instructions (chunks c+1, c+2) are outstanding -- vmcnt retires in issue
// order, and anything else the wave queued since only makes the wait stricter; the barrier then makes
// every wave's share visible and proves that all waves have left chunk c-1, whose buffer chunk c+3 takes.
__device__ __forceinline__ void enter_chunk_b(int c, const float *__restrict__ stream, float *wbuf, int wave, int lane) {
    WAIT_VMCNT(2 * DMA_PER_WAVE);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage_chunk_b(stream, wbuf, c + RING - 1, wave, lane);
}

__device__ __forceinline__ void gate_b(int f, const float *__restrict__ stream, float *wbuf, int wave, int lane) {
    if (f % P2N_CHUNK == 0) enter_chunk_b(f / P2N_CHUNK, stream, wbuf, wave, lane);
}

__device__ __forceinline__ bf16x8 frag_b(const float *wbuf, int f, int lane) {
    return *reinterpret_cast<const bf16x8 *>(wbuf + ((f / P2N_CHUNK) % RING) * BCHUNK_FLOATS + (f % P2N_CHUNK) * P2N_FRAG_FLOATS +
                                             lane * 4);
}

// a layer's accumulators start at its shift (channel of register 4g+j in lane half h: 32ct + 8g + 4h + j) ...
__device__ __forceinline__ f32x16 shift_tile_b(const float *shift, int ct, int h) {
    f32x16 out;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 32 * ct + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * g + j] = sh[j];
    }
    return out;
}
// ... and the epilogue is the ReLU alone, as one integer max on the bit pattern (p2n_fwd.hip: relu_tile)
__device__ __forceinline__ f32x16 relu_tile_b(const f32x16 acc) {
    f32x16 out;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int b = __float_as_int(acc[i]);
        out[i] = __int_as_float(b > 0 ? b : 0);
    }
    return out;
}

// accumulator tile -> the two operand fragments (k-steps 2ct, 2ct+1) of the next layer
__device__ __forceinline__ void pack_tile(const f32x16 v, bf16x8 &s0, bf16x8 &s1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = (__bf16)v[j]; s1[j] = (__bf16)v[8 + j]; }
}
__device__ __forceinline__ void pack_tile_split(const f32x16 v, bf16x8 &h0, bf16x8 &h1, bf16x8 &l0, bf16x8 &l1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h0[j] = (__bf16)v[j];
        l0[j] = (__bf16)(v[j] - (float)h0[j]);
        h1[j] = (__bf16)v[8 + j];
        l1[j] = (__bf16)(v[8 + j] - (float)h1[j]);
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__device__ __forceinline__ unsigned row_max_u32_b(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    v = v > t ? v : t;
    return v;
}

// Channel maxima of one tile over each 16-lane row of points, as a halving transpose-reduce: at every
// step lanes pair up (xor 1, xor 2, rotate 4, rotate 8 inside the DPP row), each lane keeps one register
// of a pair and takes the partner's copy of it, so 16 registers shrink to 8, 4, 2, 1 with 15 DPP moves
// instead of 64.  Lane i of a row ends up with the row maximum of register i, i.e. of channel
// 32ct + 8(i>>2) + 4h + (i&3).  Post-ReLU values are >= 0: their bit patterns order as unsigned integers and
// 0 is the neutral element for masked points.  No LDS traffic here: an LDS atomic issued while weight DMAs
// are in flight makes the compiler drain them first (s_waitcnt vmcnt(0)).
template <int N, int CTRL>
__device__ __forceinline__ void halve_max(unsigned (&m)[16], bool upper) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const unsigned keep = upper ? m[2 * q + 1] : m[2 * q];
        const unsigned give = upper ? m[2 * q] : m[2 * q + 1];
        const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give, CTRL, 0xF, 0xF, true);
        m[q] = keep > t ? keep : t;
    }
}

__device__ __forceinline__ unsigned tile_rowmax(const f32x16 v, bool valid, int lane) {
    unsigned m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = valid ? __float_as_uint(v[i]) : 0u;   // (valid is wave-uniformly true except in a cloud's last wave)
    halve_max<8, 0xB1>(m, lane & 1);    // quad_perm [1,0,3,2]
    halve_max<4, 0x4E>(m, lane & 2);    // quad_perm [2,3,0,1]
    halve_max<2, 0x124>(m, lane & 4);   // row_ror:4
    halve_max<1, 0x128>(m, lane & 8);   // row_ror:8
    return m[0];
}

template <int C>
__device__ __forceinline__ void store_feat_b(float *__restrict__ f, const f32x16 v, int b, int ct, int h, long n, int N) {
    if (n >= N) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[((long)b * C + 32 * ct + 8 * g + 4 * h + j) * N + n] = v[4 * g + j];
}

// ------------------------------------------------------------------ mid (fp32; same arithmetic as p2n_fwd.hip)
static __global__ __launch_bounds__(256) void p2n_mid_b_kernel(const float *__restrict__ partial, int tiles,
                                                        const float *__restrict__ packed, long w6g, long scale6, long shift6,
                                                        float *__restrict__ shift6c, float *__restrict__ global_f) {
    __shared__ __attribute__((aligned(16))) float g[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float m = 0.f;
    const float *pp = partial + (long)b * tiles * 256 + t;
    for (int i = 0; i < tiles; ++i) m = fmaxf(m, pp[(long)i * 256]);
    g[t] = m;
    if (blockIdx.y == 0 && global_f) global_f[(long)b * 256 + t] = m;
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(&g[4 * lane]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int co = blockIdx.y * 32 + wave * 8 + q;
        const f32x4 w = *reinterpret_cast<const f32x4 *>(packed + w6g + (long)co * 256 + 4 * lane);
        float s = (w[0] * gv[0] + w[1] * gv[1]) + (w[2] * gv[2] + w[3] * gv[3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) shift6c[(long)b * 256 + co] = fmaf(s, packed[scale6 + co], packed[shift6 + co]);
    }
}

