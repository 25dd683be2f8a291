// Point-set operators for gfx950 (MI355X): farthest-point sampling, ball query,
// brute-force kNN / three-NN, grouping gather, inverse-distance interpolation.
//
// The reference has none of these except the brute-force kNN
// inside its smoothness loss (reference losses.py:475-479); their specification is
// oracle/p2n_oracle.c: fp32 squared distance ((dx*dx + dy*dy) + dz*dz) with separate
// multiply and add, lowest index wins ties.  All kernels below evaluate exactly that
// expression (explicit __fmul_rn/__fadd_rn), so indices are bit-exact by construction.
#include <stdlib.h>
#include "p2n_common.h"

// Tuning overrides for the sweep tools (tools/fps_sweep.py, tools/sweep_gather.sh), read from the
// environment once per process: 0 / unset = the built-in choice.
struct P2nTuning { int fps_g, gl_ch, gl_threads, gl_split, gl_off, fps_t, fps_spin, fps_fault, fps_noxcd, fps_prune, fps_pw, fps_rounds, fps_pd, fps_exp; };
static const P2nTuning &p2n_tuning() {
    static const P2nTuning t = [] {
        auto num = [](const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; };
        return P2nTuning{num("P2N_FPS_G"), num("P2N_GL_CH"), num("P2N_GL_THREADS"), num("P2N_GL_SPLIT"),
                         getenv("P2N_GL_OFF") ? 1 : 0, num("P2N_FPS_T"), num("P2N_FPS_SPIN"),
                         num("P2N_FPS_FAULT"), num("P2N_FPS_NOXCD"), num("P2N_FPS_PRUNE"), num("P2N_FPS_PW"), num("P2N_FPS_ROUNDS"), num("P2N_FPS_PD"), num("P2N_FPS_EXP")};
    }();
    return t;
}

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

typedef unsigned long long u64;

// =========================================================================== FPS
// A cluster of G workgroups owns one cloud; every thread keeps PPT points (x, y, z, running min
// distance) in REGISTERS for the whole kernel, so an iteration touches no memory except the
// winner exchange.  What an iteration costs is instruction issue on one CU plus the exchange, so:
//  * few, fat waves: 256 threads x up to 48 points (one wave per SIMD) -- the reductions are paid
//    per wave, the distance updates per point;
//  * distances for two points at a time (v_pk_add/mul_f32: same IEEE roundings as the scalar
//    form), and the running argmax tracks only (distance, slot number); the winner's coordinates
//    are read back from a copy of the chunk in LDS instead of being selected per point;
//  * workgroup argmax: DPP maximum over each 16-lane row, then one 64-bit LDS atomic max per row
//    into a key that rotates over three words (the word two iterations ahead is cleared behind
//    the barrier, so no extra synchronisation) -- one barrier per iteration, every wave reads the
//    winner with one ds_read_b64;
//  * G > 1: four lanes of wave 0 publish {key, x, y, z} as four tagged 8-byte granules (one
//    agent-scope atomic store each, tag = low key word, never zero) into slot[it][g]; wave 0 polls
//    the G slots of this iteration (zero = not yet written; slots are per iteration, so no resets),
//    reduces the keys by DPP and takes the winner's coordinates straight from its granules -- one
//    fabric hop per iteration, bounded spin with a status word.
// key = (bits(min_dist) << 32) | (0xFFFFFFFF - index): max key = farthest point, lowest index on
// ties.  min_dist >= 0 so its bit pattern orders as an unsigned.
// Large clouds that need more than 12,288 points per workgroup (B*G limited to 256 co-resident
// workgroups, G <= 64) run 1024-thread workgroups with up to 20 points per thread; such a chunk
// no longer fits LDS and the winner's coordinates come from global memory.
#define FPS_SPIN_LIMIT (1 << 22)
#define FPS_LDS_POINTS 12288   // chunk that fits LDS as xyz (147,456 B)

// DPP reductions on 32-bit unsigned values: one VALU instruction per step, no LDS.
#define DPP_MAX_STEP(v, ctrl, rmask, keep)                                                            \
    do {                                                                                              \
        const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp((keep) ? (int)(v) : 0, (int)(v), (ctrl), (rmask), 0xF, !(keep)); \
        (v) = (v) > t_ ? (v) : t_;                                                                    \
    } while (0)

// max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    DPP_MAX_STEP(v, 0xB1, 0xF, false);    // quad_perm [1,0,3,2]
    DPP_MAX_STEP(v, 0x4E, 0xF, false);    // quad_perm [2,3,0,1]
    DPP_MAX_STEP(v, 0x141, 0xF, false);   // row_half_mirror
    DPP_MAX_STEP(v, 0x140, 0xF, false);   // row_mirror
    return v;
}

// max over the 64 lanes, returned wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = row16_max_u32(v);
    DPP_MAX_STEP(v, 0x142, 0xA, true);    // row_bcast15 into rows 1,3
    DPP_MAX_STEP(v, 0x143, 0xC, true);    // row_bcast31 into rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 64-bit LDS atomic max without a return value, as the instruction itself: written through
// atomicMax the compiler serialises the active lanes in a scalar loop first (uniform address).
// The wait makes the update visible to the barrier that follows.
__device__ __forceinline__ void lds_max_u64(u64 *p, u64 v) {
    const unsigned off = (unsigned)(uintptr_t)p;   // low half of a flat LDS address = the LDS offset
    asm volatile("ds_max_u64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(off), "v"(v) : "memory");
}

struct FpsArgs {
    const float *xyz;   // (B,N,3)
    int32_t *idx;       // (B,M)
    u64 *slots;         // [clouds of this launch][M+1][G][4], zeroed per launch (G > 1 only); row M: the XCD census
    int *status;        // set to 1 when an exchange timed out (never cleared here)
    int N, M, G, chunk, start_idx, b0, spin_limit;
    int xcd_mode, nb;   // xcd_mode: grid = 8*G*ceil(nb/8) blocks, a cloud's workgroups share blockIdx % 8 (one XCD under round-robin dealing)
    int poll_delay;     // s_sleep units between a round's publish and its first poll (rounds kernels)
    int exp;            // P2N_FPS_TIMING builds only: switch parts of the decision off to time them (results are then wrong)
    int fault_it;       // test hook (P2N_FPS_FAULT=k): the cloud's last workgroup skips its publish in iteration k; 0 = off
};

template <int T, int PPT, bool LDSXYZ>
__global__ __launch_bounds__(T) void p2n_fps_kernel(FpsArgs A) {
    static_assert(PPT % 2 == 0, "points are processed in pairs");
    extern __shared__ __attribute__((aligned(16))) float lxyz[];   // LDSXYZ: this workgroup's chunk, [point][3]
    __shared__ u64 wkey[3];
    __shared__ float bc[2][4];   // broadcast of the iteration's winner (G > 1)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // every iteration is a chain of dependent steps on a handful of CUs: waves of other kernels that share those
    // CUs (the level kernels run beside the FPS chain) must not take its issue slots
    __builtin_amdgcn_s_setprio(3);
    int b, g;
    if (A.xcd_mode) {
        // Workgroups are dealt round-robin over the 8 XCDs (observed, not promised): cloud c takes the blocks
        // with blockIdx % 8 == c % 8, so that its cluster shares one L2 -- verified below before it is relied on.
        const int r = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int cl = (q / A.G) * 8 + r;
        if (cl >= A.nb) return;
        b = A.b0 + cl; g = q % A.G;
    } else {
        b = A.b0 + blockIdx.x / A.G; g = blockIdx.x % A.G;
    }
    const float *P = A.xyz + (long)b * A.N * 3;
    const int base = g * A.chunk;
    int nloc = A.N - base;
    if (nloc > A.chunk) nloc = A.chunk;
    if (nloc < 0) nloc = 0;

    // Slot s of thread t is point base + t + T*s.  Slots past the chunk hold the origin with running
    // minimum 0: min(0, d) stays 0, and the workgroup's first real point has a lower index, so they
    // lose every tie and need no special casing in the loop below.
    f32x2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], md[PPT / 2];
    if (LDSXYZ) {   // one coalesced copy of the chunk into LDS, registers filled from there
        for (int i = t; i < nloc * 3; i += T) lxyz[i] = P[(long)base * 3 + i];
    }
    if (t < 3) wkey[t] = 0ull;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = t + T * s;
        const bool ok = i < nloc;
        const float *src = LDSXYZ ? lxyz + i * 3 : P + (long)(base + i) * 3;
        px[s / 2][s & 1] = ok ? src[0] : 0.f;
        py[s / 2][s & 1] = ok ? src[1] : 0.f;
        pz[s / 2][s & 1] = ok ? src[2] : 0.f;
        md[s / 2][s & 1] = ok ? __int_as_float(0x7f800000) : 0.f;
    }

    int cur = A.start_idx;
    float cx = P[(long)cur * 3 + 0], cy = P[(long)cur * 3 + 1], cz = P[(long)cur * 3 + 2];
    bool dead = false;
    // XCD census: every workgroup of the cluster publishes the id of the XCD it runs on (write-through store,
    // the form that is valid across XCDs) and reads everybody's.  Only if all agree may the per-iteration
    // exchange use plain stores, which stay in that XCD's L2 where the polling loads (L1 bypassed) find them a
    // few hundred cycles later; across XCDs a plain store would never become visible, so any disagreement (or
    // any other placement than the expected one) keeps the write-through form.  Every workgroup reads the
    // same ids, hence takes the same decision.
    bool fast = false, census_dead = false;
    if (A.G > 1 && A.xcd_mode) {
        if (wave == 0) {
            u64 *cs = A.slots + (((long)(b - A.b0) * (A.M + 1) + A.M) * A.G) * 4;
            const unsigned myx = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 0xF;   // hwreg(HW_REG_XCC_ID, 0, 4)
            if (lane == 0) __hip_atomic_store(cs + g * 4, (1ull << 32) | (u64)myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64 v = 1ull << 32 | myx;
            int spins = 0;
            bool ok;
            for (;;) {
                if (lane < A.G) v = __hip_atomic_load(cs + lane * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = __all(lane >= A.G || v != 0ull);
                if (ok || ++spins > A.spin_limit) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const bool same = __all(lane >= A.G || (unsigned)v == myx);
            if (lane == 0) bc[0][0] = __int_as_float(ok ? (same ? 1 : 0) : -1);
        }
        __syncthreads();
        const int verdict = __float_as_int(bc[0][0]);
        __syncthreads();
        fast = verdict == 1;
        if (verdict < 0) dead = census_dead = true;
    }
    const unsigned inv0 = 0xFFFFFFFFu - (unsigned)(base + t);
    int kb = 0;   // it % 3

    int it = 0;
    for (; it < A.M && !dead; ++it) {
        if (g == 0 && t == 0) A.idx[(long)b * A.M + it] = cur;
        if (it + 1 == A.M) break;
        // ---- update running minima (v_min_f32 as the instruction: fminf() would canonicalise its
        // operands first, one more VALU op per point)
        const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
        u64 tk[PPT / 2];   // per pair: (distance bits << 32) | (63 - slot): the maximum is the farthest, lowest slot
#pragma unroll
        for (int p = 0; p < PPT / 2; ++p) {
            const f32x2 dx = px[p] - c2x, dy = py[p] - c2y, dz = pz[p] - c2z;
            const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
            float m0, m1;
            asm("v_min_f32 %0, %1, %2" : "=v"(m0) : "v"(md[p][0]), "v"(dd[0]));
            asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(md[p][1]), "v"(dd[1]));
            md[p][0] = m0;
            md[p][1] = m1;
            const bool take = m1 > m0;   // first maximum wins: slots (and indices) ascend
            tk[p] = ((u64)__float_as_uint(take ? m1 : m0) << 32) | (unsigned)(take ? 62 - 2 * p : 63 - 2 * p);
        }
        // ---- thread-local argmax as a tournament of 64-bit keys (independent compares instead of
        // one chain of PPT dependent ones)
#pragma unroll
        for (int w = 1; w < PPT / 2; w <<= 1)
#pragma unroll
            for (int p = 0; p < PPT / 2; ++p)   // constant trip count so that every index is a constant
                if (p % (2 * w) == 0 && p + w < PPT / 2) tk[p] = tk[p + w] > tk[p] ? tk[p + w] : tk[p];
        const float best = __uint_as_float((unsigned)(tk[0] >> 32));
        const int bs = 63 - (int)(unsigned)(tk[0] & 0xFFFFFFFFull);
        // ---- workgroup argmax on the 64-bit key: row maxima by DPP, rows by LDS atomic max
        const unsigned hi = __float_as_uint(best);
        const unsigned binv = inv0 - (unsigned)(T * bs);
        const unsigned rhi = row16_max_u32(hi);
        const unsigned rlo = row16_max_u32(hi == rhi ? binv : 0u);
        if ((lane & 15) == 0) lds_max_u64(&wkey[kb], ((u64)rhi << 32) | rlo);
        __syncthreads();
        const u64 k = wkey[kb];
        const int kclr = kb == 0 ? 2 : kb - 1;   // (it + 2) % 3: last read before this barrier, next used behind the next one
        if (t == 0) wkey[kclr] = 0ull;
        kb = kb == 2 ? 0 : kb + 1;
        const int li = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu));   // this workgroup's winner
        if (A.G == 1) {
            cur = li;
            if (LDSXYZ) {
                cx = lxyz[(li - base) * 3 + 0]; cy = lxyz[(li - base) * 3 + 1]; cz = lxyz[(li - base) * 3 + 2];
            } else {
                cx = P[(long)li * 3 + 0]; cy = P[(long)li * 3 + 1]; cz = P[(long)li * 3 + 2];
            }
        } else {
            const int buf = it & 1;
            if (wave == 0) {
                u64 *slot = A.slots + (((long)(b - A.b0) * (A.M + 1) + it) * A.G) * 4;
                if (lane < 4 && !(A.fault_it > 0 && it == A.fault_it && g == A.G - 1)) {
                    u64 payload = k;
                    if (lane > 0) {
                        // a workgroup without real points (li outside its chunk) publishes zeros with its losing key
                        const bool real = li >= base && li < base + nloc;
                        const float c = !real ? 0.f : (LDSXYZ ? lxyz[(li - base) * 3 + lane - 1] : P[(long)li * 3 + lane - 1]);
                        payload = ((k & 0xFFFFFFFFull) << 32) | (u64)__float_as_uint(c);
                    }
                    if (fast) {   // same XCD: a plain store, served to the pollers from the shared L2
                        u64 *dst = slot + g * 4 + lane;
                        asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(payload) : "memory");
                    } else {
                        __hip_atomic_store(slot + g * 4 + lane, payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                u64 v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                int spins = 0;
                bool ok;
                for (;;) {
                    // lanes >= G read nothing and count as ready; G <= 64
                    bool ready = true;
                    if (lane < A.G) {
                        const u64 *sl = slot + lane * 4;
                        // two 16-byte sc1 loads (L1 bypassed, as the agent-scope atomic loads they replace) instead of
                        // four 8-byte ones: half the requests in the polling CU's memory queue.  Every 8-byte granule
                        // carries its own tag, so it does not matter which of the two halves of a load is newer.
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        u32x4 a, c;
                        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                                     "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(a), "=&v"(c) : "v"(sl) : "memory");
                        v0 = ((u64)a[1] << 32) | a[0]; v1 = ((u64)a[3] << 32) | a[2];
                        v2 = ((u64)c[1] << 32) | c[0]; v3 = ((u64)c[3] << 32) | c[2];
                        ready = v0 != 0ull && (v1 >> 32) != 0ull && (v2 >> 32) != 0ull && (v3 >> 32) != 0ull;
                    }
                    ok = __all(ready);
                    if (ok || ++spins > A.spin_limit) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const bool mine = lane < A.G;
                const unsigned khi = mine ? (unsigned)(v0 >> 32) : 0u, klo = mine ? (unsigned)v0 : 0u;
                const unsigned whi = wave_max_u32(khi);
                const unsigned wlo = wave_max_u32(khi == whi ? klo : 0u);
                const u64 wmask = __ballot(mine && khi == whi && klo == wlo);
                const int wl = wmask ? __builtin_ctzll(wmask) : 0;
                int widx = (int)(0xFFFFFFFFu - wlo);
                if (!ok || !wmask || widx < 0 || widx >= A.N) widx = -1;
                const float wx = __int_as_float(__builtin_amdgcn_readlane((int)(unsigned)v1, wl));
                const float wy = __int_as_float(__builtin_amdgcn_readlane((int)(unsigned)v2, wl));
                const float wz = __int_as_float(__builtin_amdgcn_readlane((int)(unsigned)v3, wl));
                if (lane == 0) {
                    bc[buf][0] = wx; bc[buf][1] = wy; bc[buf][2] = wz;
                    bc[buf][3] = __int_as_float(widx);
                }
            }
            __syncthreads();
            cur = __float_as_int(bc[buf][3]);
            cx = bc[buf][0]; cy = bc[buf][1]; cz = bc[buf][2];
            if (cur < 0) { dead = true; break; }   // uniform: every thread reads the same word
        }
    }
    if (dead) {
        // an exchange timed out (the cloud's workgroups were not all resident): say so, and leave no index
        // unwritten -- the rest of the row repeats the start point, so that no consumer gathers out of bounds
        if (t == 0) atomicExch(A.status, 1);
        if (g == 0)
            for (int i = (census_dead ? 0 : it + 1) + t; i < A.M; i += T) A.idx[(long)b * A.M + i] = A.start_idx;
    }
}


// --------------------------------------------------------------------------- several samples per exchange round
// The cluster kernel above pays one exchange (publish -> poll, two barriers) per SAMPLE.  This one pays it per ROUND
// and lets a round accept as many samples as it can PROVE the sequential algorithm would take next, in order:
//  * every workgroup publishes its best point (key k1 = distance bits | inverted index, coordinates) and an upper
//    bound d2 of its second-best running minimum;
//  * S = max d2 over the cluster.  Sort the published bests by key: c0 > c1 > ...  c0 is the next sample.  c_j is the
//    sample after c_0..c_{j-1} if (a) dist(c_j) > S strictly -- no point that is not somebody's best can beat or tie
//    it, and running minima only ever decrease -- and (b) none of c_0..c_{j-1} lowers c_j's own running minimum, i.e.
//    !(|c_i - c_j|^2 < dist(c_j)) in the very arithmetic of the update (and every c_i has finite coordinates, so that
//    its own running minimum dropped to 0 when it was taken).  Then c_j still holds its published distance,
//    every best with a smaller key is still below it in key order (= the tie rule), and every other point is <= S.
//    The round stops at the first c_j that fails; that point simply stays its workgroup's best for the next round.
//  * every workgroup takes the same decision from the same records, applies the accepted samples to its running
//    minima and starts the next round.  Indices are those of the sequential algorithm bit for bit (tools/
//    sim_fps_rounds.py is the CPU model: 5.5 samples per round at 40,000 points with 32 workgroups).
// Records live in a ring of FR_RING rounds; every 8-byte granule carries the round number (16 bits, never 0) as its tag, so
// nothing is cleared between rounds (a workgroup can be at most one round ahead of the slowest: it cannot publish round r+1
// before it has read everybody's round r).
#define FR_RING 4
#define FR_SLOT 4      // u64 per record: four granules, each {payload word, tag16 << 16 | half of k1 lo / d2}: 32 bytes, so that one
                       // 128-byte line holds four records (every member polls every record: the L2 request count is what a poll costs)
#define FR_MAXW 64


typedef unsigned fr_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned fr_tag(unsigned round) { return (round % 65535u) + 1u; }
// lanes 0..3 of the publishing wave store one granule each
__device__ __forceinline__ void fr_publish(u64 *rec, int lane, unsigned tag, unsigned k1hi, unsigned k1lo, unsigned x, unsigned y,
                                           unsigned z, unsigned d2, bool fast) {
    const unsigned lo = lane == 0 ? k1hi : lane == 1 ? x : lane == 2 ? y : z;
    const unsigned half = lane == 0 ? (k1lo & 0xFFFFu) : lane == 1 ? (k1lo >> 16) : lane == 2 ? (d2 & 0xFFFFu) : (d2 >> 16);
    const u64 gran = ((u64)((tag << 16) | half) << 32) | (u64)lo;
    u64 *dst = rec + lane;
    if (fast) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(gran) : "memory");   // same XCD: plain store, served from the shared L2
    else __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one poll of one record (L1 bypassed); true when all four granules carry this round's tag
__device__ __forceinline__ bool fr_poll(const u64 *rec, unsigned tag, fr_u32x4 &a, fr_u32x4 &c) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(c) : "v"(rec) : "memory");
    const unsigned T = tag << 16;
    return (((a[1] ^ T) | (a[3] ^ T) | (c[1] ^ T) | (c[3] ^ T)) >> 16) == 0u;
}

// The round's decision, taken by one wave from the records it polled (lane = cluster member): which published bests are
// the next samples, in which order.  Candidates = bests above every member's second best (S); when there are at most
// 16 of them -- the usual case -- they are compacted into LDS and all candidate pairs are evaluated at once (lane =
// (candidate i = lane & 15, other candidate j = lane >> 4 + 4 * pass): is j earlier in key order, and would accepting j
// lower i's running minimum?); ballots turn the pair results into every candidate's rank and "blocked" flag.  The result
// lives in candidate space (lane p < n holds candidate p's record).  More than 16 candidates are first cut to the 16 largest
// distances (a serial count over the candidate mask; rare).  acc lanes hold an accepted sample's coordinates, rank and index.
struct FrDecision { bool acc; unsigned rank; unsigned klo; float x, y, z; };
// R records per lane (record lane + 64 r): clusters of up to 64 R members.
// the instruction itself: __ballot() takes an int and costs a select and a compare on top
__device__ __forceinline__ u64 fr_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// best key (whi, wlo) of a wave and the second largest running minimum among its points: lanes at the top distance hand in
// their own second best, everybody else their best; several lanes at the top distance make the second best that distance
// itself.  The two reductions behind whi do not depend on each other, so their DPP chains interleave.
__device__ __forceinline__ void fr_wave_top2(unsigned hi, unsigned lo, unsigned sec, unsigned &whi, unsigned &wlo, unsigned &d2) {
    whi = wave_max_u32(hi);
    const bool top = hi == whi;
    const u64 tm = fr_ballot(top);
    wlo = wave_max_u32(top ? lo : 0u);
    const unsigned o = wave_max_u32(top ? sec : hi);
    d2 = (tm & (tm - 1)) ? whi : o;
}

template <int R>
__device__ __forceinline__ FrDecision fr_decide(const bool (&mine)[R], const unsigned (&khi)[R], const unsigned (&klo)[R], const float (&x)[R],
                                                const float (&y)[R], const float (&z)[R], const unsigned (&d2v)[R], const int room, float4 *rec,
                                                u64 *reck, const int lane, const int exp = 0) {
    FrDecision D;
    unsigned dmax = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) dmax = (mine[r] && d2v[r] > dmax) ? d2v[r] : dmax;
#ifdef P2N_FPS_TIMING
    const unsigned S = (exp & 2) ? 0x3c000000u : wave_max_u32(dmax);
#else
    const unsigned S = wave_max_u32(dmax);
#endif
    bool cand[R];
    u64 cmask[R];
    int n = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) { cand[r] = mine[r] && khi[r] > S; cmask[r] = fr_ballot(cand[r]); n += (int)__builtin_popcountll(cmask[r]); }
    if (n > 16) {
        // keep the candidates with the 16 largest distances (a set closed from the top, ties kept or dropped together:
        // the accepted samples are a prefix of the key order anyway)
        unsigned ge[R];
#pragma unroll
        for (int r = 0; r < R; ++r) ge[r] = 0u;
#pragma unroll
        for (int r2 = 0; r2 < R; ++r2) {
            u64 cm = cmask[r2];
            while (cm) {
                const int i = __builtin_ctzll(cm);
                cm &= cm - 1;
                const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)khi[r2], i);
#pragma unroll
                for (int r = 0; r < R; ++r) ge[r] += v >= khi[r] ? 1u : 0u;
            }
        }
        n = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) { cand[r] = cand[r] && ge[r] <= 16u; cmask[r] = fr_ballot(cand[r]); n += (int)__builtin_popcountll(cmask[r]); }
    }
    if (n == 0) {   // the largest distance ties somebody's second best (or 17 bests tie): only the overall best is provably next
        unsigned hm = 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) hm = (mine[r] && khi[r] > hm) ? khi[r] : hm;
        const unsigned ahi = wave_max_u32(hm);
        unsigned lm = 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) lm = (mine[r] && khi[r] == ahi && klo[r] > lm) ? klo[r] : lm;
        const unsigned alo = wave_max_u32(lm);
#pragma unroll
        for (int r = 0; r < R; ++r) { cand[r] = mine[r] && khi[r] == ahi && klo[r] == alo; cmask[r] = fr_ballot(cand[r]); n += (int)__builtin_popcountll(cmask[r]); }
    }
    // candidates compacted into LDS in (r, lane) order; the entries behind them are zero (key 0 is never "earlier", distance
    // 0 is never undercut), so the pair evaluation below needs no validity tests
    if (lane < 16) { rec[lane] = make_float4(0.f, 0.f, 0.f, 0.f); reck[lane] = 0ull; }
    int off = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int pos = off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cmask[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cmask[r], 0u));
        if (cand[r]) { rec[pos] = make_float4(x[r], y[r], z[r], __uint_as_float(khi[r])); reck[pos] = ((u64)khi[r] << 32) | klo[r]; }
        off += (int)__builtin_popcountll(cmask[r]);
    }
    // one wave writes and reads: LDS serves a wave's instructions in order, so only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int i = lane & 15, jb = lane >> 4;
    // all reads first, so that their latencies overlap
    const float4 Ri = rec[i];
    const u64 ki = reck[i];
    float4 Rj[4];
    u64 kj[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) { Rj[pass] = rec[jb + 4 * pass]; kj[pass] = reck[jb + 4 * pass]; }
    const u64 col = 0x0001000100010001ull << i;
    const unsigned colo = (unsigned)col, cohi = (unsigned)(col >> 32);
    u64 many = 0ull;
    unsigned rank = 0u;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#ifdef P2N_FPS_TIMING
        if (4 * pass < n && !(exp & 1)) {
#else
        if (4 * pass < n) {   // uniform
#endif
            const bool bigger = kj[pass] > ki;
            const float dd = sqdist3(Ri.x, Ri.y, Ri.z, Rj[pass].x, Rj[pass].y, Rj[pass].z);   // the update's own expression
            // a sample with a non-finite coordinate never clears its own running minimum (p - p is NaN): the
            // sequential algorithm picks it again at once, so nothing may follow it within a round
            const bool sticky = !__builtin_isfinite((Rj[pass].x + Rj[pass].y) + Rj[pass].z);
            const u64 mb = fr_ballot(bigger);
            many |= fr_ballot(bigger && (dd < Ri.w || sticky));
            rank += (unsigned)__builtin_popcount((unsigned)mb & colo) + (unsigned)__builtin_popcount((unsigned)(mb >> 32) & cohi);
        }
    }
    const bool candp = lane < n;
    int rstar = 64;   // rank of the first candidate an earlier one would change
    if (many != 0ull) {
        const unsigned rblk = (candp && (many & col) != 0ull) ? rank : 64u;
        rstar = 64 - (int)wave_max_u32(64u - rblk);
    }
    const int limit = rstar < room ? rstar : room;
    D.acc = candp && (int)rank < limit;
    D.rank = rank;
    D.x = Ri.x; D.y = Ri.y; D.z = Ri.z; D.klo = (unsigned)ki;
    return D;
}

template <int T, int PPT, bool LDSXYZ, int R>
__global__ __launch_bounds__(T) void p2n_fps_rounds_kernel(FpsArgs A) {   // R: records per decision lane (members <= 64 R)
    constexpr int NWV = T / 64;
    static_assert(PPT % 2 == 0, "points are processed in pairs");
    extern __shared__ __attribute__((aligned(16))) float lxyz[];
    __shared__ unsigned wv[NWV][4];                                     // per wave: best distance bits, inverted index, second best
    __shared__ __attribute__((aligned(16))) float wl[2][FR_MAXW][4];  // the samples a round accepted
    __shared__ int nwl[2];
    __shared__ int verdict_s;
    __shared__ float4 rec[16];
    __shared__ u64 reck[16];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    __builtin_amdgcn_s_setprio(3);
    int cl, g;
    if (A.xcd_mode) {
        const int r = blockIdx.x & 7, q = blockIdx.x >> 3;
        cl = (q / A.G) * 8 + r;
        if (cl >= A.nb) return;
        g = q % A.G;
    } else {
        cl = blockIdx.x / A.G; g = blockIdx.x % A.G;
    }
    const int b = A.b0 + cl;
    const float *P = A.xyz + (long)b * A.N * 3;
    const int base = g * A.chunk;
    int nloc = A.N - base;
    if (nloc > A.chunk) nloc = A.chunk;
    if (nloc < 0) nloc = 0;
    u64 *ring = A.slots + (long)cl * (FR_RING + 1) * A.G * FR_SLOT;   // [FR_RING + 1][G][FR_SLOT]; the last row is the XCD census

    f32x2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], md[PPT / 2];
    if (LDSXYZ)
        for (int i = t; i < nloc * 3; i += T) lxyz[i] = P[(long)base * 3 + i];
    if (t < 4) wl[0][0][t] = t < 3 ? P[(long)A.start_idx * 3 + t] : 0.f;
    if (t == 0) nwl[0] = 1;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = t + T * s;
        const bool ok = i < nloc;
        const float *src = LDSXYZ ? lxyz + i * 3 : P + (long)(base + (ok ? i : 0)) * 3;
        px[s / 2][s & 1] = ok ? src[0] : 0.f;
        py[s / 2][s & 1] = ok ? src[1] : 0.f;
        pz[s / 2][s & 1] = ok ? src[2] : 0.f;
        md[s / 2][s & 1] = ok ? __int_as_float(0x7f800000) : 0.f;
    }
    bool dead = false, census_dead = false, fast = false;
    if (A.xcd_mode && A.G <= 64) {   // XCD census, as in the kernel above
        if (wave == 0) {
            u64 *cs = ring + (long)FR_RING * A.G * FR_SLOT;
            const unsigned myx = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 0xF;
            if (lane == 0) __hip_atomic_store(cs + g * FR_SLOT, (1ull << 32) | (u64)myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64 v = 1ull << 32 | myx;
            int spins = 0;
            bool ok;
            for (;;) {
                if (lane < A.G) v = __hip_atomic_load(cs + lane * FR_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = __all(lane >= A.G || v != 0ull);
                if (ok || ++spins > A.spin_limit) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const bool same = __all(lane >= A.G || (unsigned)v == myx);
            if (lane == 0) verdict_s = ok ? (same ? 1 : 0) : -1;
        }
        __syncthreads();
        const int verdict = verdict_s;
        fast = verdict == 1;
        if (verdict < 0) dead = census_dead = true;
    }
    if (g == 0 && t == 0) A.idx[(long)b * A.M] = A.start_idx;
    const unsigned inv0 = 0xFFFFFFFFu - (unsigned)(base + t);
    int it = 1, nw = 1, buf = 0;
    unsigned round = 0;
    while (!dead && it < A.M) {
        // ---- apply the samples of the last round to the running minima
        for (int j = 0; j < nw; ++j) {
            const float4 c = *reinterpret_cast<const float4 *>(&wl[buf][j][0]);
            const f32x2 c2x = {c.x, c.x}, c2y = {c.y, c.y}, c2z = {c.z, c.z};
#pragma unroll
            for (int p = 0; p < PPT / 2; ++p) {
                const f32x2 dx = px[p] - c2x, dy = py[p] - c2y, dz = pz[p] - c2z;
                const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
                float m0, m1;
                asm("v_min_f32 %0, %1, %2" : "=v"(m0) : "v"(md[p][0]), "v"(dd[0]));
                asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(md[p][1]), "v"(dd[1]));
                md[p][0] = m0;
                md[p][1] = m1;
            }
        }
        ++round;
        // ---- this thread's best (tournament of 64-bit keys) and the largest of its other running minima
        u64 tk[PPT / 2];
#pragma unroll
        for (int p = 0; p < PPT / 2; ++p) {
            const bool take = md[p][1] > md[p][0];
            tk[p] = ((u64)__float_as_uint(take ? md[p][1] : md[p][0]) << 32) | (unsigned)(take ? 62 - 2 * p : 63 - 2 * p);
        }
#pragma unroll
        for (int w = 1; w < PPT / 2; w <<= 1)
#pragma unroll
            for (int p = 0; p < PPT / 2; ++p)
                if (p % (2 * w) == 0 && p + w < PPT / 2) tk[p] = tk[p + w] > tk[p] ? tk[p + w] : tk[p];
        const unsigned bcode = (unsigned)(tk[0] & 0xFFFFFFFFull);
        unsigned sec = 0u;
#pragma unroll
        for (int p = 0; p < PPT / 2; ++p) {
            const unsigned v0 = bcode == (unsigned)(63 - 2 * p) ? 0u : __float_as_uint(md[p][0]);
            const unsigned v1 = bcode == (unsigned)(62 - 2 * p) ? 0u : __float_as_uint(md[p][1]);
            const unsigned v = v0 > v1 ? v0 : v1;
            sec = sec > v ? sec : v;
        }
        const unsigned hi = (unsigned)(tk[0] >> 32);
        const unsigned lo = inv0 - (unsigned)(T * (63 - (int)bcode));
        // ---- wave: best key, and the second best = max over lanes of (the winner lane's second, everybody else's best)
        unsigned whi, wlo, wsec;
        fr_wave_top2(hi, lo, sec, whi, wlo, wsec);
        if (lane == 0) { wv[wave][0] = whi; wv[wave][1] = wlo; wv[wave][2] = wsec; }
        __syncthreads();
        if (wave == 0) {
            const int nbuf = buf ^ 1;
            // workgroup: the same combination over the four waves
            const unsigned h4 = lane < NWV ? wv[lane][0] : 0u, l4 = lane < NWV ? wv[lane][1] : 0u, s4 = lane < NWV ? wv[lane][2] : 0u;
            unsigned k1hi, k1lo, d2;
            fr_wave_top2(h4, l4, s4, k1hi, k1lo, d2);   // lanes past the last wave hand in zeros
            const int li = (int)(0xFFFFFFFFu - k1lo);
            u64 *slot = ring + (long)(round % FR_RING) * A.G * FR_SLOT;
            const unsigned tag = fr_tag(round);
            if (lane < 4 && !(A.fault_it > 0 && it == A.fault_it + 1 && g == A.G - 1)) {
                const bool real = li >= base && li < base + nloc;   // a workgroup without real points publishes zeros with its losing key
                const float *wsrc = LDSXYZ ? lxyz + (real ? li - base : 0) * 3 : P + (long)(real ? li : 0) * 3;
                const unsigned cxb = real ? __float_as_uint(wsrc[0]) : 0u, cyb = real ? __float_as_uint(wsrc[1]) : 0u, czb = real ? __float_as_uint(wsrc[2]) : 0u;
                fr_publish(slot + g * FR_SLOT, lane, tag, k1hi, k1lo, cxb, cyb, czb, d2, fast);
            }
            fr_u32x4 ra[R], rc[R];
            bool mine[R], ready[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { ra[r] = fr_u32x4{0, 0, 0, 0}; rc[r] = fr_u32x4{0, 0, 0, 0}; mine[r] = lane + 64 * r < A.G; ready[r] = !mine[r]; }
            int spins = 0;
            bool ok;
            for (int d = 0; d < A.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            for (;;) {
                bool all = true;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!ready[r]) ready[r] = fr_poll(slot + (lane + 64 * r) * FR_SLOT, tag, ra[r], rc[r]);
                    all = all && ready[r];
                }
                ok = __all(all);
                if (ok || ++spins > A.spin_limit) break;
                __builtin_amdgcn_s_sleep(1);
            }
            unsigned khi[R], klo[R], d2v[R];
            float cx_[R], cy_[R], cz_[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                khi[r] = mine[r] ? ra[r][0] : 0u;
                klo[r] = mine[r] ? ((ra[r][1] & 0xFFFFu) | (ra[r][3] << 16)) : 0u;
                d2v[r] = (rc[r][1] & 0xFFFFu) | (rc[r][3] << 16);
                cx_[r] = __uint_as_float(ra[r][2]); cy_[r] = __uint_as_float(rc[r][0]); cz_[r] = __uint_as_float(rc[r][2]);
            }
            int room = A.M - it;
            if (A.fault_it > 0 && it <= A.fault_it && room > A.fault_it + 1 - it) room = A.fault_it + 1 - it;   // test hook: a round boundary at the faulted sample
            const FrDecision D = fr_decide<R>(mine, khi, klo, cx_, cy_, cz_, d2v, room, rec, reck, lane);
            const int widx = (int)(0xFFFFFFFFu - D.klo);
            int m = (int)__builtin_popcountll(__ballot(D.acc));
            if (!ok || m < 1) m = -1;
            if (D.acc && m > 0) {
                wl[nbuf][D.rank][0] = D.x; wl[nbuf][D.rank][1] = D.y; wl[nbuf][D.rank][2] = D.z; wl[nbuf][D.rank][3] = 0.f;
                if (g == 0) A.idx[(long)b * A.M + it + (int)D.rank] = widx;
            }
            if (lane == 0) nwl[nbuf] = m;
        }
        __syncthreads();
        buf ^= 1;
        nw = nwl[buf];
        if (nw < 0) { dead = true; break; }
        it += nw;
    }
    if (dead) {
        if (t == 0) atomicExch(A.status, 1);
        if (g == 0)
            for (int i = (census_dead ? 0 : it) + t; i < A.M; i += T) A.idx[(long)b * A.M + i] = A.start_idx;
    }
}



#ifdef P2N_FPS_TIMING   // phase clocks of member 0 of cloud 0 (development aid: make HIPFLAGS+=-DP2N_FPS_TIMING)
#define FR_T_DECL long long frt_[6] = {0, 0, 0, 0, 0, 0}, frs_[6] = {0, 0, 0, 0, 0, 0}; int frn_ = 0
#define FR_T(i) do { frt_[i] = (long long)__builtin_readcyclecounter(); if ((i) > 0) frs_[i] += frt_[i] - frt_[(i) - 1]; if ((i) == 5) ++frn_; } while (0)
#define FR_T_DUMP do { if (g == 0 && cl == 0 && lane == 0) printf("fps rounds=%d samples=%d  cycles per round: update %lld reduce %lld publish %lld poll %lld decide %lld\n", frn_, it, frs_[1] / (frn_ ? frn_ : 1), frs_[2] / (frn_ ? frn_ : 1), frs_[3] / (frn_ ? frn_ : 1), frs_[4] / (frn_ ? frn_ : 1), frs_[5] / (frn_ ? frn_ : 1)); } while (0)
#else
#define FR_T_DECL
#define FR_T(i)
#define FR_T_DUMP
#endif
// --------------------------------------------------------------------------- the same rounds with ONE WAVE per member
// A cluster member is a single wave (a 64-thread workgroup): no barrier, no LDS hand-over inside a round -- the wave
// updates its points, reduces, publishes, polls, decides and keeps the accepted samples in the registers it polled them
// into.  Up to 64 members per cloud (one decision lane each), placed on one XCD, one per SIMD.  More members mean
// more candidates per round (tools/sim_fps_rounds.py: 7 samples per round at 64 members) and fewer points per lane.
template <int PPT, int R>
__global__ __launch_bounds__(64) void p2n_fps_wave_rounds_kernel(FpsArgs A) {   // R: records per decision lane (members <= 64 R)
    constexpr int T = 64;
    static_assert(PPT % 2 == 0, "points are processed in pairs");
    extern __shared__ __attribute__((aligned(16))) float lxyz[];
    __shared__ float4 rec[16];
    __shared__ u64 reck[16];
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    int cl, g;
    if (A.xcd_mode) {
        const int r = blockIdx.x & 7, q = blockIdx.x >> 3;
        cl = (q / A.G) * 8 + r;
        if (cl >= A.nb) return;
        g = q % A.G;
    } else {
        cl = blockIdx.x / A.G; g = blockIdx.x % A.G;
    }
    const int b = A.b0 + cl;
    const float *P = A.xyz + (long)b * A.N * 3;
    const int base = g * A.chunk;
    int nloc = A.N - base;
    if (nloc > A.chunk) nloc = A.chunk;
    if (nloc < 0) nloc = 0;
    u64 *ring = A.slots + (long)cl * (FR_RING + 1) * A.G * FR_SLOT;

    f32x2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], md[PPT / 2];
    for (int i = lane; i < nloc * 3; i += T) lxyz[i] = P[(long)base * 3 + i];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = lane + T * s;
        const bool ok = i < nloc;
        const float *src = lxyz + i * 3;
        px[s / 2][s & 1] = ok ? src[0] : 0.f;
        py[s / 2][s & 1] = ok ? src[1] : 0.f;
        pz[s / 2][s & 1] = ok ? src[2] : 0.f;
        md[s / 2][s & 1] = ok ? __int_as_float(0x7f800000) : 0.f;
    }
    bool dead = false, census_dead = false, fast = false;
    if (A.xcd_mode) {
        u64 *cs = ring + (long)FR_RING * A.G * FR_SLOT;
        const unsigned myx = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 0xF;
        if (lane == 0) __hip_atomic_store(cs + g * FR_SLOT, (1ull << 32) | (u64)myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        bool ok, same;
        for (;;) {
            ok = true; same = true;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                u64 v = 1ull << 32 | myx;
                if (lane + 64 * r < A.G) v = __hip_atomic_load(cs + (lane + 64 * r) * FR_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && v != 0ull;
                same = same && (unsigned)v == myx;
            }
            ok = __all(ok);
            if (ok || ++spins > A.spin_limit) break;
            __builtin_amdgcn_s_sleep(1);
        }
        fast = ok && __all(same);
        if (!ok) dead = census_dead = true;
    }
    if (g == 0 && lane == 0) A.idx[(long)b * A.M] = A.start_idx;
    const unsigned inv0 = 0xFFFFFFFFu - (unsigned)(base + lane);
    // the accepted samples of the last round: coordinates in the lanes named by `am` (first round: the start point)
    float x = P[(long)A.start_idx * 3 + 0], y = P[(long)A.start_idx * 3 + 1], z = P[(long)A.start_idx * 3 + 2];
    u64 am = 1ull;
    int it = 1;
    unsigned round = 0;
    FR_T_DECL;
    while (!dead && it < A.M) {
        FR_T(0);
        while (am) {
            const int i = __builtin_ctzll(am);
            am &= am - 1;
            const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
            const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), i));
            const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), i));
            const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
            for (int p = 0; p < PPT / 2; ++p) {
                const f32x2 dx = px[p] - c2x, dy = py[p] - c2y, dz = pz[p] - c2z;
                const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
                float m0, m1;
                asm("v_min_f32 %0, %1, %2" : "=v"(m0) : "v"(md[p][0]), "v"(dd[0]));
                asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(md[p][1]), "v"(dd[1]));
                md[p][0] = m0;
                md[p][1] = m1;
            }
        }
        ++round;
        FR_T(1);
        u64 tk[PPT / 2];
#pragma unroll
        for (int p = 0; p < PPT / 2; ++p) {
            const bool take = md[p][1] > md[p][0];
            tk[p] = ((u64)__float_as_uint(take ? md[p][1] : md[p][0]) << 32) | (unsigned)(take ? 62 - 2 * p : 63 - 2 * p);
        }
#pragma unroll
        for (int w = 1; w < PPT / 2; w <<= 1)
#pragma unroll
            for (int p = 0; p < PPT / 2; ++p)
                if (p % (2 * w) == 0 && p + w < PPT / 2) tk[p] = tk[p + w] > tk[p] ? tk[p + w] : tk[p];
        const unsigned bcode = (unsigned)(tk[0] & 0xFFFFFFFFull);
        unsigned sec = 0u;
#pragma unroll
        for (int p = 0; p < PPT / 2; ++p) {
            const unsigned v0 = bcode == (unsigned)(63 - 2 * p) ? 0u : __float_as_uint(md[p][0]);
            const unsigned v1 = bcode == (unsigned)(62 - 2 * p) ? 0u : __float_as_uint(md[p][1]);
            const unsigned v = v0 > v1 ? v0 : v1;
            sec = sec > v ? sec : v;
        }
        const unsigned hi = (unsigned)(tk[0] >> 32);
        const unsigned lo = inv0 - (unsigned)(T * (63 - (int)bcode));
        unsigned k1hi, k1lo, d2;
        fr_wave_top2(hi, lo, sec, k1hi, k1lo, d2);
        FR_T(2);
        const int li = (int)(0xFFFFFFFFu - k1lo);
        u64 *slot = ring + (long)(round % FR_RING) * A.G * FR_SLOT;
        const unsigned tag = fr_tag(round);
        if (lane < 4 && !(A.fault_it > 0 && it == A.fault_it + 1 && g == A.G - 1)) {
            const bool real = li >= base && li < base + nloc;
            const unsigned cxb = real ? __float_as_uint(lxyz[(li - base) * 3 + 0]) : 0u, cyb = real ? __float_as_uint(lxyz[(li - base) * 3 + 1]) : 0u,
                           czb = real ? __float_as_uint(lxyz[(li - base) * 3 + 2]) : 0u;
            fr_publish(slot + g * FR_SLOT, lane, tag, k1hi, k1lo, cxb, cyb, czb, d2, fast);
        }
        FR_T(3);
        fr_u32x4 ra[R], rc[R];
        bool mine[R], ready[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { ra[r] = fr_u32x4{0, 0, 0, 0}; rc[r] = fr_u32x4{0, 0, 0, 0}; mine[r] = lane + 64 * r < A.G; ready[r] = !mine[r]; }
        int spins = 0;
        bool ok;
        for (int d = 0; d < A.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
        for (;;) {
            bool all = true;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!ready[r]) ready[r] = fr_poll(slot + (lane + 64 * r) * FR_SLOT, tag, ra[r], rc[r]);
                all = all && ready[r];
            }
            ok = __all(all);
            if (ok || ++spins > A.spin_limit) break;
            __builtin_amdgcn_s_sleep(1);
        }
        FR_T(4);
        unsigned khi[R], klo[R], d2v[R];
        float cx_[R], cy_[R], cz_[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            khi[r] = mine[r] ? ra[r][0] : 0u;
            klo[r] = mine[r] ? ((ra[r][1] & 0xFFFFu) | (ra[r][3] << 16)) : 0u;
            d2v[r] = (rc[r][1] & 0xFFFFu) | (rc[r][3] << 16);
            cx_[r] = __uint_as_float(ra[r][2]); cy_[r] = __uint_as_float(rc[r][0]); cz_[r] = __uint_as_float(rc[r][2]);
        }
        int room = A.M - it;
        if (A.fault_it > 0 && it <= A.fault_it && room > A.fault_it + 1 - it) room = A.fault_it + 1 - it;
        const FrDecision D = fr_decide<R>(mine, khi, klo, cx_, cy_, cz_, d2v, room, rec, reck, lane, A.exp);
        x = D.x; y = D.y; z = D.z;
        const int widx = (int)(0xFFFFFFFFu - D.klo);
        am = __ballot(D.acc);
        const int m = (int)__builtin_popcountll(am);
        if (!ok || m < 1) { dead = true; break; }
        if (D.acc && g == 0) A.idx[(long)b * A.M + it + (int)D.rank] = widx;
        it += m;
        FR_T(5);
    }
    FR_T_DUMP;
    if (dead) {
        if (lane == 0) atomicExch(A.status, 1);
        if (g == 0)
            for (int i = (census_dead ? 0 : it) + lane; i < A.M; i += T) A.idx[(long)b * A.M + i] = A.start_idx;
    }
}

static int fps_cus();
static const int kFpsPptW[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48};
struct FpsWavePlan { int G, ppt; };
// members (single waves) per cloud when nb clouds share the launch: one XCD per cloud, at most one member per SIMD
static FpsWavePlan fps_wave_plan(int nb, int N) {
    FpsWavePlan w = {0, 0};
    const int ncu = fps_cus();
    const int cap = (ncu / 8) * 4, per = (nb + 7) / 8;
    int G = cap / per;
    // one cloud per XCD and 50,000 points or more: a member on every SIMD (128: two records per decision lane) -- measured
    // 60k 1.83 -> 1.63 ms, 100k 2.59 -> 2.04, 150k 2.94 -> 2.28; below, 64 members (40k: 3.29 vs 3.16, 8 x 40k: 0.89 vs 0.94, 8,192: 0.60 vs 0.74)
    if (G > (N >= 50000 ? 128 : 64)) G = N >= 50000 ? 128 : 64;
    if (const int g = p2n_tuning().fps_g) { if (g >= 2 && g <= G) G = g; }
    while (G > 2 && (N + G - 1) / G < 64) G >>= 1;   // at least one point per lane
    if (G < 2) return w;
    const int chunk = (N + G - 1) / G, need = (chunk + 63) / 64;
    for (int i = 0; i < 12; ++i)
        if (kFpsPptW[i] >= need) { w.ppt = kFpsPptW[i]; break; }
    if (w.ppt == 0 || (size_t)chunk * 12 > 64 * 1024) return w;
    w.G = G;
    return w;
}

// which clouds take the wave clusters (P2N_FPS_ROUNDS: 2 = always where a plan exists, 1 / -1 = never)
static bool fps_wave_rounds_wanted(int B, int N) {
    const int mode = p2n_tuning().fps_rounds;
    if (mode == 1 || mode < 0) return false;
    const FpsWavePlan w = fps_wave_plan(B < 64 ? B : 64, N);
    if (w.G < 2) return false;
    if (mode == 2) return true;
    // measured (profiles/r03_h_fps_rounds_ab*.txt): below ~4,096 points one workgroup without any exchange is as fast;
    // beyond ~160,000 a member's 1,200+ points per wave make the update the round, and 256-thread members are as fast
    return N >= 4096 && N <= 160000 && w.G >= 8;
}

template <int PPT, int R>
static hipError_t fps_wave_launch_r(const FpsArgs &a, int nb, hipStream_t s) {
    const size_t lds = (size_t)a.chunk * 12;
    const int ps = p2n_prof_begin(P2N_K_FPS, s);
    const unsigned grid = a.xcd_mode ? 8u * a.G * ((nb + 7) / 8) : (unsigned)(nb * a.G);
    hipLaunchKernelGGL((p2n_fps_wave_rounds_kernel<PPT, R>), dim3(grid), dim3(64), lds, s, a);
    p2n_prof_end(ps, s);
    return hipGetLastError();
}
template <int PPT>
static hipError_t fps_wave_launch(const FpsArgs &a, int nb, hipStream_t s) {
    return a.G > 64 ? fps_wave_launch_r<PPT, 2>(a, nb, s) : fps_wave_launch_r<PPT, 1>(a, nb, s);
}

// bucket-pruned single-workgroup variant (defined behind the Morton sort it builds on)
static size_t fps_pruned_bytes(int B, int N, int M);
static int fps_pruned_run(const float *xyz, int B, int N, int M, int start_idx, int32_t *idx, void *scratch, hipStream_t s);

// Launch plan: workgroups per cloud, threads, points per thread.
struct FpsPlan { int G, T, ppt; };
static const int kFpsPpt256[] = {2, 4, 8, 16, 24, 32, 40, 48};
static const int kFpsPpt512[] = {2, 4, 8, 12, 16, 24};
static const int kFpsPpt1024[] = {16, 20};   // only chunks beyond FPS_LDS_POINTS get here
// compute units of the current device (all workgroups of a launch must be co-resident: one per CU)
static int fps_cus() {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 64;
        cus[dev] = n;
    }
    return cus[dev];
}

static FpsPlan fps_plan(int B, int N) {
    FpsPlan p = {0, 256, 0};
    const int ncu = fps_cus();
    int gmax = ncu / (B < ncu ? B : ncu);   // all B*G workgroups must be co-resident
    if (gmax < 1) gmax = 1;
    const int gcap = p2n_tuning().fps_rounds >= 0 ? 256 : 64;   // members one decision wave can serve: 4 records per lane in the round kernels, 1 in the one-sample kernel
    if (gmax > gcap) gmax = gcap;
    int G;
    const int gmin256 = (N + FPS_LDS_POINTS - 1) / FPS_LDS_POINTS;
    if (gmin256 <= gmax) {
        // fits 256-thread workgroups.  One workgroup whenever the cloud fits it (the exchange costs
        // more than the distance updates it saves); otherwise ~4096 points per workgroup.
        G = gmin256;
        if (G > 1) {
            // Measured, not guessed (profiles/r02_k_fps_sweep.txt, r03_*_fps_sweep.txt): once a cloud needs a cluster at
            // all, an iteration is the exchange plus the reduction chain, and both shrink with fewer points per thread
            // down to ~1,024 points per workgroup -- 40,000 points: G = 10 13.6 ms, 16 13.0, 32 11.7; 150,000 points:
            // G = 16 6.9 ms, 32 5.7, 64 (two XCDs, write-through exchange) 8.7.  So: every CU of ONE XCD that the
            // clouds of this launch leave to a cloud (8 XCDs, clouds dealt over them), never beyond the XCD while the
            // cloud still fits it.
            const int per_xcd = ncu / 8;
            const int share = per_xcd / ((B + 7) / 8);          // CUs of an XCD per cloud when the launch's clouds are dealt over the 8 XCDs
            G = (N + 1023) / 1024;
            // the one-sample kernel stays inside one XCD (plain-store exchange through its L2) whenever the cloud fits it; the
            // round kernels only see clouds beyond 100,000 points by default, where ~1,000 points per member across XCDs
            // measured faster than 32 fat members in one (profiles/r03_i_fps_large.txt: 150k 2.96 -> 2.53 ms, 300k 6.76 -> 5.20)
            if (share >= gmin256 && (p2n_tuning().fps_rounds < 0 || N <= 100000)) { if (G > share) G = share; }
            if (G > gmax) G = gmax;
            if (G < gmin256) G = gmin256;
        }
        p.T = 256;
    } else {
        const int cap = 1024 * 20;
        G = (N + cap - 1) / cap;
        const int g1024 = gmax < 64 ? gmax : 64;   // the 1024-thread kernels read one record per decision lane
        if (G <= g1024) {   // spare CUs: ~8 points per thread
            int gw = (N + 8191) / 8192;
            if (gw > g1024) gw = g1024;
            if (gw > G) G = gw;
        }
        if (G > g1024) return p;
        p.T = 1024;
    }
    if (const int g = p2n_tuning().fps_g) {   // tuning override (tools/fps_sweep.py)
        if (g >= 1 && g <= gmax) {
            const int chunk = (N + g - 1) / g;
            if (chunk <= FPS_LDS_POINTS) { G = g; p.T = 256; }
            else if (chunk <= 1024 * 20 && gmin256 > gmax) { G = g; p.T = 1024; }
        }
    }
    if (G > gmax) return p;   // does not fit one launch: p.G stays 0
    const int chunk = (N + G - 1) / G;
    if (p.T == 256 && p2n_tuning().fps_t == 512 && chunk <= 512 * 24) p.T = 512;
    const int need = (chunk + p.T - 1) / p.T;
    const int *tab = p.T == 256 ? kFpsPpt256 : p.T == 512 ? kFpsPpt512 : kFpsPpt1024;
    const int ntab = p.T == 256 ? 8 : p.T == 512 ? 6 : 2;
    for (int i = 0; i < ntab; ++i)
        if (tab[i] >= need) { p.ppt = tab[i]; break; }
    if (p.ppt == 0) return p;
    p.G = G;
    return p;
}

// exchange slots of B clouds at the largest cluster (64 workgroups): per-iteration slots of the one-sample kernel or
// the record ring of the rounds kernel, whichever is larger
static size_t fps_slot_bytes(int B, int M, bool one_sample_kernel) {
    const size_t a = one_sample_kernel ? sizeof(u64) * 4 * (size_t)(M + 1) * 64 : 0, r = sizeof(u64) * FR_SLOT * (size_t)(FR_RING + 1) * 256;
    return (size_t)B * (a > r ? a : r);
}

extern "C" size_t p2n_fps_workspace_bytes(int B, int N, int M) {
    if (B < 1 || N < 1 || M < 1) return 16;
    // slots of the largest sub-batch: at most 64 workgroups per cloud, 32 bytes each per sample
    const FpsPlan p = fps_plan(1, N);
    const int gmax = p.G > 1 ? 64 : 1;
    return 256 + fps_slot_bytes(B, M, gmax > 1) + fps_pruned_bytes(B, N, M);
}

template <int T, int PPT, bool LDSXYZ>
static hipError_t fps_launch(const FpsArgs &a, int nb, hipStream_t s) {
    const size_t lds = LDSXYZ ? (size_t)a.chunk * 12 : 0;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(p2n_fps_kernel<T, PPT, LDSXYZ>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int ps = p2n_prof_begin(P2N_K_FPS, s);
    const unsigned grid = a.xcd_mode ? 8u * a.G * ((nb + 7) / 8) : (unsigned)(nb * a.G);
    hipLaunchKernelGGL((p2n_fps_kernel<T, PPT, LDSXYZ>), dim3(grid), dim3(T), lds, s, a);
    p2n_prof_end(ps, s);
    return hipGetLastError();
}

template <int T, int PPT, bool LDSXYZ, int R>
static hipError_t fps_rounds_launch(const FpsArgs &a, int nb, hipStream_t s) {
    const size_t lds = LDSXYZ ? (size_t)a.chunk * 12 : 0;
    if (lds > 32 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(p2n_fps_rounds_kernel<T, PPT, LDSXYZ, R>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int ps = p2n_prof_begin(P2N_K_FPS, s);
    const unsigned grid = a.xcd_mode ? 8u * a.G * ((nb + 7) / 8) : (unsigned)(nb * a.G);
    hipLaunchKernelGGL((p2n_fps_rounds_kernel<T, PPT, LDSXYZ, R>), dim3(grid), dim3(T), lds, s, a);
    p2n_prof_end(ps, s);
    return hipGetLastError();
}

extern "C" int p2n_fps_cluster_kind(int B, int N) {
    if (B < 1 || N < 1) return 0;
    if (fps_wave_rounds_wanted(B, N)) return 1;
    return p2n_fps_workgroups_per_cloud(B, N) > 1 ? 2 : 0;
}

extern "C" int p2n_fps_workgroups_per_cloud(int B, int N) {
    if (B >= 1 && N >= 1 && fps_wave_rounds_wanted(B, N)) return fps_wave_plan(B < 64 ? B : 64, N).G;
    // the LARGEST cluster p2n_fps(B, N, ...) will actually launch: the same sub-batching and halving as below, so a
    // cloud that only fits with fewer clouds in flight (B = 64 x 100,000 points) reports its clusters, not 0
    if (B < 1 || N < 1) return 0;
    const int ncu = fps_cus();
    int done = 0, gmax = 0;
    while (done < B) {
        int nb = B - done;
        if (nb > ncu) nb = ncu;
        FpsPlan p = fps_plan(nb, N);
        while (p.G == 0 && nb > 1) { nb = nb / 2; p = fps_plan(nb, N); }
        if (p.G == 0) return 0;   // p2n_fps refuses this shape
        gmax = p.G > gmax ? p.G : gmax;
        done += nb;
    }
    return gmax;
}

extern "C" int p2n_fps(const float *xyz, int B, int N, int M, int start_idx, int32_t *idx, int32_t *status,
                       void *workspace, size_t workspace_bytes, void *stream_) {
    P2N_REQUIRE(B >= 0 && N >= 1 && M >= 0, "p2n_fps: bad shape B=%d N=%d M=%d", B, N, M);
    P2N_REQUIRE(start_idx >= 0 && start_idx < N, "p2n_fps: start_idx %d outside [0,%d)", start_idx, N);
    if (B == 0 || M == 0) return P2N_OK;
    P2N_REQUIRE(xyz && idx, "p2n_fps: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t need = p2n_fps_workspace_bytes(B, N, M);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_fps: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    if (!status) P2N_CHECK_HIP(hipMemsetAsync(workspace, 0, 16, stream));   // no private status word: the workspace's
    const int spin = p2n_tuning().fps_spin > 0 ? p2n_tuning().fps_spin : FPS_SPIN_LIMIT;
    if (fps_wave_rounds_wanted(B, N)) {
        int done = 0;
        while (done < B) {
            const int nb = B - done < 64 ? B - done : 64;
            const FpsWavePlan w = fps_wave_plan(nb, N);
            P2N_REQUIRE(w.G >= 2, "p2n_fps: no wave-cluster plan for N=%d", N);
            FpsArgs a;
            a.xyz = xyz; a.idx = idx; a.status = status ? status : (int *)workspace; a.spin_limit = spin; a.fault_it = p2n_tuning().fps_fault; a.poll_delay = p2n_tuning().fps_pd; a.exp = p2n_tuning().fps_exp;
            a.nb = nb; a.xcd_mode = !p2n_tuning().fps_noxcd && fps_cus() % 8 == 0;
            a.slots = (u64 *)((char *)workspace + 256);
            a.N = N; a.M = M; a.G = w.G; a.chunk = (N + w.G - 1) / w.G; a.start_idx = start_idx; a.b0 = done;
            P2N_CHECK_HIP(hipMemsetAsync((char *)workspace + 256, 0, sizeof(u64) * FR_SLOT * (size_t)nb * (FR_RING + 1) * w.G, stream));
            hipError_t e = hipErrorInvalidValue;
            switch (w.ppt) {
                case 2: e = fps_wave_launch<2>(a, nb, stream); break;
                case 4: e = fps_wave_launch<4>(a, nb, stream); break;
                case 6: e = fps_wave_launch<6>(a, nb, stream); break;
                case 8: e = fps_wave_launch<8>(a, nb, stream); break;
                case 10: e = fps_wave_launch<10>(a, nb, stream); break;
                case 12: e = fps_wave_launch<12>(a, nb, stream); break;
                case 16: e = fps_wave_launch<16>(a, nb, stream); break;
                case 20: e = fps_wave_launch<20>(a, nb, stream); break;
                case 24: e = fps_wave_launch<24>(a, nb, stream); break;
                case 32: e = fps_wave_launch<32>(a, nb, stream); break;
                case 40: e = fps_wave_launch<40>(a, nb, stream); break;
                case 48: e = fps_wave_launch<48>(a, nb, stream); break;
            }
            P2N_CHECK_HIP(e);
            done += nb;
        }
        return P2N_OK;
    }
    if (fps_pruned_bytes(B, N, M)) {   // one workgroup per cloud, no exchange: nothing can time out, the status stays as it is
        const FpsPlan p1 = fps_plan(1, N);
        const size_t off = 256 + fps_slot_bytes(B, M, p1.G > 1);
        return fps_pruned_run(xyz, B, N, M, start_idx, idx, (char *)workspace + off, stream);
    }
    // sub-batches so that every launch's workgroups are co-resident (<= 256)
    int done = 0;
    const int ncu = fps_cus();
    while (done < B) {
        int nb = B - done;
        if (nb > ncu) nb = ncu;
        FpsPlan p = fps_plan(nb, N);
        while (p.G == 0 && nb > 1) {   // cloud too large to be resident with this many clouds in flight
            nb = nb / 2;
            p = fps_plan(nb, N);
        }
        P2N_REQUIRE(p.G >= 1, "p2n_fps: N=%d exceeds the on-chip capacity of one launch (%d points)", N, 256 * FPS_LDS_POINTS);
        FpsArgs a;
        a.xyz = xyz; a.idx = idx; a.status = status ? status : (int *)workspace; a.spin_limit = spin; a.fault_it = p2n_tuning().fps_fault; a.poll_delay = p2n_tuning().fps_pd; a.exp = p2n_tuning().fps_exp;
        a.nb = nb;
        // one XCD per cloud when its cluster and the other clouds sharing that XCD fit the XCD's 32 CUs
        a.xcd_mode = p.G > 1 && p.G * ((nb + 7) / 8) <= ncu / 8 && !p2n_tuning().fps_noxcd;
        a.slots = (u64 *)((char *)workspace + 256);
        a.N = N; a.M = M; a.G = p.G; a.chunk = (N + p.G - 1) / p.G; a.start_idx = start_idx; a.b0 = done;
        const bool rounds = p.G > 1 && p.T != 512 && p2n_tuning().fps_rounds >= 0;   // P2N_FPS_ROUNDS=-1: one sample per exchange (A/B)
        if (p.G > 1)   // launches run in stream order, so they can share the slot region
            P2N_CHECK_HIP(hipMemsetAsync((char *)workspace + 256, 0, rounds ? sizeof(u64) * FR_SLOT * (size_t)nb * (FR_RING + 1) * p.G
                                                                            : sizeof(u64) * 4 * (size_t)nb * (M + 1) * p.G, stream));
        hipError_t e = hipErrorInvalidValue;
        if (rounds && p.T == 256 && p.G <= 64) {
            switch (p.ppt) {
                case 2: e = fps_rounds_launch<256, 2, true, 1>(a, nb, stream); break;
                case 4: e = fps_rounds_launch<256, 4, true, 1>(a, nb, stream); break;
                case 8: e = fps_rounds_launch<256, 8, true, 1>(a, nb, stream); break;
                case 16: e = fps_rounds_launch<256, 16, true, 1>(a, nb, stream); break;
                case 24: e = fps_rounds_launch<256, 24, true, 1>(a, nb, stream); break;
                case 32: e = fps_rounds_launch<256, 32, true, 1>(a, nb, stream); break;
                case 40: e = fps_rounds_launch<256, 40, true, 1>(a, nb, stream); break;
                case 48: e = fps_rounds_launch<256, 48, true, 1>(a, nb, stream); break;
            }
        } else if (rounds && p.T == 256) {   // more than 64 members: four records per decision lane
            switch (p.ppt) {
                case 2: e = fps_rounds_launch<256, 2, true, 4>(a, nb, stream); break;
                case 4: e = fps_rounds_launch<256, 4, true, 4>(a, nb, stream); break;
                case 8: e = fps_rounds_launch<256, 8, true, 4>(a, nb, stream); break;
                case 16: e = fps_rounds_launch<256, 16, true, 4>(a, nb, stream); break;
                case 24: e = fps_rounds_launch<256, 24, true, 4>(a, nb, stream); break;
                case 32: e = fps_rounds_launch<256, 32, true, 4>(a, nb, stream); break;
                case 40: e = fps_rounds_launch<256, 40, true, 4>(a, nb, stream); break;
                case 48: e = fps_rounds_launch<256, 48, true, 4>(a, nb, stream); break;
            }
        } else if (rounds) {
            switch (p.ppt) {
                case 16: e = fps_rounds_launch<1024, 16, false, 1>(a, nb, stream); break;
                case 20: e = fps_rounds_launch<1024, 20, false, 1>(a, nb, stream); break;
            }
        } else if (p.T == 256) {
            switch (p.ppt) {
                case 2: e = fps_launch<256, 2, true>(a, nb, stream); break;
                case 4: e = fps_launch<256, 4, true>(a, nb, stream); break;
                case 8: e = fps_launch<256, 8, true>(a, nb, stream); break;
                case 16: e = fps_launch<256, 16, true>(a, nb, stream); break;
                case 24: e = fps_launch<256, 24, true>(a, nb, stream); break;
                case 32: e = fps_launch<256, 32, true>(a, nb, stream); break;
                case 40: e = fps_launch<256, 40, true>(a, nb, stream); break;
                case 48: e = fps_launch<256, 48, true>(a, nb, stream); break;
            }
        } else if (p.T == 512) {
            switch (p.ppt) {
                case 2: e = fps_launch<512, 2, true>(a, nb, stream); break;
                case 4: e = fps_launch<512, 4, true>(a, nb, stream); break;
                case 8: e = fps_launch<512, 8, true>(a, nb, stream); break;
                case 12: e = fps_launch<512, 12, true>(a, nb, stream); break;
                case 16: e = fps_launch<512, 16, true>(a, nb, stream); break;
                case 24: e = fps_launch<512, 24, true>(a, nb, stream); break;
            }
        } else {
            switch (p.ppt) {
                case 16: e = fps_launch<1024, 16, false>(a, nb, stream); break;
                case 20: e = fps_launch<1024, 20, false>(a, nb, stream); break;
            }
        }
        P2N_CHECK_HIP(e);
        done += nb;
    }
    return P2N_OK;
}

// =========================================================================== ball query
// One WAVE serves two queries (the kNN kernel's shape): the cloud streams through LDS as SoA tiles,
// the 64 lanes test 64 consecutive points against both queries with packed fp32 arithmetic, a ballot
// gives the hits in index order and a lane's rank among them (popcount of the lower lanes) is its
// output slot.  A query is finished as soon as it has nsample hits (wave-uniform, no divergence), a
// workgroup as soon as all its queries are.  The first version ran one THREAD per query: 2 waves per
// CU at the set-abstraction shapes and a scattered store per hit.
#define BQ_WAVES 8
#define BQ_TILE 4096   // points per LDS tile (48 KiB)

__global__ __launch_bounds__(BQ_WAVES * 64) void p2n_ball_query_kernel(const float *__restrict__ xyz,
                                                                      const float *__restrict__ new_xyz, int N, int M,
                                                                      float r2, int ns, int32_t *__restrict__ idx) {
    __shared__ __attribute__((aligned(16))) float sx[BQ_TILE], sy[BQ_TILE], sz[BQ_TILE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar, see p2n_knn_sorted_kernel
    const int b = blockIdx.y;
    const long m0 = 2 * ((long)blockIdx.x * BQ_WAVES + wave);
    const float *P = xyz + (long)b * N * 3;
    f32x2 qx = {0.f, 0.f}, qy = {0.f, 0.f}, qz = {0.f, 0.f};
    int cnt[2], first[2] = {0, 0};
    int32_t *o[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const bool act = m0 + q < M;
        const long m = act ? m0 + q : 0;
        const float *c = new_xyz + ((long)b * M + m) * 3;
        qx[q] = c[0]; qy[q] = c[1]; qz[q] = c[2];
        o[q] = idx + ((long)b * M + m) * ns;
        cnt[q] = act ? 0 : ns;   // a query past the end counts as finished and writes nothing
    }
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));   // lanes lower than this one
    for (int t0 = 0; t0 < N; t0 += BQ_TILE) {
        const int tn = min(BQ_TILE, N - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += BQ_WAVES * 64) {
            sx[i] = P[(long)(t0 + i) * 3 + 0];
            sy[i] = P[(long)(t0 + i) * 3 + 1];
            sz[i] = P[(long)(t0 + i) * 3 + 2];
        }
        __syncthreads();
        if (__syncthreads_and(cnt[0] >= ns && cnt[1] >= ns)) break;
        for (int i0 = 0; i0 < tn && (cnt[0] < ns || cnt[1] < ns); i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < tn;
            const float x = in ? sx[i] : 0.f, y = in ? sy[i] : 0.f, z = in ? sz[i] : 0.f;
            // ((dx*dx + dy*dy) + dz*dz) with the candidate first, as sqdist3(point, query): every operation
            // rounded on its own (-ffp-contract=off), both queries at once
            const f32x2 dx = f32x2{x, x} - qx, dy = f32x2{y, y} - qy, dz = f32x2{z, z} - qz;
            const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (cnt[q] >= ns) continue;   // wave-uniform
                const bool hit = in && dd[q] < r2;
                const unsigned long long mask = __ballot(hit);
                if (!mask) continue;
                if (cnt[q] == 0) first[q] = t0 + i0 + __builtin_ctzll(mask);
                const int slot = cnt[q] + __builtin_popcountll(mask & below);
                if (hit && slot < ns) o[q][slot] = t0 + i;
                cnt[q] += __builtin_popcountll(mask);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (m0 + q < M)
            for (int s = cnt[q] + lane; s < ns; s += 64) o[q][s] = first[q];   // pad with the first hit (0 if none)
}

extern "C" int p2n_ball_query(const float *xyz, const float *new_xyz, int B, int N, int M, float radius, int nsample,
                              int32_t *idx, void *stream) {
    P2N_REQUIRE(B >= 0 && N >= 1 && M >= 0 && nsample >= 1, "p2n_ball_query: bad shape B=%d N=%d M=%d nsample=%d", B,
                N, M, nsample);
    if (B == 0 || M == 0) return P2N_OK;
    P2N_REQUIRE(xyz && new_xyz && idx, "p2n_ball_query: null pointer");
    P2N_REQUIRE(B <= 65535, "p2n_ball_query: B > 65535");
    const long blocks = ((long)M + 2 * BQ_WAVES - 1) / (2 * BQ_WAVES);
    const int ps = p2n_prof_begin(P2N_K_BALL, (hipStream_t)stream);
    hipLaunchKernelGGL(p2n_ball_query_kernel, dim3((unsigned)blocks, B), dim3(BQ_WAVES * 64), 0, (hipStream_t)stream, xyz,
                       new_xyz, N, M, radius * radius, nsample, idx);
    p2n_prof_end(ps, (hipStream_t)stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// =========================================================================== kNN
// One WAVE per query.  The wave's current k best (d2, index) pairs live one per lane,
// sorted by lane number, in two VGPRs: no LDS lists, no divergence.  Each step the 64
// lanes test 64 consecutive reference points (SoA tile in LDS, conflict-free reads)
// against the k-th best; survivors (a ballot mask, walked in index order) are inserted
// by one DPP shift-right-by-one of the sorted list.  Candidates arrive in ascending
// index order, so "strictly smaller distance goes in front" is the lowest-index
// tie-break of the oracle.
#define KNN_WAVES 8
#define KNN_TILE 4096   // reference points per LDS tile (48 KiB)

__device__ __forceinline__ float dpp_wave_shr1(float v, float fill) {
    // lane l <- lane l-1 across the whole wave (DPP wave_shr:1, one VALU op); lane 0 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xF, 0xF, false));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One wave serves TWO queries: every lane measures its candidate against both with packed fp32
// arithmetic (v_pk_add/mul_f32: same IEEE roundings as the scalar form, half the instructions), so the
// LDS reads and the loop overhead are shared as well.  Each query keeps its own sorted list.
__global__ __launch_bounds__(KNN_WAVES * 64) void p2n_knn_kernel(const float *__restrict__ query,
                                                                const float *__restrict__ ref, int M, int N, int k,
                                                                int32_t *__restrict__ idx, float *__restrict__ d2out,
                                                                float *__restrict__ wout) {
    __shared__ __attribute__((aligned(16))) float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // (as a scalar: 5 % slower at 1M reference points, 4-10 % faster on small clouds)
    const int b = blockIdx.y;
    const long m0 = 2 * ((long)blockIdx.x * KNN_WAVES + wave);
    const bool active = m0 < M;
    const bool two = m0 + 1 < M;
    const float *P = ref + (long)b * N * 3;
    f32x2 qx = {0.f, 0.f}, qy = {0.f, 0.f}, qz = {0.f, 0.f};
    if (active) {
        const float *q = query + ((long)b * M + m0) * 3;
        qx[0] = q[0]; qy[0] = q[1]; qz[0] = q[2];
        const float *q1 = two ? q + 3 : q;   // an odd tail repeats the query; its second list is not written
        qx[1] = q1[0]; qy[1] = q1[1]; qz[1] = q1[2];
    }
    const float INF = __int_as_float(0x7f800000);
    float ld[2] = {INF, INF};   // lane l holds the l-th smallest distance seen so far, per query
    int li[2] = {-1, -1};
    float kth[2] = {INF, INF};
    for (int t0 = 0; t0 < N; t0 += KNN_TILE) {
        const int tn = min(KNN_TILE, N - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += KNN_WAVES * 64) {
            sx[i] = P[(long)(t0 + i) * 3 + 0];
            sy[i] = P[(long)(t0 + i) * 3 + 1];
            sz[i] = P[(long)(t0 + i) * 3 + 2];
        }
        __syncthreads();
        if (!active) continue;
        for (int i0 = 0; i0 < tn; i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < tn;
            const float x = in ? sx[i] : 0.f, y = in ? sy[i] : 0.f, z = in ? sz[i] : 0.f;
            // ((dx*dx + dy*dy) + dz*dz), every operation rounded on its own (-ffp-contract=off), both queries at once
            const f32x2 dx = f32x2{x, x} - qx, dy = f32x2{y, y} - qy, dz = f32x2{z, z} - qz;
            const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float d = in ? dd[q] : INF;
                unsigned long long mask = __ballot(d < kth[q]);
                while (mask) {
                    const int src = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float cd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
                    if (!(cd < kth[q])) continue;          // the k-th best tightened since the ballot
                    const int ci = t0 + i0 + src;
                    const bool gt = ld[q] > cd;            // this lane's entry moves one rank down
                    const float pd = dpp_wave_shr1(ld[q], -1.f);
                    const int pi = __float_as_int(dpp_wave_shr1(__int_as_float(li[q]), 0.f));
                    const bool pgt = pd > cd;
                    ld[q] = gt ? (pgt ? pd : cd) : ld[q];
                    li[q] = gt ? (pgt ? pi : ci) : li[q];
                    kth[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld[q]), k - 1));
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const long m = m0 + q;
        if (!(active && (q == 0 || two))) continue;
        if (lane < k) {
            idx[((long)b * M + m) * k + lane] = li[q];
            if (d2out) d2out[((long)b * M + m) * k + lane] = ld[q];
        }
        if (wout) {   // three-NN weights (k == 3): w_j = (1/(d2_j+1e-8)) / sum
            const float r = __fdiv_rn(1.0f, __fadd_rn(ld[q], 1e-8f));
            const float r0 = __shfl(r, 0), r1 = __shfl(r, 1), r2 = __shfl(r, 2);
            const float s = __fadd_rn(__fadd_rn(r0, r1), r2);
            if (lane < 3) wout[((long)b * M + m) * 3 + lane] = __fdiv_rn(r, s);
        }
    }
}

static int knn_launch(const float *query, const float *ref, int B, int M, int N, int k, int32_t *idx, float *d2,
                      float *w, hipStream_t stream) {
    const long blocks = ((long)M + 2 * KNN_WAVES - 1) / (2 * KNN_WAVES);   // two queries per wave
    P2N_REQUIRE(B <= 65535, "p2n_knn: B > 65535");
    const int ps = p2n_prof_begin(P2N_K_KNN, stream);
    hipLaunchKernelGGL(p2n_knn_kernel, dim3((unsigned)blocks, B, 1), dim3(KNN_WAVES * 64), 0, stream, query, ref, M, N,
                       k, idx, d2, w);
    p2n_prof_end(ps, stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_knn(const float *query, const float *ref, int B, int M, int N, int k, int32_t *idx, float *d2,
                       void *stream) {
    P2N_REQUIRE(B >= 0 && M >= 0 && N >= 1, "p2n_knn: bad shape B=%d M=%d N=%d", B, M, N);
    P2N_REQUIRE(k >= 1 && k <= 64, "p2n_knn: k=%d not in [1,64]", k);
    P2N_REQUIRE(k <= N, "p2n_knn: k=%d exceeds the number of reference points %d", k, N);
    if (B == 0 || M == 0) return P2N_OK;
    P2N_REQUIRE(query && ref && idx, "p2n_knn: null pointer");
    return knn_launch(query, ref, B, M, N, k, idx, d2, nullptr, (hipStream_t)stream);
}

// --------------------------------------------------------------------------- self kNN (query set == cloud)
// The smoothness loss searches every point's neighbours inside its own cloud.  Scanning all N
// candidates per query is N^2 distance tests plus ~k*ln(N/k) list insertions; here the cloud is
// first put in Morton-cell order (one workgroup per cloud, counting sort through an LDS histogram)
// with a bounding box per 64-point tile; a wave seeds its two (adjacent, hence nearby) queries' lists
// with one 64-key bitonic sort of their own tile and then visits only the tiles whose box can still
// hold a point within the current k-th distance of either query.
// The order of the copy affects speed only: results are the k smallest (d2, index) pairs in
// lexicographic order whatever the visiting order, and the box bound is computed with the same
// ((ex*ex + ey*ey) + ez*ez) roundings as the point distance, so by monotonicity of IEEE rounding it
// never exceeds the computed distance of a point inside the box.
#define KS_THREADS 1024
#define KS_MAX_N 16384      // points of one cloud held in the sorting workgroup's registers (16 per thread)
#define KS_PPT (KS_MAX_N / KS_THREADS)
#define KS_AXIS_BITS 5      // Morton cells: 32 x 32 x 32 over the cloud's bounding box
#define KS_BINS (1 << (3 * KS_AXIS_BITS))
#define KS_BIN_WORDS (KS_BINS + KS_BINS / 32)   // one pad word per 32 bins: a thread's 32 consecutive bins are conflict-free
#define KSW 4               // waves per workgroup of the search kernel (no LDS, no barriers)

__device__ __forceinline__ unsigned morton_spread5(unsigned v) {   // bits 0..4 -> bits 0,3,6,9,12
    v = (v | (v << 8)) & 0x0000100Fu;
    v = (v | (v << 4)) & 0x000010C3u;
    v = (v | (v << 2)) & 0x00001249u;
    return v;
}

// Counting sort of one cloud by Morton cell (one workgroup per cloud; the cloud stays in registers
// between the passes): bounding box, cell histogram in LDS, exclusive scan, scatter of (x, y, z, index)
// into the ordered copy, then one bounding box per 64-point tile.  Points of one cell land in
// whatever order the LDS atomics resolve: the order only steers the search, never its result.
__global__ __launch_bounds__(KS_THREADS) void p2n_knn_sort_kernel(const float *__restrict__ xyz, int N, int ntiles,
                                                                  float4 *__restrict__ sorted, float4 *__restrict__ boxes) {
    extern __shared__ int ks_bin[];
    __shared__ float red[6][KS_THREADS / 64];
    __shared__ int wtot[KS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float *P = xyz + (long)b * N * 3;
    const float INF = __int_as_float(0x7f800000);
    float px[KS_PPT], py[KS_PPT], pz[KS_PPT];
    float mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
#pragma unroll
    for (int u = 0; u < KS_PPT; ++u) {
        const int i = tid + u * KS_THREADS;
        px[u] = py[u] = pz[u] = 0.f;
        if (i < N) {
            px[u] = P[(long)i * 3 + 0]; py[u] = P[(long)i * 3 + 1]; pz[u] = P[(long)i * 3 + 2];
            mn[0] = fminf(mn[0], px[u]); mx[0] = fmaxf(mx[0], px[u]);
            mn[1] = fminf(mn[1], py[u]); mx[1] = fmaxf(mx[1], py[u]);
            mn[2] = fminf(mn[2], pz[u]); mx[2] = fmaxf(mx[2], pz[u]);
        }
    }
    for (int i = tid; i < KS_BIN_WORDS; i += KS_THREADS) ks_bin[i] = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
        if (lane == 0) { red[c][wave] = mn[c]; red[3 + c][wave] = mx[c]; }
    }
    __syncthreads();
    float sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = INF, z = -INF;
        for (int w = 0; w < KS_THREADS / 64; ++w) { a = fminf(a, red[c][w]); z = fmaxf(z, red[3 + c][w]); }
        mn[c] = a;
        const float ext = z - a;
        sc[c] = (ext > 0.f && ext < INF) ? (float)((1 << KS_AXIS_BITS) - 1) / ext : 0.f;
    }
    int cell[KS_PPT];
    const float top = (float)((1 << KS_AXIS_BITS) - 1);
#pragma unroll
    for (int u = 0; u < KS_PPT; ++u) {
        // fmaxf/fminf drop a NaN operand: a NaN coordinate lands in cell 0
        const unsigned cx = (unsigned)fminf(fmaxf((px[u] - mn[0]) * sc[0], 0.f), top);
        const unsigned cy = (unsigned)fminf(fmaxf((py[u] - mn[1]) * sc[1], 0.f), top);
        const unsigned cz = (unsigned)fminf(fmaxf((pz[u] - mn[2]) * sc[2], 0.f), top);
        const int code = (int)(morton_spread5(cx) | (morton_spread5(cy) << 1) | (morton_spread5(cz) << 2));
        cell[u] = code + (code >> 5);
        if (tid + u * KS_THREADS < N) atomicAdd(&ks_bin[cell[u]], 1);
    }
    __syncthreads();
    // exclusive scan of the histogram: 32 consecutive bins per thread, wave scan, wave totals
    int mine = 0;
    const int b0 = tid * 33;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) mine += ks_bin[b0 + j];
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        const int c = ks_bin[b0 + j];
        ks_bin[b0 + j] = base;
        base += c;
    }
    __syncthreads();
    float4 *S = sorted + (long)b * ntiles * 64;
#pragma unroll
    for (int u = 0; u < KS_PPT; ++u) {
        const int i = tid + u * KS_THREADS;
        if (i < N) {
            const int pos = atomicAdd(&ks_bin[cell[u]], 1);
            S[pos] = float4{px[u], py[u], pz[u], __int_as_float(i)};
        }
    }
    if (N + tid < ntiles * 64) S[N + tid] = float4{0.f, 0.f, 0.f, __int_as_float(-1)};   // fewer than 64 pad entries
    __syncthreads();   // the workgroup's stores are visible to its own loads past this point
    for (int tile = wave; tile < ntiles; tile += KS_THREADS / 64) {
        const float4 p = S[tile * 64 + lane];
        const bool valid = __float_as_int(p.w) >= 0;
        float lo[3] = {valid ? p.x : INF, valid ? p.y : INF, valid ? p.z : INF};
        float hi[3] = {valid ? p.x : -INF, valid ? p.y : -INF, valid ? p.z : -INF};
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int o = 32; o > 0; o >>= 1) {
                lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
                hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
            }
        if (lane == 0) {
            boxes[((long)b * ntiles + tile) * 2 + 0] = float4{lo[0], lo[1], lo[2], 0.f};
            boxes[((long)b * ntiles + tile) * 2 + 1] = float4{hi[0], hi[1], hi[2], 0.f};
        }
    }
}

// =========================================================================== FPS, bucket-pruned
// Farthest point sampling of a cloud that one workgroup holds, with the work of an iteration cut to the buckets
// the new sample can change.  A point's running minimum changes only if the new sample is closer than that
// minimum, so a 64-point bucket whose bounding box is at least as far from the sample as the bucket's LARGEST
// running minimum keeps every minimum, hence its cached (maximum, lowest index) key.  On Morton-ordered points
// (p2n_knn_sort_kernel: ordered copy + one box per 64-point tile) a sample reaches ~10 of the 128 buckets of an
// 8,192-point cloud (~36 in the first 64 iterations), so the iteration is the reduction chain plus a few buckets
// instead of 32 points per thread (0.80 -> 0.58 us per sample at 8,192 points, 1.07 -> see DESIGN_HISTORY section 4 at 12,288).
// Layout: wave w owns buckets w, w+W, w+2W, ...; slot s of lane j is point j of bucket s*W+w (registers, as in the
// dense kernel); lane s of the wave keeps the box and the key of the wave's slot s.  Per iteration: every lane tests its bucket's box
// (same ((ex*ex + ey*ey) + ez*ez) roundings as a point's distance and IEEE rounding is monotonic, so the bound never
// exceeds the computed distance of a point inside the box; a bucket is skipped only if bound >= its maximum, where no
// minimum can change), the wave walks the set bits of the ballot and, per active bucket, updates its 64 minima,
// takes their maximum by DPP and the lowest original index attaining it, and writes the key into lane s.
// Results are those of the dense kernel bit for bit: the same distances, and ties go to the lowest ORIGINAL index
// (the keys carry it; the order of the copy steers speed only).
struct FpsPrArgs {
    const float *xyz;      // (B,N,3)
    const float4 *sorted;  // [B][ntiles*64] (x,y,z,original index; -1 = pad)
    const float4 *boxes;   // [B][ntiles][2]
    int32_t *idx;          // (B,M)
    int N, M, ntiles, start_idx;
};

template <int W, int S>
__global__ __launch_bounds__(W * 64) void p2n_fps_pruned_kernel(FpsPrArgs A) {
    static_assert(S >= 4 && S <= 32 && S % 4 == 0, "bucket keys live in lanes 0..S-1, slots are walked in groups of four");
    extern __shared__ __attribute__((aligned(16))) float lxyz[];   // the cloud by ORIGINAL index, [point][3]: the winner's coordinates
    __shared__ u64 wkey[3];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    const float *P = A.xyz + (long)b * A.N * 3;
    for (int i = t; i < A.N * 3; i += W * 64) lxyz[i] = P[i];
    if (t < 3) wkey[t] = 0ull;
    const float INF = __int_as_float(0x7f800000);
    const float4 *Sd = A.sorted + (long)b * A.ntiles * 64;
    float px[S], py[S], pz[S], md[S];
    unsigned pv[S];   // 0xFFFFFFFF - original index (0 for pads): maxima everywhere
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int tile = s * W + wave;   // neighbouring buckets go to different waves: a sample's few active buckets are neighbours
        float4 p = {0.f, 0.f, 0.f, __int_as_float(-1)};
        if (tile < A.ntiles) p = Sd[(long)tile * 64 + lane];
        const bool valid = __float_as_int(p.w) >= 0;
        px[s] = p.x; py[s] = p.y; pz[s] = p.z;
        md[s] = valid ? INF : 0.f;   // pads: minimum 0 and the lowest key, they lose every tie
        pv[s] = valid ? 0xFFFFFFFFu - (unsigned)__float_as_int(p.w) : 0u;
    }
    float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};   // the box of bucket `lane` of this wave
    {
        const int tile = lane * W + wave;
        if (lane < S && tile < A.ntiles) {
            const float4 l4 = A.boxes[((long)b * A.ntiles + tile) * 2], h4 = A.boxes[((long)b * A.ntiles + tile) * 2 + 1];
            lo[0] = l4.x; lo[1] = l4.y; lo[2] = l4.z; hi[0] = h4.x; hi[1] = h4.y; hi[2] = h4.z;
        }
    }
    unsigned bkd = 0u, bkv = 0u;   // lane s: bits of bucket s's largest running minimum, and its 0xFFFFFFFF - index
    __syncthreads();
    int cur = A.start_idx;
    float cx = lxyz[cur * 3 + 0], cy = lxyz[cur * 3 + 1], cz = lxyz[cur * 3 + 2];
    int kb = 0;
    for (int it = 0; it < A.M; ++it) {
        if (t == 0) A.idx[(long)b * A.M + it] = cur;
        if (it + 1 == A.M) break;
        // ---- which buckets can the new sample change?
        const float ex = fmaxf(fmaxf(__fsub_rn(lo[0], cx), __fsub_rn(cx, hi[0])), 0.f);
        const float ey = fmaxf(fmaxf(__fsub_rn(lo[1], cy), __fsub_rn(cy, hi[1])), 0.f);
        const float ez = fmaxf(fmaxf(__fsub_rn(lo[2], cz), __fsub_rn(cz, hi[2])), 0.f);
        const float bound = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
        u64 m = __ballot(bound < __uint_as_float(bkd));
        if (it == 0) m = (1ull << S) - 1;   // the first sample sets every minimum (and every bucket key)
        m &= (1ull << S) - 1;
        const unsigned mlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m);
        // wave-uniform walk over the active buckets: groups of four slots, then single slots (scalar tests; a `switch`
        // on the slot number made the compiler compute every slot's distance ahead of the branch and copy the
        // minima around).  The sample's coordinates pass through an opaque move inside the block, so that nothing
        // of a slot's work can be hoisted out of its branch.
#pragma unroll
        for (int g4 = 0; g4 < S / 4; ++g4) {
            if ((mlo >> (4 * g4)) & 0xFu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sl = 4 * g4 + j;
                    if ((mlo >> sl) & 1u) {
                        float qx = cx, qy = cy, qz = cz;
                        asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));
                        const float d = sqdist3(px[sl], py[sl], pz[sl], qx, qy, qz);
                        float mnew;
                        asm("v_min_f32 %0, %1, %2" : "=v"(mnew) : "v"(md[sl]), "v"(d));
                        md[sl] = mnew;
                        const unsigned bits = __float_as_uint(mnew);
                        const unsigned mx = wave_max_u32(bits);
                        const u64 eq = __ballot(bits == mx);
                        unsigned bv;
                        if (__builtin_popcountll(eq) == 1) bv = (unsigned)__builtin_amdgcn_readlane((int)pv[sl], __builtin_ctzll(eq));
                        else bv = wave_max_u32(bits == mx ? pv[sl] : 0u);
                        bkd = lane == sl ? mx : bkd;   // mx, bv are wave-uniform
                        bkv = lane == sl ? bv : bkv;
                    }
                }
            }
        }
        // ---- workgroup argmax over the bucket keys (lanes >= S hold the zero key), as in the dense kernel
        const unsigned rhi = row16_max_u32(bkd);
        const unsigned rlo = row16_max_u32(bkd == rhi ? bkv : 0u);
        if ((lane & 15) == 0) lds_max_u64(&wkey[kb], ((u64)rhi << 32) | rlo);
        __syncthreads();
        const u64 k = wkey[kb];
        const int kclr = kb == 0 ? 2 : kb - 1;
        if (t == 0) wkey[kclr] = 0ull;
        kb = kb == 2 ? 0 : kb + 1;
        cur = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu));
        cx = lxyz[cur * 3 + 0]; cy = lxyz[cur * 3 + 1]; cz = lxyz[cur * 3 + 2];
    }
}

// Waves per workgroup x slots per wave.  Sixteen thin waves: what an iteration costs beyond the reduction chain is the
// SERIAL walk of a wave over its active buckets (~250 dependent cycles each: distance, minimum, DPP maximum, ballot,
// lane read), and neighbouring buckets sit in different waves, so more waves = fewer active buckets per wave
// (8,192 points: 0.93 us per sample with 4 waves, 0.72 with 8, 0.58 with 16; dense kernel 0.80).
static void fps_pruned_shape(int N, int *W, int *S) {
    const int ntiles = (N + 63) / 64;
    const int pw = p2n_tuning().fps_pw;   // P2N_FPS_PW: waves per workgroup (4 / 8), sweeps only
    if (pw == 4 && ntiles <= 128) { *W = 4; *S = ntiles <= 32 ? 8 : ntiles <= 64 ? 16 : ntiles <= 96 ? 24 : 32; return; }
    if (pw == 8) { *W = 8; *S = ntiles <= 64 ? 8 : ntiles <= 128 ? 16 : 24; return; }
    *W = 16;
    *S = ntiles <= 128 ? 8 : 12;
}

// scratch bytes of the pruned path, 0 = this shape takes the dense kernels
static size_t fps_pruned_bytes(int B, int N, int M) {
    const int mode = p2n_tuning().fps_prune;   // P2N_FPS_PRUNE: 1 = never, 2 = whenever the shape allows
    if (mode == 1 || B < 1 || N > FPS_LDS_POINTS || N > KS_MAX_N) return 0;
    if (mode != 2 && (N < 5120 || M < 64)) return 0;   // smaller clouds: the dense kernel's iteration (0.42-0.6 us) is at the chain's floor already
    const size_t ntiles = ((size_t)N + 63) / 64;
    return (((size_t)B * ntiles * 64 * sizeof(float4) + 255) & ~(size_t)255) + (size_t)B * ntiles * 2 * sizeof(float4) + 256;
}

template <int W, int S>
static hipError_t fps_pruned_launch(const FpsPrArgs &a, int B, hipStream_t s) {
    const size_t lds = (size_t)a.N * 12;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(p2n_fps_pruned_kernel<W, S>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((p2n_fps_pruned_kernel<W, S>), dim3(B), dim3(W * 64), lds, s, a);
    return hipGetLastError();
}

static int fps_pruned_run(const float *xyz, int B, int N, int M, int start_idx, int32_t *idx, void *scratch, hipStream_t s) {
    P2N_REQUIRE(((uintptr_t)scratch & 15) == 0, "p2n_fps: workspace must be 16-byte aligned");
    const int ntiles = (N + 63) / 64;
    float4 *sorted = (float4 *)scratch;
    float4 *boxes = (float4 *)((char *)scratch + (((size_t)B * ntiles * 64 * sizeof(float4) + 255) & ~(size_t)255));
    const size_t lds = (size_t)KS_BIN_WORDS * sizeof(int);
    if (lds > 48 * 1024)
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_knn_sort_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ps = p2n_prof_begin(P2N_K_FPS, s);
    hipLaunchKernelGGL(p2n_knn_sort_kernel, dim3(B), dim3(KS_THREADS), lds, s, xyz, N, ntiles, sorted, boxes);
    FpsPrArgs a = {xyz, sorted, boxes, idx, N, M, ntiles, start_idx};
    int W, S;
    fps_pruned_shape(N, &W, &S);
    hipError_t e = hipErrorInvalidValue;
    if (W == 4 && S == 8) e = fps_pruned_launch<4, 8>(a, B, s);
    else if (W == 4 && S == 16) e = fps_pruned_launch<4, 16>(a, B, s);
    else if (W == 4 && S == 24) e = fps_pruned_launch<4, 24>(a, B, s);
    else if (W == 4 && S == 32) e = fps_pruned_launch<4, 32>(a, B, s);
    else if (W == 8 && S == 8) e = fps_pruned_launch<8, 8>(a, B, s);
    else if (W == 8 && S == 16) e = fps_pruned_launch<8, 16>(a, B, s);
    else if (W == 16 && S == 8) e = fps_pruned_launch<16, 8>(a, B, s);
    else if (W == 16 && S == 12) e = fps_pruned_launch<16, 12>(a, B, s);
    else e = fps_pruned_launch<8, 24>(a, B, s);
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(e);
    return P2N_OK;
}

// The wave-resident candidate list holds one 64-bit key per lane, (distance bits << 32) | index:
// distances are non-negative floats, so unsigned key order is (distance, index) order.
#define KS_EMPTY 0x7f800000ffffffffull   // (inf, -1): an unfilled slot

__device__ __forceinline__ unsigned long long ks_key(float d, int i) {
    return ((unsigned long long)(unsigned)__float_as_int(d) << 32) | (unsigned)i;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}

// Bitonic sort of 64 keys, one per lane, ascending by lane.
__device__ __forceinline__ unsigned long long wave_sort_u64(unsigned long long key, int lane) {
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const unsigned long long other = shfl_xor_u64(key, j);
            const bool take_min = ((lane & kk) == 0) == ((lane & j) == 0);
            const bool oth_less = other < key;
            key = (take_min == oth_less) ? other : key;
        }
    }
    return key;
}

// One insertion of candidate key ck into the sorted list: entries behind it move one lane up.
__device__ __forceinline__ void knn_insert_key(unsigned long long &lk, unsigned long long ck) {
    const bool gt = lk > ck;
    const unsigned plo = (unsigned)__float_as_int(dpp_wave_shr1(__int_as_float((int)(unsigned)lk), 0.f));
    const unsigned phi = (unsigned)__float_as_int(dpp_wave_shr1(__int_as_float((int)(unsigned)(lk >> 32)), 0.f));
    const unsigned long long pk = ((unsigned long long)phi << 32) | plo;   // lane 0 sees key 0: never "greater"
    const bool pgt = pk > ck;
    lk = gt ? (pgt ? pk : ck) : lk;
}

__global__ __launch_bounds__(KSW * 64) void p2n_knn_sorted_kernel(const float4 *__restrict__ sorted,
                                                                  const float4 *__restrict__ boxes, int N, int ntiles, int k,
                                                                  int32_t *__restrict__ idx, float *__restrict__ d2out) {
    // the wave number as a scalar: everything derived from it (the queries' position, their tile, the tile masks'
    // split) then lives in SGPRs and the tile walk is scalar code
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const long m0 = 2 * ((long)blockIdx.x * KSW + wave);   // position in the ordered copy
    if (m0 >= N) return;                                    // whole wave; the kernel has no barrier
    const bool two = m0 + 1 < N;
    const float4 *S = sorted + (long)b * ntiles * 64;
    const float4 *BX = boxes + (long)b * ntiles * 2;
    const float4 q0 = S[m0], q1 = S[two ? m0 + 1 : m0];
    const f32x2 qx = {q0.x, q1.x}, qy = {q0.y, q1.y}, qz = {q0.z, q1.z};
    const float INF = __int_as_float(0x7f800000);
    unsigned long long lk[2];
    float kth[2];
    const int own = (int)(m0 >> 6);

    auto distances = [&](const float4 &p) {
        const f32x2 dx = f32x2{p.x, p.x} - qx, dy = f32x2{p.y, p.y} - qy, dz = f32x2{p.z, p.z} - qz;
        return (dx * dx + dy * dy) + dz * dz;
    };
    auto kth_of = [&](unsigned long long key) {
        return __int_as_float(__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), k - 1));
    };
    {   // the queries' own tile seeds both lists: one 64-key sort instead of ~40 serial insertions
        const float4 p = S[(long)own * 64 + lane];
        const int cil = __float_as_int(p.w);
        const f32x2 dd = distances(p);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // an infinite (overflowed) or NaN distance never enters the list, as in the streaming kernel
            const unsigned long long key = (cil >= 0 && dd[q] < INF) ? ks_key(dd[q], cil) : KS_EMPTY;
            lk[q] = wave_sort_u64(key, lane);
            kth[q] = kth_of(lk[q]);
        }
    }
    auto visit = [&](int t) {
        const float4 p = S[(long)t * 64 + lane];
        const int cil = __float_as_int(p.w);
        const f32x2 dd = distances(p);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float d = dd[q];
            unsigned long long mask = __ballot(cil >= 0 && d <= kth[q] && d < INF);
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float cd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
                if (!(cd <= kth[q])) continue;             // the k-th best tightened since the ballot
                knn_insert_key(lk[q], ks_key(cd, __builtin_amdgcn_readlane(cil, src)));
                kth[q] = kth_of(lk[q]);
            }
        }
    };
    auto reaches = [&](float l0, float l1) { return l0 <= kth[0] || l1 <= kth[1]; };

    const int own_blk = own >> 6;
    for (int bi = 0; bi * 64 < ntiles; ++bi) {
        // the 64-tile block that holds the queries' own tile first, then the others in order
        const int blk = bi == 0 ? own_blk : (bi <= own_blk ? bi - 1 : bi);
        const int t = blk * 64 + lane;
        const bool tv = t < ntiles && t != own;
        float lb0 = INF, lb1 = INF;
        if (tv) {
            const float4 lo = BX[2 * t], hi = BX[2 * t + 1];
            f32x2 ex = f32x2{lo.x, lo.x} - qx, ey = f32x2{lo.y, lo.y} - qy, ez = f32x2{lo.z, lo.z} - qz;
            const f32x2 fx = qx - f32x2{hi.x, hi.x}, fy = qy - f32x2{hi.y, hi.y}, fz = qz - f32x2{hi.z, hi.z};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ex[q] = fmaxf(fmaxf(ex[q], fx[q]), 0.f);
                ey[q] = fmaxf(fmaxf(ey[q], fy[q]), 0.f);
                ez[q] = fmaxf(fmaxf(ez[q], fz[q]), 0.f);
            }
            const f32x2 lb = (ex * ex + ey * ey) + ez * ez;
            lb0 = lb[0];
            lb1 = lb[1];
        }
        const unsigned long long mask = __ballot(tv && reaches(lb0, lb1));
        // in the own block, nearest in Morton order first: alternate upwards / downwards from the own tile
        unsigned long long up = mask, dn = 0ull;
        if (blk == own_blk) {
            const int ol = own & 63;
            up = ol == 63 ? 0ull : mask & ~((2ull << ol) - 1ull);
            dn = mask & ((1ull << ol) - 1ull);
        }
        bool flip = false;
        while (up | dn) {
            int src;
            if (up && (!flip || !dn)) { src = __builtin_ctzll(up); up &= up - 1; }
            else { src = 63 - __builtin_clzll(dn); dn &= ~(1ull << src); }
            flip = !flip;
            const float l0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lb0), src));
            const float l1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lb1), src));
            if (reaches(l0, l1)) visit(blk * 64 + src);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q == 1 && !two) continue;
        const int o = __float_as_int(q == 0 ? q0.w : q1.w);
        if (lane < k) {
            idx[((long)b * N + o) * k + lane] = (int)(unsigned)lk[q];
            if (d2out) d2out[((long)b * N + o) * k + lane] = __int_as_float((int)(unsigned)(lk[q] >> 32));
        }
    }
}

extern "C" size_t p2n_knn_self_workspace_bytes(int B, int N) {
    if (B < 1 || N < 1 || N > KS_MAX_N) return 256;   // larger clouds take the streaming kernel: no scratch
    const size_t ntiles = ((size_t)N + 63) / 64;
    return (((size_t)B * ntiles * 64 * sizeof(float4) + 255) & ~(size_t)255) + (size_t)B * ntiles * 2 * sizeof(float4) + 256;
}

extern "C" int p2n_knn_self(const float *xyz, int B, int N, int k, int32_t *idx, float *d2, void *workspace,
                            size_t workspace_bytes, void *stream) {
    P2N_REQUIRE(B >= 0 && N >= 1, "p2n_knn_self: bad shape B=%d N=%d", B, N);
    P2N_REQUIRE(k >= 1 && k <= 64, "p2n_knn_self: k=%d not in [1,64]", k);
    P2N_REQUIRE(k <= N, "p2n_knn_self: k=%d exceeds the number of points %d", k, N);
    if (B == 0) return P2N_OK;
    P2N_REQUIRE(xyz && idx, "p2n_knn_self: null pointer");
    P2N_REQUIRE(B <= 65535, "p2n_knn_self: B > 65535");
    hipStream_t s = (hipStream_t)stream;
    if (N > KS_MAX_N) return knn_launch(xyz, xyz, B, N, N, k, idx, d2, nullptr, s);
    const size_t need = p2n_knn_self_workspace_bytes(B, N);
    if (!workspace || workspace_bytes < need) {
        p2n_set_error("p2n_knn_self: workspace %zu bytes < required %zu", workspace_bytes, need);
        return P2N_EWORKSPACE;
    }
    const int ntiles = (N + 63) / 64;
    float4 *sorted = (float4 *)workspace;
    float4 *boxes = (float4 *)((char *)workspace + (((size_t)B * ntiles * 64 * sizeof(float4) + 255) & ~(size_t)255));
    const size_t lds = (size_t)KS_BIN_WORDS * sizeof(int);
    if (lds > 48 * 1024)
        P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_knn_sort_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ps = p2n_prof_begin(P2N_K_KNN, s);
    hipLaunchKernelGGL(p2n_knn_sort_kernel, dim3(B), dim3(KS_THREADS), lds, s, xyz, N, ntiles, sorted, boxes);
    const unsigned blocks = (unsigned)((N + 2 * KSW - 1) / (2 * KSW));
    hipLaunchKernelGGL(p2n_knn_sorted_kernel, dim3(blocks, B, 1), dim3(KSW * 64), 0, s, sorted, boxes, N, ntiles, k, idx, d2);
    p2n_prof_end(ps, s);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2, int32_t *idx,
                            float *weight, void *stream) {
    P2N_REQUIRE(B >= 0 && n >= 0, "p2n_three_nn: bad shape B=%d n=%d", B, n);
    P2N_REQUIRE(m >= 3, "p2n_three_nn: need at least 3 known points, got %d", m);
    if (B == 0 || n == 0) return P2N_OK;
    P2N_REQUIRE(unknown && known && dist2 && idx, "p2n_three_nn: null pointer");
    return knn_launch(unknown, known, B, n, m, 3, idx, dist2, weight, (hipStream_t)stream);
}

// =========================================================================== gather / interpolate
// HBM-bound streaming kernels: indices (and weights) are read once per output
// column and reused across a chunk of channels; writes are contiguous along the
// fastest output dimension.
#define GG_CH 8

// Indices are the caller's contract (0 <= i < limit).  A wild index would fault the
// GPU, so it is clamped instead of trusted; the result for such an index is undefined.
__device__ __forceinline__ int clamp_idx(int i, int limit) { return i < 0 ? 0 : (i >= limit ? limit - 1 : i); }

__global__ __launch_bounds__(256) void p2n_group_gather_kernel(const float *__restrict__ feat,
                                                              const int32_t *__restrict__ idx, int C, int N, long MS,
                                                              float *__restrict__ out) {
    const int b = blockIdx.z;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= MS) return;
    const int i = clamp_idx(idx[(long)b * MS + j], N);
    const int c0 = blockIdx.y * GG_CH;
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) out[((long)b * C + c0 + c) * MS + j] = feat[((long)b * C + c0 + c) * N + i];
}

// Four output columns per thread: one 16-byte index load, GG_CH x 4 independent gathers in flight,
// GG_CH 16-byte stores (the output is 88 % of the traffic).  Needs MS % 4 == 0 and 16-byte-aligned
// idx / out (the launcher checks; otherwise the scalar kernel above runs).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void p2n_group_gather_v4_kernel(const float *__restrict__ feat,
                                                                 const int32_t *__restrict__ idx, int C, int N, long MS,
                                                                 float *__restrict__ out) {
    const int b = blockIdx.z;
    const long j = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= MS) return;
    const i32x4 iv = *reinterpret_cast<const i32x4 *>(idx + (long)b * MS + j);
    int i[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) i[q] = clamp_idx(iv[q], N);
    const int c0 = blockIdx.y * GG_CH;
    f32x4 v[GG_CH];
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) {
            const float *f = feat + ((long)b * C + c0 + c) * N;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[c][q] = f[i[q]];
        }
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) __builtin_nontemporal_store(v[c], reinterpret_cast<f32x4 *>(out + ((long)b * C + c0 + c) * MS + j));
}

// LDS-staged gather.  A divergent dword gather from global memory costs one L1 tag lookup per
// distinct cache line (about one lane per clock per CU), which caps the two kernels above at
// 35-44 % of the HBM rate.  When whole feature rows fit in LDS (CH * N floats <= 160 KiB) a
// workgroup copies the rows of CH consecutive channels in once (they are contiguous in feat) and
// gathers from LDS banks instead; HBM then sees one linear read of feat and linear 16-byte stores.
// Grid: x = column split, y = channel chunk, z = cloud.  out has CC channels per cloud and this
// call fills channels coff + [0, C) of it (CC = C, coff = 0 for the plain gather; the fused
// set-abstraction grouping writes behind its three coordinate channels).
// Needs N % 4 == 0, MS % 4 == 0 and 16-byte-aligned feat / idx / out.
#define GL_MAX_LDS (160 * 1024)
#define GL_U 4   // steps of the column walk whose indices are prefetched together
template <int CH>
__global__ __launch_bounds__(1024) void p2n_group_gather_lds_kernel(const float *__restrict__ feat,
                                                                   const int32_t *__restrict__ idx, int C, int N,
                                                                   long MS, float *__restrict__ out, int CC, int coff,
                                                                   int quads_per_split) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int b = blockIdx.z, c0 = blockIdx.y * CH;
    const int nch = C - c0 < CH ? C - c0 : CH;
    const long nquad = MS / 4;
    long q = (long)blockIdx.x * quads_per_split;
    const long qend = q + quads_per_split < nquad ? q + quads_per_split : nquad;
    const int32_t *ib = idx + (long)b * MS;
    float *ob = out + ((long)b * CC + coff + c0) * MS;
    // Rows of the output are M*ns floats apart -- a power of two for the usual shapes -- so workgroups
    // that walk their rows in step would all be writing to the same few memory channels.  Every
    // workgroup starts at its own phase of its column range and wraps around.  The walk is software
    // pipelined: the indices of the next GL_U steps are in flight while this block's gathers and
    // stores issue (one step is far shorter than an index load's latency).
    const long q0 = q;
    const int niter = (int)((qend - q0 + blockDim.x - 1) / blockDim.x);
    const int rot = niter > 0 ? (int)((blockIdx.x * 3 + blockIdx.y + (long)blockIdx.z * gridDim.y) * 7 % niter) : 0;
    const i32x4 *iq = reinterpret_cast<const i32x4 *>(ib);
    auto quad_of = [&](int k) -> long {   // k-th step of this workgroup's walk, -1 past the range
        if (k >= niter) return -1;
        int it = k + rot;
        if (it >= niter) it -= niter;
        const long qq = q0 + (long)it * blockDim.x + threadIdx.x;
        return qq < qend ? qq : -1;
    };
    i32x4 ivn[GL_U];
#pragma unroll
    for (int u = 0; u < GL_U; ++u) {
        const long qq = quad_of(u);
        ivn[u] = qq >= 0 ? iq[qq] : i32x4{0, 0, 0, 0};
    }
    // the row copy goes behind the first index loads, so both latencies overlap
    const f32x4 *src = reinterpret_cast<const f32x4 *>(feat + ((long)b * C + c0) * N);
    const int nv = nch * (N / 4);
    for (int v = threadIdx.x; v < nv; v += blockDim.x) reinterpret_cast<f32x4 *>(rows)[v] = src[v];
    __syncthreads();
    for (int k0 = 0; k0 < niter; k0 += GL_U) {
        i32x4 ivc[GL_U];
        long qc[GL_U];
#pragma unroll
        for (int u = 0; u < GL_U; ++u) {
            ivc[u] = ivn[u];
            qc[u] = quad_of(k0 + u);
            const long qq = quad_of(k0 + GL_U + u);
            ivn[u] = qq >= 0 ? iq[qq] : i32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < GL_U; ++u) {
            if (qc[u] < 0) continue;
            int i[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) i[e] = clamp_idx(ivc[u][e], N);
            f32x4 v[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] = rows[(c < nch ? c : 0) * N + i[e]];
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < nch) __builtin_nontemporal_store(v[c], reinterpret_cast<f32x4 *>(ob + (long)c * MS) + qc[u]);
        }
    }
}

// Launch geometry of the LDS-staged kernels, from sweeps on MI355X (tools/sweep_gather.sh):
//  - gather: few channels per workgroup (two rows, one when two do not fit half of the LDS) and
//    1024 threads -- many small workgroups overlap their row loads with the others' stores;
//  - interpolation: eight channels per workgroup, because the 96-byte neighbour table of
//    a column quad is re-read once per channel chunk and is as large as six channels of output;
//  - a column split that brings the grid to ~4 workgroups per CU.
// P2N_GL_CH / P2N_GL_THREADS / P2N_GL_SPLIT override.
struct GlGeom { int ch, threads, split; };
static GlGeom gl_geometry(bool interp, int B, int C, long row_floats, long cols4) {
    GlGeom g = {0, interp ? 512 : 1024, 1};
    const long row_bytes = row_floats * 4;
    const int ch_min = interp ? 4 : 1;   // the interpolation kernel reads channel quads from LDS
    for (int ch = interp ? 8 : 2; ch >= ch_min; ch >>= 1)
        if (ch * row_bytes <= GL_MAX_LDS / 2 || (ch == ch_min && ch * row_bytes <= GL_MAX_LDS)) { g.ch = ch; break; }
    if (const int ch = p2n_tuning().gl_ch) {
        if ((ch == 1 || ch == 2 || ch == 4 || ch == 8 || ch == 16) && ch * row_bytes <= GL_MAX_LDS) g.ch = ch;
    }
    if (g.ch == 0) return g;
    while (g.ch > ch_min && g.ch / 2 >= C) g.ch >>= 1;
    if (const int t = p2n_tuning().gl_threads) {
        if (t == 256 || t == 512 || (t == 1024 && !interp)) g.threads = t;
    }
    const long groups = (long)B * ((C + g.ch - 1) / g.ch);
    long split = (1024 + groups - 1) / groups;
    const long max_split = cols4 / (4L * g.threads);   // keep >= 4 column quads per thread
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (const int sp = p2n_tuning().gl_split) {
        if (sp >= 1 && sp <= 65535) split = sp;
    }
    if (split > 65535) split = 65535;
    g.split = (int)split;
    return g;
}

template <typename K>
static hipError_t gl_allow_lds(K kernel, size_t bytes) {
    return bytes > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
                             : hipSuccess;
}

// Returns 1 when the LDS-staged kernel was launched, 0 when the shape does not qualify, < 0 on error.
static int gather_lds_launch(const float *feat, const int32_t *idx, int B, int C, int N, long MS, float *out, int CC,
                             int coff, hipStream_t stream) {
    if (N % 4 || MS % 4 || ((((uintptr_t)feat | (uintptr_t)idx | (uintptr_t)out)) & 15)) return 0;
    if (p2n_tuning().gl_off) return 0;
    const GlGeom g = gl_geometry(false, B, C, N, MS / 4);
    if (g.ch == 0 || g.ch > 8) return 0;
    const int quads_per_split = (int)((MS / 4 + g.split - 1) / g.split);
    const dim3 grid((unsigned)g.split, (unsigned)((C + g.ch - 1) / g.ch), (unsigned)B);
    const size_t lds = (size_t)g.ch * N * 4;
    hipError_t e = hipSuccess;
#define GL_LAUNCH(CHV)                                                                                              \
    do {                                                                                                            \
        e = gl_allow_lds(p2n_group_gather_lds_kernel<CHV>, lds);                                                    \
        if (e == hipSuccess)                                                                                        \
            hipLaunchKernelGGL(p2n_group_gather_lds_kernel<CHV>, grid, dim3(g.threads), lds, stream, feat, idx, C, N, MS, \
                               out, CC, coff, quads_per_split);                                                     \
    } while (0)
    switch (g.ch) {
        case 8: GL_LAUNCH(8); break;
        case 4: GL_LAUNCH(4); break;
        case 2: GL_LAUNCH(2); break;
        default: GL_LAUNCH(1); break;
    }
#undef GL_LAUNCH
    if (e != hipSuccess) { p2n_set_error("p2n gather: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return -1; }
    return 1;
}

__global__ __launch_bounds__(256) void p2n_group_gather_bwd_kernel(const float *__restrict__ gout,
                                                                  const int32_t *__restrict__ idx, int C, int N,
                                                                  long MS, float *__restrict__ gfeat) {
    const int b = blockIdx.z;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= MS) return;
    const int i = clamp_idx(idx[(long)b * MS + j], N);
    const int c0 = blockIdx.y * GG_CH;
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) atomicAdd(&gfeat[((long)b * C + c0 + c) * N + i], gout[((long)b * C + c0 + c) * MS + j]);
}

// Set-abstraction grouping, fused: out (B, 3+C, M, ns) = [ xyz[idx] - new_xyz ; feat[idx] ]
// (relative coordinates of every group member followed by its features).
__global__ __launch_bounds__(256) void p2n_sa_group_kernel(const float *__restrict__ xyz,
                                                          const float *__restrict__ new_xyz,
                                                          const float *__restrict__ feat,
                                                          const int32_t *__restrict__ idx, int C, int N, int M, int ns,
                                                          float *__restrict__ out) {
    const int b = blockIdx.z;
    const long MS = (long)M * ns;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= MS) return;
    const int i = clamp_idx(idx[(long)b * MS + j], N);
    const int CC = C + 3;
    if (blockIdx.y == 0) {
        const int m = (int)(j / ns);
        const float *q = xyz + ((long)b * N + i) * 3;
        const float *c = new_xyz + ((long)b * M + m) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) out[((long)b * CC + a) * MS + j] = __fsub_rn(q[a], c[a]);
    } else {
        const int c0 = (blockIdx.y - 1) * GG_CH;
#pragma unroll
        for (int c = 0; c < GG_CH; ++c)
            if (c0 + c < C) out[((long)b * CC + 3 + c0 + c) * MS + j] = feat[((long)b * C + c0 + c) * N + i];
    }
}

extern "C" int p2n_sa_group(const float *xyz, const float *new_xyz, const float *feat, const int32_t *idx, int B, int C,
                            int N, int M, int ns, float *out, void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && ns >= 1, "p2n_sa_group: bad shape");
    if (B == 0 || M == 0) return P2N_OK;
    P2N_REQUIRE(xyz && new_xyz && idx && out && (feat || C == 0), "p2n_sa_group: null pointer");
    P2N_REQUIRE(B <= 65535 && (C + GG_CH - 1) / GG_CH + 1 <= 65535, "p2n_sa_group: B or C too large");
    const long MS = (long)M * ns;
    // feature channels through the LDS-staged gather when the shape qualifies; the same kernel then
    // only writes the three relative-coordinate channels (grid y = 1)
    int staged = 0;
    if (C > 0) staged = gather_lds_launch(feat, idx, B, C, N, MS, out, C + 3, 3, (hipStream_t)stream);
    if (staged < 0) return P2N_EHIP;
    const int ychunks = staged ? 1 : (C + GG_CH - 1) / GG_CH + 1;
    hipLaunchKernelGGL(p2n_sa_group_kernel, dim3((unsigned)((MS + 255) / 256), ychunks, B), dim3(256), 0,
                       (hipStream_t)stream, xyz, new_xyz, feat, idx, C, N, M, ns, out);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

extern "C" int p2n_group_gather(const float *feat, const int32_t *idx, int B, int C, int N, int M, int ns, float *out,
                                void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && ns >= 0, "p2n_group_gather: bad shape");
    const long MS = (long)M * ns;
    if (B == 0 || C == 0 || MS == 0) return P2N_OK;
    P2N_REQUIRE(feat && idx && out, "p2n_group_gather: null pointer");
    P2N_REQUIRE(B <= 65535 && (C + GG_CH - 1) / GG_CH <= 65535, "p2n_group_gather: B or C too large");
    P2N_REQUIRE(MS / 4 <= 0x7fffffffL * 256, "p2n_group_gather: M*ns too large");
    const int ps = p2n_prof_begin(P2N_K_GATHER, (hipStream_t)stream);
    const int staged = gather_lds_launch(feat, idx, B, C, N, MS, out, C, 0, (hipStream_t)stream);
    if (staged < 0) return P2N_EHIP;
    if (staged) {
    } else if (MS % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(p2n_group_gather_v4_kernel, dim3((unsigned)((MS / 4 + 255) / 256), (C + GG_CH - 1) / GG_CH, B),
                           dim3(256), 0, (hipStream_t)stream, feat, idx, C, N, MS, out);
    else
        hipLaunchKernelGGL(p2n_group_gather_kernel, dim3((unsigned)((MS + 255) / 256), (C + GG_CH - 1) / GG_CH, B),
                           dim3(256), 0, (hipStream_t)stream, feat, idx, C, N, MS, out);
    p2n_prof_end(ps, (hipStream_t)stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

// Adjoints of the two gathers with the accumulation in LDS.  The first versions (kept below for rows that do not fit)
// sent one fp32 atomic per (channel, source element) to L2: 50-70 M atomics per call at the set-abstraction shapes,
// 0.6 / 1.0 ms.  Here a workgroup owns CH whole rows of the result -- (cloud b, channels c0..c0+CH) -- keeps them in
// LDS, streams the sources past them (index and gradient reads coalesced, ds_add_f32 on LDS banks) and writes the
// rows out once with plain stores: no global atomic, no zero fill.  With few (b, channel) pairs the source range is
// cut into S slices whose workgroups add their rows with global atomics (gfeat zeroed first).  Summation order
// inside a row is still that of the LDS atomics (not reproducible run to run, like the first versions).
#define ADJ_MAX_LDS (128 * 1024)

template <int CH>
__global__ __launch_bounds__(512) void p2n_group_gather_bwd_lds_kernel(const float *__restrict__ gout,
                                                                      const int32_t *__restrict__ idx, int C, int N,
                                                                      long MS, long slice, int S,
                                                                      float *__restrict__ gfeat) {
    extern __shared__ float acc[];   // [CH][N]
    const int b = blockIdx.z, c0 = blockIdx.y * CH;
    for (int i = threadIdx.x; i < CH * N; i += 512) acc[i] = 0.f;
    __syncthreads();
    const long j0 = (long)blockIdx.x * slice;
    long j1 = j0 + slice;
    if (j1 > MS) j1 = MS;
    const int32_t *ib = idx + (long)b * MS;
    for (long j = j0 + threadIdx.x; j < j1; j += 512) {
        const int i = clamp_idx(ib[j], N);
        float g[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) g[c] = c0 + c < C ? gout[((long)b * C + c0 + c) * MS + j] : 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) atomicAdd(&acc[c * N + i], g[c]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CH * N; i += 512) {
        const int c = i / N, k = i - c * N;
        if (c0 + c >= C) break;
        float *dst = &gfeat[((long)b * C + c0 + c) * N + k];
        if (S == 1) *dst = acc[i];
        else if (acc[i] != 0.f) atomicAdd(dst, acc[i]);
    }
}

// channels per workgroup for rows of `len` floats (0: a row does not fit the LDS budget), slices for `pairs` (b, block) pairs
static int adj_channels(long len) {
    for (int ch = 8; ch >= 1; ch >>= 1)
        if ((long)ch * len * (long)sizeof(float) <= ADJ_MAX_LDS) return ch;
    return 0;
}

static int adj_slices(long pairs, long sources) {
    int S = 1;
    while (pairs * S < 256 && sources / (S * 2) >= 4096 && S < 64) S *= 2;
    return S;
}

extern "C" int p2n_group_gather_bwd(const float *gout, const int32_t *idx, int B, int C, int N, int M, int ns,
                                    float *gfeat, void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && ns >= 0, "p2n_group_gather_bwd: bad shape");
    const long MS = (long)M * ns;
    if (B == 0 || C == 0) return P2N_OK;
    P2N_REQUIRE(gfeat, "p2n_group_gather_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int ch = adj_channels(N);
    const long nblk = ch ? (C + ch - 1) / ch : 0;
    const int S = ch ? adj_slices((long)B * nblk, MS) : 0;
    if (!ch || S > 1 || MS == 0) P2N_CHECK_HIP(hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * N, s));
    if (MS == 0) return P2N_OK;
    P2N_REQUIRE(gout && idx, "p2n_group_gather_bwd: null pointer");
    P2N_REQUIRE(B <= 65535 && (C + GG_CH - 1) / GG_CH <= 65535 && C <= 65535, "p2n_group_gather_bwd: B or C too large");
    if (ch) {
        const long slice = ((MS + S - 1) / S + 511) / 512 * 512;
        const size_t lds = sizeof(float) * (size_t)ch * N;
        const dim3 grid((unsigned)((MS + slice - 1) / slice), (unsigned)nblk, B);
#define ADJ_GO(CHV)                                                                                                     \
    do {                                                                                                                \
        if (lds > 48 * 1024)                                                                                            \
            P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_group_gather_bwd_lds_kernel<CHV>),    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
        hipLaunchKernelGGL(p2n_group_gather_bwd_lds_kernel<CHV>, grid, dim3(512), lds, s, gout, idx, C, N, MS, slice,    \
                           (int)grid.x, gfeat);                                                                         \
    } while (0)
        if (ch == 8) ADJ_GO(8); else if (ch == 4) ADJ_GO(4); else if (ch == 2) ADJ_GO(2); else ADJ_GO(1);
#undef ADJ_GO
    } else {
        hipLaunchKernelGGL(p2n_group_gather_bwd_kernel, dim3((unsigned)((MS + 255) / 256), (C + GG_CH - 1) / GG_CH, B),
                           dim3(256), 0, s, gout, idx, C, N, MS, gfeat);
    }
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

__global__ __launch_bounds__(256) void p2n_three_interp_kernel(const float *__restrict__ feat,
                                                              const int32_t *__restrict__ idx,
                                                              const float *__restrict__ w, int C, int m, int n,
                                                              float *__restrict__ out) {
    const int b = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t *id = idx + ((long)b * n + i) * 3;
    const float *ww = w + ((long)b * n + i) * 3;
    const int i0 = clamp_idx(id[0], m), i1 = clamp_idx(id[1], m), i2 = clamp_idx(id[2], m);
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    const int c0 = blockIdx.y * GG_CH;
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) {
            const float *f = feat + ((long)b * C + c0 + c) * m;
            out[((long)b * C + c0 + c) * n + i] =
                __fadd_rn(__fadd_rn(__fmul_rn(w0, f[i0]), __fmul_rn(w1, f[i1])), __fmul_rn(w2, f[i2]));
        }
}

// Four unknown points per thread (n % 4 == 0, 16-byte-aligned idx / weight / out): 3 + 3 16-byte
// loads for the neighbour table, GG_CH x 12 gathers, GG_CH 16-byte stores.  Same arithmetic order.
__global__ __launch_bounds__(256) void p2n_three_interp_v4_kernel(const float *__restrict__ feat,
                                                                 const int32_t *__restrict__ idx,
                                                                 const float *__restrict__ w, int C, int m, int n,
                                                                 float *__restrict__ out) {
    const int b = blockIdx.z;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    int id[12];
    float ww[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const i32x4 iv = *reinterpret_cast<const i32x4 *>(idx + ((long)b * n + i) * 3 + 4 * q);
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + ((long)b * n + i) * 3 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { id[4 * q + e] = clamp_idx(iv[e], m); ww[4 * q + e] = wv[e]; }
    }
    const int c0 = blockIdx.y * GG_CH;
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) {
            const float *f = feat + ((long)b * C + c0 + c) * m;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = __fadd_rn(__fadd_rn(__fmul_rn(ww[3 * e], f[id[3 * e]]), __fmul_rn(ww[3 * e + 1], f[id[3 * e + 1]])),
                                 __fmul_rn(ww[3 * e + 2], f[id[3 * e + 2]]));
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(out + ((long)b * C + c0 + c) * n + i));
        }
}

// LDS-staged interpolation (same reasoning as p2n_group_gather_lds_kernel).  The rows of CH channels
// of the known level are transposed into LDS as [point][CH], so that one ds_read_b128 fetches four
// channels of a neighbour (a quarter of the LDS instructions of a dword gather -- the random bank
// conflicts of those were what bounded the first version).  Every thread owns four unknown points and
// keeps their 12 indices / weights in registers across the channels; the next step's table is
// prefetched.  Same arithmetic order as the oracle: (w0*f0 + w1*f1) + w2*f2 without fma.
// Needs m % 4 == 0, n % 4 == 0 and 16-byte-aligned feat / idx / weight / out.  CH = 4 or 8.
template <int CH>
__global__ __launch_bounds__(512) void p2n_three_interp_lds_kernel(const float *__restrict__ feat,
                                                                  const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ w, int C, int m, int n,
                                                                  float *__restrict__ out, int quads_per_split) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [m][CH]
    const int b = blockIdx.z, c0 = blockIdx.y * CH;
    const int nch = C - c0 < CH ? C - c0 : CH;
    const int nquad = n / 4;
    const int q0 = blockIdx.x * quads_per_split;
    const int qend = q0 + quads_per_split < nquad ? q0 + quads_per_split : nquad;
    float *ob = out + ((long)b * C + c0) * n;
    const int niter = (qend - q0 + (int)blockDim.x - 1) / (int)blockDim.x;
    const int rot = niter > 0 ? (int)((blockIdx.x * 3 + blockIdx.y + (long)blockIdx.z * gridDim.y) * 7 % niter) : 0;
    auto quad_of = [&](int k) -> int {
        if (k >= niter) return -1;
        int it = k + rot;
        if (it >= niter) it -= niter;
        const int qq = q0 + it * (int)blockDim.x + (int)threadIdx.x;
        return qq < qend ? qq : -1;
    };
    i32x4 ivn[3];
    f32x4 wvn[3];
    auto fetch = [&](int qq) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            ivn[t] = qq >= 0 ? reinterpret_cast<const i32x4 *>(idx + ((long)b * n + 4L * qq) * 3)[t] : i32x4{0, 0, 0, 0};
            wvn[t] = qq >= 0 ? reinterpret_cast<const f32x4 *>(w + ((long)b * n + 4L * qq) * 3)[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(quad_of(0));
    // the transposing row copy goes behind the first table loads, so both latencies overlap
    {   // fill: lane -> (channel c, point quad p4), c fastest; a lane reads 16 bytes of row c
        const float *base = feat + ((long)b * C + c0) * m;
        const int nv = CH * (m / 4);
        for (int v = threadIdx.x; v < nv; v += blockDim.x) {
            const int c = v % CH, p4 = v / CH;
            const f32x4 val = c < nch ? reinterpret_cast<const f32x4 *>(base + (long)c * m)[p4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) rows[(4 * p4 + e) * CH + c] = val[e];
        }
    }
    __syncthreads();
    for (int k = 0; k < niter; ++k) {
        const int q = quad_of(k);
        int id[12];
        float ww[12];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { id[4 * t + e] = clamp_idx(ivn[t][e], m); ww[4 * t + e] = wvn[t][e]; }
        fetch(quad_of(k + 1));
        if (q < 0) continue;
        f32x4 o[CH];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 *r0 = reinterpret_cast<const f32x4 *>(rows + id[3 * e] * CH);
            const f32x4 *r1 = reinterpret_cast<const f32x4 *>(rows + id[3 * e + 1] * CH);
            const f32x4 *r2 = reinterpret_cast<const f32x4 *>(rows + id[3 * e + 2] * CH);
#pragma unroll
            for (int h = 0; h < CH / 4; ++h) {
                const f32x4 f0 = r0[h], f1 = r1[h], f2 = r2[h];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[4 * h + j][e] = __fadd_rn(__fadd_rn(__fmul_rn(ww[3 * e], f0[j]), __fmul_rn(ww[3 * e + 1], f1[j])),
                                                __fmul_rn(ww[3 * e + 2], f2[j]));
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (c < nch) __builtin_nontemporal_store(o[c], reinterpret_cast<f32x4 *>(ob + (long)c * n) + q);
    }
}

static int interp_lds_launch(const float *feat, const int32_t *idx, const float *w, int B, int C, int m, int n,
                             float *out, hipStream_t stream) {
    if (m % 4 || n % 4 || ((((uintptr_t)feat | (uintptr_t)idx | (uintptr_t)w | (uintptr_t)out)) & 15)) return 0;
    if (p2n_tuning().gl_off) return 0;
    const GlGeom g = gl_geometry(true, B, C, m, n / 4);
    if (g.ch != 4 && g.ch != 8) return 0;
    const int quads_per_split = (n / 4 + g.split - 1) / g.split;
    const dim3 grid((unsigned)g.split, (unsigned)((C + g.ch - 1) / g.ch), (unsigned)B);
    const size_t lds = (size_t)g.ch * m * 4;
    hipError_t e = hipSuccess;
#define GL_LAUNCH(CHV)                                                                                               \
    do {                                                                                                             \
        e = gl_allow_lds(p2n_three_interp_lds_kernel<CHV>, lds);                                                     \
        if (e == hipSuccess)                                                                                         \
            hipLaunchKernelGGL(p2n_three_interp_lds_kernel<CHV>, grid, dim3(g.threads), lds, stream, feat, idx, w, C, m, n, \
                               out, quads_per_split);                                                                \
    } while (0)
    if (g.ch == 8) GL_LAUNCH(8);
    else GL_LAUNCH(4);
#undef GL_LAUNCH
    if (e != hipSuccess) { p2n_set_error("p2n interpolate: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return -1; }
    return 1;
}

__global__ __launch_bounds__(256) void p2n_three_interp_bwd_kernel(const float *__restrict__ gout,
                                                                  const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ w, int C, int m, int n,
                                                                  float *__restrict__ gfeat) {
    const int b = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t *id = idx + ((long)b * n + i) * 3;
    const float *ww = w + ((long)b * n + i) * 3;
    const int i0 = clamp_idx(id[0], m), i1 = clamp_idx(id[1], m), i2 = clamp_idx(id[2], m);
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    const int c0 = blockIdx.y * GG_CH;
#pragma unroll
    for (int c = 0; c < GG_CH; ++c)
        if (c0 + c < C) {
            const float g = gout[((long)b * C + c0 + c) * n + i];
            float *f = gfeat + ((long)b * C + c0 + c) * m;
            atomicAdd(&f[i0], g * w0);
            atomicAdd(&f[i1], g * w1);
            atomicAdd(&f[i2], g * w2);
        }
}

extern "C" int p2n_three_interpolate(const float *feat, const int32_t *idx, const float *weight, int B, int C, int m,
                                     int n, float *out, void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && m >= 1 && n >= 0, "p2n_three_interpolate: bad shape");
    if (B == 0 || C == 0 || n == 0) return P2N_OK;
    P2N_REQUIRE(feat && idx && weight && out, "p2n_three_interpolate: null pointer");
    P2N_REQUIRE(B <= 65535 && (C + GG_CH - 1) / GG_CH <= 65535, "p2n_three_interpolate: B or C too large");
    const int ps = p2n_prof_begin(P2N_K_INTERP, (hipStream_t)stream);
    const int staged = interp_lds_launch(feat, idx, weight, B, C, m, n, out, (hipStream_t)stream);
    if (staged < 0) return P2N_EHIP;
    if (staged) {
    } else if (n % 4 == 0 && (((uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(p2n_three_interp_v4_kernel, dim3((n / 4 + 255) / 256, (C + GG_CH - 1) / GG_CH, B), dim3(256), 0,
                           (hipStream_t)stream, feat, idx, weight, C, m, n, out);
    else
        hipLaunchKernelGGL(p2n_three_interp_kernel, dim3((n + 255) / 256, (C + GG_CH - 1) / GG_CH, B), dim3(256), 0,
                           (hipStream_t)stream, feat, idx, weight, C, m, n, out);
    p2n_prof_end(ps, (hipStream_t)stream);
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}

template <int CH>
__global__ __launch_bounds__(512) void p2n_three_interp_bwd_lds_kernel(const float *__restrict__ gout,
                                                                      const int32_t *__restrict__ idx,
                                                                      const float *__restrict__ w, int C, int m, int n,
                                                                      int slice, int S, float *__restrict__ gfeat) {
    extern __shared__ float acc[];   // [CH][m]
    const int b = blockIdx.z, c0 = blockIdx.y * CH;
    for (int i = threadIdx.x; i < CH * m; i += 512) acc[i] = 0.f;
    __syncthreads();
    const int p0 = blockIdx.x * slice;
    const int p1 = p0 + slice < n ? p0 + slice : n;
    for (int i = p0 + threadIdx.x; i < p1; i += 512) {
        const int32_t *id = idx + ((long)b * n + i) * 3;
        const float *ww = w + ((long)b * n + i) * 3;
        const int i0 = clamp_idx(id[0], m), i1 = clamp_idx(id[1], m), i2 = clamp_idx(id[2], m);
        const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
        float g[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) g[c] = c0 + c < C ? gout[((long)b * C + c0 + c) * n + i] : 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            atomicAdd(&acc[c * m + i0], g[c] * w0);
            atomicAdd(&acc[c * m + i1], g[c] * w1);
            atomicAdd(&acc[c * m + i2], g[c] * w2);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CH * m; i += 512) {
        const int c = i / m, k = i - c * m;
        if (c0 + c >= C) break;
        float *dst = &gfeat[((long)b * C + c0 + c) * m + k];
        if (S == 1) *dst = acc[i];
        else if (acc[i] != 0.f) atomicAdd(dst, acc[i]);
    }
}

extern "C" int p2n_three_interpolate_bwd(const float *gout, const int32_t *idx, const float *weight, int B, int C,
                                         int m, int n, float *gfeat, void *stream) {
    P2N_REQUIRE(B >= 0 && C >= 0 && m >= 1 && n >= 0, "p2n_three_interpolate_bwd: bad shape");
    if (B == 0 || C == 0) return P2N_OK;
    P2N_REQUIRE(gfeat, "p2n_three_interpolate_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int ch = adj_channels(m);
    const long nblk = ch ? (C + ch - 1) / ch : 0;
    const int S = ch ? adj_slices((long)B * nblk, n) : 0;
    if (!ch || S > 1 || n == 0) P2N_CHECK_HIP(hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * m, s));
    if (n == 0) return P2N_OK;
    P2N_REQUIRE(gout && idx && weight, "p2n_three_interpolate_bwd: null pointer");
    P2N_REQUIRE(B <= 65535 && (C + GG_CH - 1) / GG_CH <= 65535 && C <= 65535, "p2n_three_interpolate_bwd: B or C too large");
    if (ch) {
        const int slice = ((n + S - 1) / S + 511) / 512 * 512;
        const size_t lds = sizeof(float) * (size_t)ch * m;
        const dim3 grid((unsigned)((n + slice - 1) / slice), (unsigned)nblk, B);
#define ADJ_GO(CHV)                                                                                                     \
    do {                                                                                                                \
        if (lds > 48 * 1024)                                                                                            \
            P2N_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&p2n_three_interp_bwd_lds_kernel<CHV>),    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
        hipLaunchKernelGGL(p2n_three_interp_bwd_lds_kernel<CHV>, grid, dim3(512), lds, s, gout, idx, weight, C, m, n,    \
                           slice, (int)grid.x, gfeat);                                                                  \
    } while (0)
        if (ch == 8) ADJ_GO(8); else if (ch == 4) ADJ_GO(4); else if (ch == 2) ADJ_GO(2); else ADJ_GO(1);
#undef ADJ_GO
    } else {
        hipLaunchKernelGGL(p2n_three_interp_bwd_kernel, dim3((n + 255) / 256, (C + GG_CH - 1) / GG_CH, B), dim3(256), 0,
                           s, gout, idx, weight, C, m, n, gfeat);
    }
    P2N_CHECK_HIP(hipGetLastError());
    return P2N_OK;
}
