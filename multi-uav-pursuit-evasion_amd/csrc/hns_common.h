// hns_common.h — what the step, reset and auxiliary kernels of libhns.so share: launch-parameter blocks, workgroup geometry, LDS
// carve-ups, the wave-private staged stores, the one-pass cylinder sweep and the reset-time observation.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hns_device.h"

namespace hns {

constexpr int kEPB = 64;   // envs per workgroup = lanes of the env wave
constexpr int kMaxK = 4;   // top-k insertion network width of the step kernels (obs_max_cylinder <= 4: the reference's default is 3)
constexpr int kWideK = 16;  // the same network for every k the cylinder count allows (the generic step instantiation + the reset kernel): no staging of the k-nearest rows
// per-agent scalars handed to the env wave: 8 live values at any time (odd stride: conflict-free), 11 with a second evader
__host__ __device__ constexpr int red_stride(int NT) { return NT == 2 ? 11 : 9; }
enum { R_AERR = 0, R_TD, R_DIST, R_SPEED, R_CC, R_CD, R_CW, R_COLL,
       // phase 3 reuses the two slots the env wave drained right behind the first barrier (action error, throttle difference)
       R_SMOOTH = R_AERR, R_FLAGS = R_TD,
       // the pursuer's push on the evader is consumed before phase 3 writes the reward terms: same slots
       R_FX = R_DIST, R_FY = R_SPEED, R_FZ = R_CC,
       // so is the thrust vector handed to the downwash partners (3 consecutive slots)
       R_TWX = R_CD,
       // two-evader extension: the push on the second evader (slots 8..10)
       R_F1X = 8 };
enum { F_CAP = 1, F_BLOCKED = 2, F_DET = 4, F_DET1 = 8 };
constexpr int kGridStride = 516;   // bytes of reset scratch per env: 2 x 256 + 4 (an odd dword stride: lanes = envs hit different LDS banks)
constexpr int kSmallWgPerCu = 2;   // the small-batch mapping (hns_step_small_kernel.h) serves grids of up to this many workgroups per CU (one with four and more pursuers; hns_inst.hip)
constexpr int kMaxT = 2;   // evaders per env (1 = the reference; 2 = BASELINE config 5's extension)

template <int A>
struct Geo {
    static constexpr int NA = kEPB * A;         // agent threads
    static constexpr int T = kEPB * (A + 1);    // + the env wave
};

// Reset: passed by value (~900 B of kernel arguments).  A launch with <= 64 B of arguments is 0.5-0.6 us faster
// (tools/microbench/launch_gap.hip), but a block in device memory read through the scalar cache in front of a wave's first
// global load costs more than that (+1.8 us measured) — so the step kernel takes StepArgs: the pointers behind its first loads
// by value, everything else through `rest`, whose scalar loads travel beside those loads.
struct Params {
    hns_cfg cfg;
    hns_buffers buf;
    const float *action;        // step
    const uint8_t *reset_mask;  // reset (nullable)
    uint32_t seed_lo, seed_hi, epoch;
    unsigned long long *prof;   // optional per-wave phase timestamps (diagnostics), else null
    uint32_t cyl_magic;         // ceil(2^32 / (3*C)): index / (3*C) as a multiply-high
    float *tasks;               // reset: optional [E, 3A+3NT+3C] task vectors (envgen: pursuers | evader(s) | cylinder slots; rows < task_first are written), else null
    int32_t task_first;         // envs >= task_first take their placement from `tasks`
    uint32_t prio_boost;        // step, tile mapping: pursuer waves run at priority 1 until their integration is done (hns_step_kernel.h; chosen by hns_inst.hip)
};

struct StepArgs {               // 64 B
    const float *action;
    float *prev_action, *drone_state, *pid_integ, *pid_last_rate, *throttle;
    const void *aux;            // the eighth slot, by evader count.  Two evaders: `cylinders` (16-byte aligned; the low four bits carry
                                // num_cylinders - 1 — the 64-byte block has no room for another word, and that kernel's first loads need the
                                // count before the parameter block is warm).  One evader: `reset_pid` ([E] u8, nullable; transforms.py:449-454),
                                // so that the byte travels with the first loads; with two evaders it is read through `rest`.
    const Params *rest;         // device copy of the launch's Params (action = null), kept by the env handle
};
static_assert(sizeof(StepArgs) == 64, "the argument block of the step kernel is sized for the fast launch path");
// The step kernels take the eight words as SEPARATE parameters: pointer parameters can be preloaded into SGPRs by the dispatcher
// (-mllvm -amdgpu-kernarg-preload-count=16; __graft_entry__.py sets it for the small-batch mapping's translation units), so a wave issues its
// first loads without a round trip to the kernel-argument segment; an aggregate passed by value is not eligible.  The body packs them back into a StepArgs.
// (14 dwords are preloaded: the parameter block — the env wave's first need — leads, `throttle`, the last of a pursuer wave's first loads, trails)
#define HNS_STEP_PARAMS                                                                                                                    \
    const Params *ka_rest, const float *ka_action, float *ka_prev_action, float *ka_drone_state, const void *ka_aux, float *ka_pid_integ, \
        float *ka_pid_last_rate, float *ka_throttle
#define HNS_STEP_ARGS_PACK const StepArgs ka{ka_action, ka_prev_action, ka_drone_state, ka_pid_integ, ka_pid_last_rate, ka_throttle, ka_aux, ka_rest}

// The parameter block is read through the scalar cache, which every launch starts with cold: a wave that meets a field of a line nobody
// has touched yet waits for an L2 round trip, and a phase that needs six lines one after the other pays six of them (the controller phase of a
// pursuer wave does; at small batches, where nothing else runs on the SIMD meanwhile, that is a third of the phase).  A wave with nothing
// better to do at its start (env wave, helpers) touches one word of every 64-byte line at once: one round trip, then everybody hits.
HNS_DEV void warm_params(const Params *rest) {
    typedef const uint32_t __attribute__((address_space(4))) WordC;
    WordC *w = (WordC *)rest;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < (int)((sizeof(Params) + 63) / 64); ++i) acc |= w[i * 16];
    asm volatile("" : : "s"(acc));
}

constexpr int kProfSlots = 16;
// lane 0 of every wave stamps s_memtime at a phase boundary (only when a buffer is attached)
HNS_DEV void prof_mark(unsigned long long *prof, int slot) {
    if (prof && (threadIdx.x & 63) == 0) {
        int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        // slots 14/15 use the chip-wide constant 100 MHz clock (comparable across XCDs)
        prof[(size_t)wave * kProfSlots + slot] = (slot >= 14) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();
#ifdef HNS_PROF_HWID       // (measurement arm: where the wave runs — HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13] — into the free slot 13; tools/wave_placement.py)
        if (slot == 0) prof[(size_t)wave * kProfSlots + 13] = __builtin_amdgcn_s_getreg(63492);
#endif
    }
}

// LDS carve-up (float offsets, every region 16-byte aligned)
struct Lds {
    int ds, cyl, cyl_stride, tp, red, ocyl, total;
};
__host__ __device__ inline int slab_floats(int A, int K, int NT);
__host__ __device__ inline int r4(int n) { return (n + 3) & ~3; }
// rows per staging pass: the whole wave (64) or, for wide workgroups whose slabs would otherwise push the workgroup past half
// of the CU's LDS, half a wave at a time (two passes per output, half the slab)
__host__ __device__ constexpr int slab_rows(int A) { return A > 4 ? 32 : 64; }
__host__ __device__ inline int slab_floats(int A, int K, int NT) {
    const int rows = slab_rows(A);
    int m = rows * (NT == 2 ? 24 : HNS_SELF_DIM);
    if (K > kMaxK) K = 0;                       // wide selections are stored by their threads, not staged
    if (rows * K * 5 > m) m = rows * K * 5;
    if (rows * (A - 1) * 3 > m) m = rows * (A - 1) * 3;
    return r4(m);
}
__host__ __device__ inline Lds lds_layout(int A, int C, int K, int NT = 1) {
    Lds L;
    int o = 0;
    L.ds = o;    o += r4(kEPB * A * 13);
    L.cyl_stride = (3 * C) | 1;                 // odd per-env stride: env-wave reads are conflict-free
    L.cyl = o;   o += r4(kEPB * L.cyl_stride);
    L.tp = o;    o += r4(kEPB * 3 * NT);
    L.red = o;   o += r4(kEPB * A * red_stride(NT));
    // obs_cylinders staging [64*A][K*5] (reset kernel, ragged tiles) / one wave-private slab per agent wave (step kernel):
    // the slab holds the widest of a wave's three output slices (64 rows of state_self / k-nearest rows / state_others)
    const int rows = K > kMaxK ? 0 : kEPB * A * K * 5, slabs = A * slab_floats(A, K, NT);
    L.ocyl = o;  o += r4(rows > slabs ? rows : slabs);
    L.total = o;
    return L;
}

// ---- workgroup-cooperative contiguous copies (16 B per lane where alignment allows) ----------
template <int T>
HNS_DEV void coop_g2s(float *__restrict__ dst, const float *__restrict__ src, int n) {
    const int n4 = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) ? (n >> 2) : 0;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int i = threadIdx.x; i < n4; i += T) d4[i] = s4[i];
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += T) dst[i] = src[i];
}
template <int T>
HNS_DEV void coop_s2g(float *__restrict__ dst, const float *__restrict__ src, int n) {
    const int n4 = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? (n >> 2) : 0;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int i = threadIdx.x; i < n4; i += T) d4[i] = s4[i];
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += T) dst[i] = src[i];
}
// full-workgroup fast path: N floats (N % 4 == 0, both sides 16-byte aligned) known at compile time ->
// fixed trip count, immediate offsets, no tail code
template <int T, int N>
HNS_DEV void coop_copy_full(float *__restrict__ dst, const float *__restrict__ src) {
    static_assert(N % 4 == 0, "float4 granularity");
    constexpr int N4 = N / 4;
    const float4 *s4 = reinterpret_cast<const float4 *>(src) + threadIdx.x;
    float4 *d4 = reinterpret_cast<float4 *>(dst) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < (N4 + T - 1) / T; ++k)
        if (k * T + (int)threadIdx.x < N4) d4[k * T] = s4[k * T];
}
// same, but only the elements of envs whose mask byte is set (per_env floats per env)
template <int T>
HNS_DEV void coop_s2g_masked(float *__restrict__ dst, const float *__restrict__ src, int n, int per_env,
                             const uint8_t *__restrict__ smask) {
    for (int i = threadIdx.x; i < n; i += T)
        if (smask[i / per_env]) dst[i] = src[i];
}
// cylinders [nenv, 3C] (contiguous) <-> LDS rows of odd stride
template <int T, bool TO_LDS>
HNS_DEV void coop_cyl(float *__restrict__ lds, float *__restrict__ g, int nenv, int c3, int stride, unsigned magic,
                      const uint8_t *__restrict__ smask) {
    for (int i = threadIdx.x; i < nenv * c3; i += T) {
        int le = (int)__umulhi((unsigned)i, magic), j = i - le * c3;     // i / c3 by multiply-high
        if (TO_LDS) lds[le * stride + j] = g[i];
        else if (smask[le]) g[i] = lds[le * stride + j];
    }
}
// failure detection (include/hns.h: hns_buffers.nonfinite): left-to-right sum of the 13 state values; (s - s) != 0 <=> not finite
HNS_DEV bool rigid_not_finite(const Rigid &s) {
    float a = s.pos.x;
    a = a + s.pos.y; a = a + s.pos.z; a = a + s.q.w; a = a + s.q.x; a = a + s.q.y; a = a + s.q.z;
    a = a + s.lin.x; a = a + s.lin.y; a = a + s.lin.z; a = a + s.ang.x; a = a + s.ang.y; a = a + s.ang.z;
    return (a - a) != 0.0f;
}
HNS_DEV void flag_nonfinite(uint32_t *word, bool bad, uint32_t bit) {
    if (word && bad) atomicOr(word, bit);          // rare: no traffic when everything is finite
}
HNS_DEV void load_rigid(const float *r, Rigid &s) {
    s.pos = {r[0], r[1], r[2]};
    s.q = {r[3], r[4], r[5], r[6]};
    s.lin = {r[7], r[8], r[9]};
    s.ang = {r[10], r[11], r[12]};
}
HNS_DEV void store_rigid(float *r, const Rigid &s) {
    r[0] = s.pos.x; r[1] = s.pos.y; r[2] = s.pos.z;
    r[3] = s.q.w; r[4] = s.q.x; r[5] = s.q.y; r[6] = s.q.z;
    r[7] = s.lin.x; r[8] = s.lin.y; r[9] = s.lin.z;
    r[10] = s.ang.x; r[11] = s.ang.y; r[12] = s.ang.z;
}

// ---- write-through stores (sc1): the bytes leave the XCD's L2 while the launch still computes instead of waiting for the
// end-of-kernel write-back (MI355X_MICROARCH.md, stores of each flavour; A/B in DESIGN.md §8.1: plain 28.3, nt 27.9, sc1 27.2 us) ----
HNS_DEV void st_f4(float4 *p, const float4 &v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}
HNS_DEV void st_f1(float *p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
}

// ---- output rows of one agent wave: registers -> wave-private LDS slab -> ONE contiguous slice of global memory -----
// Every [E,A,...] output keeps the reference's layout, so the 64 rows a wave produces are one contiguous slice of
// 64*NF floats.  A thread storing its own row issues 16- (or 8-/4-) byte pieces at a stride of NF floats: every lane of
// the store instruction lands on a different cache line (measured: state_self 1.6 us, state_others 1.0 us of the
// 28 us step).  Instead the wave parks its rows in its slab and stores the slice back linearly, 16 B per lane, whole
// lines per instruction.  The slab is private to the wave: LDS operations of one wave execute in order, no barrier.
// `valid_rows` (ragged last tile): rows of this wave that exist; the buffer descriptor ends behind them and the hardware drops what lies
// beyond (raw buffers range-check multi-dword stores per component), so the same 16-byte pieces serve a slice that ends anywhere.
template <int NF, int ROWS = 64>
HNS_DEV void wave_store_rows(float *__restrict__ slab, float *__restrict__ gslice, const float (&row)[NF], int lane, int valid_rows = 64) {
    static_assert(ROWS == 64 || ROWS == 32, "whole wave or half a wave per pass");
    // the slice start is the same for all lanes, but derived from per-lane values: hand the compiler a provably uniform
    // pointer, or it wraps every buffer store in a waterfall loop (4 readfirstlane + compare + exec mask, ~10 instructions each)
    const uintptr_t gaddr = reinterpret_cast<uintptr_t>(gslice);
    const uintptr_t guni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gaddr >> 32)) << 32) |
                           (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gaddr);
    const int vrows = __builtin_amdgcn_readfirstlane(valid_rows);
#pragma unroll
    for (int half = 0; half < 64 / ROWS; ++half) {
        if (ROWS == 64 || (lane >> 5) == half) {
            float *mine = slab + (lane & (ROWS - 1)) * NF;
            if constexpr (NF % 4 == 0) {
#pragma unroll
                for (int i = 0; i < NF / 4; ++i) reinterpret_cast<float4 *>(mine)[i] = make_float4(row[4 * i], row[4 * i + 1], row[4 * i + 2], row[4 * i + 3]);
            } else if constexpr (NF % 2 == 0) {
#pragma unroll
                for (int i = 0; i < NF / 2; ++i) reinterpret_cast<float2 *>(mine)[i] = make_float2(row[2 * i], row[2 * i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < NF; ++i) mine[i] = row[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int N4 = ROWS * NF / 4;                    // float4 pieces in this pass's slice (ROWS*NF is a multiple of 4)
        const float4 *s4 = reinterpret_cast<const float4 *>(slab) + lane;
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        int rows_here = vrows - half * ROWS;
        rows_here = rows_here < 0 ? 0 : (rows_here > ROWS ? ROWS : rows_here);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(guni + (uintptr_t)half * ROWS * NF * 4), 0, rows_here * NF * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < (N4 + 63) / 64; ++j)
            if (j * 64 + lane < N4) {
                const float4 v = s4[j * 64];
                __builtin_amdgcn_raw_buffer_store_b128((u4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, rs,
                                                       (j * 64 + lane) * 16, 0, 16 /* sc1: write-through */);
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                     // the next pass may overwrite the slab only behind these reads
    }
}

// ONE pass over the env's cylinders: line of sight to the evader (:786, LOS) and the k nearest by
// (3-D distance - size), ties -> lower index (:767-778).  The ordering is decided on SQUARED
// distances (no sqrt): md = RN(RN(sqrt(d2)) - size) is monotone in d2, so both orders agree
// whenever consecutive candidates differ by more than 2^-16 relative (then their md differ by
// >= 4 ulp and cannot tie); otherwise the exact md insertion below decides.
// Fast path: 32-bit keys = squared-distance bits with the cylinder index in the 4 low mantissa
// bits (non-negative floats order like unsigned ints), kept sorted by a branch-free min/max
// insertion network.  The 2^-19 truncation is covered by the 2^-16 gap test below.
// KT = keys tracked: one more than the largest k served (KM + 1), or exactly k + 1 where k is a compile-time constant (the fixed-shape
// instantiations of the step kernel: k = 3 tracks 4 keys — two instructions fewer per cylinder); only bi[0 .. KT) are written.
template <int NT, bool LOS, int KM = kMaxK, int KT = KM + 1, class Cfg>
HNS_DEV void cylinder_pass(const Cfg &c, int C, int K, const V3 &pos, const V3 &tp, const V3 &tpB, const float *cyl,
                           int bi[KM + 1], bool &any_block, bool &any_block1) {
    static_assert(KT >= 2 && KT <= KM + 1, "tracked keys");
    constexpr int kTrack = KT;                  // one more than k: guards the k-th/(k+1)-th boundary
    uint32_t key[kTrack];
#pragma unroll
    for (int i = 0; i < kTrack; ++i) key[i] = 0x7F80000Fu;             // +inf | 15
    LosLine los = {}, los1 = {};
    if constexpr (LOS) los = d_los_setup(c, pos, tp);
    bool los_uncertain = false, los_uncertain1 = false;
    float mb0 = kInf, mt0 = kInf, mb1 = kInf, mt1 = kInf;      // running minima behind the uncertainty test (d_los_cylinder_fast_acc)
    any_block = false; any_block1 = false;
    if constexpr (LOS && NT == 2) los1 = d_los_setup(c, pos, tpB);
#pragma unroll 4
    for (int k = 0; k < C; ++k) {
        const float ccx = cyl[3 * k], ccy = cyl[3 * k + 1], ccz = cyl[3 * k + 2];
        const float ex = pos.x - ccx, ey = pos.y - ccy, ez = pos.z - ccz;
        // (the line-of-sight tests take the pursuer-relative offsets the key below needs anyway: los.dpx/dpy ARE pos.x/pos.y — d_los_cylinder_fast_rel)
        if constexpr (LOS) any_block = d_los_cylinder_fast_acc(los, ccx, ccy, ccz, ex, ey, mb0, mt0) || any_block;
        if constexpr (LOS && NT == 2) any_block1 = d_los_cylinder_fast_acc(los1, ccx, ccy, ccz, ex, ey, mb1, mt1) || any_block1;
        const float d2 = HNS_FMA(ez, ez, HNS_FMA(ey, ey, ex * ex));          // the radicand of d_norm3
        uint32_t nk = (__float_as_uint(d2) & 0xFFFFFFF0u) | (uint32_t)k;
#pragma unroll
        for (int i = 0; i < kTrack; ++i) {
            uint32_t lo = min(key[i], nk);
            nk = max(key[i], nk);
            key[i] = lo;
        }
    }
    if constexpr (LOS) {
        los_uncertain = d_los_uncertain(los, mb0, mt0);
        if constexpr (NT == 2) los_uncertain1 = d_los_uncertain(los1, mb1, mt1);
        if (los_uncertain) any_block = d_blocked_exact(c, C, los, cyl);
        if (NT == 2 && los_uncertain1) any_block1 = d_blocked_exact(c, C, los1, cyl);
    }
    float bd[kTrack];
#pragma unroll
    for (int i = 0; i < kTrack; ++i) { bd[i] = __uint_as_float(key[i] & 0xFFFFFFF0u); bi[i] = (int)(key[i] & 15u); }
    bool order_safe = bd[0] > 1e-5f;
#pragma unroll
    for (int i = 0; i < kTrack - 1; ++i)
        if (i < K) order_safe = order_safe && (bd[i + 1] > bd[i] * 1.0000152587890625f);   // 1 + 2^-16
    if (!order_safe) {                          // rare: exact (distance - size) keys, as the reference sorts
#pragma unroll
        for (int i = 0; i < kTrack; ++i) { bd[i] = kInf; bi[i] = 0; }
        for (int k = 0; k < C; ++k) {
            float md = d_norm3(pos.x - cyl[3 * k], pos.y - cyl[3 * k + 1], pos.z - cyl[3 * k + 2]) - c.cylinder_size;
            if (md < bd[kTrack - 2]) {         // (the last tracked slot but one = position KM - 1 in the general instantiation: slots beyond k are not read)
                bd[kTrack - 2] = md; bi[kTrack - 2] = k;
#pragma unroll
                for (int i = kTrack - 2; i > 0; --i) {
                    if (bd[i] < bd[i - 1]) {
                        float td = bd[i]; bd[i] = bd[i - 1]; bd[i - 1] = td;
                        int ti = bi[i]; bi[i] = bi[i - 1]; bi[i - 1] = ti;
                    }
                }
            }
        }
    }
}

// ---- A8 (agent thread): observation of one pursuer on the post-physics state -------------------
// multirotor.py:599-633, hideandseek.py:746-917.  obs_self / state_drones are stored straight to
// global memory (5 float4 per thread, thread-contiguous); the relative position of the evader is
// written UNMASKED and the env wave re-masks it in the rare case that no pursuer detects the
// evader (:791-794).  Returns the flags and the k-nearest selection the reward pass needs.
// Two-evader extension (NT = 2, not in the reference): rows grow to 24 values = the reference's 20 +
// the relative position of the second evader + one zero; line of sight / detection per evader.
// STAGED (step kernel, full tiles): every output slice goes through the wave's slab (wave_store_rows); `sOCyl` is then
// the slab of this wave and gOth / gSelf / gState / gOCyl are still the THREAD's rows (the wave's slice starts `lane` rows earlier).
// KM > kMaxK (wide selections): the k-nearest rows go straight from the thread to its row of `gOCyl` (never staged).
template <int A, int NT, bool STAGED = false, int PS = 13, int KM = kMaxK, class Cfg = hns_cfg>
HNS_DEV void agent_obs(const Cfg &c, int C, int K, int le, int a, const Rigid &s, const V3 &tp, const V3 &tpB, float progress,
                       const float *cyl, const float *sDS, float *gOth, float *sOCyl, float *gSelf, float *gState,
                       bool &blocked, bool &det, bool &blockedB, bool &detB, int knn_idx[KM], bool knn_masked[KM], bool st = true, bool st_oth = true,
                       float *gOCyl = nullptr, float *dist_out = nullptr, bool st_ocyl = true) {
    static_assert(!(STAGED && KM > kMaxK), "wide k-nearest selections are not staged");
    constexpr int SDW = NT == 2 ? 24 : HNS_SELF_DIM;
    const int lane = threadIdx.x & 63;
    float rtx = s.pos.x - tp.x, rty = s.pos.y - tp.y, rtz = s.pos.z - tp.z;
    float dist = d_norm3(rtx, rty, rtz);
    if (dist_out) *dist_out = dist;
    const float t = progress * c.inv_max_episode_length;              // :796 (CUDA scalar-division form)
    V3 heading = d_quat_rot_x(s.q);                                   // multirotor.py:613-614
    V3 up = d_quat_rot_z(s.q, 1.0f);
    float4 v0 = make_float4(rtx, rty, rtz, s.q.w);
    float4 v1 = make_float4(s.q.x, s.q.y, s.q.z, s.lin.x);
    float4 v2 = make_float4(s.lin.y, s.lin.z, heading.x, heading.y);
    float4 v3 = make_float4(heading.z, up.x, up.y, up.z);
    float4 v4 = make_float4(t, t, t, t);
    float4 *so = reinterpret_cast<float4 *>(gSelf);                   // :856-863
    if constexpr (!STAGED) {
        if (st) { so[0] = v0; so[1] = v1; so[2] = v2; so[3] = v3; so[4] = v4; }
        if (gState && st) {                                                      // :871-886 (never masked)
            float4 *ss = reinterpret_cast<float4 *>(gState);
            ss[0] = v0; ss[1] = v1; ss[2] = v2; ss[3] = v3; ss[4] = v4;
        }
    }
    float dist1 = 0.0f;
    float4 v5 = make_float4(0, 0, 0, 0);
    if constexpr (NT == 2) {
        const float r1x = s.pos.x - tpB.x, r1y = s.pos.y - tpB.y, r1z = s.pos.z - tpB.z;
        dist1 = d_norm3(r1x, r1y, r1z);
        v5 = make_float4(r1x, r1y, r1z, 0.0f);
        if constexpr (!STAGED) {
            if (st) so[5] = v5;
            if (gState && st) reinterpret_cast<float4 *>(gState)[5] = v5;
        }
    }
    if constexpr (STAGED) {
        float row[SDW] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w, v4.x, v4.y, v4.z, v4.w};
        if constexpr (NT == 2) { row[20] = v5.x; row[21] = v5.y; row[22] = v5.z; row[23] = v5.w; }
        if (st) wave_store_rows<SDW, slab_rows(A)>(sOCyl, gSelf - lane * SDW, row, lane);
        if (gState && st) wave_store_rows<SDW, slab_rows(A)>(sOCyl, gState - lane * SDW, row, lane);
    }
    // state_others: p_i - p_j, j != i ascending (:750-751, utils/torch.py:41-53); (A-1)*3 floats per
    // thread, thread-contiguous in global memory
    if (A > 1) {
        float o[(A > 1 ? A - 1 : 1) * 3];
#pragma unroll
        for (int w = 0; w < A - 1; ++w) {
            const int j = w + (w >= a ? 1 : 0);
            const float *rj = sDS + (le * A + j) * PS;        // partner positions: rows of PS floats
            o[3 * w] = s.pos.x - rj[0]; o[3 * w + 1] = s.pos.y - rj[1]; o[3 * w + 2] = s.pos.z - rj[2];
        }
        if constexpr (STAGED) {
            if (st_oth) wave_store_rows<(A > 1 ? A - 1 : 1) * 3, slab_rows(A)>(sOCyl, gOth - lane * (A - 1) * 3, o, lane);
        } else if (!st_oth) {
        } else if ((((A - 1) * 3) & 1) == 0) {
            float2 *g2 = reinterpret_cast<float2 *>(gOth);
#pragma unroll
            for (int i = 0; i < (A - 1) * 3 / 2; ++i) g2[i] = make_float2(o[2 * i], o[2 * i + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < (A - 1) * 3; ++i) gOth[i] = o[i];
        }
    }
    int bi[KM + 1];
    bool any_block, any_block1;
    cylinder_pass<NT, true, KM>(c, C, K, s.pos, tp, tpB, cyl, bi, any_block, any_block1);
    blocked = any_block;
    det = (dist < c.drone_detect_radius) && !blocked;                 // :787-789
    if constexpr (NT == 2) {
        blockedB = any_block1;
        detB = (dist1 < c.drone_detect_radius) && !any_block1;
    }
    float *oc = KM > kMaxK ? gOCyl : sOCyl + (le * A + a) * K * 5;
    float krow[STAGED ? kMaxK * 5 : 1];
#pragma unroll
    for (int sidx = 0; sidx < KM; ++sidx) {
        if (sidx < K) {
            const float *cc = cyl + 3 * bi[sidx];
            bool masked = cc[2] < 0.0f;                                // :759,775-778
            knn_idx[sidx] = bi[sidx];
            knn_masked[sidx] = masked;
            float *row = STAGED ? krow + sidx * 5 : oc + sidx * 5;
            const float mv = c.mask_value, ch = c.cylinder_height, cs = c.cylinder_size;   // values, not lvalues (see d_rotor)
            row[0] = masked ? mv : s.pos.x - cc[0];
            row[1] = masked ? mv : s.pos.y - cc[1];
            row[2] = masked ? mv : s.pos.z - cc[2];
            row[3] = masked ? mv : ch;
            row[4] = masked ? mv : cs;
        }
    }
    if constexpr (STAGED) {
        if (st_ocyl) {
            if (K == 3) {
                float r[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) r[i] = krow[i];
                wave_store_rows<15, slab_rows(A)>(sOCyl, gOCyl - lane * 15, r, lane);
            } else if (K == 4) {
                wave_store_rows<20, slab_rows(A)>(sOCyl, gOCyl - lane * 20, krow, lane);
            } else if (K == 2) {
                float r[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) r[i] = krow[i];
                wave_store_rows<10, slab_rows(A)>(sOCyl, gOCyl - lane * 10, r, lane);
            } else {
                float r[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) r[i] = krow[i];
                wave_store_rows<5, slab_rows(A)>(sOCyl, gOCyl - lane * 5, r, lane);
            }
        }
    }
}

}  // namespace hns
