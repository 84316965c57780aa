// hns_kernels.hip — fused HideAndSeek environment step + reset for gfx950 (MI355X), and the
// C ABI of include/hns.h.
//
// Mapping.  One workgroup owns 64 consecutive envs and is WAVE-SPECIALISED:
//   * waves 0..A-1  ("agent waves"): thread i <-> pursuer (env i/A, agent i%A) — every lane busy
//     with per-drone math (controller, rotors, downwash, rigid-body integration, observation);
//   * wave A        ("env wave"):    lane l   <-> env l — evader policy (line of sight, potential
//     field), per-env reductions over the pursuers, reward assembly, done, statistics.
// The two roles are different instruction streams that run CONCURRENTLY on different SIMDs of the
// CU and meet at workgroup barriers; nothing env-level is recomputed per agent lane.
// All tensors keep the reference's [E,A,...] layouts: a workgroup's slice of every array is one
// contiguous byte range, moved HBM<->LDS with coalesced 16-byte-per-lane accesses (per-agent
// float4 records go straight to registers) and picked apart / assembled in LDS.
//
// One launch does the whole step (reference call tree: transforms.py:425-459 ->
// lee_position_controller.py:476-550 -> hideandseek.py:725-744 -> multirotor.py:466-508 ->
// rotor_group.py:55-71 -> [PhysX sim.step() replaced by d_integrate] -> hideandseek.py:746-917
// -> :919-1065).  No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "hns_device.h"
#include "hns_host.h"

namespace hns {

constexpr int kEPB = 64;   // envs per workgroup = lanes of the env wave
constexpr int kMaxK = 4;   // top-k insertion network width of the step kernels (obs_max_cylinder <= 4: the reference's default is 3)
constexpr int kWideK = 16;  // the same network for every k the cylinder count allows: first-design step kernel + reset kernel, no staging of the k-nearest rows
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
constexpr int kMaxT = 2;   // evaders per env (1 = the reference; 2 = BASELINE config 5's extension)

template <int A>
struct Geo {
    static constexpr int NA = kEPB * A;         // agent threads
    static constexpr int T = kEPB * (A + 1);    // + the env wave
};

// Reset / first-design step: passed by value (~900 B of kernel arguments).  A launch with <= 64 B of arguments is 0.5-0.6 us
// faster (tools/microbench/launch_gap.hip), but a block in device memory read through the scalar cache in front of a wave's
// first global load costs more than that (+1.8 us measured) — so the step kernel of the fourth design takes StepArgs: the
// pointers behind its first loads by value, everything else through `rest`, whose scalar loads travel beside those loads.
struct Params {
    hns_cfg cfg;
    hns_buffers buf;
    const float *action;        // step
    const uint8_t *reset_mask;  // reset (nullable)
    uint32_t seed_lo, seed_hi, epoch;
    unsigned long long *prof;   // optional per-wave phase timestamps (diagnostics), else null
    uint32_t cyl_magic;         // ceil(2^32 / (3*C)): index / (3*C) as a multiply-high
    const float *tasks;         // reset: optional [E, 3A+3NT+3C] task vectors (envgen: pursuers | evader(s) | cylinder slots), else null
    int32_t task_first;         // envs >= task_first take their placement from `tasks`
    uint32_t lab_stagger;
    uint32_t lab;               // ablation switches of the measurement build (-DHNS_LAB, tools/step_lab.py); unused otherwise
};

struct StepArgs {               // 64 B
    const float *action;
    float *prev_action, *drone_state, *pid_integ, *pid_last_rate, *throttle;
    const float *cylinders;     // 16-byte aligned; the low four bits carry num_cylinders - 1 (the 64-byte block has no room for another word,
                                // and the two-evader kernel's first loads need the count before the parameter block is warm)
    const Params *rest;         // device copy of the launch's Params (action = null), kept by the env handle
};
static_assert(sizeof(StepArgs) == 64, "the argument block of the step kernel is sized for the fast launch path");

// Measurement build only: parts of the step kernel can be switched off at run time (HNS_LAB_FLAGS) to time what is left.
#ifdef HNS_LAB
#define LAB(bit) ((p.lab & (bit)) != 0u)
#else
#define LAB(bit) false
#endif
enum { LAB_NOSTORE = 1, LAB_NOP1 = 2, LAB_NOP2 = 4, LAB_NOP3A = 8, LAB_NOP3B = 16, LAB_NOLOAD = 32,
       LAB_NOST_SELF = 64, LAB_NOST_OTH = 128, LAB_NOST_REC = 256, LAB_NOST_DS = 512, LAB_NOST_OCYL = 1024, LAB_NOST_STATS = 2048, LAB_HWID = 4096, LAB_NOLOS2 = 16384 };

constexpr int kProfSlots = 16;
// lane 0 of every wave stamps s_memtime at a phase boundary (only when a buffer is attached)
HNS_DEV void prof_mark(unsigned long long *prof, int slot) {
    if (prof && (threadIdx.x & 63) == 0) {
        int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        // slots 14/15 use the chip-wide constant 100 MHz clock (comparable across XCDs)
        prof[(size_t)wave * kProfSlots + slot] = (slot >= 14) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();
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

// ---- stores with an explicit cache policy --------------------------------------------------------------------------
// HNS_ST_POLICY: 0 = plain (write-back: the lines stay dirty in the XCD's L2 and are flushed at the end of the launch),
// 1 = sc1 (write-through: the bytes leave the L2 while the launch still computes; MI355X_MICROARCH.md, stores of each flavour)
#ifndef HNS_ST_POLICY
#define HNS_ST_POLICY 1
#endif
HNS_DEV void st_f4(float4 *p, const float4 &v) {
#if HNS_ST_POLICY == 1
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
#else
    *p = v;
#endif
}
HNS_DEV void st_f1(float *p, float v) {
#if HNS_ST_POLICY == 1
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

// ---- output rows of one agent wave: registers -> wave-private LDS slab -> ONE contiguous slice of global memory -----
// Every [E,A,...] output keeps the reference's layout, so the 64 rows a wave produces are one contiguous slice of
// 64*NF floats.  A thread storing its own row issues 16- (or 8-/4-) byte pieces at a stride of NF floats: every lane of
// the store instruction lands on a different cache line (measured: state_self 1.6 us, state_others 1.0 us of the
// 28 us step).  Instead the wave parks its rows in its slab and stores the slice back linearly, 16 B per lane, whole
// lines per instruction.  The slab is private to the wave: LDS operations of one wave execute in order, no barrier.
template <int NF, int ROWS = 64>
HNS_DEV void wave_store_rows(float *__restrict__ slab, float *__restrict__ gslice, const float (&row)[NF], int lane) {
    static_assert(ROWS == 64 || ROWS == 32, "whole wave or half a wave per pass");
    // the slice start is the same for all lanes, but derived from per-lane values: hand the compiler a provably uniform
    // pointer, or it wraps every buffer store in a waterfall loop (4 readfirstlane + compare + exec mask, ~10 instructions each)
    const uintptr_t gaddr = reinterpret_cast<uintptr_t>(gslice);
    const uintptr_t guni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gaddr >> 32)) << 32) |
                           (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gaddr);
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
#ifndef HNS_OUT_AUX
#define HNS_OUT_AUX 16
#endif
#if HNS_OUT_AUX != 0
        // cache policy of the output stores (aux: 1 = sc0, 2 = nt, 16 = sc1 write-through; measured A/B: plain 28.3, nt 27.9, sc1 27.2 us)
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(guni + (uintptr_t)half * ROWS * NF * 4), 0, ROWS * NF * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < (N4 + 63) / 64; ++j)
            if (j * 64 + lane < N4) {
                const float4 v = s4[j * 64];
                __builtin_amdgcn_raw_buffer_store_b128((u4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, rs,
                                                       (j * 64 + lane) * 16, 0, HNS_OUT_AUX);
            }
#else
        float4 *g4 = reinterpret_cast<float4 *>(guni + (uintptr_t)half * ROWS * NF * 4) + lane;
#pragma unroll
        for (int j = 0; j < (N4 + 63) / 64; ++j)
            if (j * 64 + lane < N4) g4[j * 64] = s4[j * 64];
#endif
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
template <int NT, bool LOS, int KM = kMaxK, class Cfg>
HNS_DEV void cylinder_pass(const Cfg &c, int C, int K, const V3 &pos, const V3 &tp, const V3 &tpB, const float *cyl,
                           int bi[KM + 1], bool &any_block, bool &any_block1) {
    constexpr int kTrack = KM + 1;              // one more than k: guards the k-th/(k+1)-th boundary
    uint32_t key[kTrack];
#pragma unroll
    for (int i = 0; i < kTrack; ++i) key[i] = 0x7F80000Fu;             // +inf | 15
    LosLine los = {}, los1 = {};
    if constexpr (LOS) los = d_los_setup(c, pos, tp);
    bool los_uncertain = false, los_uncertain1 = false;
    any_block = false; any_block1 = false;
    if constexpr (LOS && NT == 2) los1 = d_los_setup(c, pos, tpB);
#pragma unroll 4
    for (int k = 0; k < C; ++k) {
        const float ccx = cyl[3 * k], ccy = cyl[3 * k + 1], ccz = cyl[3 * k + 2];
        if constexpr (LOS) any_block = d_los_cylinder_fast(los, ccx, ccy, ccz, los_uncertain) || any_block;
        if constexpr (LOS && NT == 2) any_block1 = d_los_cylinder_fast(los1, ccx, ccy, ccz, los_uncertain1) || any_block1;
        const float ex = pos.x - ccx, ey = pos.y - ccy, ez = pos.z - ccz;
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
        if (los_uncertain) any_block = d_blocked_exact(c, C, los, cyl);
        if (NT == 2 && los_uncertain1) any_block1 = d_blocked_exact(c, C, los1, cyl);
    }
    float bd[kTrack];
#pragma unroll
    for (int i = 0; i < kTrack; ++i) { bd[i] = __uint_as_float(key[i] & 0xFFFFFFF0u); bi[i] = (int)(key[i] & 15u); }
    bool order_safe = bd[0] > 1e-5f;
#pragma unroll
    for (int i = 0; i < KM; ++i)
        if (i < K) order_safe = order_safe && (bd[i + 1] > bd[i] * 1.0000152587890625f);   // 1 + 2^-16
    if (!order_safe) {                          // rare: exact (distance - size) keys, as the reference sorts
#pragma unroll
        for (int i = 0; i < kTrack; ++i) { bd[i] = kInf; bi[i] = 0; }
        for (int k = 0; k < C; ++k) {
            float md = d_norm3(pos.x - cyl[3 * k], pos.y - cyl[3 * k + 1], pos.z - cyl[3 * k + 2]) - c.cylinder_size;
            if (md < bd[KM - 1]) {
                bd[KM - 1] = md; bi[KM - 1] = k;
#pragma unroll
                for (int i = KM - 1; i > 0; --i) {
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

// =================================================================================================
// The fused step kernel
// =================================================================================================
// (the second evader costs ~15 registers: without the cap of 128 the 7-wave workgroups of the 6-pursuer
//  shape drop from two per CU to one)
// FULL: the batch is a whole number of 64-env tiles (E % 64 == 0) — no lane is ever idle, so every `valid` test and
// every default value behind it is compiled out; the generic instantiation serves ragged batches.
template <int A, int NT, bool FULL, int KM = kMaxK>
__global__ __launch_bounds__(Geo<A>::T, (NT == 2 && KM == kMaxK) ? 4 : 1) void hns_step_kernel(const Params p) {
    static_assert(!(FULL && KM > kMaxK), "wide k-nearest selections: the generic (unstaged) instantiation");
    constexpr int T = Geo<A>::T, NA = Geo<A>::NA;
    constexpr int SD = NT == 2 ? 24 : HNS_SELF_DIM;      // floats per state_self / state_drones row
    constexpr int kRedS = red_stride(NT);
    extern __shared__ __align__(16) float smem[];
    const hns_cfg &c = p.cfg;
    const hns_buffers &b = p.buf;
    const int C = c.num_cylinders, K = c.obs_max_cylinder, E = c.num_envs;
    const bool with_state = c.write_critic_state && b.state_drones != nullptr;
    const Lds L = lds_layout(A, C, K, NT);
    float *sDS = smem + L.ds, *sCyl = smem + L.cyl, *sTp = smem + L.tp;
    float *sRed = smem + L.red, *sOCyl = smem + L.ocyl;

    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * kEPB;
    const int nenv = FULL ? kEPB : min(kEPB, E - e0);
    const bool env_wave = tid >= NA;
    const int le = env_wave ? tid - NA : tid / A;      // local env
    const int a = env_wave ? 0 : tid - le * A;         // agent index (agent threads)
    const int e = e0 + le;
    const bool valid = FULL ? true : le < nenv;
    const size_t ia = (size_t)e0 * A + (env_wave ? 0 : tid);

    prof_mark(p.prof, 0);
    prof_mark(p.prof, 14);
#ifdef HNS_LAB
    if (p.lab_stagger) {            // experiment: workgroups of later dispatch rounds start their loads later
        const int slot = (blockIdx.x >> 8) & 3;
        for (int i = 0; i < slot * (int)p.lab_stagger; ++i) __builtin_amdgcn_s_sleep(8);   // 8 x 64 cycles
    }
#endif
    // ---- load: per-agent float4 records straight to registers, the rest through LDS ------------
    float4 act4 = make_float4(0, 0, 0, 0), thr4 = act4, integ4 = act4, last4 = act4, prev4 = act4;
    float progress = 0.0f;
    if (valid && !LAB(LAB_NOLOAD)) {
        progress = b.progress[e];
        if (!env_wave) {
            act4 = reinterpret_cast<const float4 *>(p.action)[ia];
            thr4 = reinterpret_cast<const float4 *>(b.throttle)[ia];
            integ4 = reinterpret_cast<const float4 *>(b.pid_integ)[ia];
            last4 = reinterpret_cast<const float4 *>(b.pid_last_rate)[ia];
            prev4 = reinterpret_cast<const float4 *>(b.prev_action)[ia];
        }
    }
    const bool full = FULL ? true : (KM == kMaxK && nenv == kEPB);
    if (LAB(LAB_NOLOAD)) {
    } else if (full) {
        coop_copy_full<T, kEPB * A * 13>(sDS, b.drone_state + (size_t)e0 * A * 13);
        coop_copy_full<T, kEPB * 3 * NT>(sTp, b.target_pos + (size_t)e0 * 3 * NT);
    } else {
        coop_g2s<T>(sDS, b.drone_state + (size_t)e0 * A * 13, nenv * A * 13);
        coop_g2s<T>(sTp, b.target_pos + (size_t)e0 * 3 * NT, nenv * 3 * NT);
    }
    if (!LAB(LAB_NOLOAD)) coop_cyl<T, true>(sCyl, b.cylinders + (size_t)e0 * C * 3, nenv, 3 * C, L.cyl_stride, p.cyl_magic, nullptr);
    // the env wave keeps its env's statistics in registers: row-major [S][E] makes every row a
    // fully coalesced 256-byte wave access, no LDS staging needed
    float st[HNS_NUM_STATS];
#pragma unroll
    for (int i = 0; i < HNS_NUM_STATS; ++i) st[i] = 0.0f;
    // tanh of the raw action needs only the first record loaded: evaluated here, under the load burst (VALU idle)
    float4 ta = make_float4(0, 0, 0, 0);
    if (!env_wave && valid) ta = d_action_tanh(act4);
    __syncthreads();
    prof_mark(p.prof, 1);
    // issued after the barrier: the statistics are first needed behind phase 1, so their 6 MB stay out of the
    // bandwidth-bound load burst at the head of the launch and stream in under the phase-1 arithmetic
    if (env_wave && valid && !LAB(LAB_NOLOAD)) {
#pragma unroll
        for (int i = 0; i < HNS_NUM_STATS; ++i) st[i] = b.stats[(size_t)i * E + e];
    }

    const float *cyl = sCyl + le * L.cyl_stride;
    Rigid s = {};
    s.q.w = 1.0f;
    float thrust[4] = {0, 0, 0, 0}, moment[4] = {0, 0, 0, 0};
    V3 tw = {0.f, 0.f, 0.f}, tvel = {0.f, 0.f, 0.f}, tpn = {0.f, 0.f, 0.f};
    V3 Fenv1 = {0.f, 0.f, 0.f}, tvel1 = {0.f, 0.f, 0.f}, tpn1 = {0.f, 0.f, 0.f}, tp1 = {0.f, 0.f, 0.f};   // second evader (NT == 2)
    bool out_of_arena = false;
    float aerr = 0.f;

    // ================= phase 1: pre-physics on S_t =================================================
    V3 Fenv = {0.f, 0.f, 0.f};
    V3 tp0 = {0.f, 0.f, 0.f};
    if (valid) tp0 = {sTp[le * 3 * NT], sTp[le * 3 * NT + 1], sTp[le * 3 * NT + 2]};
    if (NT == 2 && valid) tp1 = {sTp[le * 3 * NT + 3], sTp[le * 3 * NT + 4], sTp[le * 3 * NT + 5]};
    if (LAB(LAB_NOP1)) {
        if (!env_wave && valid) load_rigid(sDS + tid * 13, s);
    } else if (!env_wave) {
        if (valid) {
            load_rigid(sDS + tid * 13, s);
            float cmd[4], thr_diff;
            float ctbr4[4], trate[3];
            d_ctbr_pid_squashed(c, ta, s.q, s.ang, prev4, integ4, last4, cmd, aerr, ctbr4, trate);    // A1 + A2
            if (b.ctbr) reinterpret_cast<float4 *>(b.ctbr)[ia] = make_float4(ctbr4[0], ctbr4[1], ctbr4[2], ctbr4[3]);           // transforms.py:456
            if (b.target_rate) reinterpret_cast<float4 *>(b.target_rate)[ia] = make_float4(trate[0], trate[1], trate[2], 0.0f);  // :457
            prof_mark(p.prof, 10);
            d_rotor(c, cmd, thr4, thrust, moment, thr_diff);                      // A3
            prof_mark(p.prof, 11);
            float ts = ((thrust[0] + thrust[1]) + thrust[2]) + thrust[3];
            tw = d_quat_rot_z(s.q, ts);                                           // multirotor.py:491
            // this pursuer's push on the evader (hideandseek.py:1074-1088), summed by the env wave
            bool blocked_pre = d_blocked(c, C, s.pos, tp0, cyl);                  // :1080
            V3 fp = d_prey_pursuer_term(c, s.pos, tp0, blocked_pre);
            float *red = sRed + tid * kRedS;
            red[R_AERR] = aerr; red[R_TD] = thr_diff;
            red[R_FX] = fp.x; red[R_FY] = fp.y; red[R_FZ] = fp.z;
            red[R_TWX] = tw.x; red[R_TWX + 1] = tw.y; red[R_TWX + 2] = tw.z;
            if constexpr (NT == 2) {                                                        // the same pursuer's push on the second evader
                const bool blocked1 = d_blocked(c, C, s.pos, tp1, cyl);
                const V3 f1 = d_prey_pursuer_term(c, s.pos, tp1, blocked1);
                red[R_F1X] = f1.x; red[R_F1X + 1] = f1.y; red[R_F1X + 2] = f1.z;
            }
        }
    } else if (valid) {
        // A6: arena + cylinder terms of the evader's potential field (hideandseek.py:1090-1136)
        Fenv = d_prey_arena_term(c, tp0, out_of_arena);
        float fcx = 0.f, fcy = 0.f;
#pragma unroll 4
        for (int k = 0; k < C; ++k) {
            float tx, ty;
            d_prey_cylinder_term(c, tp0, cyl[3 * k], cyl[3 * k + 1], cyl[3 * k + 2], tx, ty);
            fcx += tx;
            fcy += ty;
        }
        tvel = {fcx, fcy, 0.0f};     // parked until the pursuer terms arrive
        if constexpr (NT == 2) {               // each evader runs the potential field on its own (they ignore each other)
            bool out1 = false;
            Fenv1 = d_prey_arena_term(c, tp1, out1);
            out_of_arena = out_of_arena || out1;
            float gx = 0.f, gy = 0.f;
#pragma unroll 4
            for (int k = 0; k < C; ++k) {
                float tx, ty;
                d_prey_cylinder_term(c, tp1, cyl[3 * k], cyl[3 * k + 1], cyl[3 * k + 2], tx, ty);
                gx += tx;
                gy += ty;
            }
            tvel1 = {gx, gy, 0.0f};
        }
    }
    __syncthreads();
    if (env_wave && valid && !LAB(LAB_NOP1)) {
        // force = sum over pursuers (ascending) + arena + cylinders; per-axis velocity (:741)
        V3 F = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float *red = sRed + (le * A + j) * kRedS;
            F.x = (j == 0) ? red[R_FX] : F.x + red[R_FX];
            F.y = (j == 0) ? red[R_FY] : F.y + red[R_FY];
            F.z = (j == 0) ? red[R_FZ] : F.z + red[R_FZ];
        }
        F.x = F.x + Fenv.x; F.y = F.y + Fenv.y; F.z = F.z + Fenv.z;
        F.x = F.x + tvel.x; F.y = F.y + tvel.y; F.z = F.z + 0.0f;
        tvel = {(c.v_prey * F.x) / (__builtin_fabsf(F.x) + 1e-5f), (c.v_prey * F.y) / (__builtin_fabsf(F.y) + 1e-5f),
                (c.v_prey * F.z) / (__builtin_fabsf(F.z) + 1e-5f)};
        tpn = {tp0.x + tvel.x * c.dt, tp0.y + tvel.y * c.dt, tp0.z + tvel.z * c.dt};   // evader: p += v dt
        { const float sf = (tpn.x + tpn.y) + tpn.z; flag_nonfinite(b.nonfinite, (sf - sf) != 0.0f, 2u); }
        if constexpr (NT == 2) {
            V3 G = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float *red = sRed + (le * A + j) * kRedS;
                G.x = (j == 0) ? red[R_F1X] : G.x + red[R_F1X];
                G.y = (j == 0) ? red[R_F1X + 1] : G.y + red[R_F1X + 1];
                G.z = (j == 0) ? red[R_F1X + 2] : G.z + red[R_F1X + 2];
            }
            G.x = G.x + Fenv1.x; G.y = G.y + Fenv1.y; G.z = G.z + Fenv1.z;
            G.x = G.x + tvel1.x; G.y = G.y + tvel1.y; G.z = G.z + 0.0f;
            tvel1 = {(c.v_prey * G.x) / (__builtin_fabsf(G.x) + 1e-5f), (c.v_prey * G.y) / (__builtin_fabsf(G.y) + 1e-5f),
                     (c.v_prey * G.z) / (__builtin_fabsf(G.z) + 1e-5f)};
            tpn1 = {tp1.x + tvel1.x * c.dt, tp1.y + tvel1.y * c.dt, tp1.z + tvel1.z * c.dt};
            { const float sf = (tpn1.x + tpn1.y) + tpn1.z; flag_nonfinite(b.nonfinite, (sf - sf) != 0.0f, 2u); }
        }
        // statistics that only need phase-1 data are folded in now, while the agent waves integrate
        // (A10 hideandseek.py:731-733, :1097-1098, :996-997)
        float sum_ae = 0.f, sum_td = 0.f, max_td = 0.f;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float *red = sRed + (le * A + j) * kRedS;
            const float td = red[R_TD];
            sum_ae = (j == 0) ? red[R_AERR] : sum_ae + red[R_AERR];
            sum_td = (j == 0) ? td : sum_td + td;
            max_td = (j == 0) ? td : (td > max_td ? td : max_td);
        }
        const float mae = sum_ae * c.inv_num_agents;
        st[HNS_ST_ACTION_ERROR_ORDER1_MEAN] += mae;
        if (mae > st[HNS_ST_ACTION_ERROR_ORDER1_MAX]) st[HNS_ST_ACTION_ERROR_ORDER1_MAX] = mae;
        st[HNS_ST_OUT_OF_ARENA] = ((st[HNS_ST_OUT_OF_ARENA] != 0.0f) || out_of_arena) ? 1.0f : 0.0f;
        st[HNS_ST_SMOOTHNESS_COEF] = c.smoothness_coef;
        st[HNS_ST_SMOOTHNESS_MEAN] += sum_td * c.inv_num_agents;
        if (max_td > st[HNS_ST_SMOOTHNESS_MAX]) st[HNS_ST_SMOOTHNESS_MAX] = max_td;
    }
    prof_mark(p.prof, 2);

    // ================= phase 2: forces, torques, integration (agent waves) ===========================
    if (!env_wave) {
        if (valid && !LAB(LAB_NOP2)) {
            V3 fdw = {0.f, 0.f, 0.f};                                             // A4: downwash, partners ascending
#pragma unroll
            for (int o = 0; o < A - 1; ++o) {
                const int j = o + (o >= a ? 1 : 0);
                const float *rj = sDS + (le * A + j) * 13;
                const float *tj = sRed + (le * A + j) * kRedS + R_TWX;
                V3 pj = {rj[0], rj[1], rj[2]};
                V3 twj = {tj[0], tj[1], tj[2]};
                V3 fj = d_downwash_pair(s.pos, pj, twj);
                fdw.x = (o == 0) ? fj.x : fdw.x + fj.x;
                fdw.y = (o == 0) ? fj.y : fdw.y + fj.y;
                fdw.z = (o == 0) ? fj.z : fdw.z + fj.z;
            }
            V3 fw = {tw.x + fdw.x, tw.y + fdw.y, tw.z + fdw.z};
            V3 tb;
            tb.x = ((c.rotor_py[0] * thrust[0] + c.rotor_py[1] * thrust[1]) + c.rotor_py[2] * thrust[2]) + c.rotor_py[3] * thrust[3];
            tb.y = -(((c.rotor_px[0] * thrust[0] + c.rotor_px[1] * thrust[1]) + c.rotor_px[2] * thrust[2]) + c.rotor_px[3] * thrust[3]);
            tb.z = ((moment[0] + moment[1]) + moment[2]) + moment[3];
            prof_mark(p.prof, 12);
            d_integrate(c, s, fw, tb);                                            // A5
            flag_nonfinite(b.nonfinite, rigid_not_finite(s), 1u);
            prof_mark(p.prof, 13);
        }
        if (valid && !LAB(LAB_NOSTORE | LAB_NOST_REC)) {
#if defined(HNS_REC_PLAIN)
            reinterpret_cast<float4 *>(b.throttle)[ia] = thr4;
            reinterpret_cast<float4 *>(b.pid_integ)[ia] = integ4;
            reinterpret_cast<float4 *>(b.pid_last_rate)[ia] = last4;
            reinterpret_cast<float4 *>(b.prev_action)[ia] = prev4;
            b.action_error[ia] = aerr;
#elif defined(HNS_REC_NT)
            __builtin_nontemporal_store(thr4, reinterpret_cast<float4 *>(b.throttle) + ia);
            __builtin_nontemporal_store(integ4, reinterpret_cast<float4 *>(b.pid_integ) + ia);
            __builtin_nontemporal_store(last4, reinterpret_cast<float4 *>(b.pid_last_rate) + ia);
            __builtin_nontemporal_store(prev4, reinterpret_cast<float4 *>(b.prev_action) + ia);
            __builtin_nontemporal_store(aerr, b.action_error + ia);
#else
            st_f4(reinterpret_cast<float4 *>(b.throttle) + ia, thr4);
            st_f4(reinterpret_cast<float4 *>(b.pid_integ) + ia, integ4);
            st_f4(reinterpret_cast<float4 *>(b.pid_last_rate) + ia, last4);
            st_f4(reinterpret_cast<float4 *>(b.prev_action) + ia, prev4);
            st_f1(b.action_error + ia, aerr);
#endif
        }
    }
    progress += 1.0f;                                                             // isaac_env.py:236
    prof_mark(p.prof, 3);
    __syncthreads();            // every read of S_t from sDS / sTp is done
    if (!env_wave) {
        if (valid) store_rigid(sDS + tid * 13, s);
    } else if (valid) {
        sTp[le * 3 * NT] = tpn.x; sTp[le * 3 * NT + 1] = tpn.y; sTp[le * 3 * NT + 2] = tpn.z;
        if constexpr (NT == 2) { sTp[le * 3 * NT + 3] = tpn1.x; sTp[le * 3 * NT + 4] = tpn1.y; sTp[le * 3 * NT + 5] = tpn1.z; }
    }
    __syncthreads();
    prof_mark(p.prof, 8);
    if (env_wave && !LAB(LAB_NOSTORE | LAB_NOST_DS)) {
        // the env wave is idle during phase 3a: it writes back S_{t+1} (drone_state slice, evader)
        const int lane = tid - NA;
        const float4 *s4 = reinterpret_cast<const float4 *>(sDS);
        float4 *g4 = reinterpret_cast<float4 *>(b.drone_state + (size_t)e0 * A * 13);
        if (full) {
#pragma unroll
            for (int k = 0; k < (kEPB * A * 13 / 4 + 63) / 64; ++k)
                if (k * 64 + lane < kEPB * A * 13 / 4) st_f4(g4 + k * 64 + lane, s4[k * 64 + lane]);
        } else {
            float *g = b.drone_state + (size_t)e0 * A * 13;
            for (int i = lane; i < nenv * A * 13; i += 64) g[i] = sDS[i];
        }
        if (valid) {
            float *gp = b.target_pos + (size_t)e * 3 * NT, *gv = b.target_vel + (size_t)e * 3 * NT;
            st_f1(gp, tpn.x); st_f1(gp + 1, tpn.y); st_f1(gp + 2, tpn.z);
            st_f1(gv, tvel.x); st_f1(gv + 1, tvel.y); st_f1(gv + 2, tvel.z);
            if constexpr (NT == 2) {
                gp[3] = tpn1.x; gp[4] = tpn1.y; gp[5] = tpn1.z;
                gv[3] = tvel1.x; gv[4] = tvel1.y; gv[5] = tvel1.z;
            }
        }
    }

    // ================= phase 3a: observation + per-agent reward terms on S_{t+1} =====================
    if (!env_wave && valid && !LAB(LAB_NOP3A)) {
        V3 tp = {sTp[le * 3 * NT], sTp[le * 3 * NT + 1], sTp[le * 3 * NT + 2]};
        V3 tpB = tp;
        if constexpr (NT == 2) tpB = {sTp[le * 3 * NT + 3], sTp[le * 3 * NT + 4], sTp[le * 3 * NT + 5]};
        bool blocked, det, blockedB = false, detB = false;
        int knn_idx[KM];
        bool knn_masked[KM];
        if constexpr (KM > kMaxK)
            agent_obs<A, NT, false, 13, KM>(c, C, K, le, a, s, tp, tpB, progress, cyl, sDS, b.obs_others + ia * (A - 1) * 3, sOCyl, b.obs_self + ia * SD,
                                            with_state ? b.state_drones + ia * SD : nullptr, blocked, det, blockedB, detB, knn_idx, knn_masked, true, true,
                                            b.obs_cylinders + ia * K * 5);
        else if (full)    // every output slice of the wave through its private slab: whole cache lines per store instruction
            agent_obs<A, NT, true>(c, C, K, le, a, s, tp, tpB, progress, cyl, sDS, b.obs_others + ia * (A - 1) * 3, sOCyl + (tid >> 6) * slab_floats(A, K, NT),
                                   b.obs_self + ia * SD, with_state ? b.state_drones + ia * SD : nullptr, blocked, det, blockedB, detB, knn_idx, knn_masked,
                                   !LAB(LAB_NOSTORE | LAB_NOST_SELF), !LAB(LAB_NOSTORE | LAB_NOST_OTH), b.obs_cylinders + ia * K * 5);
        else
            agent_obs<A, NT>(c, C, K, le, a, s, tp, tpB, progress, cyl, sDS, b.obs_others + ia * (A - 1) * 3, sOCyl, b.obs_self + ia * SD,
                             with_state ? b.state_drones + ia * SD : nullptr, blocked, det, blockedB, detB, knn_idx, knn_masked, !LAB(LAB_NOSTORE | LAB_NOST_SELF),
                             !LAB(LAB_NOSTORE | LAB_NOST_OTH));
        prof_mark(p.prof, 9);
        // the line of sight just taken is the next step's test at ITS t: carried in the fourth column of the controller record (hns.h)
        if (!LAB(LAB_NOSTORE | LAB_NOST_REC)) b.pid_last_rate[ia * 4 + 3] = (float)((blocked ? 1 : 0) + (blockedB ? 2 : 0));
        // hideandseek.py:919-995
        float d = d_norm3(tp.x - s.pos.x, tp.y - s.pos.y, tp.z - s.pos.z);
        float act = (d > c.catch_radius) ? 1.0f : 0.0f;
        float dist_rew = (-c.dist_reward_coef * d) * act;
        bool cap_ok = (d < c.catch_radius) && !blocked;
        if constexpr (NT == 2) {
            // extension: distance term to the NEAREST evader, capture of ANY evader, `blocked` = no line of sight to either
            const float d1 = d_norm3(tpB.x - s.pos.x, tpB.y - s.pos.y, tpB.z - s.pos.z);
            cap_ok = cap_ok || ((d1 < c.catch_radius) && !blockedB);
            blocked = blocked && blockedB;
            d = d1 < d ? d1 : d;
            act = (d > c.catch_radius) ? 1.0f : 0.0f;
            dist_rew = (-c.dist_reward_coef * d) * act;
        }
        float sp = d_norm3(s.lin.x, s.lin.y, s.lin.z);
        float speed_rew = -c.speed_coef * ((sp > c.v_drone) ? 1.0f : 0.0f);
        float cc = 0.f, cd = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < KM; ++sidx) {
            if (sidx < K) {
                const float *cy = cyl + 3 * knn_idx[sidx];
                float rx = s.pos.x - cy[0], ry = s.pos.y - cy[1];
                float dxy = d_norm2(rx, ry);
                float hit = ((dxy - c.cylinder_size) < c.collision_radius) ? 1.0f : 0.0f;
                if (knn_masked[sidx]) hit = 0.0f;
                cc = (sidx == 0) ? hit : cc + hit;
            }
        }
        float cr = -c.collision_coef * cc;
#pragma unroll
        for (int o = 0; o < A - 1; ++o) {
            const int j = o + (o >= a ? 1 : 0);
            const float *rj = sDS + (le * A + j) * 13;
            float dd = d_norm3(s.pos.x - rj[0], s.pos.y - rj[1], s.pos.z - rj[2]);
            float hit = (dd < c.coll_drone_dist) ? 1.0f : 0.0f;
            cd = (o == 0) ? hit : cd + hit;
        }
        cr = cr + -c.collision_coef * cd;
        float cw = ((s.pos.z > c.max_height) ? 1.0f : 0.0f) + ((HNS_FMA(s.pos.y, s.pos.y, s.pos.x * s.pos.x) > c.arena_sq) ? 1.0f : 0.0f);
        cr = cr + -c.collision_coef * cw;
        float sm = c.smoothness_coef * d_expf(-aerr);
        if (!c.use_deployment) sm = 0.0f;
        float *red = sRed + tid * kRedS;
        red[R_DIST] = dist_rew; red[R_SPEED] = speed_rew; red[R_CC] = cc; red[R_CD] = cd; red[R_CW] = cw;
        red[R_COLL] = cr; red[R_SMOOTH] = sm;
        if constexpr (NT == 2)
            red[R_FLAGS] = __int_as_float((cap_ok ? F_CAP : 0) | (blocked ? F_BLOCKED : 0) | (det ? F_DET : 0) | (detB ? F_DET1 : 0));
        else
            red[R_FLAGS] = __int_as_float((cap_ok ? F_CAP : 0) | (blocked ? F_BLOCKED : 0) | (det ? F_DET : 0));
    }
    prof_mark(p.prof, 4);
    __syncthreads();
    prof_mark(p.prof, 5);

    // ================= phase 3b: per-env reductions, reward, done, stats (env wave) ===================
    if (env_wave && valid && !LAB(LAB_NOP3B)) {
        const float iA = c.inv_num_agents;             // mean over agents = sum * (1/A), as torch's CUDA mean
        bool any_cap = false, all_blocked = true, any_coll = false, det_any = false, det_any1 = false;
        float sum_dist = 0, sum_speed = 0, sum_cc = 0, sum_cd = 0, sum_cw = 0, sum_coll = 0, sum_smooth = 0;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float *red = sRed + (le * A + j) * kRedS;
            int fl = __float_as_int(red[R_FLAGS]);
            any_cap |= (fl & F_CAP) != 0;
            all_blocked &= (fl & F_BLOCKED) != 0;
            det_any |= (fl & F_DET) != 0;
            det_any1 |= (fl & F_DET1) != 0;
            any_coll |= red[R_COLL] < 0.0f;
            if (j == 0) {
                sum_dist = red[R_DIST]; sum_speed = red[R_SPEED]; sum_cc = red[R_CC]; sum_cd = red[R_CD]; sum_cw = red[R_CW];
                sum_coll = red[R_COLL]; sum_smooth = red[R_SMOOTH];
            } else {
                sum_dist += red[R_DIST]; sum_speed += red[R_SPEED]; sum_cc += red[R_CC]; sum_cd += red[R_CD]; sum_cw += red[R_CW];
                sum_coll += red[R_COLL]; sum_smooth += red[R_SMOOTH];
            }
        }
        float detf = det_any ? 1.0f : 0.0f;
        if constexpr (NT == 2) detf = (det_any || det_any1) ? 1.0f : 0.0f;
        const float detect_rew = c.detect_reward_coef * detf;
        const float catch_rew = c.catch_reward_coef * (any_cap ? 1.0f : 0.0f);
        float sum_rew = 0.f;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float *red = sRed + (le * A + j) * kRedS;
            float r = ((((red[R_DIST] + detect_rew) + catch_rew) + red[R_COLL]) + red[R_SPEED]) + red[R_SMOOTH];
            if (!LAB(LAB_NOSTORE)) st_f1(b.reward + (size_t)e * A + j, r);
            sum_rew = (j == 0) ? r : sum_rew + r;
        }
        flag_nonfinite(b.nonfinite, (sum_rew - sum_rew) != 0.0f, 4u);
        if (!det_any && !LAB(LAB_NOSTORE)) {                   // hideandseek.py:791-794: mask the evader's rpos
#pragma unroll
            for (int j = 0; j < A; ++j) {
                float *o = b.obs_self + ((size_t)e * A + j) * SD;
                o[0] = c.mask_value; o[1] = c.mask_value; o[2] = c.mask_value;
            }
        }
        if (NT == 2 && !det_any1) {
#pragma unroll
            for (int j = 0; j < A; ++j) {
                float *o = b.obs_self + ((size_t)e * A + j) * SD + HNS_SELF_DIM;
                o[0] = c.mask_value; o[1] = c.mask_value; o[2] = c.mask_value;
            }
        }
#define ST(i) st[i]
        ST(HNS_ST_DISTANCE_REWARD) += sum_dist * iA;
        ST(HNS_ST_SUM_DETECT_STEP) += 1.0f * detf;
        float sdet = detect_rew, scat = catch_rew;
#pragma unroll
        for (int j = 1; j < A; ++j) { sdet += detect_rew; scat += catch_rew; }
        ST(HNS_ST_DETECT_REWARD) += sdet * iA;
        const bool capture_flag = catch_rew != 0.0f;                              // :945
        ST(HNS_ST_BLOCKED) += all_blocked ? 1.0f : 0.0f;
        ST(HNS_ST_SUCCESS) = (capture_flag || ST(HNS_ST_SUCCESS) != 0.0f) ? 1.0f : 0.0f;
        float cur = (capture_flag ? 1.0f : 0.0f) * progress + (capture_flag ? 0.0f : 1.0f) * (float)c.max_episode_length;
        if (cur < ST(HNS_ST_FIRST_CAPTURE_STEP)) ST(HNS_ST_FIRST_CAPTURE_STEP) = cur;
        ST(HNS_ST_CATCH_REWARD) += scat * iA;
        ST(HNS_ST_SPEED_REWARD) += sum_speed * iA;
        ST(HNS_ST_COLLISION_CYLINDER) += sum_cc * iA;
        ST(HNS_ST_COLLISION_DRONE) += sum_cd * iA;
        ST(HNS_ST_COLLISION) += any_coll ? 1.0f : 0.0f;
        ST(HNS_ST_COLLISION_WALL) += sum_cw * iA;
        ST(HNS_ST_COLLISION_REWARD) += sum_coll * iA;
        ST(HNS_ST_SMOOTHNESS_REWARD) += sum_smooth * iA;
        const bool done = progress >= (float)c.max_episode_length;                // :1008-1010
        if (done) {                                                               // :1017-1056
            ST(HNS_ST_COLLISION) = ST(HNS_ST_COLLISION) / progress;
            ST(HNS_ST_ACTION_ERROR_ORDER1_MEAN) = ST(HNS_ST_ACTION_ERROR_ORDER1_MEAN) / progress;
            ST(HNS_ST_TARGET_PREDICTED_ERROR) = ST(HNS_ST_TARGET_PREDICTED_ERROR) / progress;
            ST(HNS_ST_SMOOTHNESS_MEAN) = ST(HNS_ST_SMOOTHNESS_MEAN) / progress;
            ST(HNS_ST_SMOOTHNESS_REWARD) = ST(HNS_ST_SMOOTHNESS_REWARD) / progress;
            ST(HNS_ST_DISTANCE_REWARD) = ST(HNS_ST_DISTANCE_REWARD) / progress;
            ST(HNS_ST_DETECT_REWARD) = ST(HNS_ST_DETECT_REWARD) / progress;
            ST(HNS_ST_CATCH_REWARD) = ST(HNS_ST_CATCH_REWARD) / progress;
            ST(HNS_ST_COLLISION_REWARD) = ST(HNS_ST_COLLISION_REWARD) / progress;
            ST(HNS_ST_COLLISION_WALL) = ST(HNS_ST_COLLISION_WALL) / progress;
            ST(HNS_ST_COLLISION_DRONE) = ST(HNS_ST_COLLISION_DRONE) / progress;
            ST(HNS_ST_COLLISION_CYLINDER) = ST(HNS_ST_COLLISION_CYLINDER) / progress;
            ST(HNS_ST_SPEED_REWARD) = ST(HNS_ST_SPEED_REWARD) / progress;
        }
        ST(HNS_ST_RETURN) += sum_rew * iA;
#undef ST
        b.done[e] = (uint8_t)done;
        if constexpr (NT == 2) { if (b.detect) b.detect[e] = (uint8_t)((det_any ? 1 : 0) | (det_any1 ? 2 : 0)); }   // bit k: evader k detected
        else { if (b.detect) b.detect[e] = (uint8_t)det_any; }
        b.progress[e] = progress;
        if (!LAB(LAB_NOSTORE | LAB_NOST_STATS)) {
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i) {
#ifdef HNS_ST_STATS_PLAIN
                b.stats[(size_t)i * E + e] = st[i];
#else
                st_f1(b.stats + (size_t)i * E + e, st[i]);
#endif
            }
        }
    }
    prof_mark(p.prof, 6);
    if (KM == kMaxK && !full) { // ragged last tile: the k-nearest rows were staged per workgroup, store the slice now
        __syncthreads();
        if (!LAB(LAB_NOSTORE | LAB_NOST_OCYL)) coop_s2g<T>(b.obs_cylinders + (size_t)e0 * A * K * 5, sOCyl, nenv * A * K * 5);
    }
    prof_mark(p.prof, 7);
    prof_mark(p.prof, 15);
}

// =================================================================================================
// The fused step kernel, third design (one evader, whole 64-env tiles): no workgroup barrier in front of the
// controller, loads issued in the order they are needed.
// =================================================================================================
// Wave-specialised step kernel (full tiles, one evader).  What the profiles of the first design showed: a launch is one
// residency round, so the load burst, each wave's serial instruction stream and the store drain were paid one after the other.  Here
//   * a pursuer wave needs nobody else's data for the controller: it loads ITS OWN 64 rigid-state rows (one contiguous
//     3.3 KB slice) through its private LDS slab, so the controller starts as soon as the first-issued loads have landed;
//   * the env wave owns everything about the evader: it fetches its envs' cylinders and evader position itself, stages
//     the cylinders for phase 3 and runs the potential field while the pursuer waves run controller and integration;
//   * pursuer <-> pursuer and pursuer <-> evader exchange goes through small published records (position at t, thrust
//     vector, position at t+1, line-of-sight flag), three workgroup barriers in all (six before);
//   * every store is a whole-line store from a wave-private slab.
// Arithmetic, evaluation order and results are those of hns_step_kernel (bit-identical; tests/test_hip_parity.py).
#ifndef HNS_ENV_PRIO
#define HNS_ENV_PRIO 2
#endif
constexpr int kPub = 11;  // published per pursuer: position at t (3), thrust vector (3), position at t+1 (3), 1 / (|thrust| + 1e-6); odd stride
struct LdsV3 { int slab, slab_stride, pub, cyl, cyl_stride, tp, red, envout, term, total; };
constexpr int kTermStride = 2 * HNS_MAX_CYLINDERS + 1;   // per env: (tx, ty) of every cylinder's push on the second evader; odd stride
__host__ __device__ inline LdsV3 lds_layout_v3(int A, int C, int K, int NT = 1) {
    LdsV3 L;
    int o = 0;
    L.slab_stride = slab_floats(A, K, NT);
    if (L.slab_stride < 64 * 13 + 4) L.slab_stride = r4(64 * 13 + 4);
    L.slab = o;  o += A * L.slab_stride;
    L.pub = o;   o += r4(kEPB * A * kPub);
    L.cyl_stride = (3 * C) | 1;
    L.cyl = o;   o += r4(kEPB * L.cyl_stride);
    L.tp = o;    o += r4(kEPB * (3 * NT + 1));               // evader(s) at t+1 ([64][3 NT], contiguous: stored as one slice) + the step counter
    L.red = o;   o += r4(kEPB * A * red_stride(NT));
    L.envout = o; o += r4(kEPB * (A > 3 * NT ? A : 3 * NT)); // the env wave's own staging: evader velocity [64,3 NT], rewards [64,A]
    L.term = o;  if (NT == 2) o += r4(kEPB * kTermStride);    // two evaders: the pursuer lanes' share of the evader policy (below)
    L.total = o;
    return L;
}

// env wave: `n` floats that sit contiguously in LDS -> one contiguous, 16-byte aligned slice of global memory, 16 B per lane
// (n % 4 == 0), write-through.  Per-lane 4-byte stores at a 12-byte stride would each be a partial-line write.
HNS_DEV void env_store_slice(const float *lds, float *g, int n, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n / 4; i += 64) st_f4(reinterpret_cast<float4 *>(g) + i, reinterpret_cast<const float4 *>(lds)[i]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Relative to the third design (same structure) the env wave — ONE wave, issuing one instruction every 5-7 cycles — relieved of most of its serial
// work (tools/phase_timeline.py: in the third design 5.8 us of the 14.5 us of a workgroup's life were env-wave work that the pursuer
// waves waited for):
//   * the line of sight evader -> pursuer at t (:1080) is not evaluated at all: it is the test the previous step (or the reset) ran on
//     the same positions for the observation (:786), carried in the fourth column of pid_last_rate (hns.h);
//   * the pursuers publish their reward terms BEFORE they build and store their observation rows, so the env wave's reductions, reward,
//     statistics and done run beside those stores instead of behind them; the detection mask of the evader's relative position
//     (:791-794) is applied by the pursuers themselves (was: the env wave patched the stored rows);
//   * everything but the pointers behind the first loads comes from a device-resident block through the scalar cache (StepArgs).
// Three workgroup barriers.
template <int A, int NT, bool PROF>
__global__ __launch_bounds__(Geo<A>::T, NT == 2 ? 4 : 1) void hns_step_v4_kernel(const StepArgs ka) {
    // the block behind `rest` is never written while the kernel runs: read it as constant memory (scalar loads, placed like kernel-argument loads)
    typedef const Params __attribute__((address_space(4))) ParamsC;
    ParamsC &p = *(ParamsC *)ka.rest;
    constexpr int NA = Geo<A>::NA, SD = NT == 2 ? 24 : HNS_SELF_DIM, kRedS = red_stride(NT), T3 = 3 * NT;
    extern __shared__ __align__(16) float smem[];
    const auto &c = p.cfg;
    const auto &b = p.buf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int e0 = blockIdx.x * kEPB;
    // NOTHING that reads the block (`p`, `c`, `b`) may precede a wave's first global loads: those scalar loads are cold, and a wait for
    // them in front of the vector loads is what a device-resident configuration used to cost (+1.8 us).
    if (tid < NA) {
        // ================================= pursuer waves ==================================================
        const int le = tid / A, a = tid - le * A;
        const unsigned ia = (unsigned)e0 * A + tid;
        // loads, first needed first: action, previous action, the wave's 64 rigid-state rows, PID state, throttle
        const float4 act4 = reinterpret_cast<const float4 *>(ka.action)[ia];
        float4 prev4 = reinterpret_cast<const float4 *>(ka.prev_action)[ia];
        constexpr int N4 = 64 * 13 / 4;                   // 208 float4 pieces per wave
        const float4 *rows4 = reinterpret_cast<const float4 *>(ka.drone_state + ((size_t)e0 * A + (tid & ~63)) * 13) + lane;
        static_assert(N4 > 192 && N4 <= 256, "three full passes and a partial one");
        const float4 rr0 = rows4[0], rr1 = rows4[64], rr2 = rows4[128];      // (named values: an array with a predicated element went to scratch)
        float4 rr3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < N4 - 192) rr3 = rows4[192];
        float4 integ4 = reinterpret_cast<const float4 *>(ka.pid_integ)[ia];
        float4 last4 = reinterpret_cast<const float4 *>(ka.pid_last_rate)[ia];
        float4 thr4 = reinterpret_cast<const float4 *>(ka.throttle)[ia];
        // (two evaders, see below: this wave's cylinder passes and this pursuer's cylinders — through the kernel argument, issued with the
        //  first loads; the count of cylinders is only known from the parameter block, so the loads cover HNS_MAX_CYLINDERS slots of the
        //  workgroup's OWN range and are clamped to it)
        constexpr int kStage = NT == 2 ? (3 * HNS_MAX_CYLINDERS + A - 1) / A : 1, kOwnCyl = NT == 2 ? (HNS_MAX_CYLINDERS + A - 1) / A : 1;
        float stage_v[kStage], own_c[kOwnCyl][3];
        if constexpr (NT == 2) {
            const uintptr_t cw = reinterpret_cast<uintptr_t>(ka.cylinders);
            const int Cq = (int)(cw & 15) + 1;
            const float *cyl0 = reinterpret_cast<const float *>(cw & ~(uintptr_t)15);
            const float *gc = cyl0 + (size_t)e0 * Cq * 3 + lane;
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int pass = (tid >> 6) + i * A;
                stage_v[i] = pass < 3 * Cq ? gc[pass * 64] : 0.0f;
            }
            const float *gcy = cyl0 + (size_t)(e0 + le) * Cq * 3;
#pragma unroll
            for (int i = 0; i < kOwnCyl; ++i) {
                const int k = a + i * A;
                const int kc = k < Cq ? k : 0;
                own_c[i][0] = gcy[3 * kc]; own_c[i][1] = gcy[3 * kc + 1]; own_c[i][2] = gcy[3 * kc + 2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = c.num_cylinders, K = c.obs_max_cylinder;
        const bool with_state = c.write_critic_state && b.state_drones != nullptr;
        const LdsV3 L = lds_layout_v3(A, C, K, NT);
        float *sPub = smem + L.pub, *sCyl = smem + L.cyl, *sTp = smem + L.tp, *sRed = smem + L.red;
        float *slab = smem + L.slab + (tid >> 6) * L.slab_stride;
        // Two evaders: ONE env wave running both potential fields (2 x (C cylinders + A pursuers) terms, one instruction per 5-7
        // cycles) and staging 3 C x 64 cylinder values kept the A pursuer waves waiting for ~9 k of a workgroup's 44 k cycles
        // (tools/phase_profile.py --targets=2, round 3).  The pursuer lanes take over what does not need the env wave's order:
        //   * wave w stages the cylinder passes w, w + A, ... for phase 3 (the env wave reads ITS cylinders straight from memory);
        //   * every pursuer evaluates its own push on both evaders (hideandseek.py:1074-1088) and pursuer a the second evader's
        //     cylinder terms of cylinders a, a + A, ... (:1114-1136); the env wave only adds them up, in the reference's order.
        // Two evaders: ONE env wave running both potential fields (2 x (C cylinders + A pursuers) terms, one instruction per 5-7
        // cycles) and staging 3 C x 64 cylinder values kept the A pursuer waves waiting for ~9 k of a workgroup's 44 k cycles
        // (tools/phase_profile.py --targets=2, round 3).  The pursuer lanes take over what does not need the env wave's order:
        //   * wave w stages the cylinder passes w, w + A, ... for phase 3 (the env wave reads ITS cylinders straight from memory);
        //   * every pursuer evaluates its own push on both evaders (hideandseek.py:1074-1088) and pursuer a the second evader's
        //     cylinder terms of cylinders a, a + A, ... (:1114-1136); the env wave only adds them up, in the reference's order.
        V3 etp0 = {0.f, 0.f, 0.f}, etp1 = {0.f, 0.f, 0.f};
        if constexpr (NT == 2) {
            const float *gt = b.target_pos + (size_t)(e0 + le) * T3;
            etp0 = V3{gt[0], gt[1], gt[2]};
            etp1 = V3{gt[3], gt[4], gt[5]};
        }
        const float4 ta = d_action_tanh(act4);           // needs the action only: evaluated while the rest is in flight
        // own rows through the private slab
        {
            float4 *s4 = reinterpret_cast<float4 *>(slab) + lane;
            s4[0] = rr0; s4[64] = rr1; s4[128] = rr2;
            if (lane < N4 - 192) s4[192] = rr3;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        Rigid s;
        load_rigid(slab + lane * 13, s);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (PROF) prof_mark(p.prof, 1);
        // ---- phase 1: controller, rotors, thrust vector (A1-A3) ----
        // line of sight evader -> this pursuer at t (:1080): positions, evader and cylinders are those the previous step (or the reset)
        // evaluated it on for the observation, so that result is carried in the spare fourth column of the controller record
        const float los_t = last4.w;
        float cmd[4], thr_diff, aerr, thrust[4], moment[4];
        float ctbr4[4], trate[3];
        d_ctbr_pid_squashed(c, ta, s.q, s.ang, prev4, integ4, last4, cmd, aerr, ctbr4, trate);
        if (b.ctbr) reinterpret_cast<float4 *>(b.ctbr)[ia] = make_float4(ctbr4[0], ctbr4[1], ctbr4[2], ctbr4[3]);           // transforms.py:456
        if (b.target_rate) reinterpret_cast<float4 *>(b.target_rate)[ia] = make_float4(trate[0], trate[1], trate[2], 0.0f);  // :457
        d_rotor(c, cmd, thr4, thrust, moment, thr_diff);
        const float ts = ((thrust[0] + thrust[1]) + thrust[2]) + thrust[3];
        const V3 tw = d_quat_rot_z(s.q, ts);                                        // multirotor.py:491
        const float inv_ntw = d_downwash_inv_norm(tw);
        {
            float *pub = sPub + tid * kPub;
            pub[0] = s.pos.x; pub[1] = s.pos.y; pub[2] = s.pos.z;
            pub[3] = tw.x; pub[4] = tw.y; pub[5] = tw.z;
            pub[9] = inv_ntw;
            pub[10] = los_t;
            float *red = sRed + tid * kRedS;
            red[R_AERR] = aerr; red[R_TD] = thr_diff;
            if constexpr (NT == 2) {
                const int los = (int)los_t;                                          // bit k: line of sight to evader k blocked at t
                const V3 f0 = d_prey_pursuer_term(c, s.pos, etp0, (los & 1) != 0);
                const V3 f1 = d_prey_pursuer_term(c, s.pos, etp1, (los & 2) != 0);
                red[R_FX] = f0.x; red[R_FY] = f0.y; red[R_FZ] = f0.z;
                red[R_F1X] = f1.x; red[R_F1X + 1] = f1.y; red[R_F1X + 2] = f1.z;
                float *term = smem + L.term + le * kTermStride;
#pragma unroll
                for (int i = 0; i < kOwnCyl; ++i) {
                    const int k = a + i * A;
                    if (k < C) {
                        float tx, ty;
                        d_prey_cylinder_term(c, etp1, own_c[i][0], own_c[i][1], own_c[i][2], tx, ty);
                        term[2 * k] = tx; term[2 * k + 1] = ty;
                    }
                }
                const int c3 = 3 * C;
#pragma unroll
                for (int i = 0; i < kStage; ++i) {
                    const int pass = (tid >> 6) + i * A;
                    if (pass < c3) {
                        const int idx = pass * 64 + lane;
                        const int row = (int)__umulhi((unsigned)idx, p.cyl_magic), col = idx - row * c3;
                        sCyl[row * L.cyl_stride + col] = stage_v[i];
                    }
                }
            }
        }
        if constexpr (PROF) prof_mark(p.prof, 2);
        __syncthreads();                                                            // barrier 1
        if constexpr (PROF) prof_mark(p.prof, 12);
        // ---- phase 2: downwash, torques, integration (A4, A5) ----
        V3 fdw = {0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < A - 1; ++o) {
            const int j = o + (o >= a ? 1 : 0);
            const float *pj = sPub + (le * A + j) * kPub;
            const V3 posj = {pj[0], pj[1], pj[2]}, twj = {pj[3], pj[4], pj[5]};
            const V3 fj = d_downwash_pair(s.pos, posj, twj, pj[9]);
            fdw.x = (o == 0) ? fj.x : fdw.x + fj.x;
            fdw.y = (o == 0) ? fj.y : fdw.y + fj.y;
            fdw.z = (o == 0) ? fj.z : fdw.z + fj.z;
        }
        const V3 fw = {tw.x + fdw.x, tw.y + fdw.y, tw.z + fdw.z};
        V3 tb;
        tb.x = ((c.rotor_py[0] * thrust[0] + c.rotor_py[1] * thrust[1]) + c.rotor_py[2] * thrust[2]) + c.rotor_py[3] * thrust[3];
        tb.y = -(((c.rotor_px[0] * thrust[0] + c.rotor_px[1] * thrust[1]) + c.rotor_px[2] * thrust[2]) + c.rotor_px[3] * thrust[3]);
        tb.z = ((moment[0] + moment[1]) + moment[2]) + moment[3];
        d_integrate(c, s, fw, tb);
        flag_nonfinite(b.nonfinite, rigid_not_finite(s), 1u);
        {
            float *pub = sPub + tid * kPub;
            pub[6] = s.pos.x; pub[7] = s.pos.y; pub[8] = s.pos.z;
        }
        // controller / rotor state: plain stores (they are early: write-through here stalls the wave, measured +0.45 us)
        if (LAB(LAB_NOSTORE | LAB_NOST_REC)) {
        } else
#ifdef HNS_V3_REC_SC1
        {
        st_f4(reinterpret_cast<float4 *>(b.throttle) + ia, thr4);
        st_f4(reinterpret_cast<float4 *>(b.pid_integ) + ia, integ4);
        st_f4(reinterpret_cast<float4 *>(b.prev_action) + ia, prev4);
        st_f1(b.action_error + ia, aerr);
        }
#else
        {
        reinterpret_cast<float4 *>(b.throttle)[ia] = thr4;
        reinterpret_cast<float4 *>(b.pid_integ)[ia] = integ4;
        reinterpret_cast<float4 *>(b.prev_action)[ia] = prev4;
        b.action_error[ia] = aerr;
        }
#endif
        if (!LAB(LAB_NOSTORE | LAB_NOST_DS)) {   // S_{t+1}: the wave's 64 rows back through the slab, one contiguous slice
            const float row[13] = {s.pos.x, s.pos.y, s.pos.z, s.q.w, s.q.x, s.q.y, s.q.z, s.lin.x, s.lin.y, s.lin.z, s.ang.x, s.ang.y, s.ang.z};
            wave_store_rows<13>(slab, b.drone_state + ((size_t)e0 * A + (tid & ~63)) * 13, row, lane);
        }
        if constexpr (PROF) prof_mark(p.prof, 3);
        __syncthreads();                                                            // barrier 2
        if constexpr (PROF) prof_mark(p.prof, 8);
        // ---- phase 3a: distance and line of sight to the evader, the k nearest cylinders, per-pursuer reward terms on S_{t+1} ----
        const float progress = sTp[kEPB * T3 + le];                                 // progress + 1, published by the env wave
        const V3 tp = {sTp[le * T3], sTp[le * T3 + 1], sTp[le * T3 + 2]};
        V3 tpB = tp;                                                                // second evader (extension, include/hns.h)
        if constexpr (NT == 2) tpB = V3{sTp[le * T3 + 3], sTp[le * T3 + 4], sTp[le * T3 + 5]};
        const float *cyl = sCyl + le * L.cyl_stride;
        const float rtx = s.pos.x - tp.x, rty = s.pos.y - tp.y, rtz = s.pos.z - tp.z;
        const float d = d_norm3(rtx, rty, rtz);                                     // |evader - pursuer| (hideandseek.py:921, :780)
        int knn_idx[kMaxK + 1];
        bool knn_masked[kMaxK];
        bool blocked, blockedB;
#ifdef HNS_LAB
        if (LAB(LAB_NOLOS2)) { cylinder_pass<NT, false>(c, C, K, s.pos, tp, tpB, cyl, knn_idx, blocked, blockedB); } else
#endif
        cylinder_pass<NT, true>(c, C, K, s.pos, tp, tpB, cyl, knn_idx, blocked, blockedB);
        const bool det = (d < c.drone_detect_radius) && !blocked;                   // :787-789
        last4.w = (float)((blocked ? 1 : 0) + (NT == 2 && blockedB ? 2 : 0));       // = the next step's line of sight at ITS t
        if (!LAB(LAB_NOSTORE | LAB_NOST_REC)) reinterpret_cast<float4 *>(b.pid_last_rate)[ia] = last4;
#pragma unroll
        for (int sidx = 0; sidx < kMaxK; ++sidx) knn_masked[sidx] = (sidx < K) ? cyl[3 * knn_idx[sidx] + 2] < 0.0f : false;   // :759,775-778
        if constexpr (PROF) prof_mark(p.prof, 9);
        bool cap_ok = (d < c.catch_radius) && !blocked, all_blk = blocked, detB = false;   // hideandseek.py:919-995
        float dn = d;
        float r1x = 0.f, r1y = 0.f, r1z = 0.f;
        if constexpr (NT == 2) {
            // extension: distance term to the NEAREST evader, capture of ANY evader, `blocked` = no line of sight to either
            r1x = s.pos.x - tpB.x; r1y = s.pos.y - tpB.y; r1z = s.pos.z - tpB.z;
            const float d1 = d_norm3(r1x, r1y, r1z);
            detB = (d1 < c.drone_detect_radius) && !blockedB;
            cap_ok = cap_ok || ((d1 < c.catch_radius) && !blockedB);
            all_blk = blocked && blockedB;
            dn = d1 < d ? d1 : d;
        }
        const float act = (dn > c.catch_radius) ? 1.0f : 0.0f;
        const float dist_rew = (-c.dist_reward_coef * dn) * act;
        // Threshold tests on norms: RN(sqrt(x)) compared with a limit is decided on x itself unless x lies within 2^-19 of
        // the squared limit; only then the correctly rounded square root is taken (same booleans as the plain form).
        bool fast = false;
        {
            const float sp2 = HNS_FMA(s.lin.z, s.lin.z, HNS_FMA(s.lin.y, s.lin.y, s.lin.x * s.lin.x));
            const float v2 = c.v_drone * c.v_drone;
            fast = sp2 > v2 * 1.00000190734863f;
            if (!fast && !(sp2 < v2 * 0.99999809265137f)) fast = __builtin_sqrtf(sp2) > c.v_drone;
        }
        const float speed_rew = -c.speed_coef * (fast ? 1.0f : 0.0f);
        float cc = 0.f, cd = 0.f;
        const float rc = c.cylinder_size + c.collision_radius, rc2 = rc * rc;
#pragma unroll
        for (int sidx = 0; sidx < kMaxK; ++sidx) {
            if (sidx < K) {
                const float *cy = cyl + 3 * knn_idx[sidx];
                const float rx = s.pos.x - cy[0], ry = s.pos.y - cy[1];
                const float s2 = HNS_FMA(ry, ry, rx * rx);
                bool h = s2 < rc2 * 0.99999618530273f;                               // 1 - 2^-18: covers the roundings of rc, dxy - size
                if (!h && !(s2 > rc2 * 1.00000381469727f)) h = (__builtin_sqrtf(s2) - c.cylinder_size) < c.collision_radius;
                float hit = h ? 1.0f : 0.0f;
                if (knn_masked[sidx]) hit = 0.0f;
                cc = (sidx == 0) ? hit : cc + hit;
            }
        }
        float cr = -c.collision_coef * cc;
        const float dd2 = c.coll_drone_dist * c.coll_drone_dist;
#pragma unroll
        for (int o = 0; o < A - 1; ++o) {
            const int j = o + (o >= a ? 1 : 0);
            const float *rj = sPub + (le * A + j) * kPub + 6;
            const float ex = s.pos.x - rj[0], ey = s.pos.y - rj[1], ez = s.pos.z - rj[2];
            const float s3 = HNS_FMA(ez, ez, HNS_FMA(ey, ey, ex * ex));
            bool h = s3 < dd2 * 0.99999809265137f;
            if (!h && !(s3 > dd2 * 1.00000190734863f)) h = __builtin_sqrtf(s3) < c.coll_drone_dist;
            const float hit = h ? 1.0f : 0.0f;
            cd = (o == 0) ? hit : cd + hit;
        }
        cr = cr + -c.collision_coef * cd;
        const float cw = ((s.pos.z > c.max_height) ? 1.0f : 0.0f) + ((HNS_FMA(s.pos.y, s.pos.y, s.pos.x * s.pos.x) > c.arena_sq) ? 1.0f : 0.0f);
        cr = cr + -c.collision_coef * cw;
        float sm = 0.0f;
        if (c.use_deployment) sm = c.smoothness_coef * d_expf(-aerr);
        {
            float *red = sRed + tid * kRedS;
            red[R_DIST] = dist_rew; red[R_SPEED] = speed_rew; red[R_CC] = cc; red[R_CD] = cd; red[R_CW] = cw;
            red[R_COLL] = cr; red[R_SMOOTH] = sm;
            red[R_FLAGS] = __int_as_float((cap_ok ? F_CAP : 0) | (all_blk ? F_BLOCKED : 0) | (det ? F_DET : 0) | (detB ? F_DET1 : 0));
        }
        if constexpr (PROF) prof_mark(p.prof, 4);
        __syncthreads();                                                            // barrier 3
        if constexpr (PROF) prof_mark(p.prof, 5);
        // ---- phase 3c: the observation rows, beside the env wave's reductions (A8 hideandseek.py:741-886) ----
        bool det_any = false, det_any1 = false;                                     // :787-794: any pursuer sees the evader
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const int fl = __float_as_int(sRed[(le * A + j) * kRedS + R_FLAGS]);
            det_any |= (fl & F_DET) != 0;
            det_any1 |= (fl & F_DET1) != 0;
        }
        {
            const float t = progress * c.inv_max_episode_length;                  // :796
            const V3 heading = d_quat_rot_x(s.q), up = d_quat_rot_z(s.q, 1.0f);   // multirotor.py:613-614
            const float m = c.mask_value;
            float row[SD] = {det_any ? rtx : m, det_any ? rty : m, det_any ? rtz : m, s.q.w, s.q.x, s.q.y, s.q.z, s.lin.x, s.lin.y, s.lin.z,
                             heading.x, heading.y, heading.z, up.x, up.y, up.z, t, t, t, t};                  // :856-863
            if constexpr (NT == 2) { row[20] = det_any1 ? r1x : m; row[21] = det_any1 ? r1y : m; row[22] = det_any1 ? r1z : m; row[23] = 0.0f; }
            if (!LAB(LAB_NOSTORE | LAB_NOST_SELF)) wave_store_rows<SD, slab_rows(A)>(slab, b.obs_self + ((size_t)e0 * A + (tid & ~63)) * SD, row, lane);
            if (with_state && !LAB(LAB_NOSTORE | LAB_NOST_SELF)) {                  // :871-886 (never masked)
                float rs[SD];
#pragma unroll
                for (int i = 0; i < SD; ++i) rs[i] = row[i];
                rs[0] = rtx; rs[1] = rty; rs[2] = rtz;
                if constexpr (NT == 2) { rs[20] = r1x; rs[21] = r1y; rs[22] = r1z; }
                wave_store_rows<SD, slab_rows(A)>(slab, b.state_drones + ((size_t)e0 * A + (tid & ~63)) * SD, rs, lane);
            }
        }
        if constexpr (A > 1) {                                                      // p_i - p_j, j != i ascending (:750-751)
            float o[(A > 1 ? A - 1 : 1) * 3];
#pragma unroll
            for (int w = 0; w < A - 1; ++w) {
                const int j = w + (w >= a ? 1 : 0);
                const float *rj = sPub + (le * A + j) * kPub + 6;
                o[3 * w] = s.pos.x - rj[0]; o[3 * w + 1] = s.pos.y - rj[1]; o[3 * w + 2] = s.pos.z - rj[2];
            }
            if (!LAB(LAB_NOSTORE | LAB_NOST_OTH))
                wave_store_rows<(A > 1 ? A - 1 : 1) * 3, slab_rows(A)>(slab, b.obs_others + ((size_t)e0 * A + (tid & ~63)) * (A - 1) * 3, o, lane);
        }
        if (!LAB(LAB_NOSTORE | LAB_NOST_OCYL)) {                                     // the k nearest cylinders (:767-778)
            float krow[kMaxK * 5];
            const float mv = c.mask_value, ch = c.cylinder_height, cs = c.cylinder_size;   // values, not lvalues (see d_rotor)
#pragma unroll
            for (int sidx = 0; sidx < kMaxK; ++sidx) {
                const float *cc = cyl + 3 * ((sidx < K) ? knn_idx[sidx] : 0);
                const bool masked = knn_masked[sidx];
                const float rx = s.pos.x - cc[0], ry = s.pos.y - cc[1], rz = s.pos.z - cc[2];   // loaded whether masked or not: no branch per value
                krow[sidx * 5] = masked ? mv : rx;
                krow[sidx * 5 + 1] = masked ? mv : ry;
                krow[sidx * 5 + 2] = masked ? mv : rz;
                krow[sidx * 5 + 3] = masked ? mv : ch;
                krow[sidx * 5 + 4] = masked ? mv : cs;
            }
            float *g = b.obs_cylinders + ((size_t)e0 * A + (tid & ~63)) * K * 5;
            if (K == 3) {
                float r[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) r[i] = krow[i];
                wave_store_rows<15, slab_rows(A)>(slab, g, r, lane);
            } else if (K == 4) {
                wave_store_rows<20, slab_rows(A)>(slab, g, krow, lane);
            } else if (K == 2) {
                float r[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) r[i] = krow[i];
                wave_store_rows<10, slab_rows(A)>(slab, g, r, lane);
            } else {
                float r[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) r[i] = krow[i];
                wave_store_rows<5, slab_rows(A)>(slab, g, r, lane);
            }
        }
        if constexpr (PROF) prof_mark(p.prof, 6);
    } else {
        // ================================= env wave: lane <-> env ========================================
        if (HNS_ENV_PRIO) __builtin_amdgcn_s_setprio(HNS_ENV_PRIO);   // one wave in four, but every barrier of its workgroup waits for it
        const int le = lane, e = e0 + le;
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = c.num_cylinders, K = c.obs_max_cylinder, E = c.num_envs;
        const LdsV3 L = lds_layout_v3(A, C, K, NT);
        float *sPub = smem + L.pub, *sCyl = smem + L.cyl, *sTp = smem + L.tp, *sRed = smem + L.red, *sEnvOut = smem + L.envout;
        float *cylw = sCyl + le * L.cyl_stride;
        // the evader at t
        const float *gt = b.target_pos + (size_t)e * T3;
        const V3 tp0 = {gt[0], gt[1], gt[2]};
        V3 tp1 = tp0;
        if constexpr (NT == 2) tp1 = V3{gt[3], gt[4], gt[5]};
        float progress = b.progress[e];
        if constexpr (NT == 1) {   // this workgroup's cylinders are one contiguous slice [64][3C]: coalesced 4-byte loads (lane <-> consecutive floats), scattered
            // into rows of odd stride (lane = env reads its row conflict-free); index / 3C by multiply-high.  Eight cylinders (24 passes)
            // at a time with every load issued before the first LDS write: one memory round trip per chunk.
            const float *gc = b.cylinders + (size_t)e0 * C * 3 + lane;
            const int c3 = 3 * C;                                   // = the number of 64-float passes
            int i0 = 0;
            for (; i0 + 24 <= c3; i0 += 24) {
                float cv[24];
#pragma unroll
                for (int i = 0; i < 24; ++i) cv[i] = gc[(i0 + i) * 64];
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    const int idx = (i0 + i) * 64 + lane;
                    const int row = (int)__umulhi((unsigned)idx, p.cyl_magic), col = idx - row * c3;
                    sCyl[row * L.cyl_stride + col] = cv[i];
                }
            }
            for (; i0 < c3; i0 += 3) {                              // the cylinders beyond a multiple of eight
                float cv[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) cv[i] = gc[(i0 + i) * 64];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int idx = (i0 + i) * 64 + lane;
                    const int row = (int)__umulhi((unsigned)idx, p.cyl_magic), col = idx - row * c3;
                    sCyl[row * L.cyl_stride + col] = cv[i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // (two evaders: the pursuer waves stage the cylinders for phase 3; this wave reads its envs' cylinders from memory below)
        progress += 1.0f;                                                           // isaac_env.py:236
        sTp[kEPB * T3 + le] = progress;
        if constexpr (PROF) prof_mark(p.prof, 1);
        float st[HNS_NUM_STATS];                  // the statistics rows of these envs: needed behind barrier 1 (not earlier: the first microseconds
        if constexpr (NT == 1) {                  // of the launch are HBM-bound and these 6 MB are not on the critical path)
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i) st[i] = b.stats[(size_t)i * E + e];
        }
        // A6: arena + cylinder terms of the potential field (hideandseek.py:1090-1136)
        bool out_of_arena = false;
        const V3 Fenv = d_prey_arena_term(c, tp0, out_of_arena);
        float fcx = 0.f, fcy = 0.f;
        if constexpr (NT == 2) {
            // this wave's own cylinders straight from memory: lane = env reads its row of 3 C floats, 16 bytes at a time when the row is made of whole
            // quads (every lane of a load instruction in another cache line, but 12 instructions instead of 48)
            const float *cylg = b.cylinders + (size_t)e * C * 3;
            if ((C & 3) == 0) {
                const float4 *g4 = reinterpret_cast<const float4 *>(cylg);
                for (int k0 = 0; k0 < C; k0 += 4) {
                    const float4 q0 = g4[3 * (k0 >> 2)], q1 = g4[3 * (k0 >> 2) + 1], q2 = g4[3 * (k0 >> 2) + 2];
                    const float v[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float tx, ty;
                        d_prey_cylinder_term(c, tp0, v[3 * i], v[3 * i + 1], v[3 * i + 2], tx, ty);
                        fcx += tx;
                        fcy += ty;
                    }
                }
            } else {
                for (int k = 0; k < C; ++k) {
                    float tx, ty;
                    d_prey_cylinder_term(c, tp0, cylg[3 * k], cylg[3 * k + 1], cylg[3 * k + 2], tx, ty);
                    fcx += tx;
                    fcy += ty;
                }
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < C; ++k) {
                float tx, ty;
                d_prey_cylinder_term(c, tp0, cylw[3 * k], cylw[3 * k + 1], cylw[3 * k + 2], tx, ty);
                fcx += tx;
                fcy += ty;
            }
        }
        V3 Fenv1 = {0.f, 0.f, 0.f};
        float gcx = 0.f, gcy = 0.f;
        if constexpr (NT == 2) {               // each evader runs the potential field on its own (they ignore each other)
            bool out1 = false;
            Fenv1 = d_prey_arena_term(c, tp1, out1);
            out_of_arena = out_of_arena || out1;               // (its cylinder terms come from the pursuer lanes, summed behind barrier 1)
            // the statistics rows only now: loads return in order, and the cylinder rows above must not queue behind 6 MB from HBM
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i) st[i] = b.stats[(size_t)i * E + e];
        }
        if constexpr (PROF) prof_mark(p.prof, 2);
        __syncthreads();                                                            // barrier 1: positions at t, line-of-sight flags, action errors
        if constexpr (PROF) prof_mark(p.prof, 12);
        // the pursuers' pushes (hideandseek.py:1074-1088), ascending; then arena, then cylinders
        V3 F = {0.f, 0.f, 0.f}, G = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < A; ++j) {
            if constexpr (NT == 2) {                                                // evaluated by the pursuer's own lane, same arithmetic
                const float *red = sRed + (le * A + j) * kRedS;
                F.x = (j == 0) ? red[R_FX] : F.x + red[R_FX];
                F.y = (j == 0) ? red[R_FY] : F.y + red[R_FY];
                F.z = (j == 0) ? red[R_FZ] : F.z + red[R_FZ];
                G.x = (j == 0) ? red[R_F1X] : G.x + red[R_F1X];
                G.y = (j == 0) ? red[R_F1X + 1] : G.y + red[R_F1X + 1];
                G.z = (j == 0) ? red[R_F1X + 2] : G.z + red[R_F1X + 2];
            } else {
                const float *pj = sPub + (le * A + j) * kPub;
                const V3 dp = {pj[0], pj[1], pj[2]};
                const int los = (int)pj[10];                                        // :1080, carried over from the previous step's observation (bit k: evader k)
                const V3 fp = d_prey_pursuer_term(c, dp, tp0, (los & 1) != 0);
                F.x = (j == 0) ? fp.x : F.x + fp.x;
                F.y = (j == 0) ? fp.y : F.y + fp.y;
                F.z = (j == 0) ? fp.z : F.z + fp.z;
            }
        }
        if constexpr (NT == 2) {
            const float *term = smem + L.term + le * kTermStride;
#pragma unroll 4
            for (int k = 0; k < C; ++k) {
                gcx += term[2 * k];
                gcy += term[2 * k + 1];
            }
        }
        F.x = F.x + Fenv.x; F.y = F.y + Fenv.y; F.z = F.z + Fenv.z;
        F.x = F.x + fcx; F.y = F.y + fcy; F.z = F.z + 0.0f;
        const V3 tvel = {(c.v_prey * F.x) / (__builtin_fabsf(F.x) + 1e-5f), (c.v_prey * F.y) / (__builtin_fabsf(F.y) + 1e-5f),
                         (c.v_prey * F.z) / (__builtin_fabsf(F.z) + 1e-5f)};        // per-axis speed (:741)
        const V3 tpn = {tp0.x + tvel.x * c.dt, tp0.y + tvel.y * c.dt, tp0.z + tvel.z * c.dt};
        sTp[le * T3] = tpn.x; sTp[le * T3 + 1] = tpn.y; sTp[le * T3 + 2] = tpn.z;
        { const float sf = (tpn.x + tpn.y) + tpn.z; flag_nonfinite(b.nonfinite, (sf - sf) != 0.0f, 2u); }
        V3 tvel1 = {0.f, 0.f, 0.f};
        if constexpr (NT == 2) {
            G.x = G.x + Fenv1.x; G.y = G.y + Fenv1.y; G.z = G.z + Fenv1.z;
            G.x = G.x + gcx; G.y = G.y + gcy; G.z = G.z + 0.0f;
            tvel1 = V3{(c.v_prey * G.x) / (__builtin_fabsf(G.x) + 1e-5f), (c.v_prey * G.y) / (__builtin_fabsf(G.y) + 1e-5f),
                       (c.v_prey * G.z) / (__builtin_fabsf(G.z) + 1e-5f)};
            const V3 tpn1 = {tp1.x + tvel1.x * c.dt, tp1.y + tvel1.y * c.dt, tp1.z + tvel1.z * c.dt};
            sTp[le * T3 + 3] = tpn1.x; sTp[le * T3 + 4] = tpn1.y; sTp[le * T3 + 5] = tpn1.z;
            { const float sf = (tpn1.x + tpn1.y) + tpn1.z; flag_nonfinite(b.nonfinite, (sf - sf) != 0.0f, 2u); }
        }
        if (!LAB(LAB_NOSTORE)) {            // [64,3 NT] slices, whole lines: the new position is already laid out in sTp
            sEnvOut[le * T3] = tvel.x; sEnvOut[le * T3 + 1] = tvel.y; sEnvOut[le * T3 + 2] = tvel.z;
            if constexpr (NT == 2) { sEnvOut[le * T3 + 3] = tvel1.x; sEnvOut[le * T3 + 4] = tvel1.y; sEnvOut[le * T3 + 5] = tvel1.z; }
            env_store_slice(sTp, b.target_pos + (size_t)e0 * T3, kEPB * T3, lane);
            env_store_slice(sEnvOut, b.target_vel + (size_t)e0 * T3, kEPB * T3, lane);
        }
        {   // statistics that only need phase-1 data (A10 hideandseek.py:731-733, :1097-1098, :996-997)
            float sum_ae = 0.f, sum_td = 0.f, max_td = 0.f;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float *red = sRed + (le * A + j) * kRedS;
                const float td = red[R_TD];
                sum_ae = (j == 0) ? red[R_AERR] : sum_ae + red[R_AERR];
                sum_td = (j == 0) ? td : sum_td + td;
                max_td = (j == 0) ? td : (td > max_td ? td : max_td);
            }
            const float mae = sum_ae * c.inv_num_agents;
            st[HNS_ST_ACTION_ERROR_ORDER1_MEAN] += mae;
            if (mae > st[HNS_ST_ACTION_ERROR_ORDER1_MAX]) st[HNS_ST_ACTION_ERROR_ORDER1_MAX] = mae;
            st[HNS_ST_OUT_OF_ARENA] = ((st[HNS_ST_OUT_OF_ARENA] != 0.0f) || out_of_arena) ? 1.0f : 0.0f;
            st[HNS_ST_SMOOTHNESS_COEF] = c.smoothness_coef;
            st[HNS_ST_SMOOTHNESS_MEAN] += sum_td * c.inv_num_agents;
            if (max_td > st[HNS_ST_SMOOTHNESS_MAX]) st[HNS_ST_SMOOTHNESS_MAX] = max_td;
        }
        if constexpr (PROF) prof_mark(p.prof, 3);
        __syncthreads();                                                            // barrier 2
        if constexpr (PROF) prof_mark(p.prof, 8);
        if constexpr (PROF) prof_mark(p.prof, 4);
        __syncthreads();                                                            // barrier 3: reward terms
        if constexpr (PROF) prof_mark(p.prof, 5);
        // ---- phase 3b: per-env reductions, reward, done, statistics (hideandseek.py:919-1065) ----
        const float iA = c.inv_num_agents;
        bool any_cap = false, all_blocked = true, any_coll = false, det_any = false, det_any1 = false;
        float sum_dist = 0, sum_speed = 0, sum_cc = 0, sum_cd = 0, sum_cw = 0, sum_coll = 0, sum_smooth = 0;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float *red = sRed + (le * A + j) * kRedS;
            const int fl = __float_as_int(red[R_FLAGS]);
            any_cap |= (fl & F_CAP) != 0;
            all_blocked &= (fl & F_BLOCKED) != 0;
            det_any |= (fl & F_DET) != 0;
            det_any1 |= (fl & F_DET1) != 0;
            any_coll |= red[R_COLL] < 0.0f;
            if (j == 0) {
                sum_dist = red[R_DIST]; sum_speed = red[R_SPEED]; sum_cc = red[R_CC]; sum_cd = red[R_CD]; sum_cw = red[R_CW];
                sum_coll = red[R_COLL]; sum_smooth = red[R_SMOOTH];
            } else {
                sum_dist += red[R_DIST]; sum_speed += red[R_SPEED]; sum_cc += red[R_CC]; sum_cd += red[R_CD]; sum_cw += red[R_CW];
                sum_coll += red[R_COLL]; sum_smooth += red[R_SMOOTH];
            }
        }
        const float detf = (det_any || (NT == 2 && det_any1)) ? 1.0f : 0.0f;
        const float detect_rew = c.detect_reward_coef * detf;
        const float catch_rew = c.catch_reward_coef * (any_cap ? 1.0f : 0.0f);
        float sum_rew = 0.f;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float *red = sRed + (le * A + j) * kRedS;
            const float r = ((((red[R_DIST] + detect_rew) + catch_rew) + red[R_COLL]) + red[R_SPEED]) + red[R_SMOOTH];
            sEnvOut[le * A + j] = r;
            sum_rew = (j == 0) ? r : sum_rew + r;
        }
        if (!LAB(LAB_NOSTORE)) env_store_slice(sEnvOut, b.reward + (size_t)e0 * A, kEPB * A, lane);
        flag_nonfinite(b.nonfinite, (sum_rew - sum_rew) != 0.0f, 4u);
#define ST(i) st[i]
        ST(HNS_ST_DISTANCE_REWARD) += sum_dist * iA;
        ST(HNS_ST_SUM_DETECT_STEP) += 1.0f * detf;
        float sdet = detect_rew, scat = catch_rew;
#pragma unroll
        for (int j = 1; j < A; ++j) { sdet += detect_rew; scat += catch_rew; }
        ST(HNS_ST_DETECT_REWARD) += sdet * iA;
        const bool capture_flag = catch_rew != 0.0f;                              // :945
        ST(HNS_ST_BLOCKED) += all_blocked ? 1.0f : 0.0f;
        ST(HNS_ST_SUCCESS) = (capture_flag || ST(HNS_ST_SUCCESS) != 0.0f) ? 1.0f : 0.0f;
        const float cur = (capture_flag ? 1.0f : 0.0f) * progress + (capture_flag ? 0.0f : 1.0f) * (float)c.max_episode_length;
        if (cur < ST(HNS_ST_FIRST_CAPTURE_STEP)) ST(HNS_ST_FIRST_CAPTURE_STEP) = cur;
        ST(HNS_ST_CATCH_REWARD) += scat * iA;
        ST(HNS_ST_SPEED_REWARD) += sum_speed * iA;
        ST(HNS_ST_COLLISION_CYLINDER) += sum_cc * iA;
        ST(HNS_ST_COLLISION_DRONE) += sum_cd * iA;
        ST(HNS_ST_COLLISION) += any_coll ? 1.0f : 0.0f;
        ST(HNS_ST_COLLISION_WALL) += sum_cw * iA;
        ST(HNS_ST_COLLISION_REWARD) += sum_coll * iA;
        ST(HNS_ST_SMOOTHNESS_REWARD) += sum_smooth * iA;
        const bool done = progress >= (float)c.max_episode_length;                // :1008-1010
        if (done) {                                                               // :1017-1056
            ST(HNS_ST_COLLISION) = ST(HNS_ST_COLLISION) / progress;
            ST(HNS_ST_ACTION_ERROR_ORDER1_MEAN) = ST(HNS_ST_ACTION_ERROR_ORDER1_MEAN) / progress;
            ST(HNS_ST_TARGET_PREDICTED_ERROR) = ST(HNS_ST_TARGET_PREDICTED_ERROR) / progress;
            ST(HNS_ST_SMOOTHNESS_MEAN) = ST(HNS_ST_SMOOTHNESS_MEAN) / progress;
            ST(HNS_ST_SMOOTHNESS_REWARD) = ST(HNS_ST_SMOOTHNESS_REWARD) / progress;
            ST(HNS_ST_DISTANCE_REWARD) = ST(HNS_ST_DISTANCE_REWARD) / progress;
            ST(HNS_ST_DETECT_REWARD) = ST(HNS_ST_DETECT_REWARD) / progress;
            ST(HNS_ST_CATCH_REWARD) = ST(HNS_ST_CATCH_REWARD) / progress;
            ST(HNS_ST_COLLISION_REWARD) = ST(HNS_ST_COLLISION_REWARD) / progress;
            ST(HNS_ST_COLLISION_WALL) = ST(HNS_ST_COLLISION_WALL) / progress;
            ST(HNS_ST_COLLISION_DRONE) = ST(HNS_ST_COLLISION_DRONE) / progress;
            ST(HNS_ST_COLLISION_CYLINDER) = ST(HNS_ST_COLLISION_CYLINDER) / progress;
            ST(HNS_ST_SPEED_REWARD) = ST(HNS_ST_SPEED_REWARD) / progress;
        }
        ST(HNS_ST_RETURN) += sum_rew * iA;
#undef ST
        b.done[e] = (uint8_t)done;
        if (b.detect) b.detect[e] = (uint8_t)((det_any ? 1 : 0) | (NT == 2 && det_any1 ? 2 : 0));      // bit k: evader k detected
        b.progress[e] = progress;
        if (!LAB(LAB_NOSTORE | LAB_NOST_STATS)) {
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i) st_f1(b.stats + (size_t)i * E + e, st[i]);
        }
        if constexpr (PROF) prof_mark(p.prof, 6);
    }
    if constexpr (PROF) prof_mark(p.prof, 7);
    if constexpr (PROF) prof_mark(p.prof, 15);
#ifdef HNS_LAB
    if (p.prof && lane == 0 && LAB(LAB_HWID)) {                   // where the hardware put this wave (tools/wave_placement.py)
        unsigned long long *pr = p.prof + (size_t)(blockIdx.x * (A + 1) + (tid >> 6)) * kProfSlots;
        pr[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        pr[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
        pr[12] = (unsigned long long)(tid >= NA);
    }
#endif
}

// Line-of-sight column of pid_last_rate from the state the buffers hold (hns_set_state: the uploaded record may come from anywhere).
// One thread per pursuer, cylinders straight from global memory; not on any hot path.
__global__ __launch_bounds__(256) void hns_refresh_los_kernel(const hns_cfg c, const hns_buffers b) {
    const int A = c.num_agents, C = c.num_cylinders, NT = c.num_targets == 2 ? 2 : 1;
    const int ia = blockIdx.x * 256 + threadIdx.x;
    if (ia >= c.num_envs * A) return;
    const int e = ia / A;
    const float *ds = b.drone_state + (size_t)ia * 13, *cyl = b.cylinders + (size_t)e * C * 3, *tp = b.target_pos + (size_t)e * 3 * NT;
    const V3 pos = {ds[0], ds[1], ds[2]};
    int flag = d_blocked(c, C, pos, V3{tp[0], tp[1], tp[2]}, cyl) ? 1 : 0;
    if (NT == 2) flag |= d_blocked(c, C, pos, V3{tp[3], tp[4], tp[5]}, cyl) ? 2 : 0;
    b.pid_last_rate[(size_t)ia * 4 + 3] = (float)flag;
}

// =================================================================================================
// Reset kernel (A11): hideandseek.py:576-723, multirotor.py:635-650 + the reset-time obs pass
// (isaac_env.py:221).  The env wave regenerates the state of the masked envs into LDS with a
// Philox stream, then the agent waves run the same agent_obs as the step kernel.
// =================================================================================================
template <int A, int NT, int KM = kMaxK>
__global__ __launch_bounds__(Geo<A>::T) void hns_reset_kernel(const Params p) {
    // KM > kMaxK: wide k-nearest selections, rows stored by their threads (see agent_obs)
    constexpr int T = Geo<A>::T, NA = Geo<A>::NA;
    constexpr int SD = NT == 2 ? 24 : HNS_SELF_DIM;
    extern __shared__ __align__(16) float smem[];
    __shared__ uint8_t sMask[kEPB];
    __shared__ uint8_t sDet[kEPB];
    __shared__ uint8_t sDet1[kEPB];
    const hns_cfg &c = p.cfg;
    const hns_buffers &b = p.buf;
    const int C = c.num_cylinders, K = c.obs_max_cylinder, E = c.num_envs, GN = c.grid_num;
    const bool with_state = c.write_critic_state && b.state_drones != nullptr;
    const Lds L = lds_layout(A, C, K, NT);
    float *sDS = smem + L.ds, *sCyl = smem + L.cyl, *sTp = smem + L.tp;
    float *sOCyl = smem + L.ocyl;
    uint8_t *sGrid = reinterpret_cast<uint8_t *>(smem + L.total);   // 64 x kGridStride B of grid scratch after the step layout

    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * kEPB;
    const int nenv = min(kEPB, E - e0);
    const bool env_wave = tid >= NA;
    const int le = env_wave ? tid - NA : tid / A;
    const int a = env_wave ? 0 : tid - le * A;
    const int e = e0 + le;
    const bool valid = le < nenv;
    if (tid < kEPB) {
        sMask[tid] = (tid < nenv) ? (p.reset_mask ? (p.reset_mask[e0 + tid] != 0) : 1) : 0;
        sDet[tid] = 0;
        sDet1[tid] = 0;
    }
    __syncthreads();
    const bool masked = valid && sMask[le];

    if (env_wave && valid) {
        // hideandseek.py:712 resets first_capture_step for ALL envs on any reset call
        b.stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * E + e] = (float)c.max_episode_length;
    }
    if (env_wave && masked) {
        Rng rng = {p.seed_lo, p.seed_hi, (uint32_t)(e + c.env_index_offset), p.epoch, 0u, {0u, 0u, 0u, 0u}, 0};
        float *ds = sDS + le * A * 13;
        float *tp = sTp + le * 3 * NT;
        float *cyl = sCyl + le * L.cyl_stride;
        // envgen (hideandseek_envgen.py:896-898): placement given by a task vector [drones | evader(s) | cylinders]
        const float *task = (p.tasks && e >= p.task_first) ? p.tasks + (size_t)e * (3 * A + 3 * NT + 3 * C) : nullptr;
        for (int j = 0; j < A; ++j) {
            float *d = ds + 13 * j;
            if (task) {
                d[0] = task[3 * j]; d[1] = task[3 * j + 1];
            } else if (c.init_mode == HNS_INIT_RANDOM) {
                d[0] = c.drone_xy_lo[0] + rng.uniform() * (c.drone_xy_hi[0] - c.drone_xy_lo[0]);
                d[1] = c.drone_xy_lo[1] + rng.uniform() * (c.drone_xy_hi[1] - c.drone_xy_lo[1]);
            } else {
                d[0] = c.fixed_drone_pos[j][0]; d[1] = c.fixed_drone_pos[j][1];
            }
            if (task) d[2] = task[3 * j + 2];
            else if (c.init_mode == HNS_INIT_SCENARIO) d[2] = c.fixed_drone_pos[j][2];
            else d[2] = c.z_lo + rng.uniform() * (c.z_hi - c.z_lo);
            float r0 = c.rpy_lo[0] + rng.uniform() * (c.rpy_hi[0] - c.rpy_lo[0]);
            float r1 = c.rpy_lo[1] + rng.uniform() * (c.rpy_hi[1] - c.rpy_lo[1]);
            float r2 = c.rpy_lo[2] + rng.uniform() * (c.rpy_hi[2] - c.rpy_lo[2]);
            Q4 q = d_euler_to_quat(r0, r1, r2);
            d[3] = q.w; d[4] = q.x; d[5] = q.y; d[6] = q.z;
            for (int i = 7; i < 13; ++i) d[i] = 0.0f;
            size_t ja = (size_t)e * A + j;
            float thr = c.hover_throttle;                                         // multirotor.py:647-648
            float ci = 0.5f * (c.max_thrust_ratio + (2.0f * (thr * thr) - 1.0f));
            float pa = ((ci + ci) + ci) + ci;
            reinterpret_cast<float4 *>(b.throttle)[ja] = make_float4(thr, thr, thr, thr);
            reinterpret_cast<float4 *>(b.pid_integ)[ja] = make_float4(0, 0, 0, 0);
            reinterpret_cast<float4 *>(b.pid_last_rate)[ja] = make_float4(0, 0, 0, 0);
            b.prev_action[ja * 4 + 3] = pa / 4.0f;                                // hideandseek.py:714-716
        }
        if (task) {
            tp[0] = task[3 * A]; tp[1] = task[3 * A + 1]; tp[2] = task[3 * A + 2];
        } else {
            if (c.init_mode == HNS_INIT_RANDOM) {
                tp[0] = c.target_xy_lo[0] + rng.uniform() * (c.target_xy_hi[0] - c.target_xy_lo[0]);
                tp[1] = c.target_xy_lo[1] + rng.uniform() * (c.target_xy_hi[1] - c.target_xy_lo[1]);
            } else {
                tp[0] = c.fixed_target_pos[0]; tp[1] = c.fixed_target_pos[1];
            }
            if (c.init_mode == HNS_INIT_SCENARIO) tp[2] = c.fixed_target_pos[2];
            else tp[2] = c.z_lo + rng.uniform() * (c.z_hi - c.z_lo);
        }
        if constexpr (NT == 2) {               // second evader: same box as the first (its draws follow the first evader's)
            if (task) {
                tp[3] = task[3 * A + 3]; tp[4] = task[3 * A + 4]; tp[5] = task[3 * A + 5];
            } else {
                if (c.init_mode == HNS_INIT_RANDOM) {
                    tp[3] = c.target_xy_lo[0] + rng.uniform() * (c.target_xy_hi[0] - c.target_xy_lo[0]);
                    tp[4] = c.target_xy_lo[1] + rng.uniform() * (c.target_xy_hi[1] - c.target_xy_lo[1]);
                } else {             // fixed scenarios name one evader: the second mirrors it in y
                    tp[3] = c.fixed_target_pos[0]; tp[4] = -c.fixed_target_pos[1];
                }
                if (c.init_mode == HNS_INIT_SCENARIO) tp[5] = c.fixed_target_pos[2];
                else tp[5] = c.z_lo + rng.uniform() * (c.z_hi - c.z_lo);
            }
        }
        if (task) {
            for (int k = 0; k < 3 * C; ++k) cyl[k] = task[3 * A + 3 * NT + k];
        } else if (c.init_mode == HNS_INIT_SCENARIO) {
            for (int k = 0; k < C; ++k) {
                cyl[3 * k] = c.fixed_cyl_pos[k][0]; cyl[3 * k + 1] = c.fixed_cyl_pos[k][1];
                { const float iz = c.invalid_z, fz = c.fixed_cyl_pos[k][2]; cyl[3 * k + 2] = (k >= c.fixed_cyl_active) ? iz : fz; }
            }
        } else {                                                                  // hideandseek.py:576-607
            uint8_t *occ = sGrid + le * kGridStride;   // [GN*GN] occupancy, then [GN*GN] free-cell list
            uint8_t *freec = occ + 256;
            const int half = GN / 2;
            for (int i = 0; i < GN; ++i)
                for (int j = 0; j < GN; ++j) {
                    float dd = __builtin_sqrtf((float)((i - half) * (i - half) + (j - half) * (j - half)));
                    occ[i * GN + j] = dd >= (float)half;                          // :168-181
                }
            for (int j = 0; j < A; ++j) occ[d_cell(c, ds[13 * j]) * GN + d_cell(c, ds[13 * j + 1])] = 1;
            occ[d_cell(c, tp[0]) * GN + d_cell(c, tp[1])] = 1;
            if constexpr (NT == 2) occ[d_cell(c, tp[3]) * GN + d_cell(c, tp[4])] = 1;
            int n_active;
            if (c.cyl_fixed_num >= 0) n_active = c.cyl_fixed_num;
            else {
                int span = C + 1 - c.cyl_min_num;
                int r = (int)(rng.uniform() * (float)span);
                if (r > span - 1) r = span - 1;
                n_active = c.cyl_min_num + r;
            }
            int nfree = 0;
            for (int i = 0; i < GN * GN; ++i) if (!occ[i]) freec[nfree++] = (uint8_t)i;
            for (int k = 0; k < C; ++k) {           // randperm[:C] as a partial Fisher-Yates (:116)
                int span = nfree - k;
                int j = (int)(rng.uniform() * (float)span);
                if (j > span - 1) j = span - 1;
                j += k;
                uint8_t t = freec[k]; freec[k] = freec[j]; freec[j] = t;
                int gx = freec[k] / GN, gy = freec[k] % GN;
                float x = 0.0f + (float)(gx - half) * c.grid_size, y = 0.0f + (float)(gy - half) * c.grid_size;
                cyl[3 * k] = d_clamp(x, -c.boundary, c.boundary);                 // :121-141
                cyl[3 * k + 1] = d_clamp(y, -c.boundary, c.boundary);
                cyl[3 * k + 2] = (k >= n_active) ? c.invalid_z : 0.5f * c.cylinder_height;
            }
        }
        for (int sidx = 0; sidx < HNS_NUM_STATS; ++sidx) b.stats[(size_t)sidx * E + e] = 0.0f;   // :711
        b.stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * E + e] = (float)c.max_episode_length;
        b.progress[e] = 0.0f;
        b.done[e] = 0;
    }
    __syncthreads();
    if (!env_wave && masked) {
        Rigid s;
        load_rigid(sDS + tid * 13, s);
        V3 tp = {sTp[le * 3 * NT], sTp[le * 3 * NT + 1], sTp[le * 3 * NT + 2]};
        V3 tpB = tp;
        if constexpr (NT == 2) tpB = {sTp[le * 3 * NT + 3], sTp[le * 3 * NT + 4], sTp[le * 3 * NT + 5]};
        const size_t ia = (size_t)e0 * A + tid;
        bool blocked, det, blockedB = false, detB = false;
        int knn_idx[KM];
        bool knn_masked[KM];
        agent_obs<A, NT, false, 13, KM>(c, C, K, le, a, s, tp, tpB, 0.0f, sCyl + le * L.cyl_stride, sDS, b.obs_others + ia * (A - 1) * 3, sOCyl, b.obs_self + ia * SD,
                                        with_state ? b.state_drones + ia * SD : nullptr, blocked, det, blockedB, detB, knn_idx, knn_masked, true, true,
                                        b.obs_cylinders + ia * K * 5);
        b.pid_last_rate[ia * 4 + 3] = (float)((blocked ? 1 : 0) + (blockedB ? 2 : 0));   // line of sight of the new state (the env wave zeroed the record above)
        if (det) sDet[le] = 1;
        if (NT == 2 && detB) sDet1[le] = 1;
    }
    __syncthreads();
    if (env_wave && masked && !sDet[le]) {                     // hideandseek.py:791-794
        for (int j = 0; j < A; ++j) {
            float *o = b.obs_self + ((size_t)e * A + j) * SD;
            o[0] = c.mask_value; o[1] = c.mask_value; o[2] = c.mask_value;
        }
    }
    if (NT == 2 && env_wave && masked && !sDet1[le]) {
        for (int j = 0; j < A; ++j) {
            float *o = b.obs_self + ((size_t)e * A + j) * SD + HNS_SELF_DIM;
            o[0] = c.mask_value; o[1] = c.mask_value; o[2] = c.mask_value;
        }
    }
    if (env_wave && masked && b.detect) b.detect[e] = (uint8_t)(sDet[le] | (NT == 2 ? (sDet1[le] << 1) : 0));
    coop_s2g_masked<T>(b.drone_state + (size_t)e0 * A * 13, sDS, nenv * A * 13, A * 13, sMask);
    coop_cyl<T, false>(sCyl, b.cylinders + (size_t)e0 * C * 3, nenv, 3 * C, L.cyl_stride, p.cyl_magic, sMask);
    coop_s2g_masked<T>(b.target_pos + (size_t)e0 * 3 * NT, sTp, nenv * 3 * NT, 3 * NT, sMask);
    if constexpr (KM == kMaxK) coop_s2g_masked<T>(b.obs_cylinders + (size_t)e0 * A * K * 5, sOCyl, nenv * A * K * 5, A * K * 5, sMask);
}

// =================================================================================================
// Hover task (BASELINE config 1; reference omni_drones/envs/single/hover.py:322-523): one thread per
// env — a plumbing-scale task (tens of envs), written for clarity, reusing the drone math above.
// =================================================================================================
enum { HS_RETURN = 0, HS_POS_BONUS, HS_HEAD_BONUS, HS_REWARD_POS, HS_REWARD_UP, HS_REWARD_VEL, HS_REWARD_ACC, HS_REWARD_JERK,
       HS_EPISODE_LEN, HS_POS_ERROR, HS_HEADING_ALIGNMENT, HS_UPRIGHTNESS, HS_ACTION_SMOOTHNESS, HS_LINEAR_V_MAX,
       HS_ANGULAR_V_MAX, HS_LINEAR_A_MAX, HS_ANGULAR_A_MAX, HS_LINEAR_JERK_MAX, HS_ANGULAR_JERK_MAX, HS_LINEAR_V_MEAN,
       HS_ANGULAR_V_MEAN, HS_LINEAR_A_MEAN, HS_ANGULAR_A_MEAN, HS_LINEAR_JERK_MEAN, HS_ANGULAR_JERK_MEAN, HS_MOTOR1,
       HS_MOTOR2, HS_MOTOR3, HS_MOTOR4, HS_CMD_R, HS_CMD_P, HS_CMD_Y, HS_CMD_THRUST, HS_TARGET_R_RATE, HS_TARGET_P_RATE,
       HS_TARGET_Y_RATE, HS_REAL_R_RATE, HS_REAL_P_RATE, HS_REAL_Y_RATE };
enum { HA_LV_EP = 0, HA_AV_EP, HA_LA_EP, HA_AA_EP, HA_LJ_EP, HA_AJ_EP, HA_LAST_LV, HA_LAST_AV, HA_LAST_LA, HA_LAST_AA,
       HA_LAST_LJ, HA_LAST_AJ };

struct HoverParams {
    hns_cfg cfg;
    hns_hover_cfg hover;
    hns_hover_buffers buf;
    const float *action;
    const uint8_t *reset_mask;
    uint32_t seed_lo, seed_hi, epoch;
};

// hover.py:361-437 (_compute_state_and_obs) for one env
HNS_DEV void hover_obs(const hns_cfg &c, const hns_hover_cfg &h, const Rigid &s, float progress, float *st, float *ac, int E,
                       float *obs, V3 &heading, V3 &up, float &lv, float &la, float &lj) {
#define ST(i) st[(size_t)(i) * E]
#define AC(i) ac[(size_t)(i) * E]
    V3 br = d_quat_rot<true>(s.q, s.ang);
    ST(HS_REAL_R_RATE) = (br.x * 180.0f) * kInvPi;
    ST(HS_REAL_P_RATE) = (br.y * 180.0f) * kInvPi;
    ST(HS_REAL_Y_RATE) = (br.z * 180.0f) * kInvPi;
    heading = d_quat_rot_x(s.q);
    up = d_quat_rot_z(s.q, 1.0f);
    const float t = progress * c.inv_max_episode_length;
    obs[0] = h.target_pos[0] - s.pos.x; obs[1] = h.target_pos[1] - s.pos.y; obs[2] = h.target_pos[2] - s.pos.z;
    obs[3] = s.q.w; obs[4] = s.q.x; obs[5] = s.q.y; obs[6] = s.q.z;
    obs[7] = s.lin.x; obs[8] = s.lin.y; obs[9] = s.lin.z;
    obs[10] = heading.x; obs[11] = heading.y; obs[12] = heading.z;
    obs[13] = up.x; obs[14] = up.y; obs[15] = up.z;
    obs[16] = t; obs[17] = t; obs[18] = t; obs[19] = t;
    lv = d_norm3(s.lin.x, s.lin.y, s.lin.z);
    const float av = d_norm3(s.ang.x, s.ang.y, s.ang.z);
    const float n = progress + 1.0f;
    if (__builtin_fabsf(lv) > ST(HS_LINEAR_V_MAX)) ST(HS_LINEAR_V_MAX) = __builtin_fabsf(lv);
    AC(HA_LV_EP) += __builtin_fabsf(lv); ST(HS_LINEAR_V_MEAN) = AC(HA_LV_EP) / n;
    if (__builtin_fabsf(av) > ST(HS_ANGULAR_V_MAX)) ST(HS_ANGULAR_V_MAX) = __builtin_fabsf(av);
    AC(HA_AV_EP) += __builtin_fabsf(av); ST(HS_ANGULAR_V_MEAN) = AC(HA_AV_EP) / n;
    la = __builtin_fabsf(lv - AC(HA_LAST_LV)) / c.dt;
    const float aa = __builtin_fabsf(av - AC(HA_LAST_AV)) / c.dt;
    if (__builtin_fabsf(la) > ST(HS_LINEAR_A_MAX)) ST(HS_LINEAR_A_MAX) = __builtin_fabsf(la);
    AC(HA_LA_EP) += __builtin_fabsf(la); ST(HS_LINEAR_A_MEAN) = AC(HA_LA_EP) / n;
    if (__builtin_fabsf(aa) > ST(HS_ANGULAR_A_MAX)) ST(HS_ANGULAR_A_MAX) = __builtin_fabsf(aa);
    AC(HA_AA_EP) += __builtin_fabsf(aa); ST(HS_ANGULAR_A_MEAN) = AC(HA_AA_EP) / n;
    lj = __builtin_fabsf(la - AC(HA_LAST_LA)) / c.dt;
    const float aj = __builtin_fabsf(aa - AC(HA_LAST_AA)) / c.dt;
    if (__builtin_fabsf(lj) > ST(HS_LINEAR_JERK_MAX)) ST(HS_LINEAR_JERK_MAX) = __builtin_fabsf(lj);
    AC(HA_LJ_EP) += __builtin_fabsf(lj); ST(HS_LINEAR_JERK_MEAN) = AC(HA_LJ_EP) / n;
    if (__builtin_fabsf(aj) > ST(HS_ANGULAR_JERK_MAX)) ST(HS_ANGULAR_JERK_MAX) = __builtin_fabsf(aj);
    AC(HA_AJ_EP) += __builtin_fabsf(aj); ST(HS_ANGULAR_JERK_MEAN) = AC(HA_AJ_EP) / n;
    AC(HA_LAST_LV) = lv; AC(HA_LAST_AV) = av; AC(HA_LAST_LA) = la; AC(HA_LAST_AA) = aa; AC(HA_LAST_LJ) = lj; AC(HA_LAST_AJ) = aj;
#undef ST
#undef AC
}

__global__ __launch_bounds__(64) void hns_hover_step_kernel(const HoverParams p) {
    const hns_cfg &c = p.cfg;
    const hns_hover_cfg &h = p.hover;
    const hns_hover_buffers &b = p.buf;
    const int E = c.num_envs;
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    float *st = b.stats + e, *ac = b.acc + e;
#define ST(i) st[(size_t)(i) * E]
    Rigid s;
    load_rigid(b.drone_state + (size_t)e * 13, s);
    float4 act4 = reinterpret_cast<const float4 *>(p.action)[e];
    float4 thr4 = reinterpret_cast<float4 *>(b.throttle)[e], integ4 = reinterpret_cast<float4 *>(b.pid_integ)[e];
    float4 last4 = reinterpret_cast<float4 *>(b.pid_last_rate)[e], prev4 = reinterpret_cast<float4 *>(b.prev_action)[e];
    float cmd[4], aerr, ctbr[4], trate[3], thrust[4], moment[4], td;
    d_ctbr_pid(c, act4, s.q, s.ang, prev4, integ4, last4, cmd, aerr, ctbr, trate);
    ST(HS_MOTOR1) = cmd[0]; ST(HS_MOTOR2) = cmd[1]; ST(HS_MOTOR3) = cmd[2]; ST(HS_MOTOR4) = cmd[3];          // hover.py:326-329
    d_rotor(c, cmd, thr4, thrust, moment, td);
    ST(HS_CMD_R) = ctbr[0]; ST(HS_CMD_P) = ctbr[1]; ST(HS_CMD_Y) = ctbr[2]; ST(HS_CMD_THRUST) = ctbr[3];      // :334-338
    ST(HS_TARGET_R_RATE) = trate[0]; ST(HS_TARGET_P_RATE) = trate[1]; ST(HS_TARGET_Y_RATE) = trate[2];         // :341-344
    const float ts = ((thrust[0] + thrust[1]) + thrust[2]) + thrust[3];
    V3 fw = d_quat_rot_z(s.q, ts), tb;
    tb.x = ((c.rotor_py[0] * thrust[0] + c.rotor_py[1] * thrust[1]) + c.rotor_py[2] * thrust[2]) + c.rotor_py[3] * thrust[3];
    tb.y = -(((c.rotor_px[0] * thrust[0] + c.rotor_px[1] * thrust[1]) + c.rotor_px[2] * thrust[2]) + c.rotor_px[3] * thrust[3]);
    tb.z = ((moment[0] + moment[1]) + moment[2]) + moment[3];
    d_integrate(c, s, fw, tb);
    const float progress = b.progress[e] + 1.0f;
    float obs[HNS_SELF_DIM], lv, la, lj;
    V3 heading, up;
    hover_obs(c, h, s, progress, st, ac, E, obs, heading, up, lv, la, lj);
    // hover.py:439-523
    const float pos_error = d_norm3(obs[0], obs[1], obs[2]);
    const float hx = h.target_heading[0] - heading.x, hy = h.target_heading[1] - heading.y, hz = h.target_heading[2] - heading.z;
    const float head_error = d_norm3(hx, hy, hz);
    const float align = (heading.x * h.target_heading[0] + heading.y * h.target_heading[1]) + heading.z * h.target_heading[2];
    const float reward_pos = -pos_error * h.reward_distance_scale;
    const float bonus = (pos_error <= 0.02f) ? 10.0f : 0.0f;
    const float bpos = bonus > 0.0f ? 1.0f : 0.0f;
    const float reward_head = -head_error * bpos;
    const float head_bonus = ((head_error <= 0.02f) ? 10.0f : 0.0f) * bpos;
    const float u = (up.z + 1.0f) / 2.0f;
    const float reward_up = u * u;
    const float reward_v = (h.reward_v_scale * bpos) * ((lv < h.linear_vel_max) ? 1.0f : 0.0f);
    const float reward_acc = (h.reward_acc_scale * bpos) * ((la < h.linear_acc_max) ? 1.0f : 0.0f);
    const float reward_jerk = (h.reward_jerk_scale * bpos) * -lj;
    const float reward = ((((((reward_pos + bonus) + reward_head) + head_bonus) + reward_up) + reward_v) + reward_acc) + reward_jerk;
    const float w = 1.0f - h.alpha;
    ST(HS_POS_ERROR) += w * (pos_error - ST(HS_POS_ERROR));                    // lerp_ :506-509
    ST(HS_HEADING_ALIGNMENT) += w * (align - ST(HS_HEADING_ALIGNMENT));
    ST(HS_UPRIGHTNESS) += w * (up.z - ST(HS_UPRIGHTNESS));
    ST(HS_ACTION_SMOOTHNESS) += w * (-td - ST(HS_ACTION_SMOOTHNESS));
    ST(HS_RETURN) += reward;
    ST(HS_REWARD_POS) = reward_pos; ST(HS_POS_BONUS) = bonus; ST(HS_HEAD_BONUS) = head_bonus;
    ST(HS_REWARD_VEL) = reward_v; ST(HS_REWARD_ACC) = reward_acc; ST(HS_REWARD_JERK) = reward_jerk;
    ST(HS_EPISODE_LEN) = progress;
#undef ST
    store_rigid(b.drone_state + (size_t)e * 13, s);
    reinterpret_cast<float4 *>(b.throttle)[e] = thr4;
    reinterpret_cast<float4 *>(b.pid_integ)[e] = integ4;
    reinterpret_cast<float4 *>(b.pid_last_rate)[e] = last4;
    reinterpret_cast<float4 *>(b.prev_action)[e] = prev4;
    for (int i = 0; i < HNS_SELF_DIM; ++i) b.obs[(size_t)e * HNS_SELF_DIM + i] = obs[i];
    b.reward[e] = reward;
    b.done[e] = (uint8_t)(progress >= (float)c.max_episode_length);
    b.progress[e] = progress;
}

// hover.py:285-320 (_reset_idx) + the reset-time observation of the masked envs
__global__ __launch_bounds__(64) void hns_hover_reset_kernel(const HoverParams p) {
    const hns_cfg &c = p.cfg;
    const hns_hover_cfg &h = p.hover;
    const hns_hover_buffers &b = p.buf;
    const int E = c.num_envs;
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    for (int i = HA_LV_EP; i <= HA_AJ_EP; ++i) b.acc[(size_t)i * E + e] = 0.0f;      // all envs, :313-320
    if (p.reset_mask && !p.reset_mask[e]) return;
    Rng rng = {p.seed_lo, p.seed_hi, (uint32_t)(e + c.env_index_offset), p.epoch, 0u, {0u, 0u, 0u, 0u}, 0};
    Rigid s = {};
    s.pos.x = h.pos_lo[0] + rng.uniform() * (h.pos_hi[0] - h.pos_lo[0]);
    s.pos.y = h.pos_lo[1] + rng.uniform() * (h.pos_hi[1] - h.pos_lo[1]);
    s.pos.z = h.pos_lo[2] + rng.uniform() * (h.pos_hi[2] - h.pos_lo[2]);
    float r0 = h.rpy_lo[0] + rng.uniform() * (h.rpy_hi[0] - h.rpy_lo[0]);
    float r1 = h.rpy_lo[1] + rng.uniform() * (h.rpy_hi[1] - h.rpy_lo[1]);
    float r2 = h.rpy_lo[2] + rng.uniform() * (h.rpy_hi[2] - h.rpy_lo[2]);
    s.q = d_euler_to_quat(r0, r1, r2);
    store_rigid(b.drone_state + (size_t)e * 13, s);
    const float thr = c.hover_throttle;
    reinterpret_cast<float4 *>(b.throttle)[e] = make_float4(thr, thr, thr, thr);
    reinterpret_cast<float4 *>(b.pid_integ)[e] = make_float4(0, 0, 0, 0);
    reinterpret_cast<float4 *>(b.pid_last_rate)[e] = make_float4(0, 0, 0, 0);
    for (int i = 0; i < HNS_HOVER_NUM_STATS; ++i) b.stats[(size_t)i * E + e] = 0.0f;
    for (int i = HA_LAST_LV; i <= HA_LAST_AJ; ++i) b.acc[(size_t)i * E + e] = 0.0f;
    b.progress[e] = 0.0f;
    b.done[e] = 0;
    float obs[HNS_SELF_DIM], lv, la, lj;
    V3 heading, up;
    hover_obs(c, h, s, 0.0f, b.stats + e, b.acc + e, E, obs, heading, up, lv, la, lj);
    for (int i = 0; i < HNS_SELF_DIM; ++i) b.obs[(size_t)e * HNS_SELF_DIM + i] = obs[i];
}

// =================================================================================================
// Extension (not in the reference): planar ray-fan range sensor.  One thread per
// (env, pursuer, ray); a workgroup stages the cylinder sets and the ray origins/headings of its envs
// in LDS once.  Geometry exactly as oracle/hns_oracle.c::hns_oracle_raycast.
// =================================================================================================
struct RayParams {
    hns_cfg cfg;
    const float *drone_state, *cylinders;
    float *out;
    int num_rays, envs_per_block;
    float max_range;
};

__global__ __launch_bounds__(256) void hns_raycast_kernel(const RayParams p) {
    extern __shared__ __align__(16) float smem[];
    const hns_cfg &c = p.cfg;
    const int E = c.num_envs, A = c.num_agents, C = c.num_cylinders, N = p.num_rays, EPB = p.envs_per_block;
    const int e0 = blockIdx.x * EPB;
    const int nenv = min(EPB, E - e0);
    float *sCyl = smem;                         // [EPB][C][3]
    float *sOrg = smem + EPB * C * 3;           // [EPB][A][4] = ox, oy, ux0, uy0
    for (int i = threadIdx.x; i < nenv * C * 3; i += 256) sCyl[i] = p.cylinders[(size_t)e0 * C * 3 + i];
    for (int i = threadIdx.x; i < nenv * A; i += 256) {
        const float *ds = p.drone_state + ((size_t)e0 * A + i) * 13;
        Q4 q = {ds[3], ds[4], ds[5], ds[6]};
        V3 h = d_quat_rot_x(q);
        float hn = d_norm2(h.x, h.y);
        sOrg[4 * i] = ds[0]; sOrg[4 * i + 1] = ds[1];
        sOrg[4 * i + 2] = hn > 1e-6f ? h.x / hn : 1.0f;
        sOrg[4 * i + 3] = hn > 1e-6f ? h.y / hn : 0.0f;
    }
    __syncthreads();
    const float step = 6.283185307179586f / (float)N;
    for (int i = threadIdx.x; i < nenv * A * N; i += 256) {
        const int ea = i / N, r = i - ea * N, le = ea / A;
        const float ox = sOrg[4 * ea], oy = sOrg[4 * ea + 1], ux0 = sOrg[4 * ea + 2], uy0 = sOrg[4 * ea + 3];
        float sn, cs;
        d_sincosf(step * (float)r, sn, cs);
        const float ux = HNS_FMA(ux0, cs, -(uy0 * sn)), uy = HNS_FMA(ux0, sn, uy0 * cs);
        const float oo = HNS_FMA(oy, oy, ox * ox);
        const float ou = HNS_FMA(oy, uy, ox * ux);
        const float dw = HNS_FMA(ou, ou, -(oo - c.arena_sq));
        float best = dw >= 0.0f ? __builtin_sqrtf(dw) - ou : 0.0f;
        if (!(best >= 0.0f)) best = 0.0f;
        const float *cyl = sCyl + le * C * 3;
        for (int k = 0; k < C; ++k) {
            const float ccx = cyl[3 * k], ccy = cyl[3 * k + 1], ccz = cyl[3 * k + 2];
            if (!(ccz > 0.0f)) continue;
            const float mx = ccx - ox, my = ccy - oy;
            const float bq = HNS_FMA(my, uy, mx * ux);
            const float cq = HNS_FMA(my, my, mx * mx) - c.cylinder_size * c.cylinder_size;
            const float disc = HNS_FMA(bq, bq, -cq);
            if (disc >= 0.0f) {
                float t = bq - __builtin_sqrtf(disc);
                if (cq <= 0.0f) t = 0.0f;
                if (t >= 0.0f && t < best) best = t;
            }
        }
        p.out[(size_t)e0 * A * N + i] = best > p.max_range ? p.max_range : best;
    }
}

}  // namespace hns

// =================================================================================================
// Host side: the C ABI (include/hns.h)
// =================================================================================================
using hns::Params;

static thread_local std::string g_last_error;
void hns_set_error(const std::string &m) { g_last_error = m; }
static void set_error(const std::string &m) { g_last_error = m; }


template <int A>
static void select_kernels(hns_env *env) {
    const hns_cfg &c = env->cfg;
    if (c.num_targets == 2) {
        env->step_fn = (c.num_envs % hns::kEPB == 0) ? hns::hns_step_kernel<A, 2, true> : hns::hns_step_kernel<A, 2, false>;
        env->reset_fn = hns::hns_reset_kernel<A, 2>;
    } else {
        env->step_fn = (c.num_envs % hns::kEPB == 0) ? hns::hns_step_kernel<A, 1, true> : hns::hns_step_kernel<A, 1, false>;
        env->reset_fn = hns::hns_reset_kernel<A, 1>;
    }
    const bool wide = c.obs_max_cylinder > hns::kMaxK;     // k-nearest selections beyond the step kernels' network: the first design, unstaged rows
    if (wide) {
        if (c.num_targets == 2) { env->step_fn = hns::hns_step_kernel<A, 2, false, hns::kWideK>; env->reset_fn = hns::hns_reset_kernel<A, 2, hns::kWideK>; }
        else { env->step_fn = hns::hns_step_kernel<A, 1, false, hns::kWideK>; env->reset_fn = hns::hns_reset_kernel<A, 1, hns::kWideK>; }
    }
    const char *force = getenv("HNS_STEP_DESIGN");        // "1" = the first design for every shape (A/B measurements)
    const bool v3 = c.num_envs % hns::kEPB == 0 && !(force && force[0] == '1') && !wide;
    if (v3 && c.num_targets == 2) { env->step_args_fn = hns::hns_step_v4_kernel<A, 2, false>; env->step_args_prof_fn = hns::hns_step_v4_kernel<A, 2, true>; }
    else if (v3) { env->step_args_fn = hns::hns_step_v4_kernel<A, 1, false>; env->step_args_prof_fn = hns::hns_step_v4_kernel<A, 1, true>; }
    env->threads = hns::Geo<A>::T;
    env->cyl_magic = (uint32_t)(0xFFFFFFFFull / (uint32_t)(3 * c.num_cylinders) + 1ull);
    env->grid = (c.num_envs + hns::kEPB - 1) / hns::kEPB;
    hns::Lds L = hns::lds_layout(A, c.num_cylinders, c.obs_max_cylinder, c.num_targets == 2 ? 2 : 1);
    env->lds_step = (size_t)L.total * sizeof(float);
    env->lds_reset = env->lds_step + (size_t)hns::kEPB * hns::kGridStride;   // + per-env occupancy grid / free-cell list
    if (v3) env->lds_step = (size_t)hns::lds_layout_v3(A, c.num_cylinders, c.obs_max_cylinder, c.num_targets == 2 ? 2 : 1).total * sizeof(float);
}


static int upload_step_params(hns_env *env);
static int alloc_step_params(hns_env *env);

extern "C" {

int hns_abi_version(void) { return HNS_ABI_VERSION; }
size_t hns_cfg_size(void) { return sizeof(hns_cfg); }
const char *hns_last_error(void) { return g_last_error.c_str(); }

int hns_create(const hns_cfg *cfg, hns_env **out) {
    if (!cfg || !out) { set_error("hns_create: null argument"); return HNS_ERR_INVALID_ARG; }
    *out = nullptr;
    if (cfg->abi_version != HNS_ABI_VERSION) { set_error("hns_create: abi_version mismatch"); return HNS_ERR_INVALID_ARG; }
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_agents > HNS_MAX_AGENTS || cfg->num_cylinders < 1 ||
        cfg->num_cylinders > HNS_MAX_CYLINDERS || cfg->obs_max_cylinder < 1 || cfg->obs_max_cylinder > cfg->num_cylinders) {
        set_error("hns_create: num_envs/num_agents/num_cylinders/obs_max_cylinder out of range");
        return HNS_ERR_INVALID_ARG;
    }
    if (cfg->num_targets < 0 || cfg->num_targets > hns::kMaxT) { set_error("hns_create: num_targets must be 0, 1 or 2"); return HNS_ERR_INVALID_ARG; }
    if (cfg->grid_num < 1 || cfg->grid_num > 16) { set_error("hns_create: grid_num out of range"); return HNS_ERR_INVALID_ARG; }
    if (cfg->init_mode != HNS_INIT_SCENARIO) {
        int half = cfg->grid_num / 2, free_cells = 0;
        for (int i = 0; i < cfg->grid_num; ++i)
            for (int j = 0; j < cfg->grid_num; ++j)
                if (sqrtf((float)((i - half) * (i - half) + (j - half) * (j - half))) < (float)half) ++free_cells;
        if (free_cells - (cfg->num_agents + (cfg->num_targets == 2 ? 2 : 1)) < cfg->num_cylinders) {   // pursuers and evader(s) occupy cells first
            set_error("hns_create: not enough free grid cells for the cylinders (hideandseek.py:112-113)");
            return HNS_ERR_CONFIG;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("hns_create: no HIP device visible (this library has no CPU path)");
        return HNS_ERR_NO_DEVICE;
    }
    hns_env *env = new (std::nothrow) hns_env();
    if (!env) { set_error("hns_create: out of host memory"); return HNS_ERR_INVALID_ARG; }
    env->cfg = *cfg;
    if (hipGetDevice(&env->device) != hipSuccess) env->device = 0;
    std::memset(&env->buf, 0, sizeof(env->buf));
    switch (cfg->num_agents) {
        case 1: select_kernels<1>(env); break;
        case 2: select_kernels<2>(env); break;
        case 3: select_kernels<3>(env); break;
        case 4: select_kernels<4>(env); break;
        case 5: select_kernels<5>(env); break;
        case 6: select_kernels<6>(env); break;
        case 7: select_kernels<7>(env); break;
        default: delete env; set_error("hns_create: unsupported num_agents"); return HNS_ERR_INVALID_ARG;
    }
    size_t lds_max = env->lds_reset > env->lds_step ? env->lds_reset : env->lds_step;
    if (lds_max > 160 * 1024) {
        delete env;
        set_error("hns_create: configuration needs more than 160 KiB LDS per workgroup");
        return HNS_ERR_CONFIG;
    }
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->step_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_step);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->reset_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_reset);
    if (e1 == hipSuccess && env->step_args_fn) {
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->step_args_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_step);
        if (e1 == hipSuccess)
            e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->step_args_prof_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_step);
    }
    if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        delete env;
        return HNS_ERR_DEVICE;
    }
    if (alloc_step_params(env) != HNS_OK) { hns_destroy(env); return HNS_ERR_DEVICE; }
    *out = env;
    return HNS_OK;
}

void hns_destroy(hns_env *env) {
    if (!env) return;
    for (auto &p : env->events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : env->pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (env->params_dev) (void)hipFree(env->params_dev);
    if (env->params_ring) (void)hipHostFree(env->params_ring);
    for (auto *img : env->captured_images) (void)hipHostFree(img);
    for (auto &ev : env->ring_events) if (ev) (void)hipEventDestroy(ev);
    delete env->params_host;
    delete env;
}

int hns_bind(hns_env *env, const hns_buffers *buffers) {
    if (!env || !buffers) { set_error("hns_bind: null argument"); return HNS_ERR_INVALID_ARG; }
    const void *req[] = {buffers->drone_state, buffers->throttle, buffers->pid_integ, buffers->pid_last_rate,
                         buffers->prev_action, buffers->target_pos, buffers->target_vel, buffers->cylinders,
                         buffers->progress, buffers->stats, buffers->obs_self, buffers->obs_cylinders,
                         buffers->reward, buffers->action_error, buffers->done};
    for (const void *ptr : req)
        if (!ptr) { set_error("hns_bind: a required buffer pointer is null"); return HNS_ERR_INVALID_ARG; }
    if (env->cfg.num_agents > 1 && !buffers->obs_others) { set_error("hns_bind: obs_others is null"); return HNS_ERR_INVALID_ARG; }
    if (env->cfg.write_critic_state && !buffers->state_drones) {
        set_error("hns_bind: write_critic_state set but state_drones is null");
        return HNS_ERR_INVALID_ARG;
    }
    const void *al16[] = {buffers->throttle, buffers->pid_integ, buffers->pid_last_rate, buffers->prev_action,
                          buffers->drone_state, buffers->target_pos, buffers->target_vel, buffers->obs_self,
                          buffers->state_drones, buffers->obs_cylinders, buffers->ctbr, buffers->target_rate};
    for (const void *ptr : al16)
        if (reinterpret_cast<uintptr_t>(ptr) & 15) { set_error("hns_bind: buffers must be 16-byte aligned"); return HNS_ERR_INVALID_ARG; }
    if (env->cfg.num_agents > 1 && (reinterpret_cast<uintptr_t>(buffers->obs_others) & 7)) {
        set_error("hns_bind: obs_others must be 8-byte aligned");
        return HNS_ERR_INVALID_ARG;
    }
    // a host pointer (or memory of another GPU) here would fault inside the kernel: check once, at bind time
    for (const void *ptr : req)
        if (!hns_on_env_device(env, ptr)) {
            set_error("hns_bind: every buffer must be device memory of the GPU that was current at hns_create (no host pointers)");
            return HNS_ERR_INVALID_ARG;
        }
    env->buf = *buffers;
    env->bound = true;
    return upload_step_params(env);
}

// the step launch's parameter block (everything but the action)
static void fill_step_params(const hns_env *env, Params &p) {
    memset(&p, 0, sizeof(p));          // compared bytewise with the device copy: no stack garbage in the padding
    p.cfg = env->cfg;
    p.buf = env->buf;
    p.prof = env->prof;
    p.cyl_magic = env->cyl_magic;
#ifdef HNS_LAB
    { const char *f = getenv("HNS_LAB_FLAGS"); p.lab = f ? (uint32_t)atoi(f) : 0u; }
    { const char *f = getenv("HNS_LAB_STAGGER"); p.lab_stagger = f ? (uint32_t)atoi(f) : 0u; }
#endif
}

// Device copy of that block for the step kernel that reads it through `StepArgs::rest`: one allocation in hns_create, refreshed
// where the block changes (bind, the configuration setters, the profiling buffer), never from a steady-state step.
static int alloc_step_params(hns_env *env) {
    HNS_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&env->params_dev), sizeof(Params)));
    HNS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&env->params_ring), sizeof(Params) * hns_env::kParamRing, hipHostMallocDefault));
    for (auto &ev : env->ring_events) HNS_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    env->params_host = new Params;
    return HNS_OK;
}

// A change travels as ONE stream-ordered copy of the block from a pinned image, enqueued on the stream of the latest step / reset
// call: launches already enqueued there keep the old values, later ones see the new ones; no device synchronisation, no allocation,
// legal inside a stream capture (the image a capture takes is then kept for the graph's lifetime).  The very first upload (hns_bind
// before any launch) is a plain blocking copy: nothing reads the block yet.
static int upload_step_params(hns_env *env) {
    if (!env->step_args_fn || !env->bound) return HNS_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != env->device) {
        set_error("the current HIP device is not the one this env was created on (hipSetDevice first)");
        return HNS_ERR_DEVICE;
    }
    Params q;
    fill_step_params(env, q);
    if (env->params_valid && memcmp(&q, env->params_host, sizeof(Params)) == 0) return HNS_OK;
    memcpy(env->params_host, &q, sizeof(Params));
    if (!env->params_valid) {
        HNS_CHECK_HIP(hipMemcpy(env->params_dev, env->params_host, sizeof(Params), hipMemcpyHostToDevice));
        env->params_valid = true;
        return HNS_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(env->last_stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) {
        Params *img = nullptr;
        HNS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&img), sizeof(Params), hipHostMallocDefault));
        memcpy(img, &q, sizeof(Params));
        env->captured_images.push_back(img);
        HNS_CHECK_HIP(hipMemcpyAsync(env->params_dev, img, sizeof(Params), hipMemcpyHostToDevice, env->last_stream));
        return HNS_OK;
    }
    const int slot = env->ring_next;
    env->ring_next = (slot + 1) % hns_env::kParamRing;
    if (env->ring_pending[slot]) HNS_CHECK_HIP(hipEventSynchronize(env->ring_events[slot]));   // only when kParamRing changes are in flight at once
    memcpy(env->params_ring + slot, &q, sizeof(Params));
    HNS_CHECK_HIP(hipMemcpyAsync(env->params_dev, env->params_ring + slot, sizeof(Params), hipMemcpyHostToDevice, env->last_stream));
    HNS_CHECK_HIP(hipEventRecord(env->ring_events[slot], env->last_stream));
    env->ring_pending[slot] = true;
    return HNS_OK;
}

static int launch(hns_env *env, bool is_step, const Params &p, hipStream_t stream) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != env->device) {
        set_error("hns_step / hns_reset: the current HIP device is not the one this env was created on");
        return HNS_ERR_DEVICE;
    }
    env->last_stream = stream;
    std::pair<hipEvent_t, hipEvent_t> ev{};
    const bool time_it = is_step && env->timing > 0 && (env->step_count++ % (uint64_t)env->timing) == 0;
    auto fn = is_step ? env->step_fn : env->reset_fn;
    size_t lds = is_step ? env->lds_step : env->lds_reset;
    const bool split = is_step && env->step_args_fn != nullptr;
    hns::StepArgs ka{};
    if (split) {
#ifdef HNS_LAB
        { const int rc = upload_step_params(env); if (rc != HNS_OK) return rc; }     // the ablation switches come from the environment, per launch
#endif
        if (!env->params_valid) { set_error("hns_step: the device copy of the launch parameters is missing (bind first)"); return HNS_ERR_NOT_BOUND; }
        ka = hns::StepArgs{p.action, p.buf.prev_action, p.buf.drone_state, p.buf.pid_integ, p.buf.pid_last_rate, p.buf.throttle,
                           reinterpret_cast<const float *>(reinterpret_cast<uintptr_t>(p.buf.cylinders) | (uintptr_t)((env->cfg.num_cylinders - 1) & 15)), env->params_dev};
    }
    if (time_it) {
        if (!env->pool.empty()) { ev = env->pool.back(); env->pool.pop_back(); }
        else {
            HNS_CHECK_HIP(hipEventCreate(&ev.first));
            HNS_CHECK_HIP(hipEventCreate(&ev.second));
        }
        // the events ride on the dispatch itself (start / stop of THIS kernel, the timestamps a profiler reads),
        // not on separate marker packets before and after it
        if (split) hipExtLaunchKernelGGL(env->prof ? env->step_args_prof_fn : env->step_args_fn, dim3(env->grid), dim3(env->threads), (uint32_t)lds, stream, ev.first, ev.second, 0, ka);
        else hipExtLaunchKernelGGL(fn, dim3(env->grid), dim3(env->threads), (uint32_t)lds, stream, ev.first, ev.second, 0, p);
        HNS_CHECK_HIP(hipGetLastError());
        env->events.push_back(ev);
        return HNS_OK;
    }
    if (split) hipLaunchKernelGGL(env->prof ? env->step_args_prof_fn : env->step_args_fn, dim3(env->grid), dim3(env->threads), lds, stream, ka);
    else hipLaunchKernelGGL(fn, dim3(env->grid), dim3(env->threads), lds, stream, p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_step(hns_env *env, const float *action, void *stream) {
    if (!env || !action) { set_error("hns_step: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_step: buffers not bound"); return HNS_ERR_NOT_BOUND; }
    if (reinterpret_cast<uintptr_t>(action) & 15) { set_error("hns_step: action must be 16-byte aligned"); return HNS_ERR_INVALID_ARG; }
    Params p;
    fill_step_params(env, p);
    p.action = action;
    return launch(env, true, p, static_cast<hipStream_t>(stream));
}

int hns_reset(hns_env *env, const uint8_t *reset_mask, uint64_t seed, void *stream) {
    if (!env) { set_error("hns_reset: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_reset: buffers not bound"); return HNS_ERR_NOT_BOUND; }
    Params p;
    p.cfg = env->cfg;
    p.buf = env->buf;
    p.action = nullptr;
    p.reset_mask = reset_mask;
    p.seed_lo = (uint32_t)seed;
    p.seed_hi = (uint32_t)(seed >> 32);
    p.epoch = env->epoch++;
    p.prof = nullptr;
    p.cyl_magic = env->cyl_magic;
    p.tasks = nullptr;
    p.task_first = 0;
    p.lab = 0;
    p.lab_stagger = 0;
    return launch(env, false, p, static_cast<hipStream_t>(stream));
}

int hns_reset_tasks(hns_env *env, const uint8_t *reset_mask, const float *tasks, int32_t task_first, uint64_t seed, void *stream) {
    if (!env || !tasks) { set_error("hns_reset_tasks: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_reset_tasks: buffers not bound"); return HNS_ERR_NOT_BOUND; }
    if (task_first < 0 || task_first > env->cfg.num_envs) { set_error("hns_reset_tasks: task_first out of range"); return HNS_ERR_INVALID_ARG; }
    Params p;
    p.cfg = env->cfg;
    p.buf = env->buf;
    p.action = nullptr;
    p.reset_mask = reset_mask;
    p.seed_lo = (uint32_t)seed;
    p.seed_hi = (uint32_t)(seed >> 32);
    p.epoch = env->epoch++;
    p.prof = nullptr;
    p.cyl_magic = env->cyl_magic;
    p.tasks = tasks;
    p.task_first = task_first;
    p.lab = 0;
    p.lab_stagger = 0;
    return launch(env, false, p, static_cast<hipStream_t>(stream));
}

static int hover_check(const hns_cfg *cfg, const hns_hover_cfg *hover, const hns_hover_buffers *b) {
    if (!cfg || !hover || !b) { set_error("hns_hover: null argument"); return HNS_ERR_INVALID_ARG; }
    if (cfg->abi_version != HNS_ABI_VERSION || cfg->num_envs < 1 || cfg->num_agents != 1) {
        set_error("hns_hover: bad cfg (abi_version, num_envs >= 1, num_agents == 1)");
        return HNS_ERR_INVALID_ARG;
    }
    const void *req[] = {b->drone_state, b->throttle, b->pid_integ, b->pid_last_rate, b->prev_action, b->progress,
                         b->stats, b->acc, b->obs, b->reward, b->done};
    for (const void *ptr : req)
        if (!ptr) { set_error("hns_hover: a buffer pointer is null"); return HNS_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("hns_hover: no HIP device visible (this library has no CPU path)");
        return HNS_ERR_NO_DEVICE;
    }
    return HNS_OK;
}

int hns_hover_step(const hns_cfg *cfg, const hns_hover_cfg *hover, const hns_hover_buffers *buffers, const float *action,
                   void *stream) {
    int rc = hover_check(cfg, hover, buffers);
    if (rc != HNS_OK) return rc;
    if (!action) { set_error("hns_hover_step: null action"); return HNS_ERR_INVALID_ARG; }
    hns::HoverParams p;
    p.cfg = *cfg; p.hover = *hover; p.buf = *buffers; p.action = action; p.reset_mask = nullptr;
    p.seed_lo = p.seed_hi = p.epoch = 0;
    hipLaunchKernelGGL(hns::hns_hover_step_kernel, dim3((cfg->num_envs + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_hover_reset(const hns_cfg *cfg, const hns_hover_cfg *hover, const hns_hover_buffers *buffers,
                    const uint8_t *reset_mask, uint64_t seed, uint32_t epoch, void *stream) {
    int rc = hover_check(cfg, hover, buffers);
    if (rc != HNS_OK) return rc;
    hns::HoverParams p;
    p.cfg = *cfg; p.hover = *hover; p.buf = *buffers; p.action = nullptr; p.reset_mask = reset_mask;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32); p.epoch = epoch;
    hipLaunchKernelGGL(hns::hns_hover_reset_kernel, dim3((cfg->num_envs + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_raycast(hns_env *env, int num_rays, float max_range, float *out, void *stream) {
    if (!env || !out) { set_error("hns_raycast: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_raycast: buffers not bound"); return HNS_ERR_NOT_BOUND; }
    if (num_rays < 1 || num_rays > 1024 || !(max_range > 0.0f)) { set_error("hns_raycast: num_rays in [1,1024], max_range > 0"); return HNS_ERR_INVALID_ARG; }
    hns::RayParams p;
    p.cfg = env->cfg;
    p.drone_state = env->buf.drone_state;
    p.cylinders = env->buf.cylinders;
    p.out = out;
    p.num_rays = num_rays;
    p.max_range = max_range;
    const int A = env->cfg.num_agents, C = env->cfg.num_cylinders;
    int epb = 1024 / (A * num_rays);                 // ~4 rays per thread
    if (epb < 1) epb = 1;
    if (epb > 64) epb = 64;
    p.envs_per_block = epb;
    size_t lds = (size_t)epb * (C * 3 + A * 4) * sizeof(float);
    hipLaunchKernelGGL(hns::hns_raycast_kernel, dim3((env->cfg.num_envs + epb - 1) / epb), dim3(256), lds,
                       static_cast<hipStream_t>(stream), p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_set_v_prey(hns_env *env, float v_prey) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->cfg.v_prey = v_prey;
    return upload_step_params(env);
}
int hns_set_smoothness_coef(hns_env *env, float coef) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->cfg.smoothness_coef = coef;
    return upload_step_params(env);
}
int hns_set_reset_epoch(hns_env *env, uint32_t epoch) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->epoch = epoch;
    return HNS_OK;
}
uint32_t hns_get_reset_epoch(const hns_env *env) { return env ? env->epoch : 0u; }

// Fixture injection / read-back: copies between HOST arrays and the bound device buffers, field
// by field (null host fields are skipped), asynchronously on `stream`.
int hns_refresh_derived_state(hns_env *env, void *stream) {
    if (!env) { set_error("hns_refresh_derived_state: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_refresh_derived_state: hns_bind first"); return HNS_ERR_NOT_BOUND; }
    const int n = env->cfg.num_envs * env->cfg.num_agents;
    hipLaunchKernelGGL(hns::hns_refresh_los_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env->cfg, env->buf);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

static int copy_state(hns_env *env, const hns_buffers *host, void *stream, bool to_device) {
    if (!env || !host) { set_error("hns_set_state/hns_get_state: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_set_state/hns_get_state: hns_bind first"); return HNS_ERR_NOT_BOUND; }
    const hns_cfg &c = env->cfg;
    const size_t E = (size_t)c.num_envs, A = (size_t)c.num_agents, C = (size_t)c.num_cylinders, K = (size_t)c.obs_max_cylinder;
    const size_t NT = c.num_targets == 2 ? 2 : 1, SD = c.num_targets == 2 ? 24 : HNS_SELF_DIM;
    const hns_buffers &d = env->buf;
    struct Field { const void *host; void *dev; size_t bytes; };
    const Field f[] = {
        {host->drone_state, d.drone_state, E * A * 13 * 4}, {host->throttle, d.throttle, E * A * 16}, {host->pid_integ, d.pid_integ, E * A * 16},
        {host->pid_last_rate, d.pid_last_rate, E * A * 16}, {host->prev_action, d.prev_action, E * A * 16},
        {host->target_pos, d.target_pos, E * NT * 12}, {host->target_vel, d.target_vel, E * NT * 12}, {host->cylinders, d.cylinders, E * C * 12},
        {host->progress, d.progress, E * 4}, {host->stats, d.stats, (size_t)HNS_NUM_STATS * E * 4}, {host->obs_self, d.obs_self, E * A * SD * 4},
        {host->obs_others, d.obs_others, E * A * (A - 1) * 12}, {host->obs_cylinders, d.obs_cylinders, E * A * K * 20},
        {host->state_drones, d.state_drones, E * A * SD * 4}, {host->reward, d.reward, E * A * 4}, {host->action_error, d.action_error, E * A * 4},
        {host->done, d.done, E}, {host->detect, d.detect, E}, {host->nonfinite, d.nonfinite, 4}, {host->ctbr, d.ctbr, E * A * 16}, {host->target_rate, d.target_rate, E * A * 16}};
    for (const Field &x : f) {
        if (!x.host || !x.dev || x.bytes == 0) continue;
        if (to_device) HNS_CHECK_HIP(hipMemcpyAsync(x.dev, x.host, x.bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        else HNS_CHECK_HIP(hipMemcpyAsync(const_cast<void *>(x.host), x.dev, x.bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    }
    if (to_device) return hns_refresh_derived_state(env, stream);   // the line-of-sight column belongs to the positions just uploaded
    return HNS_OK;
}
int hns_set_state(hns_env *env, const hns_buffers *host, void *stream) { return copy_state(env, host, stream, true); }
int hns_get_state(hns_env *env, const hns_buffers *host, void *stream) { return copy_state(env, host, stream, false); }

int hns_set_phase_profile(hns_env *env, unsigned long long *device_buf) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->prof = device_buf;
    return upload_step_params(env);
}

int hns_enable_timing(hns_env *env, int on) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->timing = on < 0 ? 0 : on;
    if (env->timing > 0) {
        // event pairs for the first timed launches are made HERE, not inside the region the caller is about to time
        // (only when the caller's current device is the env's, as for a launch; otherwise they are made at the first timed launch)
        int dev = -1;
        while (hipGetDevice(&dev) == hipSuccess && dev == env->device && env->pool.size() < 16) {
            std::pair<hipEvent_t, hipEvent_t> ev;
            HNS_CHECK_HIP(hipEventCreate(&ev.first));
            HNS_CHECK_HIP(hipEventCreate(&ev.second));
            env->pool.push_back(ev);
        }
    }
    return HNS_OK;
}

float hns_step_kernel_ms(hns_env *env, int *num_launches) {
    if (num_launches) *num_launches = 0;
    if (!env || env->events.empty()) return -1.0f;
    if (hipEventSynchronize(env->events.back().second) != hipSuccess) return -1.0f;
    double total = 0.0;
    int n = 0;
    for (auto &p : env->events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { total += ms; ++n; }
        env->pool.push_back(p);
    }
    env->events.clear();
    if (num_launches) *num_launches = n;
    return n ? (float)(total / n) : -1.0f;
}

}  // extern "C"
