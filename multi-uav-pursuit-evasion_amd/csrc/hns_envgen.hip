// hns_envgen.hip — device side of the Adaptive Environment Generator, gfx950.
//
// Reference: GenBuffer in omni_drones/envs/hide_and_seek/hideandseek_envgen.py:209-377 — host numpy
// with a Python loop per task (`samplenearby`, :316-370) and DGL's farthest_point_sampler for
// trimming the history to 5000 entries (:291-304).  At 65 536 envs that is seconds per task batch;
// the step kernel runs a whole 800-step episode in 23 ms.  Two kernels replace it:
//
//   hns_fps_kernel     : farthest-point sampling of k of n points.  FPS is k strictly sequential
//                        rounds (update min-distance to the newest sample, take the arg-max), so the
//                        cost is the per-round latency.  One persistent launch, one workgroup per
//                        CU, points partitioned over ALL threads of the chip with their running
//                        min-distances in registers; per round every workgroup publishes its
//                        candidate as one tagged 8-byte granule (agent-scope store), every workgroup
//                        sweeps all candidates (all loads in flight at once) and computes the same global arg-max — one fabric hop
//                        per round, no grid barrier, no atomics (MI355X_MICROARCH "R2": the data is
//                        the flag; slots are double-buffered by round parity).  Every spin is bounded.
//   hns_perturb_kernel : one thread per task: draw a history entry, perturb, clip, grid sanity check,
//                        retry (Philox4x32 stream per task; the oracle reproduces it bit for bit).
#include <hip/hip_runtime.h>

#include <string>

#include "hns_device.h"
#include "hns_host.h"

namespace hns {

constexpr int kFpsThreads = 256;
constexpr int kFpsMaxGroups = 256;          // one workgroup per CU
constexpr int kFpsMaxPerThread = 8;         // points per thread (registers): n <= 256*256*8
constexpr unsigned kFpsSpinLimit = 1u << 20;     // ~1 s of polling before a workgroup gives up

typedef __attribute__((address_space(1))) unsigned long long gu64;

struct FpsParams {
    const float *points;   // [n, d]
    int n, d, k, start, groups;
    int in_lds;            // the workgroup's points are staged in LDS (rows of d+1 floats: conflict-free)
    int xcds;              // hns_fps_xcd_kernel: workgroups with blockIdx % 8 < xcds work (1 or 2 XCDs)
    int batch;             // hns_fps_xcd_kernel: candidates per exchange (1 .. kFxB; below)
    int32_t *out_idx;      // [k]
    unsigned long long *scratch;   // [0]: error word; [8 ..): granules [2 parity][groups]
};


// One round's all-to-all: reduce the workgroup's candidate, publish it as one tagged granule (slots double-buffered by round parity),
// sweep every workgroup's granule and take the same arg-max everywhere.  Returns the winner's index (its coordinates are in sQ when
// sQ is given), or -1 after reporting that a workgroup never showed up (every spin is bounded).
template <int THREADS>
HNS_DEV int fps_exchange(const FpsParams &p, gu64 *gran, int G, int g_self, int r, unsigned long long best, unsigned long long *s_best,
                         int *s_cur, int *s_fail, float *sQ, float *warm = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = p.d;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    const unsigned long long tag = (unsigned long long)(r % 4095 + 1) << 52;
    gu64 *slot = gran + (size_t)(r & 1) * G;
    if (wave == 0) {                                    // the waves' candidates: one per lane, then a shuffle tree (was a serial loop on one lane)
        best = lane < THREADS / 64 ? s_best[lane] : 0ull;
#pragma unroll
        for (int off = THREADS / 128; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o > best ? o : best;
        }
    }
    if (tid == 0) {
        __hip_atomic_store(slot + g_self, tag | best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (warm && best != 0) {                      // (a workgroup without points has no candidate: nothing to touch)
            // pull this workgroup's candidate row into the XCD's L2 while the exchange is in flight: if it wins, every workgroup of
            // this XCD fetches its coordinates next (scalar loads, on the critical path of the round) and finds them there
            const int ci = (int)(0xFFFFFu - (unsigned)(best & 0xFFFFFu));
            const float *row = p.points + (size_t)ci * d;          // (4-byte loads: rows of other widths than 36 are not 16-byte aligned)
            float a = row[0];
            for (int c = 8; c < d; c += 8) a += row[c];             // one word per 32 bytes touches every line of the row
            *warm += a + row[d - 1];
        }
    }
    // sweep every workgroup's candidate (wave 0): all loads in flight at once, same arg-max everywhere
    if (wave == 0) {
        unsigned long long v[kFpsMaxGroups / 64];
        bool fail = false;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < kFpsMaxGroups / 64; ++u) {
                const int g = u * 64 + lane;
                v[u] = g < G ? __hip_atomic_load(slot + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            }
#pragma unroll
            for (int u = 0; u < kFpsMaxGroups / 64; ++u) ok = ok && ((v[u] >> 52) == (tag >> 52));
            if (__all(ok)) break;
            if (++spins > kFpsSpinLimit) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        unsigned long long gb = 0;
#pragma unroll
        for (int u = 0; u < kFpsMaxGroups / 64; ++u) {
            const unsigned long long c52 = v[u] & 0xFFFFFFFFFFFFFull;
            gb = c52 > gb ? c52 : gb;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(gb, off);
            gb = o > gb ? o : gb;
        }
        const int gi = (int)(0xFFFFFu - (unsigned)(gb & 0xFFFFFu));
        if (lane == 0) { *s_cur = gi; if (fail) *s_fail = 1; }
        // the winner's coordinates for the next round (immutable input: plain loads)
        if (!fail && sQ)
            for (int c = lane; c < d; c += 64) sQ[c] = p.points[(size_t)gi * d + c];
    }
    __syncthreads();
    if (*s_fail) {                                   // a workgroup never showed up: give up loudly, never hang
        if (tid == 0) __hip_atomic_store((gu64 *)p.scratch, 1ull + (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
    }
    return *s_cur;
}

__global__ __launch_bounds__(kFpsThreads) void hns_fps_kernel(const FpsParams p) {
    extern __shared__ __align__(16) float s_dyn[];     // [d] the newest sample, then the staged points
    __shared__ unsigned long long s_best[kFpsThreads / 64];
    __shared__ int s_cur;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int G = p.groups, gtid = blockIdx.x * kFpsThreads + tid, stride = G * kFpsThreads;
    gu64 *gran = (gu64 *)(p.scratch + 8);
    float dist[kFpsMaxPerThread];
#pragma unroll
    for (int j = 0; j < kFpsMaxPerThread; ++j) dist[j] = kInf;
    if (tid == 0) s_fail = 0;
    const int d = p.d, ld = d + 1;
    float *sQ = s_dyn, *sPts = s_dyn + ((d + 3) & ~3);
    if (p.in_lds) {                                     // this thread's points, row (j*256 + tid)
#pragma unroll
        for (int j = 0; j < kFpsMaxPerThread; ++j) {
            const int i = gtid + j * stride;
            if (i < p.n)
                for (int c = 0; c < d; ++c) sPts[(j * kFpsThreads + tid) * ld + c] = p.points[(size_t)i * d + c];
        }
    }
    int cur = p.start;
    for (int c = tid; c < d; c += kFpsThreads) sQ[c] = p.points[(size_t)cur * d + c];
    __syncthreads();
    for (int r = 0; r < p.k; ++r) {
        if (blockIdx.x == 0 && tid == 0) p.out_idx[r] = cur;
        if (r == p.k - 1) break;
        // ---- distances to the newest sample (in sQ), running minimum, local arg-max -----------------
        // candidates travel as ONE 64-bit word: [tag 12 | key 32 | 0xFFFFF - index 20]; key = 0 for a chosen point,
        // else bits(distance) + 1 (non-negative floats order like their bit patterns), so the arg-max with
        // ties -> lower index is an integer max
        unsigned long long best = 0;
#pragma unroll
        for (int j = 0; j < kFpsMaxPerThread; ++j) {
            const int i = gtid + j * stride;
            if (i < p.n) {
                const float *x = p.in_lds ? sPts + (j * kFpsThreads + tid) * ld : p.points + (size_t)i * d;
                float acc = 0.0f;                     // one sequential fmaf chain over the coordinates (= the oracle)
                int c = 0;
                for (; c + 4 <= d; c += 4) {
                    const float x0 = x[c], x1 = x[c + 1], x2 = x[c + 2], x3 = x[c + 3];
                    const float d0 = x0 - sQ[c], d1 = x1 - sQ[c + 1], d2 = x2 - sQ[c + 2], d3 = x3 - sQ[c + 3];
                    acc = HNS_FMA(d0, d0, acc); acc = HNS_FMA(d1, d1, acc); acc = HNS_FMA(d2, d2, acc); acc = HNS_FMA(d3, d3, acc);
                }
                for (; c < d; ++c) {
                    const float df = x[c] - sQ[c];
                    acc = HNS_FMA(df, df, acc);
                }
                // a chosen point leaves the pool (-1 never wins), so the k indices are distinct even among duplicates
                const float m = (i == cur) ? -1.0f : (acc < dist[j] ? acc : dist[j]);
                dist[j] = m;
                const unsigned long long key = m < 0.0f ? 0ull : (unsigned long long)__float_as_uint(m) + 1ull;
                const unsigned long long cand = (key << 20) | (unsigned long long)(0xFFFFFu - (unsigned)i);
                best = cand > best ? cand : best;
            }
        }
        cur = fps_exchange<kFpsThreads>(p, gran, G, blockIdx.x, r, best, s_best, &s_cur, &s_fail, sQ);
        if (cur < 0) return;
    }
}

// XCD-local variant for the generator's shapes (up to 36 coordinates; up to 65 536 points on one XCD, up to 131 072 on two).  The per-round exchange between 32
// workgroups of ONE XCD costs 0.8 us against 2.6 us across the chip (tools/microbench/exchange_latency.hip) — with the same agent-scope
// stores and loads, so the result does not depend on where the workgroups actually land, only the time does.  The points (10 MB for
// 70 000 tasks) fit neither that XCD's LDS nor its L2: they live in registers, two per thread at 1024 threads per CU (a third one spills; at most 36
// coordinates = 3 pursuers + evader + 8 cylinders, the shape of BASELINE config 4; wider tasks take the chip-wide kernel); the newest
// sample's coordinates are workgroup-uniform and come in through scalar loads (`points` is immutable input).  Launched as 8 x 32 workgroups;
// those with blockIdx % 8 != 0 — by the round-robin dispatch, the ones on the other XCDs — leave at once.
constexpr int kFxThreads = 1024, kFxGroups = 32, kFxStride = 8, kFxPts = 2, kFxD = 36;

// Several samples per exchange (round 4).  Sequential farthest-point sampling is k strictly dependent rounds, and a round costs its
// latency (exchange 0.8 us + the dependent fetch of the winner + two workgroup barriers), not its arithmetic.  But the NEXT samples are
// often already decided: let p1 > p2 > ... be the candidates in key order (min-distance, ties -> lower index) BEFORE p1 is applied.  Applying
// p1 can only lower keys.  If dist(p2, p1) >= d(p2), p2's key does not move, every other key was below it and stays below it: p2 IS the
// next sample of the sequential algorithm.  By induction p_m is accepted when dist(p_m, p_j) >= d(p_m) for all accepted j < m; at the
// first candidate that fails the prefix ends (its key drops, the next sample may be any point).  So an exchange carries every workgroup's best
// candidates (round 4: its top 8; now see kFxT), every workgroup derives the same global top kFxB, wave 0 evaluates the kFxB (kFxB - 1) / 2 pair
// distances with the SAME sequential fma chain the update uses (so ">= d" decides exactly what min(d, dist) would), and the accepted
// prefix is applied in one round.  The indices are those of sequential sampling, bit for bit (tests/test_hip_envgen.py against the oracle's
// sequential loop); only the number of exchanges changes: ~k / (mean accepted) instead of k.
constexpr int kFxB = 16;                   // (8 until round 5: with the selections cheap, the accepted prefix was capped 93 % of the time)
constexpr int kFxPairs = kFxB * (kFxB - 1) / 2;
// ... of which a WORKGROUP contributes at most kFxT, plus a bound (round 5).  The global top 8 of 70 000 points come from 6-8 different workgroups
// almost always, yet every wave used to extract its top 8 (eight serial wave maxima), every workgroup its top 8 of those, and wave 0 the global top 8
// of 8 x workgroups granules — two thirds of an exchange's 12 us went into selections (tools/fps_phases.py).  Now a wave extracts its top kFxT + 1, the
// workgroup publishes its top kFxT as candidates and its (kFxT + 1)-th key as a BOUND: every point of the workgroup that is not published ranks below it.
// The global selection runs over the candidates only and stops at the first winner that does not beat the largest bound — below that, an unpublished
// point could rank higher, so the sequential algorithm's next sample is not decided by what was exchanged.  The first winner always beats every bound
// (the largest key overall is some workgroup's first candidate), so every exchange yields a sample; the indices stay those of sequential sampling, only
// (rarely: three of the global top 8 in one workgroup, ~1.4 % of the exchanges with 64 workgroups) an exchange ends a few samples early.
constexpr int kFxT = 2, kFxP = kFxT + 1;


#ifdef FPS_PHASES
#define FPS_STAMP(i) do { if ((threadIdx.x >> 6) == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); fps_ph[i] += (unsigned)(t_ - fps_last); fps_last = t_; if ((i) == 6) fps_nx += 1; } } while (0)
__device__ unsigned fps_ph[8];
__device__ unsigned fps_nx;
__device__ unsigned long long fps_last;
#else
#define FPS_STAMP(i) do {} while (0)
#endif

// Maximum of a 32-bit value over the wave, uniform result: the row_shr / row_bcast ladder on the VALU's data-parallel primitives (six
// v_max_u32_dpp and one v_readlane).  `__shfl_xor` trees go through ds_bpermute — an LDS round trip per step, two per 64-bit key: twelve
// of those max-reductions per exchange were 4-5 us of a 10 us exchange (measured), this form is ~60 cycles each.
HNS_DEV unsigned wave_max_u32(unsigned v) {
#define HNS_DPP_MAX(ctrl, rows) { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false); v = o_ > v ? o_ : v; }
    HNS_DPP_MAX(0x111, 0xf)      // row_shr:1
    HNS_DPP_MAX(0x112, 0xf)      // row_shr:2
    HNS_DPP_MAX(0x114, 0xf)      // row_shr:4
    HNS_DPP_MAX(0x118, 0xf)      // row_shr:8   -> lane 15 of every row holds its row's maximum
    HNS_DPP_MAX(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    HNS_DPP_MAX(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's
#undef HNS_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// ... of a candidate key (min-distance bits + 1 in bits 20..51, 0xFFFFF - index below): the distance part first, then the index part among
// the lanes that hold that distance (ties -> lower index)
// (round 5: the index part takes the second ladder only when several lanes hold the winning distance — equal minimum distances, or no candidate at
//  all; otherwise it is one ballot and one v_readlane.  These maxima run 24 times per exchange on a wave that issues alone, one dependent
//  instruction per ~6 cycles: the exchange's serial path, not the distance updates, is what a trim costs — 10 000 points take as long as 70 000.)
HNS_DEV unsigned long long wave_max_key(unsigned long long k) {
    const unsigned hi = (unsigned)(k >> 20), lo = (unsigned)k & 0xFFFFFu;
    const unsigned mh = wave_max_u32(hi);
    const unsigned long long eq = __ballot(hi == mh);
    unsigned ml;
    if (__popcll(eq) == 1) ml = (unsigned)__builtin_amdgcn_readlane((int)lo, (int)__builtin_ctzll(eq));
    else ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 20) | (unsigned long long)ml;
}

// The workgroup's top `pp` keys (its candidates and, last, its bound), left in s_top[0 .. kFxP) (LDS: whatever is indexed by a run-time value lives there — a register array indexed
// by the lane or by a loop counter ends up in scratch memory): every wave takes its own top nb (nb wave maxima, the lane that owns a winner pops
// it from its sorted pair), parks them in LDS, and after ONE workgroup barrier every wave takes the top nb of those (THREADS / 64) x kFxB
// values — one per lane at 1024 threads.
template <int THREADS>
HNS_DEV void fps_top(unsigned long long (&mine)[2], int pp, unsigned long long *s_wtop, unsigned long long *s_top, bool wave_active) {
    static_assert(THREADS / 64 * kFxP <= 64, "one parked key per lane");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave_active) {
#pragma unroll
        for (int pass = 0; pass < kFxP; ++pass) {
            unsigned long long best = 0ull;
            if (pass < pp) {
                best = wave_max_key(mine[0]);
                if (best != 0ull && mine[0] == best) { mine[0] = mine[1]; mine[1] = 0ull; }
            }
            if (lane == 0) s_wtop[wave * kFxP + pass] = best;
        }
    } else if (lane < kFxP) {
        s_wtop[wave * kFxP + lane] = 0ull;                   // a wave that holds no point has no candidate (and spends no issue slots on finding one)
    }
    __syncthreads();
    FPS_STAMP(1);
    unsigned long long v0 = lane < THREADS / 64 * kFxP ? s_wtop[lane] : 0ull;
    if (wave != 0) return;                                   // (the publishers are threads 0 .. pp - 1)
#pragma unroll
    for (int pass = 0; pass < kFxP; ++pass) {
        unsigned long long best = 0ull;
        if (pass < pp) {
            best = wave_max_key(v0);
            v0 = (best != 0ull && v0 == best) ? 0ull : v0;
        }
        if (lane == 0) s_top[pass] = best;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    FPS_STAMP(2);
}

// One exchange: publish the workgroup's candidates and its bound, sweep everybody's, take the global top nb among the candidates that beat the largest
// bound, accept the prefix that sequential sampling would select next (at most max_accept).  Returns the number accepted (their indices in s_acc), or
// -1 after reporting that a workgroup never showed up.  Granules of one parity: [candidates: workgroup x kFxT][bounds: workgroup].
// `pairs`: this lane's two candidate pairs (a < b), packed a0 | b0 << 8 | a1 << 16 | b1 << 24 — pair index lane and lane + 64 (the latter for lanes < 56).
HNS_DEV bool fps_bits_set(unsigned long long m0, unsigned long long m1, int base, int n) {      // bits [base, base + n) of the 128-bit mask (m1:m0), n <= 16
    const unsigned long long want = (1ull << n) - 1ull;
    if (base + n <= 64) return ((m0 >> base) & want) == want;
    if (base >= 64) return ((m1 >> (base - 64)) & want) == want;
    const int lown = 64 - base;
    return (m0 >> base) == ((1ull << lown) - 1ull) && (m1 & ((1ull << (n - lown)) - 1ull)) == ((1ull << (n - lown)) - 1ull);
}

template <int THREADS, int XM>
HNS_DEV int fps_exchange_b(const FpsParams &p, gu64 *gran, int G, int g_self, int r, const unsigned long long *s_top, int nb, int max_accept,
                           int *s_acc, unsigned *s_key, int *s_nacc, int *s_fail, float *s_rows, float *warm, unsigned pairs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = p.d;
    const unsigned long long tag = (unsigned long long)(r % 4095 + 1) << 52;
    const int tp = nb < kFxT ? nb : kFxT;                  // this workgroup's candidates; its bound is key number tp of s_top
    gu64 *slot = gran + (size_t)(r & 1) * G * kFxP;
    if (tid <= tp) {
        const unsigned long long mine = s_top[tid];
        __hip_atomic_store(tid < tp ? slot + g_self * kFxT + tid : slot + G * kFxT + g_self, tag | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine != 0ull && tid < tp) {
            // pull the candidate's row into the XCD's L2 while the exchange is in flight: if it wins, every workgroup fetches it next
            const int ci = (int)(0xFFFFFu - (unsigned)(mine & 0xFFFFFu));
            const float *row = p.points + (size_t)ci * d;
            float a = row[0];
            for (int c = 8; c < d; c += 8) a += row[c];
            *warm += a + row[d - 1];
        }
    }
    if (wave == 0) {
        constexpr int UC = kFxGroups * XM * kFxT / 64, UB = (kFxGroups * XM + 63) / 64;      // candidate / bound granules per lane with XM XCDs at work
        static_assert(UC * 64 == kFxGroups * XM * kFxT && (UC == 2 || UC == 4), "two or four candidate granules per lane");
        unsigned long long v[UC], bv[UB];
        const int ncand = G * kFxT;
        bool fail = false;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int g = u * 64 + lane;                    // candidate (workgroup g / kFxT, number g % kFxT): only the first tp of a workgroup are written
                const bool live = g < ncand && (g % kFxT) < tp;
                v[u] = live ? __hip_atomic_load(slot + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int g = u * 64 + lane;
                bv[u] = g < G ? __hip_atomic_load(slot + ncand + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            }
#pragma unroll
            for (int u = 0; u < UC; ++u) ok = ok && ((v[u] >> 52) == (tag >> 52));
#pragma unroll
            for (int u = 0; u < UB; ++u) ok = ok && ((bv[u] >> 52) == (tag >> 52));
            if (__all(ok)) break;
            if (++spins > kFpsSpinLimit) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        FPS_STAMP(3);
        // the largest bound: no unpublished point ranks above it
        unsigned long long gbound = 0ull;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned long long b52 = bv[u] & 0xFFFFFFFFFFFFFull;
            gbound = b52 > gbound ? b52 : gbound;
        }
        gbound = wave_max_key(gbound);
        // this lane's candidates in descending order (one compare-exchange, or the five of the four-element network): a pass then compares heads only,
        // and the lane whose head wins moves up its next one — per pass two instructions per granule used to go into a max and a clear over all of them
#pragma unroll
        for (int u = 0; u < UC; ++u) v[u] &= 0xFFFFFFFFFFFFFull;
#define FPS_CE(i, j) { const unsigned long long hi_ = v[i] > v[j] ? v[i] : v[j], lo_ = v[i] > v[j] ? v[j] : v[i]; v[i] = hi_; v[j] = lo_; }
        if constexpr (UC == 2) { FPS_CE(0, 1) }
        else { FPS_CE(0, 1) FPS_CE(2, 3) FPS_CE(0, 2) FPS_CE(1, 3) FPS_CE(1, 2) }
#undef FPS_CE
        // the global top nb.  A winner's row is requested the moment it is known (lane c loads coordinate c; the loads of winner m fly while winners
        // m + 1 ... are still being found) and lands in LDS behind the last pass.
        float rowv[kFxB];
        bool more = !fail;
#pragma unroll
        for (int pass = 0; pass < kFxB; ++pass) {
            unsigned long long gb = 0ull;
            if (more && pass < nb) {                            // (uniform)
                gb = wave_max_key(v[0]);
                gb = gb > gbound ? gb : 0ull;                 // at or below the largest bound an unpublished point could rank higher: the decided prefix ends here
                if (gb != 0ull && v[0] == gb) {
#pragma unroll
                    for (int u = 0; u + 1 < UC; ++u) v[u] = v[u + 1];
                    v[UC - 1] = 0ull;
                }
                more = gb != 0ull;                            // keys only fall from here on
            }
            const int gim = gb != 0ull ? (int)(0xFFFFFu - (unsigned)(gb & 0xFFFFFu)) : -1;
            if (lane == 0) {
                s_acc[pass] = gim;
                s_key[pass] = (unsigned)(gb >> 20);                // min-distance bits + 1; 0 = no candidate (or a point already chosen)
            }
            rowv[pass] = (gim >= 0 && lane < d) ? p.points[(size_t)gim * d + lane] : 0.0f;      // (rows of kFxD floats, zero-padded)
        }
        FPS_STAMP(4);
        if (lane < kFxD) {
#pragma unroll
            for (int m = 0; m < kFxB; ++m) s_rows[m * kFxD + lane] = rowv[m];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the pairs' distances — the update's own chain, dist(x = candidate b, q = candidate a) — two pairs per lane; candidate b keeps its key under
        // candidate a iff dist >= d(b): one predicate per pair, gathered with two ballots
        const int a0 = pairs & 255, b0 = (pairs >> 8) & 255, a1 = (pairs >> 16) & 255, b1 = pairs >> 24;
        float acc0 = 0.0f, acc1 = 0.0f;
        if (!fail && nb > 1) {
            const float *xa0 = s_rows + a0 * kFxD, *xb0 = s_rows + b0 * kFxD, *xa1 = s_rows + a1 * kFxD, *xb1 = s_rows + b1 * kFxD;
#pragma unroll 4
            for (int c = 0; c < kFxD; ++c) {                       // (padding: 0 - 0 leaves the chain unchanged)
                const float df0 = xb0[c] - xa0[c], df1 = xb1[c] - xa1[c];
                acc0 = HNS_FMA(df0, df0, acc0);
                acc1 = HNS_FMA(df1, df1, acc1);
            }
        }
        const unsigned kb0 = s_key[b0], kb1 = s_key[b1], kme = s_key[lane < kFxB ? lane : 0];
        const float gd0 = kb0 != 0u ? __uint_as_float(kb0 - 1u) : kInf, gd1 = kb1 != 0u ? __uint_as_float(kb1 - 1u) : kInf;
        const unsigned long long m0 = __ballot(acc0 >= gd0), m1 = __ballot(lane < kFxPairs - 64 && acc1 >= gd1);
        const unsigned long long have = __ballot(lane < kFxB && kme != 0u);
        // accepted prefix: candidate m needs dist(m, j) >= d(m) for every j < m — bits m (m - 1) / 2 ... + m - 1 of the pair mask
        int nacc = (int)(have & 1ull);
        bool open = nacc == 1;
#pragma unroll
        for (int m = 1; m < kFxB; ++m) {
            const bool okm = open && m < nb && ((have >> m) & 1ull) != 0ull && fps_bits_set(m0, m1, m * (m - 1) / 2, m);
            open = okm;
            nacc += okm ? 1 : 0;
        }
        FPS_STAMP(5);
        nacc = nacc < max_accept ? nacc : max_accept;
        if (lane == 0) {
            *s_nacc = nacc;
            if (fail || nacc == 0) *s_fail = 1;                // (no candidate at all cannot happen while samples are still due: k <= n)
        }
    }
    __syncthreads();
    FPS_STAMP(6);
    if (*s_fail) {
        if (tid == 0) __hip_atomic_store((gu64 *)p.scratch, 1ull + (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
    }
    return *s_nacc;
}

// Rows of fewer than 36 coordinates are zero-padded to 36 in registers: a zero difference leaves the sequential fma chain unchanged, bit for
// bit (acc + 0 * 0 = acc for acc >= 0).  (Round 3 had a second instantiation for exactly 36 coordinates — 16-byte row loads and wide scalar
// loads of the newest sample; with the samples' rows in LDS the only difference left was the one-time load of the points, and the
// specialised form did not fit the 128 registers of a 1024-thread workgroup beside the exchange: 37 spilled registers, 28 ms per trim
// against 13 ms.)
// PTS points per thread (1 in every instantiation that ships: 32 768 points per XCD; with 2 the eight-candidate exchange does not fit the 128
// registers of a 1024-thread workgroup), XM = the most XCDs an instantiation serves.
template <int PTS, int XM>
__global__ __launch_bounds__(kFxThreads) void hns_fps_xcd_kernel(const FpsParams p) {
    if ((int)(blockIdx.x % kFxStride) >= p.xcds) return;
    __shared__ unsigned long long s_wtop[kFxThreads / 64 * kFxP];
    __shared__ __align__(16) float s_rows[kFxB * kFxD];            // the newest samples' coordinates (written by wave 0 in the exchange)
    __shared__ unsigned long long s_top[kFxP];
    __shared__ int s_acc[kFxB];                                    // the newest samples' indices (the first: `start`)
    __shared__ unsigned s_key[kFxB];                               // ... and their distance keys (exchange)
    __shared__ int s_nacc;
    __shared__ int s_fail;
    const int G = kFxGroups * p.xcds;
    const int tid = threadIdx.x, g_self = (blockIdx.x / kFxStride) * p.xcds + blockIdx.x % kFxStride;
    // Point (slot, workgroup) = index slot * G + workgroup, slot = tid (+ j * 1024): the points are dealt round-robin over the workgroups, so that every
    // workgroup holds the same number (70 536 points on 128 workgroups: 551-552 each, the first 9 of a workgroup's 16 waves) and the waves behind
    // a workgroup's last point skip the distance update and the candidate search — they used to run both on copies of the last point.  (Before: blocks
    // of 1024 consecutive points per workgroup, 59 of 128 workgroups without any, the others full: four busy waves per SIMD where there are now two.)
    // Which thread holds a point does not enter the result: candidates are ordered by (distance, index).
    const int d = p.d;
    const int nb = p.batch < 1 ? 1 : (p.batch > kFxB ? kFxB : p.batch);
    const bool wave_active = (tid & ~63) * G + g_self < p.n;       // the wave's first lane holds a point (slot of lane 0, first pass)
    gu64 *gran = (gu64 *)(p.scratch + 8);
    float x[PTS][kFxD], dist[PTS];
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        dist[j] = kInf;
        int i = (tid + j * kFxThreads) * G + g_self;
        i = i < p.n ? i : p.n - 1;                      // beyond the end: a copy of the last point, never a candidate (see below)
        const float *row = p.points + (size_t)i * d;
#pragma unroll
        for (int c = 0; c < kFxD; ++c) x[j][c] = c < d ? row[c < d ? c : 0] : 0.0f;
    }
    if (tid == 0) s_fail = 0;
    if (tid < kFxD) {                                   // the first sample's row
        int i0 = p.start;
        s_rows[tid] = tid < d ? p.points[(size_t)i0 * d + tid] : 0.0f;
    }
    if (tid == 0) s_acc[0] = p.start;
    // this lane's two candidate pairs (a < b) for the exchange: pair index lane and lane + 64
    unsigned pairs = 0u;
    {
        const int lane = tid & 63;
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const int idx = lane + 64 * which;
            int b = 1;
            while ((b + 1) * b / 2 <= idx) ++b;
            const int a = idx - b * (b - 1) / 2;
            if (idx < kFxPairs) pairs |= ((unsigned)a | ((unsigned)b << 8)) << (16 * which);
        }
    }
    int ncur = 1, nout = 0;
    float warm = 0.0f;                                  // sum of the rows touched to warm the L2 (kept alive by the store below)
    __syncthreads();
    for (int r = 0;; ++r) {                             // r counts exchanges, nout the samples written
        if (g_self == 0 && tid < ncur) {
            p.out_idx[nout + tid] = s_acc[tid];
        }
        nout += ncur;
        if (nout >= p.k) break;
        FPS_STAMP(7);
        // the newest samples' coordinates sit in LDS (uniform addresses: broadcast reads, 4 at a time); every point's distance to each of
        // them stays ONE sequential fmaf chain over the coordinates (= the oracle).  Two samples side by side (round 5: two independent chains, 0.37 ->
        // 0.33 us per sample; a row beyond `ncur` — another candidate's, always present in LDS — is computed along and not applied).  What bounds the
        // update is the LDS: nine 16-byte broadcast reads per wave and sample, 9-16 waves per CU.  Measured and dropped: four samples side by side
        // (57 spilled registers under the 128 of a 1024-thread workgroup), and a sample's row carried in three registers with the coordinate taken
        // through the DPP row broadcast (v_sub_f32_dpp ... row_newbcast:c — a twelfth of the LDS traffic, the same 72 instructions, and slower:
        // 13.8 against 12.8 us per exchange, tools/fps_phases.py).
        constexpr int kGrp = 2;
        static_assert(kFxB % kGrp == 0, "the update works through the samples in whole groups");
#pragma unroll 1
        for (int m0 = 0; m0 < (wave_active ? ncur : 0); m0 += kGrp) {
            float acc[kGrp][PTS];
#pragma unroll
            for (int s2 = 0; s2 < kGrp; ++s2)
#pragma unroll
                for (int j = 0; j < PTS; ++j) acc[s2][j] = 0.0f;
            const float4 *qbase = reinterpret_cast<const float4 *>(s_rows + m0 * kFxD);
#pragma unroll
            for (int c0 = 0; c0 < kFxD; c0 += 4) {
                float4 q4[kGrp];
#pragma unroll
                for (int s2 = 0; s2 < kGrp; ++s2) q4[s2] = qbase[s2 * (kFxD / 4) + c0 / 4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int s2 = 0; s2 < kGrp; ++s2) {
                        const float qc = c == 0 ? q4[s2].x : c == 1 ? q4[s2].y : c == 2 ? q4[s2].z : q4[s2].w;
#pragma unroll
                        for (int j = 0; j < PTS; ++j) {
                            const float df = x[j][c0 + c] - qc;
                            acc[s2][j] = HNS_FMA(df, df, acc[s2][j]);
                        }
                    }
            }
#pragma unroll
            for (int s2 = 0; s2 < kGrp; ++s2) {
                if (m0 + s2 < ncur) {                             // (uniform)
                    const int cm = s_acc[m0 + s2];
#pragma unroll
                    for (int j = 0; j < PTS; ++j) {
                        const int i = (tid + j * kFxThreads) * G + g_self;
                        // a chosen point leaves the pool (-1 never wins), so the k indices are distinct even among duplicates
                        dist[j] = (i == cm) ? -1.0f : (acc[s2][j] < dist[j] ? acc[s2][j] : dist[j]);
                    }
                }
            }
        }
        FPS_STAMP(0);
        unsigned long long mine[2] = {0ull, 0ull};
#pragma unroll
        for (int j = 0; j < PTS; ++j) {
            const int i = (tid + j * kFxThreads) * G + g_self;
            const float m = dist[j];
            const unsigned long long key = m < 0.0f ? 0ull : (unsigned long long)__float_as_uint(m) + 1ull;
            const unsigned long long cand = (i < p.n && key != 0ull) ? ((key << 20) | (unsigned long long)(0xFFFFFu - (unsigned)i)) : 0ull;
            if (cand > mine[0]) { mine[1] = mine[0]; mine[0] = cand; }
            else if (cand > mine[1]) mine[1] = cand;
        }
        static_assert(PTS <= 2, "a thread's candidates are a sorted pair");
        fps_top<kFxThreads>(mine, (nb < kFxT ? nb : kFxT) + 1, s_wtop, s_top, wave_active);
        const int left = p.k - nout;
        ncur = fps_exchange_b<kFxThreads, XM>(p, gran, G, g_self, r, s_top, nb, left < kFxB ? left : kFxB, s_acc, s_key, &s_nacc, &s_fail, s_rows, &warm, pairs);
        if (ncur < 0) return;
    }
#ifdef FPS_PHASES
    if (g_self == 0 && tid < 6) p.scratch[2 + tid] = tid < 4 ? (unsigned long long)fps_ph[2 * tid] | ((unsigned long long)fps_ph[2 * tid + 1] << 32) : (tid == 4 ? (unsigned long long)fps_nx : 0ull);
#endif
    if (warm == -1.0f) p.scratch[1] = 1;               // never true (coordinates are normalised to [0, 1]): the loads above are not dead
}

// ---- samplenearby (hideandseek_envgen.py:316-370, grid check :187-207) -----------------------------
struct PerturbParams {
    hns_cfg cfg;
    const float *history;   // [n_hist, task_dim]
    float *out;             // [n_tasks, task_dim]
    int n_hist, n_tasks, expand_cylinders;
    float expand_step;
    uint32_t seed_lo, seed_hi;
};

// continuous -> grid cell exactly as the host GenBuffer does it (float64 rint, clip)
HNS_DEV int envgen_cell(double x, double grid_size, int num_grid) {
    int g = (int)__builtin_rint(x / grid_size) + num_grid / 2;
    return g < 0 ? 0 : (g > num_grid - 1 ? num_grid - 1 : g);
}

constexpr int kMaxBodies = HNS_MAX_AGENTS + 2 + HNS_MAX_CYLINDERS;      // pursuers, one or two evaders, cylinder slots

__global__ __launch_bounds__(256) void hns_perturb_kernel(const PerturbParams p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_tasks) return;
    const hns_cfg &c = p.cfg;
    // task vector: [pursuers | evader (two with the two-evader extension) | cylinder slots], three coordinates each
    const int A = c.num_agents, Cn = c.num_cylinders, NM = A + (c.num_targets == 2 ? 2 : 1), nb = NM + Cn, TD = 3 * nb, GN = c.grid_num, half = GN / 2;
    // the reference divides by the Python double 2*cylinder_size (0.2), not by its fp32 rounding: recover the
    // decimal the YAML holds (6 places) so that bodies exactly on a cell edge fall into the same cell
    const double gs = __builtin_rint((double)c.grid_size * 1e6) / 1e6;
    // bounds, :320-333 (incl. the reference's z window around max_height for drones / evader)
    const float cb = (float)((int)(c.arena_size / c.grid_size)) * c.grid_size;
    const float bxy = c.arena_size / 1.41421356237309515f - 0.1f;
    Rng rng{p.seed_lo, p.seed_hi, (uint32_t)t, 0x9E3779B9u, 0u, {0, 0, 0, 0}, 0};
    float *out = p.out + (size_t)t * TD;
    const float *origin = nullptr;
    for (int attempt = 0; attempt < 10; ++attempt) {
        // a fresh history entry per attempt keeps every task independent of the others
        // (the reference re-perturbs the same origin and, failing that, copies another task's result)
        int h = (int)(rng.uniform() * (float)p.n_hist);
        if (h > p.n_hist - 1) h = p.n_hist - 1;
        origin = p.history + (size_t)h * TD;
        int cells[kMaxBodies];
        bool ok = true;
        for (int b = 0; b < nb; ++b) {
            float v[3] = {origin[3 * b], origin[3 * b + 1], origin[3 * b + 2]};
            if (b < NM) {
                for (int j = 0; j < 3; ++j) v[j] += (rng.uniform() * 2.0f - 1.0f) * p.expand_step;
                v[0] = d_clamp(v[0], -bxy, bxy); v[1] = d_clamp(v[1], -bxy, bxy);
                v[2] = d_clamp(v[2], c.max_height - 0.1f, c.max_height + 0.1f);
            } else {
                if (p.expand_cylinders) {
                    for (int j = 0; j < 2; ++j) {
                        int s = (int)(rng.uniform() * 3.0f);
                        v[j] += (float)((s > 2 ? 2 : s) - 1) * c.grid_size;
                    }
                }
                v[0] = d_clamp(v[0], -cb, cb); v[1] = d_clamp(v[1], -cb, cb);
                v[2] = d_clamp(v[2], -20.0f, c.max_height * 0.5f);
            }
            out[3 * b] = v[0]; out[3 * b + 1] = v[1]; out[3 * b + 2] = v[2];
            const int gx = envgen_cell((double)v[0], gs, GN), gy = envgen_cell((double)v[1], gs, GN);
            const int dx = gx - half, dy = gy - half;
            if (dx * dx + dy * dy >= half * half) ok = false;           // outside the disc of free cells (:168-181)
            cells[b] = gx * GN + gy;
        }
        for (int b = 1; b < nb && ok; ++b)
            for (int b2 = 0; b2 < b; ++b2)
                if (cells[b] == cells[b2]) { ok = false; break; }
        if (ok) return;
    }
    // every attempt failed the grid check: fall back to the last origin unperturbed (a stored task)
    for (int j = 0; j < TD; ++j) out[j] = origin[j];
}

}  // namespace hns

extern "C" {

size_t hns_fps_scratch_bytes(void) { return (size_t)(8 + 2 * hns::kFxB * hns::kFpsMaxGroups) * sizeof(unsigned long long); }   // error word + granules [2 parities][<= 256 workgroups][<= 4]

int hns_fps(const float *points, int32_t n, int32_t d, int32_t k, int32_t start, int32_t *out_idx, void *scratch, void *stream) {
    if (!points || !out_idx || !scratch || n < 1 || d < 1 || k < 1 || k > n || start < 0 || start >= n) {
        hns_set_error("hns_fps: bad argument (need 1 <= k <= n, 0 <= start < n, non-null device pointers)");
        return HNS_ERR_INVALID_ARG;
    }
    const long cap = (long)hns::kFpsMaxGroups * hns::kFpsThreads * hns::kFpsMaxPerThread;
    if (n > cap) { hns_set_error("hns_fps: more than 524288 points"); return HNS_ERR_INVALID_ARG; }
    int dev = 0, cus = 0;
    HNS_CHECK_HIP(hipGetDevice(&dev));
    HNS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int groups = (n + hns::kFpsThreads - 1) / hns::kFpsThreads;        // every workgroup must be resident: <= one per CU
    const int max_groups = cus < hns::kFpsMaxGroups ? cus : hns::kFpsMaxGroups;
    if (groups > max_groups) groups = max_groups;
    if ((long)groups * hns::kFpsThreads * hns::kFpsMaxPerThread < n) { hns_set_error("hns_fps: too many points for this device"); return HNS_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    HNS_CHECK_HIP(hipMemsetAsync(scratch, 0, hns_fps_scratch_bytes(), s));
    hns::FpsParams p;
    p.points = points; p.n = n; p.d = d; p.k = k; p.start = start; p.groups = groups;
    p.out_idx = out_idx; p.scratch = (unsigned long long *)scratch; p.xcds = 0; p.batch = 1;
    // the generator's own shape: the XCD-local kernel (same results; HNS_FPS_KERNEL=chip keeps the chip-wide one, for A/B measurements)
    static const bool chip_only = [] { const char *e = getenv("HNS_FPS_KERNEL"); return e && e[0] == 'c'; }();
    if (!chip_only && d <= hns::kFxD && d >= 4 && n >= 2048 && cus >= hns::kFxGroups * hns::kFxStride) {
        // samples per exchange (HNS_FPS_BATCH=1 keeps one per exchange, for A/B measurements; the indices are the same either way)
        static const int batch = [] { const char *e = getenv("HNS_FPS_BATCH"); const int b = e ? atoi(e) : hns::kFxB; return b < 1 ? 1 : (b > hns::kFxB ? hns::kFxB : b); }();
        // XCDs at work, one point per thread (measured with four samples per exchange, 5000 of n samples; tools/lab/r04_batch7-8.sh): an exchange's fixed
        // cost grows with the XCDs it spans (3.7 / 4.4 / 6.5 / 7.8 us on 1 / 2 / 4 / 8), a sample's distance update shrinks (0.6 / 0.3 us on
        // two / four; 1.2 us on one XCD with two points per thread).  n <= 32 768: one XCD, <= 65 536: two, <= 131 072: four; beyond that the
        // chip-wide kernel.  HNS_FPS_XCDS=1|2|4 overrides (A/B).
        static const int forced = [] { const char *e = getenv("HNS_FPS_XCDS"); return e ? atoi(e) : 0; }();
        int xcds = (forced == 1 || forced == 2 || forced == 4) ? forced : 1;
        auto capacity = [&](int x) { return (long)x * hns::kFxGroups * hns::kFxThreads; };
        while (xcds < 4 && capacity(xcds) < n) xcds *= 2;
        if (capacity(xcds) >= n) {
            p.xcds = xcds; p.groups = hns::kFxGroups * xcds; p.in_lds = 0; p.batch = batch;
            auto fn = xcds <= 2 ? hns::hns_fps_xcd_kernel<1, 2> : hns::hns_fps_xcd_kernel<1, 4>;
            hipLaunchKernelGGL(fn, dim3(hns::kFxGroups * hns::kFxStride), dim3(hns::kFxThreads), 0, s, p);
            HNS_CHECK_HIP(hipGetLastError());
            return HNS_OK;
        }
    }
    const int per_thread = (n + groups * hns::kFpsThreads - 1) / (groups * hns::kFpsThreads);
    const size_t q_floats = (size_t)((d + 3) & ~3), pts_floats = (size_t)per_thread * hns::kFpsThreads * (d + 1);
    p.in_lds = (q_floats + pts_floats) * sizeof(float) <= 144 * 1024 ? 1 : 0;
    const size_t lds = (q_floats + (p.in_lds ? pts_floats : 0)) * sizeof(float);
    static thread_local unsigned long long lds_attr_devs = 0ull;       // the attribute is per device: remembered per device of this thread
    if (!(lds_attr_devs & (1ull << (dev & 63)))) {
        HNS_CHECK_HIP(hipFuncSetAttribute((const void *)hns::hns_fps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024 + 1024));
        lds_attr_devs |= 1ull << (dev & 63);
    }
    hipLaunchKernelGGL(hns::hns_fps_kernel, dim3(groups), dim3(hns::kFpsThreads), lds, s, p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_perturb_tasks(hns_env *env, const float *history, int32_t n_hist, float *tasks_out, int32_t n_tasks, int32_t expand_cylinders,
                      float expand_step, uint64_t seed, void *stream) {
    if (!env || !history || !tasks_out || n_hist < 1 || n_tasks < 0) { hns_set_error("hns_perturb_tasks: bad argument"); return HNS_ERR_INVALID_ARG; }
    if (n_tasks == 0) return HNS_OK;
    hns::PerturbParams p;
    p.cfg = env->cfg;
    p.history = history; p.out = tasks_out; p.n_hist = n_hist; p.n_tasks = n_tasks;
    p.expand_cylinders = expand_cylinders; p.expand_step = expand_step;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    hipLaunchKernelGGL(hns::hns_perturb_kernel, dim3((n_tasks + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

}  // extern "C"
