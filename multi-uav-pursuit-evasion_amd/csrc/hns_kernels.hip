// hns_kernels.hip — fused HideAndSeek environment step + reset for gfx950 (MI355X), and the
// C ABI of include/hns.h.
//
// Mapping: every env owns a power-of-two LANE GROUP of G lanes inside one
// wave64 (G = next pow2 >= A+1): lanes 0..A-1 are the pursuers, lane A is the "env lane"
// (evader + per-env statistics).  A 256-thread workgroup therefore holds EPB = 256/G whole envs;
// no env straddles a wave, so every intra-env exchange is a width-G shuffle / ballot, and the
// tensors keep the reference's [E,A,...] layouts: each workgroup's slice of every array is ONE
// contiguous byte range that is moved HBM<->LDS with 16-byte-per-lane coalesced accesses and
// picked apart / assembled per agent in LDS.
//
// One launch does the whole step (reference call tree: transforms.py:425-459 ->
// lee_position_controller.py:476-550 -> hideandseek.py:725-744 -> multirotor.py:466-508 ->
// rotor_group.py:55-71 -> [PhysX sim.step() replaced by d_integrate] -> hideandseek.py:746-917
// -> :919-1065).  No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "hns_device.h"

namespace hns {

constexpr int kThreads = 256;
constexpr int kMaxK = 4;   // top-k insertion network width (obs_max_cylinder <= 4)

template <int A>
struct Geo {
    static constexpr int G = (A + 1 <= 2) ? 2 : (A + 1 <= 4) ? 4 : 8;
    static constexpr int EPB = kThreads / G;
};

struct Params {
    hns_cfg cfg;
    hns_buffers buf;
    const float *action;        // step
    const uint8_t *reset_mask;  // reset (nullable)
    uint32_t seed_lo, seed_hi, epoch;
};

// LDS carve-up (float offsets, every region 16-byte aligned)
struct Lds {
    int ds, cyl, tp, tvel, stats, self, oth, ocyl, state, total;
};
__host__ __device__ inline int r4(int n) { return (n + 3) & ~3; }
__host__ __device__ inline Lds lds_layout(int EPB, int A, int C, int K, int with_state) {
    Lds L;
    int o = 0;
    L.ds = o;    o += r4(EPB * A * 13);
    L.cyl = o;   o += r4(EPB * C * 3);
    L.tp = o;    o += r4(EPB * 3);
    L.tvel = o;  o += r4(EPB * 3);
    L.stats = o; o += r4(HNS_NUM_STATS * EPB);
    L.self = o;  o += r4(EPB * A * HNS_SELF_DIM);
    L.oth = o;   o += r4(EPB * A * (A - 1) * 3);
    L.ocyl = o;  o += r4(EPB * A * K * 5);
    L.state = o; o += with_state ? r4(EPB * A * HNS_SELF_DIM) : 0;
    L.total = o;
    return L;
}

// ---- workgroup-cooperative contiguous copies (16 B per lane where alignment allows) ----------
HNS_DEV void coop_g2s(float *__restrict__ dst, const float *__restrict__ src, int n) {
    const int n4 = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) ? (n >> 2) : 0;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int i = threadIdx.x; i < n4; i += kThreads) d4[i] = s4[i];
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) dst[i] = src[i];
}
HNS_DEV void coop_s2g(float *__restrict__ dst, const float *__restrict__ src, int n) {
    const int n4 = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? (n >> 2) : 0;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int i = threadIdx.x; i < n4; i += kThreads) d4[i] = s4[i];
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) dst[i] = src[i];
}
// same, but only the elements of envs whose mask byte is set (per_env floats per env)
HNS_DEV void coop_s2g_masked(float *__restrict__ dst, const float *__restrict__ src, int n, int per_env,
                             const uint8_t *__restrict__ smask) {
    for (int i = threadIdx.x; i < n; i += kThreads)
        if (smask[i / per_env]) dst[i] = src[i];
}
// [HNS_NUM_STATS][E] rows <-> LDS [HNS_NUM_STATS][EPB]
template <int EPB, bool TO_LDS>
HNS_DEV void coop_stats(float *__restrict__ lds, float *__restrict__ g, int E, int e0, int nenv) {
    if ((E & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0 && nenv == EPB) {
        constexpr int Q = EPB / 4;
        for (int i = threadIdx.x; i < HNS_NUM_STATS * Q; i += kThreads) {
            int row = i / Q, c4 = i % Q;
            float4 *gp = reinterpret_cast<float4 *>(g + (size_t)row * E + e0) + c4;
            float4 *lp = reinterpret_cast<float4 *>(lds + row * EPB) + c4;
            if (TO_LDS) *lp = *gp; else *gp = *lp;
        }
    } else {
        for (int i = threadIdx.x; i < HNS_NUM_STATS * EPB; i += kThreads) {
            int row = i / EPB, col = i % EPB;
            if (col < nenv) {
                if (TO_LDS) lds[row * EPB + col] = g[(size_t)row * E + e0 + col];
                else g[(size_t)row * E + e0 + col] = lds[row * EPB + col];
            }
        }
    }
}

template <int G>
HNS_DEV float bcast(float v, int j) { return __shfl(v, j, G); }

// bits of the A agent lanes of this lane's group in a wave-wide ballot
template <int A, int G>
HNS_DEV unsigned group_bits(bool pred) {
    unsigned long long m = __ballot(pred);
    int shift = (threadIdx.x & 63) & ~(G - 1);
    return (unsigned)(m >> shift) & ((1u << A) - 1u);
}

struct ObsSide {
    bool blocked;     // this agent's line of sight (post-physics)
    bool bdetect;     // any agent of the env detects the evader
    int knn_idx[kMaxK];
    bool knn_masked[kMaxK];
};

// ---- A8: observation pass (agent lanes)  multirotor.py:599-633, hideandseek.py:746-917 ----------
// sDS holds the post-physics [A,13] records of this workgroup's envs.
template <int A>
HNS_DEV void obs_pass(const hns_cfg &c, int C, int K, int le, int a, bool is_agent, const Rigid &s, const V3 &tp,
                      float progress, const float *cyl, const float *sDS, float *sSelf, float *sOth, float *sOCyl,
                      float *sState, ObsSide &side) {
    constexpr int G = Geo<A>::G;
    float rtx = 0.f, rty = 0.f, rtz = 0.f;
    bool blocked = false, det = false;
    if (is_agent) {
        rtx = s.pos.x - tp.x; rty = s.pos.y - tp.y; rtz = s.pos.z - tp.z;
        float dist = d_norm3(rtx, rty, rtz);
        blocked = d_blocked(c, C, s.pos, tp, cyl);
        det = (dist < c.drone_detect_radius) && !blocked;
    }
    const bool det_any = group_bits<A, G>(det && is_agent) != 0u;     // :791
    side.blocked = blocked;
    side.bdetect = det_any;
    if (!is_agent) return;
    const float t = progress / (float)c.max_episode_length;          // :796
    const V3 ex = {1.0f, 0.0f, 0.0f}, ez = {0.0f, 0.0f, 1.0f};
    V3 heading = d_quat_rot<false>(s.q, ex);                          // multirotor.py:613-614
    V3 up = d_quat_rot<false>(s.q, ez);
    float o[HNS_SELF_DIM];
    o[0] = det_any ? rtx : c.mask_value; o[1] = det_any ? rty : c.mask_value; o[2] = det_any ? rtz : c.mask_value;
    o[3] = s.q.w; o[4] = s.q.x; o[5] = s.q.y; o[6] = s.q.z;
    o[7] = s.lin.x; o[8] = s.lin.y; o[9] = s.lin.z;
    o[10] = heading.x; o[11] = heading.y; o[12] = heading.z;
    o[13] = up.x; o[14] = up.y; o[15] = up.z;
    o[16] = t; o[17] = t; o[18] = t; o[19] = t;
    float4 *so = reinterpret_cast<float4 *>(sSelf + (le * A + a) * HNS_SELF_DIM);
#pragma unroll
    for (int i = 0; i < HNS_SELF_DIM / 4; ++i) so[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    if (sState) {                                                      // :871-886 (unmasked rpos)
        float4 *ss = reinterpret_cast<float4 *>(sState + (le * A + a) * HNS_SELF_DIM);
        ss[0] = make_float4(rtx, rty, rtz, o[3]);
#pragma unroll
        for (int i = 1; i < HNS_SELF_DIM / 4; ++i) ss[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
    // state_others: p_i - p_j, j != i ascending (:750-751, utils/torch.py:41-53)
    {
        float *w = sOth + (le * A + a) * (A - 1) * 3;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            if (j == a) continue;
            const float *rj = sDS + (le * A + j) * 13;
            w[0] = s.pos.x - rj[0]; w[1] = s.pos.y - rj[1]; w[2] = s.pos.z - rj[2];
            w += 3;
        }
    }
    // k nearest cylinders by 3-D distance - size; streaming stable insertion, ties -> lower index (:767-778)
    float bd[kMaxK];
    int bi[kMaxK];
#pragma unroll
    for (int i = 0; i < kMaxK; ++i) { bd[i] = kInf; bi[i] = 0; }
    for (int k = 0; k < C; ++k) {
        float md = d_norm3(s.pos.x - cyl[3 * k], s.pos.y - cyl[3 * k + 1], s.pos.z - cyl[3 * k + 2]) - c.cylinder_size;
        if (md < bd[kMaxK - 1]) {
            bd[kMaxK - 1] = md; bi[kMaxK - 1] = k;
#pragma unroll
            for (int i = kMaxK - 1; i > 0; --i) {
                if (bd[i] < bd[i - 1]) {
                    float td = bd[i]; bd[i] = bd[i - 1]; bd[i - 1] = td;
                    int ti = bi[i]; bi[i] = bi[i - 1]; bi[i - 1] = ti;
                }
            }
        }
    }
    float *oc = sOCyl + (le * A + a) * K * 5;
#pragma unroll
    for (int sidx = 0; sidx < kMaxK; ++sidx) {
        if (sidx < K) {
            const float *cc = cyl + 3 * bi[sidx];
            bool masked = cc[2] < 0.0f;                                // :759,775-778
            side.knn_idx[sidx] = bi[sidx];
            side.knn_masked[sidx] = masked;
            float *row = oc + sidx * 5;
            row[0] = masked ? c.mask_value : s.pos.x - cc[0];
            row[1] = masked ? c.mask_value : s.pos.y - cc[1];
            row[2] = masked ? c.mask_value : s.pos.z - cc[2];
            row[3] = masked ? c.mask_value : c.cylinder_height;
            row[4] = masked ? c.mask_value : c.cylinder_size;
        }
    }
}

// =================================================================================================
// The fused step kernel
// =================================================================================================
template <int A>
__global__ __launch_bounds__(kThreads) void hns_step_kernel(const Params p) {
    constexpr int G = Geo<A>::G, EPB = Geo<A>::EPB;
    extern __shared__ __align__(16) float smem[];
    const hns_cfg &c = p.cfg;
    const hns_buffers &b = p.buf;
    const int C = c.num_cylinders, K = c.obs_max_cylinder, E = c.num_envs;
    const bool with_state = c.write_critic_state && b.state_drones != nullptr;
    const Lds L = lds_layout(EPB, A, C, K, with_state);
    float *sDS = smem + L.ds, *sCyl = smem + L.cyl, *sTp = smem + L.tp, *sTvel = smem + L.tvel;
    float *sStats = smem + L.stats, *sSelf = smem + L.self, *sOth = smem + L.oth, *sOCyl = smem + L.ocyl;
    float *sState = with_state ? smem + L.state : nullptr;

    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * EPB;
    const int nenv = min(EPB, E - e0);
    const int le = tid / G, a = tid % G;
    const int e = e0 + le;
    const bool env_ok = le < nenv;
    const bool is_agent = env_ok && a < A;
    const bool is_envlane = env_ok && a == A;
    const size_t ia = (size_t)e * A + a;

    // ---- load: per-agent float4 records straight to registers, the rest through LDS ------------
    float4 act4 = make_float4(0, 0, 0, 0), thr4 = act4, integ4 = act4, last4 = act4, prev4 = act4;
    if (is_agent) {
        act4 = reinterpret_cast<const float4 *>(p.action)[ia];
        thr4 = reinterpret_cast<const float4 *>(b.throttle)[ia];
        integ4 = reinterpret_cast<const float4 *>(b.pid_integ)[ia];
        last4 = reinterpret_cast<const float4 *>(b.pid_last_rate)[ia];
        prev4 = reinterpret_cast<const float4 *>(b.prev_action)[ia];
    }
    float progress = env_ok ? b.progress[e] : 0.0f;
    coop_g2s(sDS, b.drone_state + (size_t)e0 * A * 13, nenv * A * 13);
    coop_g2s(sCyl, b.cylinders + (size_t)e0 * C * 3, nenv * C * 3);
    coop_g2s(sTp, b.target_pos + (size_t)e0 * 3, nenv * 3);
    coop_stats<EPB, true>(sStats, b.stats, E, e0, nenv);
    __syncthreads();

    // ---- pre-physics on S_t ------------------------------------------------------------------
    const float *cyl = sCyl + le * C * 3;
    V3 tp = {0.f, 0.f, 0.f};
    if (env_ok) tp = {sTp[le * 3], sTp[le * 3 + 1], sTp[le * 3 + 2]};
    Rigid s = {};
    s.q.w = 1.0f;
    float thrust[4] = {0, 0, 0, 0}, moment[4] = {0, 0, 0, 0};
    float thr_diff = 0.f, aerr = 0.f;
    V3 tw = {0.f, 0.f, 0.f}, fp = {0.f, 0.f, 0.f};
    if (is_agent) {
        const float *r = sDS + (le * A + a) * 13;
        s.pos = {r[0], r[1], r[2]};
        s.q = {r[3], r[4], r[5], r[6]};
        s.lin = {r[7], r[8], r[9]};
        s.ang = {r[10], r[11], r[12]};
        float cmd[4];
        d_ctbr_pid(c, act4, s.q, s.ang, prev4, integ4, last4, cmd, aerr);        // A1 + A2
        d_rotor(c, cmd, thr4, thrust, moment, thr_diff);                          // A3
        float ts = ((thrust[0] + thrust[1]) + thrust[2]) + thrust[3];
        V3 tv = {0.0f, 0.0f, ts};
        tw = d_quat_rot<false>(s.q, tv);                                          // multirotor.py:491
        bool blocked_pre = d_blocked(c, C, s.pos, tp, cyl);                       // hideandseek.py:1080
        fp = d_prey_pursuer_term(c, s.pos, tp, blocked_pre);
    }
    // A6: evader force = sum over pursuers (ascending) + arena + cylinders; per-axis velocity (:741)
    V3 F = {bcast<G>(fp.x, 0), bcast<G>(fp.y, 0), bcast<G>(fp.z, 0)};
#pragma unroll
    for (int j = 1; j < A; ++j) {
        F.x = F.x + bcast<G>(fp.x, j); F.y = F.y + bcast<G>(fp.y, j); F.z = F.z + bcast<G>(fp.z, j);
    }
    bool out_of_arena;
    V3 fr = d_prey_arena_term(c, tp, out_of_arena);
    F.x = F.x + fr.x; F.y = F.y + fr.y; F.z = F.z + fr.z;
    {
        float fcx = 0.f, fcy = 0.f;
        for (int k = 0; k < C; ++k) {
            float tx, ty;
            d_prey_cylinder_term(c, tp, cyl[3 * k], cyl[3 * k + 1], cyl[3 * k + 2], tx, ty);
            fcx = (k == 0) ? tx : fcx + tx;
            fcy = (k == 0) ? ty : fcy + ty;
        }
        F.x = F.x + fcx; F.y = F.y + fcy; F.z = F.z + 0.0f;
    }
    V3 tvel = {(c.v_prey * F.x) / (__builtin_fabsf(F.x) + 1e-5f), (c.v_prey * F.y) / (__builtin_fabsf(F.y) + 1e-5f),
               (c.v_prey * F.z) / (__builtin_fabsf(F.z) + 1e-5f)};
    V3 tpn = {tp.x + tvel.x * c.dt, tp.y + tvel.y * c.dt, tp.z + tvel.z * c.dt};   // evader: p += v dt

    // A4: downwash from the other drones of the env (positions of S_t from LDS, thrust vectors by shuffle)
    V3 fdw = {0.f, 0.f, 0.f};
    bool first = true;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        V3 twj = {bcast<G>(tw.x, j), bcast<G>(tw.y, j), bcast<G>(tw.z, j)};
        if (is_agent && j != a) {
            const float *rj = sDS + (le * A + j) * 13;
            V3 pj = {rj[0], rj[1], rj[2]};
            V3 fj = d_downwash_pair(s.pos, pj, twj);
            fdw.x = first ? fj.x : fdw.x + fj.x;
            fdw.y = first ? fj.y : fdw.y + fj.y;
            fdw.z = first ? fj.z : fdw.z + fj.z;
            first = false;
        }
    }
    if (is_agent) {
        V3 fw = {tw.x + fdw.x, tw.y + fdw.y, tw.z + fdw.z};
        V3 tb;
        tb.x = ((c.rotor_py[0] * thrust[0] + c.rotor_py[1] * thrust[1]) + c.rotor_py[2] * thrust[2]) + c.rotor_py[3] * thrust[3];
        tb.y = -(((c.rotor_px[0] * thrust[0] + c.rotor_px[1] * thrust[1]) + c.rotor_px[2] * thrust[2]) + c.rotor_px[3] * thrust[3]);
        tb.z = ((moment[0] + moment[1]) + moment[2]) + moment[3];
        d_integrate(c, s, fw, tb);                                                // A5
    }
    progress += 1.0f;                                                             // isaac_env.py:236

    // ---- publish S_{t+1} to LDS ----------------------------------------------------------------
    __syncthreads();
    if (is_agent) {
        float *r = sDS + (le * A + a) * 13;
        r[0] = s.pos.x; r[1] = s.pos.y; r[2] = s.pos.z;
        r[3] = s.q.w; r[4] = s.q.x; r[5] = s.q.y; r[6] = s.q.z;
        r[7] = s.lin.x; r[8] = s.lin.y; r[9] = s.lin.z;
        r[10] = s.ang.x; r[11] = s.ang.y; r[12] = s.ang.z;
        reinterpret_cast<float4 *>(b.throttle)[ia] = thr4;
        reinterpret_cast<float4 *>(b.pid_integ)[ia] = integ4;
        reinterpret_cast<float4 *>(b.pid_last_rate)[ia] = last4;
        reinterpret_cast<float4 *>(b.prev_action)[ia] = prev4;
        b.action_error[ia] = aerr;
    }
    if (is_envlane) {
        sTp[le * 3] = tpn.x; sTp[le * 3 + 1] = tpn.y; sTp[le * 3 + 2] = tpn.z;
        sTvel[le * 3] = tvel.x; sTvel[le * 3 + 1] = tvel.y; sTvel[le * 3 + 2] = tvel.z;
    }
    __syncthreads();

    // ---- post-physics on S_{t+1}: observation, reward, done, stats -------------------------------
    ObsSide side;
    obs_pass<A>(c, C, K, le, a, is_agent, s, tpn, progress, cyl, sDS, sSelf, sOth, sOCyl, sState, side);

    float dist_rew = 0.f, speed_rew = 0.f, cc = 0.f, cd = 0.f, cw = 0.f, coll_rew = 0.f, smooth = 0.f;
    bool cap_ok = false;
    if (is_agent) {                                                               // hideandseek.py:919-995
        float d = d_norm3(tpn.x - s.pos.x, tpn.y - s.pos.y, tpn.z - s.pos.z);
        float act = (d > c.catch_radius) ? 1.0f : 0.0f;
        dist_rew = (-c.dist_reward_coef * d) * act;
        cap_ok = (d < c.catch_radius) && !side.blocked;
        float sp = d_norm3(s.lin.x, s.lin.y, s.lin.z);
        speed_rew = -c.speed_coef * ((sp > c.v_drone) ? 1.0f : 0.0f);
#pragma unroll
        for (int sidx = 0; sidx < kMaxK; ++sidx) {
            if (sidx < K) {
                const float *cy = cyl + 3 * side.knn_idx[sidx];
                float rx = s.pos.x - cy[0], ry = s.pos.y - cy[1];
                float dxy = __builtin_sqrtf(rx * rx + ry * ry);
                float hit = ((dxy - c.cylinder_size) < c.collision_radius) ? 1.0f : 0.0f;
                if (side.knn_masked[sidx]) hit = 0.0f;
                cc = (sidx == 0) ? hit : cc + hit;
            }
        }
        float cr = -c.collision_coef * cc;
        bool firstj = true;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            if (j == a) continue;
            const float *rj = sDS + (le * A + j) * 13;
            float dd = d_norm3(s.pos.x - rj[0], s.pos.y - rj[1], s.pos.z - rj[2]);
            float hit = (dd < c.coll_drone_dist) ? 1.0f : 0.0f;
            cd = firstj ? hit : cd + hit;
            firstj = false;
        }
        cr = cr + -c.collision_coef * cd;
        cw = ((s.pos.z > c.max_height) ? 1.0f : 0.0f) + (((s.pos.x * s.pos.x + s.pos.y * s.pos.y) > c.arena_sq) ? 1.0f : 0.0f);
        cr = cr + -c.collision_coef * cw;
        coll_rew = cr;
        float sm = c.smoothness_coef * d_expf(-aerr);
        if (!c.use_deployment) sm = 0.0f;
        smooth = sm;
    }
    const bool any_cap = group_bits<A, G>(cap_ok && is_agent) != 0u;
    const bool all_blocked = group_bits<A, G>(side.blocked && is_agent) == ((1u << A) - 1u);
    const bool any_coll = group_bits<A, G>((coll_rew < 0.0f) && is_agent) != 0u;
    const float detf = side.bdetect ? 1.0f : 0.0f;
    const float detect_rew = c.detect_reward_coef * detf;
    const float catch_rew = c.catch_reward_coef * (any_cap ? 1.0f : 0.0f);
    const float rew = ((((dist_rew + detect_rew) + catch_rew) + coll_rew) + speed_rew) + smooth;
    if (is_agent) b.reward[ia] = rew;

    // per-env sums over the agents, ascending order, identical in every lane of the group
    float sum_dist = bcast<G>(dist_rew, 0), sum_speed = bcast<G>(speed_rew, 0), sum_cc = bcast<G>(cc, 0);
    float sum_cd = bcast<G>(cd, 0), sum_cw = bcast<G>(cw, 0), sum_coll = bcast<G>(coll_rew, 0);
    float sum_smooth = bcast<G>(smooth, 0), sum_td = bcast<G>(thr_diff, 0), max_td = sum_td;
    float sum_ae = bcast<G>(aerr, 0), sum_rew = bcast<G>(rew, 0);
#pragma unroll
    for (int j = 1; j < A; ++j) {
        sum_dist += bcast<G>(dist_rew, j); sum_speed += bcast<G>(speed_rew, j); sum_cc += bcast<G>(cc, j);
        sum_cd += bcast<G>(cd, j); sum_cw += bcast<G>(cw, j); sum_coll += bcast<G>(coll_rew, j);
        sum_smooth += bcast<G>(smooth, j);
        float tdj = bcast<G>(thr_diff, j);
        sum_td += tdj;
        if (tdj > max_td) max_td = tdj;
        sum_ae += bcast<G>(aerr, j); sum_rew += bcast<G>(rew, j);
    }
    if (is_envlane) {
#define ST(i) sStats[(i) * EPB + le]
        const float fA = (float)A;
        float mae = sum_ae / fA;                                                  // A10, hideandseek.py:731-733
        ST(HNS_ST_ACTION_ERROR_ORDER1_MEAN) += mae;
        if (mae > ST(HNS_ST_ACTION_ERROR_ORDER1_MAX)) ST(HNS_ST_ACTION_ERROR_ORDER1_MAX) = mae;
        ST(HNS_ST_OUT_OF_ARENA) = ((ST(HNS_ST_OUT_OF_ARENA) != 0.0f) || out_of_arena) ? 1.0f : 0.0f;   // :1097-1098
        ST(HNS_ST_DISTANCE_REWARD) += sum_dist / fA;
        ST(HNS_ST_SUM_DETECT_STEP) += 1.0f * detf;
        float sdet = detect_rew, scat = catch_rew;
#pragma unroll
        for (int j = 1; j < A; ++j) { sdet += detect_rew; scat += catch_rew; }
        ST(HNS_ST_DETECT_REWARD) += sdet / fA;
        const bool capture_flag = catch_rew != 0.0f;                              // :945
        ST(HNS_ST_BLOCKED) += all_blocked ? 1.0f : 0.0f;
        ST(HNS_ST_SUCCESS) = (capture_flag || ST(HNS_ST_SUCCESS) != 0.0f) ? 1.0f : 0.0f;
        float cur = (capture_flag ? 1.0f : 0.0f) * progress + (capture_flag ? 0.0f : 1.0f) * (float)c.max_episode_length;
        if (cur < ST(HNS_ST_FIRST_CAPTURE_STEP)) ST(HNS_ST_FIRST_CAPTURE_STEP) = cur;
        ST(HNS_ST_CATCH_REWARD) += scat / fA;
        ST(HNS_ST_SPEED_REWARD) += sum_speed / fA;
        ST(HNS_ST_COLLISION_CYLINDER) += sum_cc / fA;
        ST(HNS_ST_COLLISION_DRONE) += sum_cd / fA;
        ST(HNS_ST_COLLISION) += any_coll ? 1.0f : 0.0f;
        ST(HNS_ST_COLLISION_WALL) += sum_cw / fA;
        ST(HNS_ST_COLLISION_REWARD) += sum_coll / fA;
        ST(HNS_ST_SMOOTHNESS_COEF) = c.smoothness_coef;
        ST(HNS_ST_SMOOTHNESS_REWARD) += sum_smooth / fA;
        ST(HNS_ST_SMOOTHNESS_MEAN) += sum_td / fA;
        if (max_td > ST(HNS_ST_SMOOTHNESS_MAX)) ST(HNS_ST_SMOOTHNESS_MAX) = max_td;
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
        ST(HNS_ST_RETURN) += sum_rew / fA;
#undef ST
        b.done[e] = (uint8_t)done;
        b.progress[e] = progress;
    }
    __syncthreads();

    // ---- store: contiguous slices, 16 B per lane -------------------------------------------------
    coop_s2g(b.drone_state + (size_t)e0 * A * 13, sDS, nenv * A * 13);
    coop_s2g(b.target_pos + (size_t)e0 * 3, sTp, nenv * 3);
    coop_s2g(b.target_vel + (size_t)e0 * 3, sTvel, nenv * 3);
    coop_stats<EPB, false>(sStats, b.stats, E, e0, nenv);
    coop_s2g(b.obs_self + (size_t)e0 * A * HNS_SELF_DIM, sSelf, nenv * A * HNS_SELF_DIM);
    if (A > 1) coop_s2g(b.obs_others + (size_t)e0 * A * (A - 1) * 3, sOth, nenv * A * (A - 1) * 3);
    coop_s2g(b.obs_cylinders + (size_t)e0 * A * K * 5, sOCyl, nenv * A * K * 5);
    if (with_state) coop_s2g(b.state_drones + (size_t)e0 * A * HNS_SELF_DIM, sState, nenv * A * HNS_SELF_DIM);
}

// =================================================================================================
// Reset kernel (A11): hideandseek.py:576-723, multirotor.py:635-650 + the reset-time obs pass
// (isaac_env.py:221).  The env lane regenerates the env's state into LDS with a Philox stream,
// then the agent lanes run the same obs_pass as the step kernel.
// =================================================================================================
template <int A>
__global__ __launch_bounds__(kThreads) void hns_reset_kernel(const Params p) {
    constexpr int G = Geo<A>::G, EPB = Geo<A>::EPB;
    extern __shared__ __align__(16) float smem[];
    __shared__ uint8_t sMask[EPB];
    const hns_cfg &c = p.cfg;
    const hns_buffers &b = p.buf;
    const int C = c.num_cylinders, K = c.obs_max_cylinder, E = c.num_envs, GN = c.grid_num;
    const bool with_state = c.write_critic_state && b.state_drones != nullptr;
    const Lds L = lds_layout(EPB, A, C, K, with_state);
    float *sDS = smem + L.ds, *sCyl = smem + L.cyl, *sTp = smem + L.tp;
    float *sSelf = smem + L.self, *sOth = smem + L.oth, *sOCyl = smem + L.ocyl;
    float *sState = with_state ? smem + L.state : nullptr;
    uint8_t *sGrid = reinterpret_cast<uint8_t *>(smem + L.total);   // EPB x 512 B of grid scratch after the step layout

    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * EPB;
    const int nenv = min(EPB, E - e0);
    const int le = tid / G, a = tid % G;
    const int e = e0 + le;
    const bool env_ok = le < nenv;
    if (tid < EPB) sMask[tid] = (tid < nenv) ? (p.reset_mask ? (p.reset_mask[e0 + tid] != 0) : 1) : 0;
    __syncthreads();
    const bool masked = env_ok && sMask[le];
    const bool is_agent = masked && a < A;
    const bool is_envlane = env_ok && a == A;

    if (is_envlane) {
        // hideandseek.py:712 resets first_capture_step for ALL envs on any reset call
        b.stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * E + e] = (float)c.max_episode_length;
    }
    if (is_envlane && masked) {
        Rng rng = {p.seed_lo, p.seed_hi, (uint32_t)(e + c.env_index_offset), p.epoch, 0u, {0u, 0u, 0u, 0u}, 0};
        float *ds = sDS + le * A * 13;
        float *tp = sTp + le * 3;
        float *cyl = sCyl + le * C * 3;
        for (int j = 0; j < A; ++j) {
            float *d = ds + 13 * j;
            if (c.init_mode == HNS_INIT_RANDOM) {
                d[0] = c.drone_xy_lo[0] + rng.uniform() * (c.drone_xy_hi[0] - c.drone_xy_lo[0]);
                d[1] = c.drone_xy_lo[1] + rng.uniform() * (c.drone_xy_hi[1] - c.drone_xy_lo[1]);
            } else {
                d[0] = c.fixed_drone_pos[j][0]; d[1] = c.fixed_drone_pos[j][1];
            }
            if (c.init_mode == HNS_INIT_SCENARIO) d[2] = c.fixed_drone_pos[j][2];
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
        if (c.init_mode == HNS_INIT_RANDOM) {
            tp[0] = c.target_xy_lo[0] + rng.uniform() * (c.target_xy_hi[0] - c.target_xy_lo[0]);
            tp[1] = c.target_xy_lo[1] + rng.uniform() * (c.target_xy_hi[1] - c.target_xy_lo[1]);
        } else {
            tp[0] = c.fixed_target_pos[0]; tp[1] = c.fixed_target_pos[1];
        }
        if (c.init_mode == HNS_INIT_SCENARIO) tp[2] = c.fixed_target_pos[2];
        else tp[2] = c.z_lo + rng.uniform() * (c.z_hi - c.z_lo);
        b.target_vel[(size_t)e * 3] = 0.0f; b.target_vel[(size_t)e * 3 + 1] = 0.0f; b.target_vel[(size_t)e * 3 + 2] = 0.0f;
        if (c.init_mode == HNS_INIT_SCENARIO) {
            for (int k = 0; k < C; ++k) {
                cyl[3 * k] = c.fixed_cyl_pos[k][0]; cyl[3 * k + 1] = c.fixed_cyl_pos[k][1];
                cyl[3 * k + 2] = (k >= c.fixed_cyl_active) ? c.invalid_z : c.fixed_cyl_pos[k][2];
            }
        } else {                                                                  // hideandseek.py:576-607
            uint8_t *occ = sGrid + le * 512;        // [GN*GN] occupancy, then [GN*GN] free-cell list
            uint8_t *freec = occ + 256;
            const int half = GN / 2;
            for (int i = 0; i < GN; ++i)
                for (int j = 0; j < GN; ++j) {
                    float dd = __builtin_sqrtf((float)((i - half) * (i - half) + (j - half) * (j - half)));
                    occ[i * GN + j] = dd >= (float)half;                          // :168-181
                }
            for (int j = 0; j < A; ++j) occ[d_cell(c, ds[13 * j]) * GN + d_cell(c, ds[13 * j + 1])] = 1;
            occ[d_cell(c, tp[0]) * GN + d_cell(c, tp[1])] = 1;
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
    Rigid s = {};
    s.q.w = 1.0f;
    V3 tp = {0.f, 0.f, 0.f};
    if (masked) tp = {sTp[le * 3], sTp[le * 3 + 1], sTp[le * 3 + 2]};
    if (is_agent) {
        const float *r = sDS + (le * A + a) * 13;
        s.pos = {r[0], r[1], r[2]};
        s.q = {r[3], r[4], r[5], r[6]};
        s.lin = {r[7], r[8], r[9]};
        s.ang = {r[10], r[11], r[12]};
    }
    ObsSide side;
    obs_pass<A>(c, C, K, le, a, is_agent, s, tp, 0.0f, sCyl + le * C * 3, sDS, sSelf, sOth, sOCyl, sState, side);
    __syncthreads();
    coop_s2g_masked(b.drone_state + (size_t)e0 * A * 13, sDS, nenv * A * 13, A * 13, sMask);
    coop_s2g_masked(b.cylinders + (size_t)e0 * C * 3, sCyl, nenv * C * 3, C * 3, sMask);
    coop_s2g_masked(b.target_pos + (size_t)e0 * 3, sTp, nenv * 3, 3, sMask);
    coop_s2g_masked(b.obs_self + (size_t)e0 * A * HNS_SELF_DIM, sSelf, nenv * A * HNS_SELF_DIM, A * HNS_SELF_DIM, sMask);
    if (A > 1) coop_s2g_masked(b.obs_others + (size_t)e0 * A * (A - 1) * 3, sOth, nenv * A * (A - 1) * 3, A * (A - 1) * 3, sMask);
    coop_s2g_masked(b.obs_cylinders + (size_t)e0 * A * K * 5, sOCyl, nenv * A * K * 5, A * K * 5, sMask);
    if (with_state)
        coop_s2g_masked(b.state_drones + (size_t)e0 * A * HNS_SELF_DIM, sState, nenv * A * HNS_SELF_DIM, A * HNS_SELF_DIM, sMask);
}

}  // namespace hns

// =================================================================================================
// Host side: the C ABI (include/hns.h)
// =================================================================================================
using hns::Params;

static thread_local std::string g_last_error;
static void set_error(const std::string &m) { g_last_error = m; }

struct hns_env {
    hns_cfg cfg;
    hns_buffers buf;
    bool bound = false;
    uint32_t epoch = 0;
    int grid = 0;
    size_t lds_step = 0, lds_reset = 0;
    void (*step_fn)(const Params) = nullptr;
    void (*reset_fn)(const Params) = nullptr;
    int timing = 0;          // 0 = off, n = bracket every n-th step launch with hipEvents
    uint64_t step_count = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded, not yet harvested
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;     // free event pairs
};

template <int A>
static void select_kernels(hns_env *env) {
    env->step_fn = hns::hns_step_kernel<A>;
    env->reset_fn = hns::hns_reset_kernel<A>;
    const hns_cfg &c = env->cfg;
    constexpr int EPB = hns::Geo<A>::EPB;
    env->grid = (c.num_envs + EPB - 1) / EPB;
    hns::Lds L = hns::lds_layout(EPB, A, c.num_cylinders, c.obs_max_cylinder, c.write_critic_state);
    env->lds_step = (size_t)L.total * sizeof(float);
    env->lds_reset = env->lds_step + (size_t)EPB * 512;   // + per-env occupancy grid / free-cell list
}

#define HNS_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
            return HNS_ERR_DEVICE;                                                 \
        }                                                                          \
    } while (0)

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
    if (cfg->obs_max_cylinder > hns::kMaxK) {
        set_error("hns_create: obs_max_cylinder > 4 is not supported by the HIP kernels");
        return HNS_ERR_INVALID_ARG;
    }
    if (cfg->grid_num < 1 || cfg->grid_num > 16) { set_error("hns_create: grid_num out of range"); return HNS_ERR_INVALID_ARG; }
    if (cfg->init_mode != HNS_INIT_SCENARIO) {
        int half = cfg->grid_num / 2, free_cells = 0;
        for (int i = 0; i < cfg->grid_num; ++i)
            for (int j = 0; j < cfg->grid_num; ++j)
                if (sqrtf((float)((i - half) * (i - half) + (j - half) * (j - half))) < (float)half) ++free_cells;
        if (free_cells - (cfg->num_agents + 1) < cfg->num_cylinders) {
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
    if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        delete env;
        return HNS_ERR_DEVICE;
    }
    *out = env;
    return HNS_OK;
}

void hns_destroy(hns_env *env) {
    if (!env) return;
    for (auto &p : env->events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto &p : env->pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
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
    const void *al16[] = {buffers->throttle, buffers->pid_integ, buffers->pid_last_rate, buffers->prev_action};
    for (const void *ptr : al16)
        if (reinterpret_cast<uintptr_t>(ptr) & 15) { set_error("hns_bind: per-agent float4 buffers must be 16-byte aligned"); return HNS_ERR_INVALID_ARG; }
    env->buf = *buffers;
    env->bound = true;
    return HNS_OK;
}

static int launch(hns_env *env, bool is_step, const Params &p, hipStream_t stream) {
    std::pair<hipEvent_t, hipEvent_t> ev{};
    const bool time_it = is_step && env->timing > 0 && (env->step_count++ % (uint64_t)env->timing) == 0;
    if (time_it) {
        if (!env->pool.empty()) { ev = env->pool.back(); env->pool.pop_back(); }
        else {
            HNS_CHECK_HIP(hipEventCreate(&ev.first));
            HNS_CHECK_HIP(hipEventCreate(&ev.second));
        }
        HNS_CHECK_HIP(hipEventRecord(ev.first, stream));
    }
    auto fn = is_step ? env->step_fn : env->reset_fn;
    size_t lds = is_step ? env->lds_step : env->lds_reset;
    hipLaunchKernelGGL(fn, dim3(env->grid), dim3(hns::kThreads), lds, stream, p);
    HNS_CHECK_HIP(hipGetLastError());
    if (time_it) {
        HNS_CHECK_HIP(hipEventRecord(ev.second, stream));
        env->events.push_back(ev);
    }
    return HNS_OK;
}

int hns_step(hns_env *env, const float *action, void *stream) {
    if (!env || !action) { set_error("hns_step: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_step: buffers not bound"); return HNS_ERR_NOT_BOUND; }
    if (reinterpret_cast<uintptr_t>(action) & 15) { set_error("hns_step: action must be 16-byte aligned"); return HNS_ERR_INVALID_ARG; }
    Params p;
    p.cfg = env->cfg;
    p.buf = env->buf;
    p.action = action;
    p.reset_mask = nullptr;
    p.seed_lo = p.seed_hi = p.epoch = 0;
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
    return launch(env, false, p, static_cast<hipStream_t>(stream));
}

int hns_set_v_prey(hns_env *env, float v_prey) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->cfg.v_prey = v_prey;
    return HNS_OK;
}
int hns_set_smoothness_coef(hns_env *env, float coef) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->cfg.smoothness_coef = coef;
    return HNS_OK;
}
int hns_set_reset_epoch(hns_env *env, uint32_t epoch) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->epoch = epoch;
    return HNS_OK;
}
uint32_t hns_get_reset_epoch(const hns_env *env) { return env ? env->epoch : 0u; }

int hns_enable_timing(hns_env *env, int on) {
    if (!env) return HNS_ERR_INVALID_ARG;
    env->timing = on < 0 ? 0 : on;
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
