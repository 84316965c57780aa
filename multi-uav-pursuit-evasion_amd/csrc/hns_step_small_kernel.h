// hns_step_small_kernel.h — the step kernel's second mapping, for batches that do not fill the chip (BASELINE config 2: 4 096 envs; the
// reference's own default: 2 048, cfg/task/HideAndSeek.yaml:9).  Same call tree, arithmetic, evaluation order and buffers as
// hns_step_v4_kernel (bit-identical; tests/test_hip_parity.py runs both on the same batches).
//
// What the phase stamps of the tile mapping showed at 4 096 envs (tools/phase_profile.py, tools/lab/r04_batch22.sh): 64 workgroups on 256 CUs,
// every wave alone on its SIMD issuing one instruction per 5-6 cycles, and a workgroup's life (20.2 k cycles = the launch) is the pursuer
// lanes' serial instruction stream: load 0.6 k, controller + rotors 4.9 k, downwash + integration 4.0 k, line of sight + k nearest +
// reward terms 6.5 k, and then the env wave's reductions and statistics 3.2 k behind the last barrier.  Smaller tiles do not shorten
// that stream (lanes are envs); more lanes per (env, pursuer) do not either where the work is dot and cross products.  What does: the
// second half of the stream is made of pieces that do not depend on one another, and three SIMDs in four are idle.  So every pursuer wave
// gets a HELPER wave with the same lane <-> (env, pursuer) mapping:
//   * before barrier 1 the helpers stage the workgroup's cylinders (was: the env wave) and fetch their shares of the statistics rows;
//   * between barriers 1 and 2 one helper wave (lane <-> env) updates the statistics that only need the controller's outputs;
//   * behind barrier 2 (positions at t+1 published) a helper selects the k nearest cylinders of its pursuers and counts the cylinder
//     collisions while the owner computes the distance, speed, drone-collision, wall and smoothness terms, sweeps the cylinders for the
//     line of sight (detection / capture flags), and stores its controller state, S_{t+1} and its state_others rows;
//   * behind barrier 3 the owner stores the (masked) state_self rows, the helper its k-nearest rows; lane <-> env, helper wave 2 builds the reward
//     rows and their running sum, helper wave 0 updates the eight statistics that are plain sums over the pursuers, and the env wave keeps
//     done / progress and the seven statistics decided by the detection and capture flags.
// 2 A + 1 waves per workgroup, three workgroup barriers, no new exchange beyond the records the tile mapping already publishes (+ one flag).
#pragma once
#include "hns_step_kernel.h"

namespace hns {

template <int A>
struct GeoSmall {
    static constexpr int NA = kEPB * A;             // owner threads; env wave: [NA, NA + 64); helpers: [NA + 64, 2 NA + 64).  Waves go round the four
                                                    // SIMDs in this order: with three pursuers the env wave has a SIMD to itself and every owner shares
                                                    // one with a helper (idle while the owner computes) — with the env wave last, the owner beside it
                                                    // reached barrier 2 ~900 cycles behind the others (tools/phase_profile.py --waves)
    static constexpr int T = kEPB * (2 * A + 1);
};
// LDS (float offsets): a staging slab per owner and per helper wave (whole-line stores, as in the tile mapping: rows stored by their lanes were
// measured too — 27 store instructions of 64 scattered lines each in the last phase keep the CU's one address path busy for most of it, and
// from 16 384 envs on the step is 2-4 us slower), then the records the tile mapping publishes
struct LdsSmall { int slab, slab_stride, hslab, hslab_stride, pub, cyl, cyl_stride, tp, red, envout, total; };
__host__ __device__ inline LdsSmall lds_layout_small(int A, int C, int K) {
    LdsSmall L;
    int o = 0;
    L.slab_stride = slab_floats(A, K, 1);
    if (L.slab_stride < 64 * 13 + 4) L.slab_stride = r4(64 * 13 + 4);
    L.slab = o;   o += A * L.slab_stride;
    L.hslab_stride = r4(slab_rows(A) * K * 5);
    L.hslab = o;  o += A * L.hslab_stride;
    L.pub = o;    o += r4(kEPB * A * kPub);
    L.cyl_stride = (3 * C) | 1;
    L.cyl = o;    o += r4(kEPB * L.cyl_stride);
    L.tp = o;     o += r4(kEPB * 4);                         // evader at t+1 [64][3], then the step counter [64]
    L.red = o;    o += r4(kEPB * A * red_stride(1));
    L.envout = o; o += r4(kEPB * (A > 3 ? A : 3));           // arena flags, then evader velocity [64,3], then rewards [64,A]
    L.total = o;
    return L;
}

// the statistics helper wave 0 owns: sums of one published reward term over the env's pursuers (hideandseek.py:960-1056)
HNS_DEV constexpr bool small_helper_stat(int i) {
    return i == HNS_ST_DISTANCE_REWARD || i == HNS_ST_SPEED_REWARD || i == HNS_ST_COLLISION_CYLINDER || i == HNS_ST_COLLISION_DRONE ||
           i == HNS_ST_COLLISION || i == HNS_ST_COLLISION_WALL || i == HNS_ST_COLLISION_REWARD || i == HNS_ST_SMOOTHNESS_REWARD;
}
// ... and helper wave 1 (wave 0 with a single pursuer): what the controller phase decides (hideandseek.py:731-733, :1097-1098, :996-997)
HNS_DEV constexpr bool small_early_stat(int i) {
    return i == HNS_ST_ACTION_ERROR_ORDER1_MEAN || i == HNS_ST_ACTION_ERROR_ORDER1_MAX || i == HNS_ST_SMOOTHNESS_MEAN ||
           i == HNS_ST_SMOOTHNESS_MAX || i == HNS_ST_SMOOTHNESS_COEF || i == HNS_ST_OUT_OF_ARENA;
}
// ... and what the step never changes (the rows stay as they are: neither loaded nor stored)
HNS_DEV constexpr bool small_untouched_stat(int i) { return i == HNS_ST_DISTANCE_PREDICTED_REWARD || i == HNS_ST_DISTANCE_THRESHOLD_L; }
// ... the helper wave that builds the reward rows keeps their sum
HNS_DEV constexpr bool small_reward_stat(int i) { return i == HNS_ST_RETURN; }
HNS_DEV constexpr bool small_env_stat(int i) { return !small_helper_stat(i) && !small_early_stat(i) && !small_untouched_stat(i) && !small_reward_stat(i); }

template <int A, bool PROF, int CS = 0>
__global__ __launch_bounds__(GeoSmall<A>::T, 1) void hns_step_small_kernel(HNS_STEP_PARAMS) {
    HNS_STEP_ARGS_PACK;
    typedef const Params __attribute__((address_space(4))) ParamsC;
    ParamsC &p = *(ParamsC *)ka.rest;
    constexpr int NA = GeoSmall<A>::NA, SD = HNS_SELF_DIM, kRedS = red_stride(1), KM = kMaxK;
#ifdef HNS_NO_WARM
    constexpr bool WARM = false;
#else
    constexpr bool WARM = true;
#endif
    extern __shared__ __align__(16) float smem[];
    const auto &c = p.cfg;
    const auto &b = p.buf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int e0 = blockIdx.x * kEPB;
    if (tid < NA) {
        // ================================= owner waves: controller, physics, own-state terms =========================
        const int le = tid / A, a = tid - le * A;
        const unsigned ia = (unsigned)e0 * A + tid;
        const float4 act4 = reinterpret_cast<const float4 *>(ka.action)[ia];
        float4 prev4 = reinterpret_cast<const float4 *>(ka.prev_action)[ia];
        constexpr int N4 = 64 * 13 / 4;
        const float4 *rows4 = reinterpret_cast<const float4 *>(ka.drone_state + ((size_t)e0 * A + (tid & ~63)) * 13) + lane;
        float4 rr0 = rows4[0], rr1 = rows4[64], rr2 = rows4[128], rr3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < N4 - 192) rr3 = rows4[192];
        float4 integ4 = reinterpret_cast<const float4 *>(ka.pid_integ)[ia];
        float4 last4 = reinterpret_cast<const float4 *>(ka.pid_last_rate)[ia];
        unsigned rp = 0;
        {   // reset_pid = the incoming root `done` (transforms.py:449-454); branch-free, as in hns_step_v4_kernel
            const uint8_t *rpp = static_cast<const uint8_t *>(ka.aux);
            const uint8_t *rsafe = rpp ? rpp : reinterpret_cast<const uint8_t *>(ka.action);
            const unsigned byte = rsafe[e0 + le];
            rp = rpp ? byte : 0u;
        }
        float4 thr4 = reinterpret_cast<const float4 *>(ka.throttle)[ia];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = CS ? CS : c.num_cylinders, K = CS ? 3 : c.obs_max_cylinder;
        const bool with_state = c.write_critic_state && b.state_drones != nullptr;
        const LdsSmall L = lds_layout_small(A, C, K);
        float *sPub = smem + L.pub, *sTp = smem + L.tp, *sRed = smem + L.red;
        float *slab = smem + L.slab + (tid >> 6) * L.slab_stride;
        const float4 ta = d_action_tanh(act4);
        Rigid s;
        {
            float4 *s4 = reinterpret_cast<float4 *>(slab) + lane;
            s4[0] = rr0; s4[64] = rr1; s4[128] = rr2;
            if (lane < N4 - 192) s4[192] = rr3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            load_rigid(slab + lane * 13, s);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PROF) prof_mark(p.prof, 1);
        // ---- phase 1: controller, rotors, thrust vector (as hns_step_v4_kernel) ----
        const float los_t = last4.w;
        float cmd[4], thr_diff, aerr, thrust[4], moment[4];
        float ctbr4[4], trate[3];
        {
            const bool r = rp != 0;
            integ4.x = r ? 0.0f : integ4.x; integ4.y = r ? 0.0f : integ4.y; integ4.z = r ? 0.0f : integ4.z;
            last4.x = r ? 0.0f : last4.x; last4.y = r ? 0.0f : last4.y; last4.z = r ? 0.0f : last4.z;
        }
        d_ctbr_pid_squashed(c, ta, s.q, s.ang, prev4, integ4, last4, cmd, aerr, ctbr4, trate);
        // (the wait for the last of the first loads HERE, ahead of the two stores below: behind them it would wait for their acknowledgement too)
        asm volatile("" : "+v"(thr4.x), "+v"(thr4.y), "+v"(thr4.z), "+v"(thr4.w));
        if (b.ctbr) reinterpret_cast<float4 *>(b.ctbr)[ia] = make_float4(ctbr4[0], ctbr4[1], ctbr4[2], ctbr4[3]);
        if (b.target_rate) reinterpret_cast<float4 *>(b.target_rate)[ia] = make_float4(trate[0], trate[1], trate[2], 0.0f);
        d_rotor(c, cmd, thr4, thrust, moment, thr_diff);
        const float ts = ((thrust[0] + thrust[1]) + thrust[2]) + thrust[3];
        const V3 tw = d_quat_rot_z(s.q, ts);
        const float inv_ntw = d_downwash_inv_norm(tw);
        {
            float *pub = sPub + tid * kPub;
            pub[0] = s.pos.x; pub[1] = s.pos.y; pub[2] = s.pos.z;
            pub[3] = tw.x; pub[4] = tw.y; pub[5] = tw.z;
            pub[9] = inv_ntw;
            pub[10] = los_t;
            float *red = sRed + tid * kRedS;
            red[R_AERR] = aerr; red[R_TD] = thr_diff;
        }
        if constexpr (PROF) prof_mark(p.prof, 2);
        __syncthreads();                                                            // barrier 1
        if constexpr (PROF) prof_mark(p.prof, 12);
        // ---- phase 2: downwash, torques, integration ----
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
        if constexpr (PROF) prof_mark(p.prof, 3);
        __syncthreads();                                                            // barrier 2: the helpers start their cylinder sweep
        if constexpr (PROF) prof_mark(p.prof, 8);
        // ---- phase 3a, the owner's share: terms that need this pursuer's own state, and the line of sight ----
        const float progress = sTp[kEPB * 3 + le];
        const V3 tp = {sTp[le * 3], sTp[le * 3 + 1], sTp[le * 3 + 2]};
        const float rtx = s.pos.x - tp.x, rty = s.pos.y - tp.y, rtz = s.pos.z - tp.z;
        const float d = d_norm3(rtx, rty, rtz);                                     // hideandseek.py:921, :780
        const bool blocked = d_blocked(c, C, s.pos, tp, smem + L.cyl + le * L.cyl_stride);   // :786 — the sweep cylinder_pass runs (same tests, same fallback)
        const bool det = (d < c.drone_detect_radius) && !blocked;                   // :787-789
        const bool cap_ok = (d < c.catch_radius) && !blocked;
        const float act = (d > c.catch_radius) ? 1.0f : 0.0f;
        const float dist_rew = (-c.dist_reward_coef * d) * act;
        bool fast = false;
        {
            const float sp2 = HNS_FMA(s.lin.z, s.lin.z, HNS_FMA(s.lin.y, s.lin.y, s.lin.x * s.lin.x));
            const float v2 = c.v_drone * c.v_drone;
            fast = sp2 > v2 * 1.00000190734863f;
            if (!fast && !(sp2 < v2 * 0.99999809265137f)) fast = __builtin_sqrtf(sp2) > c.v_drone;
        }
        const float speed_rew = -c.speed_coef * (fast ? 1.0f : 0.0f);
        float cd = 0.f;
        const float dd2 = c.coll_drone_dist * c.coll_drone_dist;
        float oth[(A > 1 ? A - 1 : 1) * 3];
#pragma unroll
        for (int o = 0; o < A - 1; ++o) {
            const int j = o + (o >= a ? 1 : 0);
            const float *rj = sPub + (le * A + j) * kPub + 6;
            const float ex = s.pos.x - rj[0], ey = s.pos.y - rj[1], ez = s.pos.z - rj[2];
            oth[3 * o] = ex; oth[3 * o + 1] = ey; oth[3 * o + 2] = ez;              // p_i - p_j, j != i ascending (:750-751)
            const float s3 = HNS_FMA(ez, ez, HNS_FMA(ey, ey, ex * ex));
            bool h = s3 < dd2 * 0.99999809265137f;
            if (!h && !(s3 > dd2 * 1.00000190734863f)) h = __builtin_sqrtf(s3) < c.coll_drone_dist;
            const float hit = h ? 1.0f : 0.0f;
            cd = (o == 0) ? hit : cd + hit;
        }
        const float cw = ((s.pos.z > c.max_height) ? 1.0f : 0.0f) + ((HNS_FMA(s.pos.y, s.pos.y, s.pos.x * s.pos.x) > c.arena_sq) ? 1.0f : 0.0f);
        float sm = 0.0f;
        if (c.use_deployment) sm = c.smoothness_coef * d_expf(-aerr);
        {
            float *red = sRed + tid * kRedS;
            red[R_DIST] = dist_rew; red[R_SPEED] = speed_rew; red[R_CD] = cd; red[R_CW] = cw; red[R_SMOOTH] = sm;
            red[R_FLAGS] = __int_as_float((cap_ok ? F_CAP : 0) | (blocked ? F_BLOCKED : 0) | (det ? F_DET : 0));
        }
        if constexpr (PROF) prof_mark(p.prof, 9);
        // controller / rotor state, S_{t+1}, state_others: stored here, while the helpers still select their cylinders (behind barrier 3 — nobody
        // waits for them — they measured 0.1-0.4 us slower: the launch ends when its last store is acknowledged; tools/lab/r04_batch38.sh)
        last4.w = blocked ? 1.0f : 0.0f;                                            // = the next step's line of sight at ITS t
        reinterpret_cast<float4 *>(b.throttle)[ia] = thr4;
        reinterpret_cast<float4 *>(b.pid_integ)[ia] = integ4;
        reinterpret_cast<float4 *>(b.prev_action)[ia] = prev4;
        reinterpret_cast<float4 *>(b.pid_last_rate)[ia] = last4;
        b.action_error[ia] = aerr;
        {
            const float row[13] = {s.pos.x, s.pos.y, s.pos.z, s.q.w, s.q.x, s.q.y, s.q.z, s.lin.x, s.lin.y, s.lin.z, s.ang.x, s.ang.y, s.ang.z};
            wave_store_rows<13>(slab, b.drone_state + ((size_t)e0 * A + (tid & ~63)) * 13, row, lane, 64);
        }
        const float t = progress * c.inv_max_episode_length;                      // :796
        const V3 heading = d_quat_rot_x(s.q), up = d_quat_rot_z(s.q, 1.0f);       // multirotor.py:613-614
        if constexpr (A > 1)
            wave_store_rows<(A > 1 ? A - 1 : 1) * 3, slab_rows(A)>(slab, b.obs_others + ((size_t)e0 * A + (tid & ~63)) * (A - 1) * 3, oth, lane, 64);
        if (with_state) {                                                          // :871-886 (never masked)
            const float rs[SD] = {rtx, rty, rtz, s.q.w, s.q.x, s.q.y, s.q.z, s.lin.x, s.lin.y, s.lin.z,
                                  heading.x, heading.y, heading.z, up.x, up.y, up.z, t, t, t, t};
            wave_store_rows<SD, slab_rows(A)>(slab, b.state_drones + ((size_t)e0 * A + (tid & ~63)) * SD, rs, lane, 64);
        }
        if constexpr (PROF) prof_mark(p.prof, 4);
        __syncthreads();                                                            // barrier 3: the helpers' flags
        if constexpr (PROF) prof_mark(p.prof, 5);
        bool det_any = false;                                                       // :787-794: any pursuer sees the evader
#pragma unroll
        for (int j = 0; j < A; ++j) det_any |= (__float_as_int(sRed[(le * A + j) * kRedS + R_FLAGS]) & F_DET) != 0;
        {
            const float m = c.mask_value;
            const float row[SD] = {det_any ? rtx : m, det_any ? rty : m, det_any ? rtz : m, s.q.w, s.q.x, s.q.y, s.q.z, s.lin.x, s.lin.y, s.lin.z,
                                   heading.x, heading.y, heading.z, up.x, up.y, up.z, t, t, t, t};                  // :856-863
            wave_store_rows<SD, slab_rows(A)>(slab, b.obs_self + ((size_t)e0 * A + (tid & ~63)) * SD, row, lane, 64);
        }
        if constexpr (PROF) prof_mark(p.prof, 6);
    } else if (tid >= NA + kEPB) {
        // ================================= helper waves: cylinders ===================================================
        const int htid = tid - NA - kEPB, hw = htid >> 6;
        const int le = htid / A;
        if constexpr (WARM) warm_params(ka.rest);
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = CS ? CS : c.num_cylinders, K = CS ? 3 : c.obs_max_cylinder, E = c.stats_stride;
        const LdsSmall L = lds_layout_small(A, C, K);
        float *sPub = smem + L.pub, *sCyl = smem + L.cyl, *sTp = smem + L.tp, *sRed = smem + L.red;
        (void)sTp;
        // stage the workgroup's cylinders (one contiguous slice [64][3C]): helper wave w takes the 64-float passes w, w + A, ...;
        // every load issued before the first LDS write
        {
            const float *gc = b.cylinders + (size_t)e0 * C * 3 + lane;
            const int c3 = 3 * C;
            constexpr int kStage = (3 * (CS ? CS : HNS_MAX_CYLINDERS) + A - 1) / A;
            float sv[kStage];
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int pass = hw + i * A;
                sv[i] = pass < c3 ? gc[pass * 64] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int pass = hw + i * A;
                if (pass < c3) {
                    const int idx = pass * 64 + lane;
                    const int row = (int)__umulhi((unsigned)idx, p.cyl_magic), col = idx - row * c3;
                    sCyl[row * L.cyl_stride + col] = sv[i];
                }
            }
        }
        // lane <-> env: helper wave 0 fetches the eight statistics rows it updates behind barrier 3, helper wave 1 the six it updates behind barrier 1
        constexpr int kEarlyWave = A > 1 ? 1 : 0, kRewardWave = A > 2 ? 2 : A - 1;
        float st[HNS_NUM_STATS];
        if (hw == kRewardWave) st[HNS_ST_RETURN] = b.stats[(size_t)HNS_ST_RETURN * E + e0 + lane];
        if (hw == 0) {
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i)
                if (small_helper_stat(i)) st[i] = b.stats[(size_t)i * E + e0 + lane];
        }
        if (hw == kEarlyWave) {
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i)
                if (small_early_stat(i)) st[i] = b.stats[(size_t)i * E + e0 + lane];
        }
        if constexpr (PROF) prof_mark(p.prof, 2);
        __syncthreads();                                                            // barrier 1
        // The statistics rows fetched above are first used behind barrier 3, behind this wave's stores.  Left alone, the compiler waits for them
        // THERE with s_waitcnt vmcnt(n), n = the memory operations issued since — which makes the wave wait for its own freshly issued stores to be
        // acknowledged (measured: 2 500 cycles for 11 stores in the env wave's tail).  Pin the wait here, where only loads are outstanding.
        if (hw == 0) {
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i)
                if (small_helper_stat(i)) asm volatile("" : "+v"(st[i]));
        }
        if (hw == kRewardWave) asm volatile("" : "+v"(st[HNS_ST_RETURN]));
        if (hw == kEarlyWave) {
            // statistics that only need phase-1 data (hideandseek.py:731-733, :1097-1098, :996-997; :1017-1056 for the division at the episode's end)
            const int e = e0 + lane;
            float sum_ae = 0.f, sum_td = 0.f, max_td = 0.f;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float *red = sRed + (lane * A + j) * kRedS;
                const float td = red[R_TD];
                sum_ae = (j == 0) ? red[R_AERR] : sum_ae + red[R_AERR];
                sum_td = (j == 0) ? td : sum_td + td;
                max_td = (j == 0) ? td : (td > max_td ? td : max_td);
            }
            const float progress = sTp[kEPB * 3 + lane];
            const bool out_of_arena = smem[L.envout + lane] != 0.0f;               // the env wave's arena test (:1096-1098)
            const float mae = sum_ae * c.inv_num_agents;
            st[HNS_ST_ACTION_ERROR_ORDER1_MEAN] += mae;
            if (mae > st[HNS_ST_ACTION_ERROR_ORDER1_MAX]) st[HNS_ST_ACTION_ERROR_ORDER1_MAX] = mae;
            st[HNS_ST_OUT_OF_ARENA] = ((st[HNS_ST_OUT_OF_ARENA] != 0.0f) || out_of_arena) ? 1.0f : 0.0f;
            st[HNS_ST_SMOOTHNESS_COEF] = c.smoothness_coef;
            st[HNS_ST_SMOOTHNESS_MEAN] += sum_td * c.inv_num_agents;
            if (max_td > st[HNS_ST_SMOOTHNESS_MAX]) st[HNS_ST_SMOOTHNESS_MAX] = max_td;
            if (progress >= (float)c.max_episode_length) {
                st[HNS_ST_ACTION_ERROR_ORDER1_MEAN] = st[HNS_ST_ACTION_ERROR_ORDER1_MEAN] / progress;
                st[HNS_ST_SMOOTHNESS_MEAN] = st[HNS_ST_SMOOTHNESS_MEAN] / progress;
            }
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i)
                if (small_early_stat(i)) st_f1(b.stats + (size_t)i * E + e, st[i]);
        }
        if constexpr (PROF) prof_mark(p.prof, 3);
        __syncthreads();                                                            // barrier 2 (R_AERR / R_TD drained)
        if constexpr (PROF) prof_mark(p.prof, 8);
        // ---- phase 3a, the helper's share: the k nearest cylinders on S_{t+1}, cylinder collisions ----
        const float *pp = sPub + htid * kPub + 6;
        const V3 pos = {pp[0], pp[1], pp[2]};
        const float *cyl = sCyl + le * L.cyl_stride;
        int knn_idx[KM + 1];
        bool knn_masked[KM];
        bool blocked, blockedB;
        if constexpr (PROF) prof_mark(p.prof, 10);
        float cc = 0.f;
        cylinder_pass<1, false, KM, (CS ? 4 : KM + 1)>(c, C, K, pos, pos, pos, cyl, knn_idx, blocked, blockedB);
        if constexpr (PROF) prof_mark(p.prof, 9);
#pragma unroll
        for (int sidx = 0; sidx < KM; ++sidx) knn_masked[sidx] = (sidx < K) ? cyl[3 * knn_idx[sidx] + 2] < 0.0f : false;   // :759,775-778
        const float rc = c.cylinder_size + c.collision_radius, rc2 = rc * rc;
#pragma unroll
        for (int sidx = 0; sidx < KM; ++sidx) {
            if (sidx < K) {
                const float *cy = cyl + 3 * knn_idx[sidx];
                const float rx = pos.x - cy[0], ry = pos.y - cy[1];
                const float s2 = HNS_FMA(ry, ry, rx * rx);
                bool h = s2 < rc2 * 0.99999618530273f;
                if (!h && !(s2 > rc2 * 1.00000381469727f)) h = (__builtin_sqrtf(s2) - c.cylinder_size) < c.collision_radius;
                float hit = h ? 1.0f : 0.0f;
                if (knn_masked[sidx]) hit = 0.0f;
                cc = (sidx == 0) ? hit : cc + hit;
            }
        }
        {
            sRed[htid * kRedS + R_CC] = cc;
        }
        if constexpr (PROF) prof_mark(p.prof, 4);
        __syncthreads();                                                            // barrier 3
        if constexpr (PROF) prof_mark(p.prof, 5);
        {   // the k nearest cylinders (:767-778)
            float krow[kMaxK * 5];
            const float mv = c.mask_value, ch = c.cylinder_height, cs = c.cylinder_size;
#pragma unroll
            for (int sidx = 0; sidx < kMaxK; ++sidx) {
                const float *cy = cyl + 3 * ((sidx < K) ? knn_idx[sidx] : 0);
                const bool masked = knn_masked[sidx];
                const float rx = pos.x - cy[0], ry = pos.y - cy[1], rz = pos.z - cy[2];
                krow[sidx * 5] = masked ? mv : rx;
                krow[sidx * 5 + 1] = masked ? mv : ry;
                krow[sidx * 5 + 2] = masked ? mv : rz;
                krow[sidx * 5 + 3] = masked ? mv : ch;
                krow[sidx * 5 + 4] = masked ? mv : cs;
            }
            float *g = b.obs_cylinders + ((size_t)e0 * A + (htid & ~63)) * K * 5;
            float *slab = smem + L.hslab + hw * L.hslab_stride;
            if (K == 3) {
                float r[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) r[i] = krow[i];
                wave_store_rows<15, slab_rows(A)>(slab, g, r, lane, 64);
            } else if (K == 4) {
                wave_store_rows<20, slab_rows(A)>(slab, g, krow, lane, 64);
            } else if (K == 2) {
                float r[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) r[i] = krow[i];
                wave_store_rows<10, slab_rows(A)>(slab, g, r, lane, 64);
            } else {
                float r[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) r[i] = krow[i];
                wave_store_rows<5, slab_rows(A)>(slab, g, r, lane, 64);
            }
        }
        if (hw == kRewardWave) {
            // lane <-> env: the reward rows (hideandseek.py:919-1006) and their running sum
            float *sEnvOut = smem + L.envout;
            bool any_cap = false, det_any = false;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const int fl = __float_as_int(sRed[(lane * A + j) * kRedS + R_FLAGS]);
                any_cap |= (fl & F_CAP) != 0;
                det_any |= (fl & F_DET) != 0;
            }
            const float detect_rew = c.detect_reward_coef * (det_any ? 1.0f : 0.0f);
            const float catch_rew = c.catch_reward_coef * (any_cap ? 1.0f : 0.0f);
            float sum_rew = 0.f;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float *red = sRed + (lane * A + j) * kRedS;
                float cr = -c.collision_coef * red[R_CC];
                cr = cr + -c.collision_coef * red[R_CD];
                cr = cr + -c.collision_coef * red[R_CW];
                const float r = ((((red[R_DIST] + detect_rew) + catch_rew) + cr) + red[R_SPEED]) + red[R_SMOOTH];
                sEnvOut[lane * A + j] = r;
                sum_rew = (j == 0) ? r : sum_rew + r;
            }
            env_store_slice<false>(sEnvOut, b.reward + (size_t)e0 * A, kEPB * A, lane, kEPB * A);
            flag_nonfinite(b.nonfinite, (sum_rew - sum_rew) != 0.0f, 4u);
            st[HNS_ST_RETURN] += sum_rew * c.inv_num_agents;
            st_f1(b.stats + (size_t)HNS_ST_RETURN * E + e0 + lane, st[HNS_ST_RETURN]);
        }
        if (hw == 0) {
            // lane <-> env: the statistics that are sums of one reward term over the env's pursuers (hideandseek.py:960-1056)
            const int e = e0 + lane;
            const float iA = c.inv_num_agents;
            const float progress = sTp[kEPB * 3 + lane];
            bool any_coll = false;
            float sum_dist = 0, sum_speed = 0, sum_cc = 0, sum_cd = 0, sum_cw = 0, sum_coll = 0, sum_smooth = 0;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float *red = sRed + (lane * A + j) * kRedS;
                float cr = -c.collision_coef * red[R_CC];
                cr = cr + -c.collision_coef * red[R_CD];
                cr = cr + -c.collision_coef * red[R_CW];
                any_coll |= cr < 0.0f;
                if (j == 0) {
                    sum_dist = red[R_DIST]; sum_speed = red[R_SPEED]; sum_cc = red[R_CC]; sum_cd = red[R_CD]; sum_cw = red[R_CW];
                    sum_coll = cr; sum_smooth = red[R_SMOOTH];
                } else {
                    sum_dist += red[R_DIST]; sum_speed += red[R_SPEED]; sum_cc += red[R_CC]; sum_cd += red[R_CD]; sum_cw += red[R_CW];
                    sum_coll += cr; sum_smooth += red[R_SMOOTH];
                }
            }
            st[HNS_ST_DISTANCE_REWARD] += sum_dist * iA;
            st[HNS_ST_SPEED_REWARD] += sum_speed * iA;
            st[HNS_ST_COLLISION_CYLINDER] += sum_cc * iA;
            st[HNS_ST_COLLISION_DRONE] += sum_cd * iA;
            st[HNS_ST_COLLISION] += any_coll ? 1.0f : 0.0f;
            st[HNS_ST_COLLISION_WALL] += sum_cw * iA;
            st[HNS_ST_COLLISION_REWARD] += sum_coll * iA;
            st[HNS_ST_SMOOTHNESS_REWARD] += sum_smooth * iA;
            if (progress >= (float)c.max_episode_length) {                          // :1017-1056
#pragma unroll
                for (int i = 0; i < HNS_NUM_STATS; ++i)
                    if (small_helper_stat(i)) st[i] = st[i] / progress;
            }
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i)
                if (small_helper_stat(i)) st_f1(b.stats + (size_t)i * E + e, st[i]);
        }
        if constexpr (PROF) prof_mark(p.prof, 6);
    } else {
        // ================================= env wave: lane <-> env ========================================
        __builtin_amdgcn_s_setprio(2);
        const int le = lane;
        const int e = e0 + le;
        if constexpr (WARM) warm_params(ka.rest);
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = CS ? CS : c.num_cylinders, K = CS ? 3 : c.obs_max_cylinder, E = c.stats_stride;
        const LdsSmall L = lds_layout_small(A, C, K);
        float *sPub = smem + L.pub, *sTp = smem + L.tp, *sRed = smem + L.red, *sEnvOut = smem + L.envout;
        const float *gt = b.target_pos + (size_t)e * 3;
        const V3 tp0 = {gt[0], gt[1], gt[2]};
        float progress = b.progress[e];
        float st[HNS_NUM_STATS];
#pragma unroll
        for (int i = 0; i < HNS_NUM_STATS; ++i)
            if (small_env_stat(i)) st[i] = b.stats[(size_t)i * E + e];
        progress += 1.0f;                                                           // isaac_env.py:236
        sTp[kEPB * 3 + le] = progress;
        if constexpr (PROF) prof_mark(p.prof, 1);
        bool out_of_arena = false;
        const V3 Fenv = d_prey_arena_term(c, tp0, out_of_arena);                    // hideandseek.py:1090-1112
        sEnvOut[le] = out_of_arena ? 1.0f : 0.0f;                                  // for the helper that keeps the early statistics
        // cylinder terms (:1114-1136): this lane's env straight from memory (the helpers' staging is only complete at barrier 1, and this wave
        // has time now — behind the barrier the owners would wait for it)
        float fcx = 0.f, fcy = 0.f;
        {
            const float *cylg = b.cylinders + (size_t)e * C * 3;
#pragma unroll 4
            for (int k = 0; k < C; ++k) {
                float tx, ty;
                d_prey_cylinder_term(c, tp0, cylg[3 * k], cylg[3 * k + 1], cylg[3 * k + 2], tx, ty);
                fcx += tx;
                fcy += ty;
            }
        }
        if constexpr (PROF) prof_mark(p.prof, 2);
        __syncthreads();                                                            // barrier 1: positions at t, flags, action errors, staged cylinders
        if constexpr (PROF) prof_mark(p.prof, 12);
#pragma unroll
        for (int i = 0; i < HNS_NUM_STATS; ++i)                                     // (the wait for these rows: here, not behind the stores — see the helpers)
            if (small_env_stat(i)) asm volatile("" : "+v"(st[i]));
        V3 F = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < A; ++j) {                                               // the pursuers' pushes (:1074-1088), ascending
            const float *pj = sPub + (le * A + j) * kPub;
            const V3 dp = {pj[0], pj[1], pj[2]};
            const int los = (int)pj[10];
            const V3 fp = d_prey_pursuer_term(c, dp, tp0, (los & 1) != 0);
            F.x = (j == 0) ? fp.x : F.x + fp.x;
            F.y = (j == 0) ? fp.y : F.y + fp.y;
            F.z = (j == 0) ? fp.z : F.z + fp.z;
        }
        F.x = F.x + Fenv.x; F.y = F.y + Fenv.y; F.z = F.z + Fenv.z;
        F.x = F.x + fcx; F.y = F.y + fcy; F.z = F.z + 0.0f;
        const V3 tvel = {(c.v_prey * F.x) / (__builtin_fabsf(F.x) + 1e-5f), (c.v_prey * F.y) / (__builtin_fabsf(F.y) + 1e-5f),
                         (c.v_prey * F.z) / (__builtin_fabsf(F.z) + 1e-5f)};        // per-axis speed (:741)
        const V3 tpn = {tp0.x + tvel.x * c.dt, tp0.y + tvel.y * c.dt, tp0.z + tvel.z * c.dt};
        sTp[le * 3] = tpn.x; sTp[le * 3 + 1] = tpn.y; sTp[le * 3 + 2] = tpn.z;
        { const float sf = (tpn.x + tpn.y) + tpn.z; flag_nonfinite(b.nonfinite, (sf - sf) != 0.0f, 2u); }
        if constexpr (PROF) prof_mark(p.prof, 3);
        __syncthreads();                                                            // barrier 2: evader at t+1 published
        if constexpr (PROF) prof_mark(p.prof, 8);
        {
            sEnvOut[le * 3] = tvel.x; sEnvOut[le * 3 + 1] = tvel.y; sEnvOut[le * 3 + 2] = tvel.z;
            env_store_slice<false>(sTp, b.target_pos + (size_t)e0 * 3, kEPB * 3, lane, kEPB * 3);
            env_store_slice<false>(sEnvOut, b.target_vel + (size_t)e0 * 3, kEPB * 3, lane, kEPB * 3);
        }
        if constexpr (PROF) prof_mark(p.prof, 4);
        __syncthreads();                                                            // barrier 3: reward terms and flags
        if constexpr (PROF) prof_mark(p.prof, 5);
        // ---- phase 3b: flags, done, the env wave's seven statistics (hideandseek.py:919-1065) ----
        const float iA = c.inv_num_agents;
        bool any_cap = false, all_blocked = true, det_any = false;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const int fl = __float_as_int(sRed[(le * A + j) * kRedS + R_FLAGS]);
            any_cap |= (fl & F_CAP) != 0;
            all_blocked &= (fl & F_BLOCKED) != 0;
            det_any |= (fl & F_DET) != 0;
        }
        const float detf = det_any ? 1.0f : 0.0f;
        const float detect_rew = c.detect_reward_coef * detf;
        const float catch_rew = c.catch_reward_coef * (any_cap ? 1.0f : 0.0f);
        if constexpr (PROF) prof_mark(p.prof, 10);
#define ST(i) st[i]
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
        const bool done = progress >= (float)c.max_episode_length;                // :1008-1010
        if (done) {                                                               // :1017-1056 (the other ten: the helpers)
            ST(HNS_ST_TARGET_PREDICTED_ERROR) = ST(HNS_ST_TARGET_PREDICTED_ERROR) / progress;
            ST(HNS_ST_DETECT_REWARD) = ST(HNS_ST_DETECT_REWARD) / progress;
            ST(HNS_ST_CATCH_REWARD) = ST(HNS_ST_CATCH_REWARD) / progress;
        }
#undef ST
        if constexpr (PROF) prof_mark(p.prof, 11);
        b.done[e] = (uint8_t)done;
        if (b.detect) b.detect[e] = (uint8_t)(det_any ? 1 : 0);
        b.progress[e] = progress;
#pragma unroll
        for (int i = 0; i < HNS_NUM_STATS; ++i)
            if (small_env_stat(i)) st_f1(b.stats + (size_t)i * E + e, st[i]);
        if constexpr (PROF) prof_mark(p.prof, 6);
    }
    if constexpr (PROF) prof_mark(p.prof, 7);
    if constexpr (PROF) prof_mark(p.prof, 15);
}

}  // namespace hns
