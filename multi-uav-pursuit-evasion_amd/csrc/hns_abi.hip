// hns_abi.hip — the C ABI of include/hns.h (host side) and the small kernels that are not instantiated per pursuer count:
// Hover (BASELINE config 1), the ray fan, the line-of-sight refresh, the copy yardstick.  The step and reset kernels live in
// hns_step_kernel.h / hns_reset_kernel.h and are instantiated by hns_inst.hip, one translation unit per pursuer count.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "hns_common.h"
#include "hns_host.h"

namespace hns {

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

// hns_moments / hns_rollout_moments: [sum, sum of squares, count, success sum, env count (, sum, sum of squares, count of a second array)] in fp64, one
// workgroup, fixed order (thread t takes elements t, t + 1024, ...; then a tree over the 1024 partial sums)
HNS_DEV void moments_pass(const float *__restrict__ v, long long n, int t, double &a, double &b) {
    const long long n4 = ((reinterpret_cast<uintptr_t>(v) & 15) == 0) ? n / 4 : 0;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    for (long long i = t; i < n4; i += 1024) {
        const float4 q = v4[i];
        const double x0 = q.x, x1 = q.y, x2 = q.z, x3 = q.w;
        a += (x0 + x1) + (x2 + x3);
        b += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
    }
    for (long long i = 4 * n4 + t; i < n; i += 1024) { const double x = v[i]; a += x; b += x * x; }
}
__global__ __launch_bounds__(1024) void hns_moments_kernel(const float *__restrict__ v, long long n, const float *__restrict__ s, long long m,
                                                             const float *__restrict__ v2, long long n2, double *__restrict__ out) {
    __shared__ double red[5][1024];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0, c = 0.0, a2 = 0.0, b2 = 0.0;
    moments_pass(v, n, t, a, b);
    if (v2) moments_pass(v2, n2, t, a2, b2);
    for (long long i = t; i < m; i += 1024) c += (double)s[i];
    red[0][t] = a; red[1][t] = b; red[2][t] = c; red[3][t] = a2; red[4][t] = b2;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if (t < w) {
#pragma unroll
            for (int r = 0; r < 5; ++r) red[r][t] += red[r][t + w];
        }
        __syncthreads();
    }
    if (t == 0) {
        out[0] = red[0][0]; out[1] = red[1][0]; out[2] = (double)n; out[3] = red[2][0]; out[4] = (double)m;
        if (v2) { out[5] = red[3][0]; out[6] = red[4][0]; out[7] = (double)n2; }
    }
}

// hns_clock_probe: one wave spins for `ticks` periods of the chip-wide constant 100 MHz clock (s_memrealtime) and reports how many SHADER-clock cycles
// (s_memtime) went by: out[0] = shader cycles, out[1] = 100 MHz ticks -> MHz = 100 out[0] / out[1].  The chip clocks to its power budget (DVFS); a launch-bound
// region a few hundred microseconds into a process's life runs on clocks that are still moving — bench.py prints what this reads before and after its timed region.
__global__ __launch_bounds__(64) void hns_clock_probe_kernel(unsigned long long *__restrict__ out, unsigned ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_readcyclecounter();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// measurement yardstick (hns_copy_f4): a plain float4 copy, one piece per thread.  Of the shapes tried on this chip (tools/microbench/copy_rate.hip:
// 4 / 8 pieces per thread, persistent grid-stride grids, non-temporal accesses, hipMemcpyAsync) this simplest one is the fastest or within 5 % of the
// fastest at every size: 6.6-7.1 TB/s for footprints the Infinity Cache holds, 6.0-6.25 TB/s beyond it (MI355X_MICROARCH.md: 6.29).
__global__ __launch_bounds__(256) void hns_copy_f4_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

}  // namespace hns

// =================================================================================================
// Host side: the C ABI (include/hns.h)
// =================================================================================================
using hns::Params;

static thread_local std::string g_last_error;
void hns_set_error(const std::string &m) { g_last_error = m; }
static void set_error(const std::string &m) { g_last_error = m; }


// hns_inst.hip, one translation unit per pursuer count
void hns_select_kernels_1(hns_env *), hns_select_kernels_2(hns_env *), hns_select_kernels_3(hns_env *), hns_select_kernels_4(hns_env *),
    hns_select_kernels_5(hns_env *), hns_select_kernels_6(hns_env *), hns_select_kernels_7(hns_env *);

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
    if (cfg->action_input != HNS_ACTION_POLICY && cfg->action_input != HNS_ACTION_MOTOR) { set_error("hns_create: action_input must be HNS_ACTION_POLICY or HNS_ACTION_MOTOR"); return HNS_ERR_INVALID_ARG; }
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
    if (env->cfg.stats_stride == 0) env->cfg.stats_stride = cfg->num_envs;
    if (env->cfg.stats_stride < cfg->num_envs) { delete env; set_error("hns_create: stats_stride must be 0 or >= num_envs"); return HNS_ERR_INVALID_ARG; }
    if (hipGetDevice(&env->device) != hipSuccess) env->device = 0;
    std::memset(&env->buf, 0, sizeof(env->buf));
    if (hipDeviceGetAttribute(&env->cus, hipDeviceAttributeMultiprocessorCount, env->device) != hipSuccess || env->cus < 1) env->cus = 256;
    switch (cfg->num_agents) {
        case 1: hns_select_kernels_1(env); break;
        case 2: hns_select_kernels_2(env); break;
        case 3: hns_select_kernels_3(env); break;
        case 4: hns_select_kernels_4(env); break;
        case 5: hns_select_kernels_5(env); break;
        case 6: hns_select_kernels_6(env); break;
        case 7: hns_select_kernels_7(env); break;
        default: delete env; set_error("hns_create: unsupported num_agents"); return HNS_ERR_INVALID_ARG;
    }
    size_t lds_max = env->lds_reset > env->lds_step ? env->lds_reset : env->lds_step;
    if (lds_max > 160 * 1024) {
        delete env;
        set_error("hns_create: configuration needs more than 160 KiB LDS per workgroup");
        return HNS_ERR_CONFIG;
    }
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->step_args_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_step);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->reset_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_reset);
    if (e1 == hipSuccess && env->step_args_prof_fn)
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(env->step_args_prof_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->lds_step);
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
    if (env->capture_pool) (void)hipHostFree(env->capture_pool);
    for (auto &ev : env->region_ev) if (ev) (void)hipEventDestroy(ev);
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
    p.prio_boost = (uint32_t)env->prio_boost;
}

// Device copy of that block for the step kernel that reads it through `StepArgs::rest`: one allocation in hns_create, refreshed
// where the block changes (bind, the configuration setters, the profiling buffer), never from a steady-state step.
static int alloc_step_params(hns_env *env) {
    HNS_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&env->params_dev), sizeof(Params)));
    HNS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&env->params_ring), sizeof(Params) * hns_env::kParamRing, hipHostMallocDefault));
    HNS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&env->capture_pool), sizeof(Params) * hns_env::kCaptureImages, hipHostMallocDefault));
    for (auto &ev : env->ring_events) HNS_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    env->params_host = new Params;
    return HNS_OK;
}

// A change travels as ONE stream-ordered copy of the block from a pinned image, enqueued on the stream of the latest step / reset
// call: launches already enqueued there keep the old values, later ones see the new ones; no device synchronisation, no allocation,
// legal inside a stream capture (the image a capture takes is then kept for the graph's lifetime).  The very first upload (hns_bind
// before any launch) is a plain blocking copy: nothing reads the block yet.
static int upload_step_params(hns_env *env) {
    if (!env->bound) return HNS_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != env->device) {
        set_error("the current HIP device is not the one this env was created on (hipSetDevice first)");
        return HNS_ERR_DEVICE;
    }
    Params q;
    fill_step_params(env, q);
    // unchanged since the last upload: nothing to do — unless a stream capture once took a change: a replay of that graph rewrites the device
    // block behind the host's back, so the last image enqueued from here no longer says what the block holds (found by
    // test_setter_inside_a_capture_in_the_default_mode: the third replay flew the captured speed)
    if (env->params_valid && env->capture_used == 0 && memcmp(&q, env->params_host, sizeof(Params)) == 0) return HNS_OK;
    memcpy(env->params_host, &q, sizeof(Params));
    if (!env->params_valid) {
        HNS_CHECK_HIP(hipMemcpy(env->params_dev, env->params_host, sizeof(Params), hipMemcpyHostToDevice));
        env->params_valid = true;
        return HNS_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(env->last_stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) {
        // nothing may be allocated while a capture is active (hipHostMalloc is refused in the default capture mode): the images come from
        // a pool made by hns_create, one per captured change, never reused (a graph may replay its copy node for as long as it lives)
        if (env->capture_used >= hns_env::kCaptureImages) {
            set_error("a stream capture changed the configuration more than 16 times over this env's lifetime (pinned image pool exhausted)");
            return HNS_ERR_CONFIG;
        }
        Params *img = env->capture_pool + env->capture_used++;
        memcpy(img, &q, sizeof(Params));
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

static int check_device(hns_env *env) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != env->device) {
        set_error("hns_step / hns_reset: the current HIP device is not the one this env was created on");
        return HNS_ERR_DEVICE;
    }
    return HNS_OK;
}

// The step launch: 64 B of arguments (StepArgs), everything else through the device-resident block behind `rest`.
static int launch_step(hns_env *env, const float *action, hipStream_t stream) {
    if (const int rc = check_device(env)) return rc;
    env->last_stream = stream;
    if (!env->params_valid) { set_error("hns_step: the device copy of the launch parameters is missing (bind first)"); return HNS_ERR_NOT_BOUND; }
    const hns_buffers &b = env->buf;
    const void *aux = env->cfg.num_targets == 2
                          ? reinterpret_cast<const void *>(reinterpret_cast<uintptr_t>(b.cylinders) | (uintptr_t)((env->cfg.num_cylinders - 1) & 15))
                          : static_cast<const void *>(b.reset_pid);
    auto fn = (env->prof && env->step_args_prof_fn) ? env->step_args_prof_fn : env->step_args_fn;
    if (env->timing > 0 && (env->step_count++ % (uint64_t)env->timing) == 0) {
        std::pair<hipEvent_t, hipEvent_t> ev{};
        if (!env->pool.empty()) { ev = env->pool.back(); env->pool.pop_back(); }
        else {
            HNS_CHECK_HIP(hipEventCreate(&ev.first));
            HNS_CHECK_HIP(hipEventCreate(&ev.second));
        }
        // the events ride on the dispatch itself (start / stop of THIS kernel, the timestamps a profiler reads),
        // not on separate marker packets before and after it
        hipExtLaunchKernelGGL(fn, dim3(env->grid), dim3(env->threads_step), (uint32_t)env->lds_step, stream, ev.first, ev.second, 0, static_cast<const hns::Params *>(env->params_dev),
                              action, b.prev_action, b.drone_state, aux, b.pid_integ, b.pid_last_rate, b.throttle);
        HNS_CHECK_HIP(hipGetLastError());
        env->events.push_back(ev);
        return HNS_OK;
    }
    hipLaunchKernelGGL(fn, dim3(env->grid), dim3(env->threads_step), env->lds_step, stream, static_cast<const hns::Params *>(env->params_dev), action, b.prev_action,
                       b.drone_state, aux, b.pid_integ, b.pid_last_rate, b.throttle);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

static int launch_reset(hns_env *env, const Params &p, hipStream_t stream) {
    if (const int rc = check_device(env)) return rc;
    env->last_stream = stream;
    hipLaunchKernelGGL(env->reset_fn, dim3(env->grid), dim3(env->threads), env->lds_reset, stream, p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_step(hns_env *env, const float *action, void *stream) {
    if (!env || !action) { set_error("hns_step: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound) { set_error("hns_step: buffers not bound"); return HNS_ERR_NOT_BOUND; }
    if (reinterpret_cast<uintptr_t>(action) & 15) { set_error("hns_step: action must be 16-byte aligned"); return HNS_ERR_INVALID_ARG; }
    return launch_step(env, action, static_cast<hipStream_t>(stream));
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
    return launch_reset(env, p, static_cast<hipStream_t>(stream));
}

int hns_reset_tasks(hns_env *env, const uint8_t *reset_mask, float *tasks, int32_t task_first, uint64_t seed, void *stream) {
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
    return launch_reset(env, p, static_cast<hipStream_t>(stream));
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
        {host->progress, d.progress, E * 4}, {host->stats, d.stats, c.stats_stride == c.num_envs ? (size_t)HNS_NUM_STATS * E * 4 : 0 /* a slice's columns: the pitched copy below */}, {host->obs_self, d.obs_self, E * A * SD * 4},
        {host->obs_others, d.obs_others, E * A * (A - 1) * 12}, {host->obs_cylinders, d.obs_cylinders, E * A * K * 20},
        {host->state_drones, d.state_drones, E * A * SD * 4}, {host->reward, d.reward, E * A * 4}, {host->action_error, d.action_error, E * A * 4},
        {host->done, d.done, E}, {host->detect, d.detect, E}, {host->nonfinite, d.nonfinite, 4}, {host->ctbr, d.ctbr, E * A * 16}, {host->target_rate, d.target_rate, E * A * 16}};
    for (const Field &x : f) {
        if (!x.host || !x.dev || x.bytes == 0) continue;
        if (to_device) HNS_CHECK_HIP(hipMemcpyAsync(x.dev, x.host, x.bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        else HNS_CHECK_HIP(hipMemcpyAsync(const_cast<void *>(x.host), x.dev, x.bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    }
    if (host->stats && d.stats && c.stats_stride != c.num_envs) {
        // a handle over a slice of a larger batch (stats rows strided by the owner's env count): the HOST array is this handle's dense
        // [HNS_NUM_STATS, num_envs], the device side its columns of every row
        const size_t hp = E * 4, dp = (size_t)c.stats_stride * 4;
        if (to_device) HNS_CHECK_HIP(hipMemcpy2DAsync(d.stats, dp, host->stats, hp, hp, HNS_NUM_STATS, hipMemcpyHostToDevice, (hipStream_t)stream));
        else HNS_CHECK_HIP(hipMemcpy2DAsync(const_cast<float *>(host->stats), hp, d.stats, dp, hp, HNS_NUM_STATS, hipMemcpyDeviceToHost, (hipStream_t)stream));
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

int hns_step_mapping(const hns_env *env) { return env ? env->small_mapping : HNS_ERR_INVALID_ARG; }

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

int hns_region_begin(hns_env *env, void *stream) {
    if (!env) { set_error("hns_region_begin: null argument"); return HNS_ERR_INVALID_ARG; }
    for (auto &ev : env->region_ev)
        if (!ev) HNS_CHECK_HIP(hipEventCreate(&ev));
    HNS_CHECK_HIP(hipEventRecord(env->region_ev[0], static_cast<hipStream_t>(stream)));
    env->region_state = 1;
    return HNS_OK;
}
int hns_region_end(hns_env *env, void *stream) {
    if (!env || env->region_state != 1) { set_error("hns_region_end: no region begun"); return HNS_ERR_INVALID_ARG; }
    HNS_CHECK_HIP(hipEventRecord(env->region_ev[1], static_cast<hipStream_t>(stream)));
    env->region_state = 2;
    return HNS_OK;
}
float hns_region_ms(hns_env *env) {
    if (!env || env->region_state != 2) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(env->region_ev[1]) != hipSuccess || hipEventElapsedTime(&ms, env->region_ev[0], env->region_ev[1]) != hipSuccess) return -1.0f;
    return ms;
}

int hns_moments(const float *values, int64_t n, const float *success, int64_t m, double *out, void *stream) {
    if (!values || !out || n < 0 || m < 0 || (m > 0 && !success)) { set_error("hns_moments: bad argument"); return HNS_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(hns::hns_moments_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), values, (long long)n, success, (long long)m,
                       static_cast<const float *>(nullptr), 0ll, out);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_rollout_moments(const float *advantages, int64_t n, const float *success, int64_t m, const float *returns, int64_t n_returns, double *out, void *stream) {
    if (!advantages || !returns || !out || n < 0 || m < 0 || n_returns < 0 || (m > 0 && !success)) { set_error("hns_rollout_moments: bad argument"); return HNS_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(hns::hns_moments_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), advantages, (long long)n, success, (long long)m,
                       returns, (long long)n_returns, out);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_clock_probe(unsigned long long *out, uint32_t ticks, void *stream) {
    if (!out || ticks == 0 || ticks > 1000000u) { set_error("hns_clock_probe: out (device, 2 x u64) and 1 <= ticks <= 1e6 (10 ms)"); return HNS_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(hns::hns_clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, ticks);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

int hns_copy_f4(void *dst, const void *src, size_t bytes, void *stream) {
    if (!dst || !src || (bytes & 15) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15)) {
        set_error("hns_copy_f4: 16-byte aligned device pointers and a multiple of 16 bytes");
        return HNS_ERR_INVALID_ARG;
    }
    const size_t n4 = bytes / 16;
    if (n4 == 0) return HNS_OK;
    const size_t blocks = (n4 + 255) / 256;
    if (blocks > 0x7fffffffull) { set_error("hns_copy_f4: more than 2^31 workgroups"); return HNS_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(hns::hns_copy_f4_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<float4 *>(dst), static_cast<const float4 *>(src), n4);
    HNS_CHECK_HIP(hipGetLastError());
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

