// hns_reset_kernel.h — masked reset + reset-time observation (A11); instantiated per pursuer count by hns_inst.hip.
#pragma once
#include "hns_common.h"

namespace hns {

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
    __shared__ uint8_t sMask[kEPB];      // envs being reset
    __shared__ uint8_t sTouch[kEPB];     // envs whose state / observation this launch rewrites: the reset ones, or all with cfg.reset_extra_step
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
        sTouch[tid] = (tid < nenv) ? (sMask[tid] | (c.reset_extra_step != 0)) : 0;
        sDet[tid] = 0;
        sDet1[tid] = 0;
    }
    __syncthreads();
    const bool masked = valid && sMask[le];
    const bool touch = valid && sTouch[le];

    if (env_wave && valid) {
        // hideandseek.py:712 resets first_capture_step for ALL envs on any reset call
        b.stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * c.stats_stride + e] = (float)c.max_episode_length;
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
            if (c.pid_reset_on_reset) {       // 0 = the reference: `_reset_idx` leaves the controller alone (reset_pid at the next step clears it)
                reinterpret_cast<float4 *>(b.pid_integ)[ja] = make_float4(0, 0, 0, 0);
                reinterpret_cast<float4 *>(b.pid_last_rate)[ja] = make_float4(0, 0, 0, 0);
            }
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
        if (p.tasks && !task) {
            // envgen: a uniformly sampled task is archived as SAMPLED (hideandseek_envgen.py:883-895 inserts `tasks_unif` before the sim.step of
            // :1013) — the placement goes back into the caller's task row here, before the extra step below moves the bodies
            float *row = p.tasks + (size_t)e * (3 * A + 3 * NT + 3 * C);
            for (int j = 0; j < A; ++j) { row[3 * j] = ds[13 * j]; row[3 * j + 1] = ds[13 * j + 1]; row[3 * j + 2] = ds[13 * j + 2]; }
            for (int i = 0; i < 3 * NT; ++i) row[3 * A + i] = tp[i];
            for (int k = 0; k < 3 * C; ++k) row[3 * A + 3 * NT + k] = cyl[k];
        }
        for (int sidx = 0; sidx < HNS_NUM_STATS; ++sidx) b.stats[(size_t)sidx * c.stats_stride + e] = 0.0f;   // :711
        b.stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * c.stats_stride + e] = (float)c.max_episode_length;
        b.progress[e] = 0.0f;
        b.done[e] = 0;
    }
    if (env_wave && touch && c.reset_extra_step) {
        // hideandseek.py:722-723: `_reset_idx` ends with one sim.step() of the whole scene.  No rotor force is applied in it (apply_action
        // runs in _pre_sim_step only): every drone integrates one dt under gravity and damping, every evader (gravity disabled, :544-565)
        // moves one dt with the velocity it holds — envs that are not being reset included.
        float *ds = sDS + le * A * 13;
        float *tp = sTp + le * 3 * NT;
        if (!masked) {                                        // not regenerated above: the state the buffers hold
            const float *gds = b.drone_state + (size_t)e * A * 13;
            for (int i = 0; i < A * 13; ++i) ds[i] = gds[i];
            for (int i = 0; i < 3 * NT; ++i) tp[i] = b.target_pos[(size_t)e * 3 * NT + i];
            float *cyl = sCyl + le * L.cyl_stride;
            for (int k = 0; k < 3 * C; ++k) cyl[k] = b.cylinders[(size_t)e * C * 3 + k];
        }
        const V3 zero = {0.f, 0.f, 0.f};
        for (int j = 0; j < A; ++j) {
            Rigid s;
            load_rigid(ds + 13 * j, s);
            d_integrate(c, s, zero, zero);
            store_rigid(ds + 13 * j, s);
        }
        for (int i = 0; i < 3 * NT; ++i) tp[i] = tp[i] + b.target_vel[(size_t)e * 3 * NT + i] * c.dt;
    }
    __syncthreads();
    if (!env_wave && touch) {
        Rigid s;
        load_rigid(sDS + tid * 13, s);
        V3 tp = {sTp[le * 3 * NT], sTp[le * 3 * NT + 1], sTp[le * 3 * NT + 2]};
        V3 tpB = tp;
        if constexpr (NT == 2) tpB = {sTp[le * 3 * NT + 3], sTp[le * 3 * NT + 4], sTp[le * 3 * NT + 5]};
        const size_t ia = (size_t)e0 * A + tid;
        bool blocked, det, blockedB = false, detB = false;
        int knn_idx[KM];
        bool knn_masked[KM];
        agent_obs<A, NT, false, 13, KM>(c, C, K, le, a, s, tp, tpB, masked ? 0.0f : b.progress[e], sCyl + le * L.cyl_stride, sDS, b.obs_others + ia * (A - 1) * 3, sOCyl, b.obs_self + ia * SD,
                                        with_state ? b.state_drones + ia * SD : nullptr, blocked, det, blockedB, detB, knn_idx, knn_masked, true, true,
                                        b.obs_cylinders + ia * K * 5);
        b.pid_last_rate[ia * 4 + 3] = (float)((blocked ? 1 : 0) + (blockedB ? 2 : 0));   // line of sight of the new state (the env wave zeroed the record above)
        if (det) sDet[le] = 1;
        if (NT == 2 && detB) sDet1[le] = 1;
    }
    __syncthreads();
    if (env_wave && touch && !sDet[le]) {                      // hideandseek.py:791-794
        for (int j = 0; j < A; ++j) {
            float *o = b.obs_self + ((size_t)e * A + j) * SD;
            o[0] = c.mask_value; o[1] = c.mask_value; o[2] = c.mask_value;
        }
    }
    if (NT == 2 && env_wave && touch && !sDet1[le]) {
        for (int j = 0; j < A; ++j) {
            float *o = b.obs_self + ((size_t)e * A + j) * SD + HNS_SELF_DIM;
            o[0] = c.mask_value; o[1] = c.mask_value; o[2] = c.mask_value;
        }
    }
    if (env_wave && touch && b.detect) b.detect[e] = (uint8_t)(sDet[le] | (NT == 2 ? (sDet1[le] << 1) : 0));
    coop_s2g_masked<T>(b.drone_state + (size_t)e0 * A * 13, sDS, nenv * A * 13, A * 13, sTouch);
    coop_cyl<T, false>(sCyl, b.cylinders + (size_t)e0 * C * 3, nenv, 3 * C, L.cyl_stride, p.cyl_magic, sMask);
    coop_s2g_masked<T>(b.target_pos + (size_t)e0 * 3 * NT, sTp, nenv * 3 * NT, 3 * NT, sTouch);
    if constexpr (KM == kMaxK) coop_s2g_masked<T>(b.obs_cylinders + (size_t)e0 * A * K * 5, sOCyl, nenv * A * K * 5, A * K * 5, sTouch);
}

}  // namespace hns
