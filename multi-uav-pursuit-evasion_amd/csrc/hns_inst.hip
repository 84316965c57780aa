// hns_inst.hip — the step and reset kernels instantiated for ONE pursuer count (compiled once per count: -DHNS_INST_A=1 ... 7, side
// by side; __graft_entry__.build), and the host-side choice among them.  Two translation units per count: the tile mapping + reset
// kernels, and (-DHNS_INST_SMALL=1) the small-batch mapping.  Both are compiled with their pointer parameters preloaded into SGPRs
// (-mllvm -amdgpu-kernarg-preload-count=16, __graft_entry__.PRELOAD_FLAGS: -0.15 ... -0.4 us per step below 32 768 envs in the small-batch mapping, round 4;
// the tile mapping since round 5, 16.28 -> 16.11 us at 65 536 envs, nothing either way for the 6v2 shard — profiles/r05_kernarg_preload_ab.txt; the flag also
// covers hns_reset_kernel's by-value Params, which simply do not fit the 16 preloaded dwords).
#include "hns_host.h"

#include <cstdlib>
#include <cstring>

#ifndef HNS_INST_A
#error "compile with -DHNS_INST_A=<pursuers per env>"
#endif
#define HNS_CAT2(a, b) a##b
#define HNS_CAT(a, b) HNS_CAT2(a, b)

#ifdef HNS_INST_SMALL
#include "hns_step_small_kernel.h"

// hns_step_small_kernel for an env the caller found eligible (one evader, whole tiles, k <= 4)
void HNS_CAT(hns_select_small_, HNS_INST_A)(hns_env *env) {
    constexpr int A = HNS_INST_A;
    using namespace hns;
    const hns_cfg &c = env->cfg;
    env->small_mapping = 1;
    env->step_args_fn = hns_step_small_kernel<A, false>;
    env->step_args_prof_fn = hns_step_small_kernel<A, true>;
    if (c.obs_max_cylinder == 3) {
        switch (c.num_cylinders) {
            case 5: env->step_args_fn = hns_step_small_kernel<A, false, 5>; break;
            case 8: env->step_args_fn = hns_step_small_kernel<A, false, 8>; break;
            default: break;
        }
    }
    env->threads_step = GeoSmall<A>::T;
    env->lds_step = (size_t)lds_layout_small(A, c.num_cylinders, c.obs_max_cylinder).total * sizeof(float);
}

#else
#include "hns_reset_kernel.h"
#include "hns_step_kernel.h"

void HNS_CAT(hns_select_small_, HNS_INST_A)(hns_env *env);

// Which instantiation serves an env: whole 64-env tiles with k <= 4 take the tuned kernel (and, with a phase-profile
// buffer attached, its stamped twin); ragged batches and wider selections take the generic one.
void HNS_CAT(hns_select_kernels_, HNS_INST_A)(hns_env *env) {
    constexpr int A = HNS_INST_A;
    using namespace hns;
    const hns_cfg &c = env->cfg;
    const bool two = c.num_targets == 2, wide = c.obs_max_cylinder > kMaxK, ragged = c.num_envs % kEPB != 0;
    const bool motor = c.action_input == HNS_ACTION_MOTOR;      // include/hns.h: the caller's controller transform ran in front
    if (motor) {
        if (wide) env->step_args_fn = two ? hns_step_v4_kernel<A, 2, true, kWideK, false, 0, true> : hns_step_v4_kernel<A, 1, true, kWideK, false, 0, true>;
        else env->step_args_fn = two ? hns_step_v4_kernel<A, 2, true, kMaxK, false, 0, true> : hns_step_v4_kernel<A, 1, true, kMaxK, false, 0, true>;
        env->reset_fn = wide ? (two ? hns_reset_kernel<A, 2, kWideK> : hns_reset_kernel<A, 1, kWideK>) : (two ? hns_reset_kernel<A, 2> : hns_reset_kernel<A, 1>);
    } else if (wide) {
        env->step_args_fn = two ? hns_step_v4_kernel<A, 2, true, kWideK, false> : hns_step_v4_kernel<A, 1, true, kWideK, false>;
        env->reset_fn = two ? hns_reset_kernel<A, 2, kWideK> : hns_reset_kernel<A, 1, kWideK>;
    } else {
        if (ragged) env->step_args_fn = two ? hns_step_v4_kernel<A, 2, true, kMaxK, false> : hns_step_v4_kernel<A, 1, true, kMaxK, false>;
        else {
            env->step_args_fn = two ? hns_step_v4_kernel<A, 2, false, kMaxK, false> : hns_step_v4_kernel<A, 1, false, kMaxK, false>;
            env->step_args_prof_fn = two ? hns_step_v4_kernel<A, 2, false, kMaxK, true> : hns_step_v4_kernel<A, 1, false, kMaxK, true>;
            // the reference's / BASELINE's shapes: cylinder count and k = 3 as compile-time constants (the stamped twin stays generic in them)
            if (c.obs_max_cylinder == 3) {
                switch (c.num_cylinders) {
                    case 5: env->step_args_fn = two ? hns_step_v4_kernel<A, 2, false, kMaxK, false, 5> : hns_step_v4_kernel<A, 1, false, kMaxK, false, 5>; break;
                    case 8:
                        // (seven pursuers / two evaders / 8 slots: that one fixed-shape instantiation needed 4 more registers than two 8-wave workgroups
                        //  per CU leave and spilled them; its shape-generic twin does not — it serves the shape)
                        if constexpr (A == 7) env->step_args_fn = two ? env->step_args_fn : hns_step_v4_kernel<A, 1, false, kMaxK, false, 8>;
                        else env->step_args_fn = two ? hns_step_v4_kernel<A, 2, false, kMaxK, false, 8> : hns_step_v4_kernel<A, 1, false, kMaxK, false, 8>;
                        break;
                    case 16: env->step_args_fn = two ? hns_step_v4_kernel<A, 2, false, kMaxK, false, 16> : hns_step_v4_kernel<A, 1, false, kMaxK, false, 16>; break;
                    default: break;
                }
            }
        }
        env->reset_fn = two ? hns_reset_kernel<A, 2> : hns_reset_kernel<A, 1>;
    }
    env->threads = Geo<A>::T;
    env->threads_step = Geo<A>::T;
    env->cyl_magic = (uint32_t)(0xFFFFFFFFull / (uint32_t)(3 * c.num_cylinders) + 1ull);
    env->grid = (c.num_envs + kEPB - 1) / kEPB;
    const int NT = two ? 2 : 1;
    env->lds_step = (size_t)lds_layout_v3(A, c.num_cylinders, c.obs_max_cylinder, NT).total * sizeof(float);
    env->lds_reset = (size_t)lds_layout(A, c.num_cylinders, c.obs_max_cylinder, NT).total * sizeof(float) +
                     (size_t)kEPB * kGridStride;      // + per-env occupancy grid / free-cell list
    // Batches that leave most SIMDs idle take the second mapping (hns_step_small_kernel.h: a helper wave per pursuer wave): one evader, whole
    // tiles, k <= 4, at most kSmallWgPerCu workgroups per CU.  HNS_STEP_MAPPING=tile|small overrides the choice where the shape allows both
    // (A/B runs, tests/test_hip_parity.py).
    const char *mp = std::getenv("HNS_STEP_MAPPING");
    const bool eligible = !two && !wide && !ragged && !motor;
    // (2 A + 1 waves per workgroup: with four and more pursuers two of them no longer share a CU — 32 768 envs measured 13.4 / 19.5 / 22.2 us
    //  with the tile mapping against 17.3 / 22.9 / 25.9 us for 4 / 6 / 7 pursuers, while one tile per CU is faster in the small mapping for every count)
    bool small = eligible && env->grid <= (A <= 3 ? kSmallWgPerCu : 1) * env->cus;
    if (mp && !std::strcmp(mp, "tile")) small = false;
    if (mp && !std::strcmp(mp, "small")) small = eligible;
    if (small) HNS_CAT(hns_select_small_, HNS_INST_A)(env);
    // Tile mapping, one evader, up to three pursuers, one or two full residency rounds (4 workgroups per CU): the pursuer waves run at priority 1
    // until their integration is done (hns_step_kernel.h).  Measured with alternating blocks in one process (tools/ab_env.py, tools/lab/r04_batch46.sh):
    // 65 536 envs -3.4 %, 131 072 -4.5 %, 5 cylinder slots -1.5 %, 2 pursuers -0.6 %; 49 152 envs +1 %, 262 144 +1.2 %, 4 pursuers +1.5 %, 6v2 +5 % (off
    // there).  HNS_STEP_PRIO=0|1 overrides (A/B runs).
    env->prio_boost = !small && !two && A <= 3 && env->grid > 3 * env->cus && env->grid <= 8 * env->cus;
    if (const char *pb = std::getenv("HNS_STEP_PRIO")) { if (pb[0] == '0' || pb[0] == '1') env->prio_boost = !small && pb[0] == '1'; }
}
#endif
