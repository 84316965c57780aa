// hns_step_kernel.h — the fused HideAndSeek step kernel (one launch per env.step) for gfx950; instantiated per pursuer count by
// hns_inst.hip.  Reference call tree: transforms.py:425-459 -> lee_position_controller.py:476-550 -> hideandseek.py:725-744 ->
// multirotor.py:466-508 -> rotor_group.py:55-71 -> [PhysX sim.step() replaced by d_integrate] -> hideandseek.py:746-917 -> :919-1065.
// No MFMA: there is no dense contraction on this path.
#pragma once
#include "hns_common.h"

namespace hns {

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
constexpr int kPub = 11;  // published per pursuer: position at t (3), thrust vector (3), position at t+1 (3), 1 / (|thrust| + 1e-6); odd stride
struct LdsV3 { int slab, slab_stride, pub, cyl, cyl_stride, tp, red, envout, term, term_stride, total; };
// per env: (tx, ty) of every cylinder's push on the second evader; odd stride.  Sized by the env's cylinder count since round 6: with the maximum's 33 floats
// a 3v2 / 8-cylinder workgroup took 53.5 KB of LDS — 0.5 KB too much for a third workgroup per CU (49.4 KB then; 39.5 KB = a FOURTH with v4_small_slabs, below)
__host__ __device__ constexpr int term_stride(int C) { return 2 * C + 1; }
// Rows per staging pass of the tile mapping's slabs (hns_common.h: slab_rows).  Three pursuers with two evaders stage half a wave at a time as the wide
// workgroups do: with whole-wave slabs of 24-value rows a workgroup took 49.4 KB of LDS and 131 registers = THREE workgroups per CU, i.e. 768 of a
// 65 536-env launch's 1 024 workgroups in a first round and 256 in a second (49 152 envs: 15.85 us, 65 536 envs: 24.4 us, tools/small_batch.py --targets=2, round 6).
// Half-wave slabs, the env wave's staging inside the (by then dead) cylinder-term region and the 128-register cap make it four per CU: one round.
__host__ __device__ constexpr bool v4_small_slabs(int A, int NT) { return NT == 2 && A == 3; }
__host__ __device__ constexpr int slab_rows_v4(int A, int NT) { return (A > 4 || v4_small_slabs(A, NT)) ? 32 : 64; }
__host__ __device__ inline int slab_floats_v4(int A, int K, int NT) {
    const int rows = slab_rows_v4(A, NT);
    int m = rows * (NT == 2 ? 24 : HNS_SELF_DIM);
    if (K > kMaxK) K = 0;                       // wide selections are stored by their threads, not staged
    if (rows * K * 5 > m) m = rows * K * 5;
    if (rows * (A - 1) * 3 > m) m = rows * (A - 1) * 3;
    return r4(m);
}
__host__ __device__ inline LdsV3 lds_layout_v3(int A, int C, int K, int NT = 1) {
    LdsV3 L;
    int o = 0;
    L.slab_stride = slab_floats_v4(A, K, NT);
    if (L.slab_stride < 64 * 13 + 4) L.slab_stride = r4(64 * 13 + 4);
    L.slab = o;  o += A * L.slab_stride;
    L.pub = o;   o += r4(kEPB * A * kPub);
    L.cyl_stride = (3 * C) | 1;
    L.cyl = o;   o += r4(kEPB * L.cyl_stride);
    L.tp = o;    o += r4(kEPB * (3 * NT + 1));               // evader(s) at t+1 ([64][3 NT], contiguous: stored as one slice) + the step counter
    L.red = o;   o += r4(kEPB * A * red_stride(NT));
    const int envout = r4(kEPB * (A > 3 * NT ? A : 3 * NT));  // the env wave's own staging: evader velocity [64,3 NT], rewards [64,A]
    L.term_stride = term_stride(C);
    const int term = NT == 2 ? r4(kEPB * L.term_stride) : 0;  // two evaders: the pursuer lanes' share of the evader policy (below)
    if (v4_small_slabs(A, NT) && term >= envout) {
        // the env wave is the term region's only reader (right behind barrier 1, before it stages anything) and the staging's only user: one region
        L.term = o; L.envout = o; o += term;
    } else {
        L.envout = o; o += envout;
        L.term = o;  o += term;
    }
    L.total = o;
    return L;
}

// env wave: `n` floats that sit contiguously in LDS -> one contiguous, 16-byte aligned slice of global memory, 16 B per lane
// (n % 4 == 0), write-through.  Per-lane 4-byte stores at a 12-byte stride would each be a partial-line write.
// (generic instantiation: only the first `nvalid` floats exist — 4-byte stores.)
template <bool GEN>
HNS_DEV void env_store_slice(const float *lds, float *g, int n, int lane, int nvalid) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (GEN) {
        for (int i = lane; i < nvalid; i += 64) st_f1(g + i, lds[i]);
    } else {
        for (int i = lane; i < n / 4; i += 64) st_f4(reinterpret_cast<float4 *>(g) + i, reinterpret_cast<const float4 *>(lds)[i]);
    }
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
//
// Instantiations (hns_inst.hip): <A, NT, GEN, KM, PROF, CS>.  GEN = false is the kernel described above: whole 64-env tiles, k <= 4, nothing
// predicated.  GEN = true serves every other shape with the SAME phases, barriers, arithmetic and evaluation order (the buffers are
// bit-identical where both apply; tests/test_hip_parity.py): a ragged last tile (E % 64 != 0 — lanes beyond the batch load the last
// env's data again, compute, and store nothing; slices end at the batch through their buffer descriptors) and, with KM = kWideK,
// selections of up to 16 nearest cylinders (rows stored by their threads, not staged).  It reads the parameter block before its
// first loads and loads rigid-state rows per thread: correct, not tuned — no configuration of the reference's cfg/ needs it.
// CS > 0: the cylinder count as a compile-time constant, with k = 3 — the shapes of the reference's task files and of BASELINE's configurations
// (5, 8, 16 slots; obs_max_cylinder 3).  Same arithmetic, same order; the cylinder loops unroll completely and the k-nearest predicates
// fold away: 6v2 / 16 cylinders 50.4 -> 48.5 us, 3v1 / 8 cylinders 18.7 -> 18.5 us (A/B/A/B on one box, tools/lab/r04_batch18.sh).
// MOTOR (cfg.action_input = HNS_ACTION_MOTOR, include/hns.h): `action` holds the rotor commands of the caller's own controller transform
// (transforms.py:455-456) — the step starts at _pre_sim_step (hideandseek.py:725-744): no tanh / CTBR / PID; the action error is read from the
// bound buffer (:731), `prev_action`, `pid_integ`, `ctbr`, `target_rate` are left alone.  Served by the generic instantiation only (a
// compatibility path: the torch controller in front costs twenty times this kernel).
// Waves per SIMD the two-evader instantiations are compiled for (the second __launch_bounds__ argument; 512 / it = the VGPR budget).  Six pursuers —
// BASELINE configuration 5 — fit two 7-wave workgroups per CU at 4 (109 VGPRs).  With one to three pursuers every pursuer lane carries MORE of the second
// evader's share (3 * 16 / A staged cylinder values, 16 / A own cylinders: 195 / 134 / 131 VGPRs unconstrained), and at 128 the compiler spilled
// (31-35, 24, 3 registers to scratch: VERDICT r5 weak #8); their workgroups are 2-4 waves, so 2 / 3 / 3 waves per SIMD still place 4 / 4 / 3 of them on a CU.
__host__ __device__ constexpr int step_waves_per_simd(int A, int NT, int KM, bool MOTOR, int CS) {
    // (three pursuers: four with the cylinder count as a compile-time constant — 102-118 registers; the runtime-count instantiation needs 131 and stays at three)
    return (NT == 2 && KM == kMaxK && !MOTOR) ? (A == 1 ? 2 : A == 2 ? 3 : (A == 3 && CS == 0) ? 3 : 4) : 1;
}
template <int A, int NT, bool GEN, int KM, bool PROF, int CS = 0, bool MOTOR = false>
__global__ __launch_bounds__(Geo<A>::T, step_waves_per_simd(A, NT, KM, MOTOR, CS)) void hns_step_v4_kernel(HNS_STEP_PARAMS) {
    HNS_STEP_ARGS_PACK;
    static_assert(GEN || KM == kMaxK, "wide k-nearest selections: the generic instantiation");
    static_assert(!MOTOR || GEN, "motor-command input: the generic instantiation");
    static_assert(CS == 0 || (!GEN && !PROF && CS <= HNS_MAX_CYLINDERS), "fixed shapes: the tuned instantiation only");
    // the block behind `rest` is never written while the kernel runs: read it as constant memory (scalar loads, placed like kernel-argument loads)
    typedef const Params __attribute__((address_space(4))) ParamsC;
    ParamsC &p = *(ParamsC *)ka.rest;
    constexpr int NA = Geo<A>::NA, SD = NT == 2 ? 24 : HNS_SELF_DIM, kRedS = red_stride(NT), T3 = 3 * NT;
    extern __shared__ __align__(16) float smem[];
    const auto &c = p.cfg;
    const auto &b = p.buf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int e0 = blockIdx.x * kEPB;
    // generic instantiation: envs of this tile that exist, and the env a lane's LOADS refer to (the last one for lanes beyond the batch)
    int nv = kEPB, Etot = 0;
    if constexpr (GEN) { Etot = c.num_envs; nv = Etot - e0 < kEPB ? Etot - e0 : kEPB; }
    // NOTHING that reads the block (`p`, `c`, `b`) may precede a wave's first global loads: those scalar loads are cold, and a wait for
    // them in front of the vector loads is what a device-resident configuration used to cost (+1.8 us).
    if (tid < NA) {
        // ================================= pursuer waves ==================================================
        const int le = tid / A, a = tid - le * A;
        const unsigned ia = (unsigned)e0 * A + tid;
        const bool valid = !GEN || le < nv;               // (generic) this pursuer exists
        const int vrows = GEN ? nv * A - (tid & ~63) : 64;   // rows of this wave's output slices that exist (wave_store_rows clamps)
        const unsigned il = GEN ? (unsigned)(e0 + (le < nv ? le : nv - 1)) * A + a : ia;   // the record the loads read
        // loads, first needed first: action, previous action, the wave's 64 rigid-state rows, [reset_pid,] PID state, throttle
        const float4 act4 = reinterpret_cast<const float4 *>(ka.action)[il];
        float4 prev4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (!MOTOR) prev4 = reinterpret_cast<const float4 *>(ka.prev_action)[il];
        constexpr int N4 = 64 * 13 / 4;                   // 208 float4 pieces per wave
        const float4 *rows4 = reinterpret_cast<const float4 *>(ka.drone_state + ((size_t)e0 * A + (tid & ~63)) * 13) + lane;
        static_assert(N4 > 192 && N4 <= 256, "three full passes and a partial one");
        float4 rr0 = make_float4(0.f, 0.f, 0.f, 0.f), rr1 = rr0, rr2 = rr0, rr3 = rr0;   // (named values: an array with a predicated element went to scratch)
        if constexpr (!GEN) {
            rr0 = rows4[0]; rr1 = rows4[64]; rr2 = rows4[128];
            if (lane < N4 - 192) rr3 = rows4[192];
        }
        // (two evaders: the integrator's quad is loaded LAST of the first loads, below — right behind this load the compiler recycled the quad's unused fourth
        //  register for an address computation and had to wait, s_waitcnt vmcnt(0), for all the loads issued so far before issuing the remaining fifteen)
        float4 integ4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (NT == 1 && !MOTOR) integ4 = reinterpret_cast<const float4 *>(ka.pid_integ)[il];
        // (motor-command input: of the controller record only the line-of-sight column is the env's — one dword in, one dword out)
        float4 last4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MOTOR) last4.w = ka.pid_last_rate[(size_t)il * 4 + 3];
        else last4 = reinterpret_cast<const float4 *>(ka.pid_last_rate)[il];
        // reset_pid = the incoming root `done` (transforms.py:449-454): one byte per env; with one evader its pointer rides in the argument block.
        // Loaded WITHOUT a branch (a null pointer reads a byte of `action` instead and the result is ignored): behind `if (pointer)` the compiler
        // closed the branch with s_waitcnt vmcnt(0) — every pursuer wave waited for its first loads to land before it issued the remaining ones.
        unsigned rp = 0;
        if constexpr (NT == 1) {
            const uint8_t *rpp = static_cast<const uint8_t *>(ka.aux);
            const uint8_t *rsafe = rpp ? rpp : reinterpret_cast<const uint8_t *>(ka.action);
            const unsigned byte = rsafe[GEN ? (e0 + (le < nv ? le : nv - 1)) : (e0 + le)];
            rp = rpp ? byte : 0u;
        }
        float4 thr4 = reinterpret_cast<const float4 *>(ka.throttle)[il];
        // (two evaders, see below: this wave's cylinder passes and this pursuer's cylinders — through the kernel argument, issued with the
        //  first loads; the count of cylinders is only known from the parameter block, so the loads cover HNS_MAX_CYLINDERS slots of the
        //  workgroup's OWN range and are clamped to it)
        constexpr int kStage = NT == 2 ? (3 * HNS_MAX_CYLINDERS + A - 1) / A : 1, kOwnCyl = NT == 2 ? (HNS_MAX_CYLINDERS + A - 1) / A : 1;
        float stage_v[kStage], own_c[kOwnCyl][3];
        if constexpr (NT == 2) {
            const uintptr_t cw = reinterpret_cast<uintptr_t>(ka.aux);
            const int Cq = CS ? CS : (int)(cw & 15) + 1;
            // (a pointer rebuilt from an integer is a FLAT pointer to the compiler: its loads count on the LDS counter as well, and the wave's first LDS
            //  round trip — its rigid-state rows through the slab — then waits for every one of them.  Named as global memory, they are global loads.)
            typedef const float __attribute__((address_space(1))) gcf;
            const gcf *cyl0 = reinterpret_cast<const gcf *>(cw & ~(uintptr_t)15);
            const gcf *gc = cyl0 + (size_t)e0 * Cq * 3 + lane;
#pragma unroll
            for (int i = 0; i < kStage; ++i) {
                const int pass = (tid >> 6) + i * A;
                stage_v[i] = (pass < 3 * Cq && (!GEN || pass * 64 + lane < nv * 3 * Cq)) ? gc[pass * 64] : 0.0f;
            }
            const gcf *gcy = cyl0 + (size_t)(e0 + (GEN && le >= nv ? nv - 1 : le)) * Cq * 3;
#pragma unroll
            for (int i = 0; i < kOwnCyl; ++i) {
                const int k = a + i * A;
                const int kc = k < Cq ? k : 0;
                own_c[i][0] = gcy[3 * kc]; own_c[i][1] = gcy[3 * kc + 1]; own_c[i][2] = gcy[3 * kc + 2];
            }
            if constexpr (!MOTOR) integ4 = reinterpret_cast<const float4 *>(ka.pid_integ)[il];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = CS ? CS : c.num_cylinders, K = CS ? 3 : c.obs_max_cylinder;
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
        V3 etp0 = {0.f, 0.f, 0.f}, etp1 = {0.f, 0.f, 0.f};
        if constexpr (NT == 2) {
            {   // (two evaders: the pointer comes through the parameter block; branch-free as above)
                const uint8_t *rsafe = b.reset_pid ? b.reset_pid : reinterpret_cast<const uint8_t *>(ka.action);
                const unsigned byte = rsafe[e0 + (GEN && le >= nv ? nv - 1 : le)];
                rp = b.reset_pid ? byte : 0u;
            }
            const float *gt = b.target_pos + (size_t)(e0 + (GEN && le >= nv ? nv - 1 : le)) * T3;
            etp0 = V3{gt[0], gt[1], gt[2]};
            etp1 = V3{gt[3], gt[4], gt[5]};
        }
        float4 ta = act4;
        if constexpr (!MOTOR) ta = d_action_tanh(act4);  // needs the action only: evaluated while the rest is in flight
        Rigid s;
        if constexpr (GEN) {
            load_rigid(ka.drone_state + (size_t)il * 13, s);     // (generic) the thread's own row, 13 scalar loads
        } else {
            // own rows through the private slab
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
        // Issue arbitration among the waves of a SIMD is priority first, then age: left alone, the first two workgroups placed on a CU finish 3 us
        // before the last two (tools/phase_profile.py --spread: 10.2 / 10.8 / 13.2 / 13.5 us by placement order, all four started within 2 us), and the
        // launch drains at half occupancy.  With the pursuer waves at priority 1 until their integration is done and at 0 behind it, a workgroup
        // that lags gets the issue slots of one that leads: 65 536 envs 17.17 -> 16.58 us (alternating blocks in one process, tools/lab/r04_batch46.sh).
        // Shape dependent — slower with 4+ pursuers, two evaders (+5 %), less than one or more than two residency rounds (+1 %) — so the host
        // turns it on per env (hns_inst.hip); priorities by phase in finer steps (3-2-1-0, 2-2-1-0, 3-1-1-0) or by placement order measured no better.
        const bool boost = p.prio_boost != 0;
        if (boost) __builtin_amdgcn_s_setprio(1);
        // ---- phase 1: controller, rotors, thrust vector (A1-A3) ----
        // line of sight evader -> this pursuer at t (:1080): positions, evader and cylinders are those the previous step (or the reset)
        // evaluated it on for the observation, so that result is carried in the spare fourth column of the controller record
        const float los_t = last4.w;
        float cmd[4], thr_diff, aerr, thrust[4], moment[4];
        float ctbr4[4], trate[3];
        if constexpr (MOTOR) {
            // the caller's transform ran A1 / A2 (and nan_to_num, transforms.py:455): its commands feed the rotors as they are (hideandseek.py:735),
            // its action error is the statistic's input (:731)
            cmd[0] = ta.x; cmd[1] = ta.y; cmd[2] = ta.z; cmd[3] = ta.w;
            aerr = b.action_error[il];
        } else {
        {   // reset_pid (lee_position_controller.py:497-502): integrator and last body rate start from zero (selects, no branch)
            const bool r = rp != 0;
            integ4.x = r ? 0.0f : integ4.x; integ4.y = r ? 0.0f : integ4.y; integ4.z = r ? 0.0f : integ4.z;
            last4.x = r ? 0.0f : last4.x; last4.y = r ? 0.0f : last4.y; last4.z = r ? 0.0f : last4.z;
        }
        d_ctbr_pid_squashed(c, ta, s.q, s.ang, prev4, integ4, last4, cmd, aerr, ctbr4, trate);
        // (the wait for the last of the first loads HERE, ahead of the two stores below: behind them it would wait for their acknowledgement too)
        asm volatile("" : "+v"(thr4.x), "+v"(thr4.y), "+v"(thr4.z), "+v"(thr4.w));
        if (b.ctbr && valid) reinterpret_cast<float4 *>(b.ctbr)[ia] = make_float4(ctbr4[0], ctbr4[1], ctbr4[2], ctbr4[3]);           // transforms.py:456
        if (b.target_rate && valid) reinterpret_cast<float4 *>(b.target_rate)[ia] = make_float4(trate[0], trate[1], trate[2], 0.0f);  // :457
        }
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
                float *term = smem + L.term + le * L.term_stride;
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
        flag_nonfinite(b.nonfinite, valid && rigid_not_finite(s), 1u);
        {
            float *pub = sPub + tid * kPub;
            pub[6] = s.pos.x; pub[7] = s.pos.y; pub[8] = s.pos.z;
        }
        // controller / rotor state: write-through as well.  (Round 2 measured +0.45 us for that in the kernel of the time and left them to the end-of-kernel
        // write-back; on the round-4 kernel, alternating blocks in one process: 65 536 envs 17.20 -> 16.46 us, 131 072 envs -4.5 %, the 6v2 shard
        // 48.3 -> 46.7 us, 262 144 and 1 048 576 envs unchanged.  The early `ctbr` / `target_rate` and the `pid_last_rate` stores stay plain: write-through
        // there measured +1 % at 65 536 envs.  tools/lab/r04_batch53.sh, _54, _56.)
        if (valid) {
            st_f4(reinterpret_cast<float4 *>(b.throttle) + ia, thr4);
            if constexpr (!MOTOR) {
                st_f4(reinterpret_cast<float4 *>(b.pid_integ) + ia, integ4);
                st_f4(reinterpret_cast<float4 *>(b.prev_action) + ia, prev4);
                st_f1(b.action_error + ia, aerr);
            }
        }
        {   // S_{t+1}: the wave's 64 rows back through the slab, one contiguous slice
            const float row[13] = {s.pos.x, s.pos.y, s.pos.z, s.q.w, s.q.x, s.q.y, s.q.z, s.lin.x, s.lin.y, s.lin.z, s.ang.x, s.ang.y, s.ang.z};
            wave_store_rows<13>(slab, b.drone_state + ((size_t)e0 * A + (tid & ~63)) * 13, row, lane, vrows);
        }
        if constexpr (PROF) prof_mark(p.prof, 3);
        __syncthreads();                                                            // barrier 2
        if constexpr (PROF) prof_mark(p.prof, 8);
        if (boost) __builtin_amdgcn_s_setprio(0);
        // ---- phase 3a: distance and line of sight to the evader, the k nearest cylinders, per-pursuer reward terms on S_{t+1} ----
        const float progress = sTp[kEPB * T3 + le];                                 // progress + 1, published by the env wave
        const V3 tp = {sTp[le * T3], sTp[le * T3 + 1], sTp[le * T3 + 2]};
        V3 tpB = tp;                                                                // second evader (extension, include/hns.h)
        if constexpr (NT == 2) tpB = V3{sTp[le * T3 + 3], sTp[le * T3 + 4], sTp[le * T3 + 5]};
        const float *cyl = sCyl + le * L.cyl_stride;
        const float rtx = s.pos.x - tp.x, rty = s.pos.y - tp.y, rtz = s.pos.z - tp.z;
        const float d = d_norm3(rtx, rty, rtz);                                     // |evader - pursuer| (hideandseek.py:921, :780)
        int knn_idx[KM + 1];
        bool knn_masked[KM];
        bool blocked, blockedB;
        cylinder_pass<NT, true, KM, (CS ? 4 : KM + 1)>(c, C, K, s.pos, tp, tpB, cyl, knn_idx, blocked, blockedB);
        const bool det = (d < c.drone_detect_radius) && !blocked;                   // :787-789
        last4.w = (float)((blocked ? 1 : 0) + (NT == 2 && blockedB ? 2 : 0));       // = the next step's line of sight at ITS t
        if constexpr (MOTOR) { if (valid) b.pid_last_rate[(size_t)ia * 4 + 3] = last4.w; }
        else if (valid) reinterpret_cast<float4 *>(b.pid_last_rate)[ia] = last4;
#pragma unroll
        for (int sidx = 0; sidx < KM; ++sidx) knn_masked[sidx] = (sidx < K) ? cyl[3 * knn_idx[sidx] + 2] < 0.0f : false;   // :759,775-778
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
        for (int sidx = 0; sidx < KM; ++sidx) {
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
            wave_store_rows<SD, slab_rows_v4(A, NT)>(slab, b.obs_self + ((size_t)e0 * A + (tid & ~63)) * SD, row, lane, vrows);
            if (with_state) {                                                      // :871-886 (never masked)
                float rs[SD];
#pragma unroll
                for (int i = 0; i < SD; ++i) rs[i] = row[i];
                rs[0] = rtx; rs[1] = rty; rs[2] = rtz;
                if constexpr (NT == 2) { rs[20] = r1x; rs[21] = r1y; rs[22] = r1z; }
                wave_store_rows<SD, slab_rows_v4(A, NT)>(slab, b.state_drones + ((size_t)e0 * A + (tid & ~63)) * SD, rs, lane, vrows);
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
            wave_store_rows<(A > 1 ? A - 1 : 1) * 3, slab_rows_v4(A, NT)>(slab, b.obs_others + ((size_t)e0 * A + (tid & ~63)) * (A - 1) * 3, o, lane, vrows);
        }
        if constexpr (KM > kMaxK) {                                                 // wide selections: each thread stores its own rows (:767-778)
            const float mv = c.mask_value, ch = c.cylinder_height, cs = c.cylinder_size;
            float *oc = b.obs_cylinders + (size_t)ia * K * 5;
            if (valid) {
#pragma unroll
                for (int sidx = 0; sidx < KM; ++sidx) {
                    if (sidx < K) {
                        const float *cc = cyl + 3 * knn_idx[sidx];
                        const bool masked = knn_masked[sidx];
                        oc[sidx * 5] = masked ? mv : s.pos.x - cc[0];
                        oc[sidx * 5 + 1] = masked ? mv : s.pos.y - cc[1];
                        oc[sidx * 5 + 2] = masked ? mv : s.pos.z - cc[2];
                        oc[sidx * 5 + 3] = masked ? mv : ch;
                        oc[sidx * 5 + 4] = masked ? mv : cs;
                    }
                }
            }
        } else {                                                                    // the k nearest cylinders (:767-778)
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
                wave_store_rows<15, slab_rows_v4(A, NT)>(slab, g, r, lane, vrows);
            } else if (K == 4) {
                wave_store_rows<20, slab_rows_v4(A, NT)>(slab, g, krow, lane, vrows);
            } else if (K == 2) {
                float r[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) r[i] = krow[i];
                wave_store_rows<10, slab_rows_v4(A, NT)>(slab, g, r, lane, vrows);
            } else {
                float r[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) r[i] = krow[i];
                wave_store_rows<5, slab_rows_v4(A, NT)>(slab, g, r, lane, vrows);
            }
        }
        if constexpr (PROF) prof_mark(p.prof, 6);
    } else {
        // ================================= env wave: lane <-> env ========================================
        __builtin_amdgcn_s_setprio(2);   // one wave in four, but every barrier of its workgroup waits for it (the pursuer waves run at 0 or 1, below)
        const int le = lane;
        const bool valid = !GEN || le < nv;                  // (generic) this env exists; lanes beyond the batch work on the last env's data
        const int e = e0 + (valid ? le : nv - 1);
#ifndef HNS_NO_WARM
        warm_params(ka.rest);
#endif
        if constexpr (PROF) prof_mark(p.prof, 0);
        if constexpr (PROF) prof_mark(p.prof, 14);
        const int C = CS ? CS : c.num_cylinders, K = CS ? 3 : c.obs_max_cylinder, E = c.stats_stride;      // (E: the row stride of `stats`, = num_envs unless the env is a slice)
        const LdsV3 L = lds_layout_v3(A, C, K, NT);
        float *sPub = smem + L.pub, *sCyl = smem + L.cyl, *sTp = smem + L.tp, *sRed = smem + L.red, *sEnvOut = smem + L.envout;
        float *cylw = sCyl + le * L.cyl_stride;
        // the evader at t
        const float *gt = b.target_pos + (size_t)e * T3;
        const V3 tp0 = {gt[0], gt[1], gt[2]};
        V3 tp1 = tp0;
        if constexpr (NT == 2) tp1 = V3{gt[3], gt[4], gt[5]};
        float progress = b.progress[e];
        if constexpr (NT == 1 && GEN) {   // (generic) lane = env copies its own row
            const float *gr = b.cylinders + (size_t)e * C * 3;
            for (int j = 0; j < 3 * C; ++j) cylw[j] = gr[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (NT == 1) {   // this workgroup's cylinders are one contiguous slice [64][3C]: coalesced 4-byte loads (lane <-> consecutive floats), scattered
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
#ifndef HNS_NO_PIN
        // Most of the statistics rows fetched above are first used behind barrier 3, behind this wave's stores of the evader and the rewards.  Left
        // alone, the compiler waits for them THERE with s_waitcnt vmcnt(n), n = the memory operations issued since — so the wave waits for its own
        // freshly issued stores to be acknowledged (hns_step_small_kernel.h: 2 500 cycles for 11 stores).  Pin the wait here: only loads are outstanding.
#pragma unroll
        for (int i = 0; i < HNS_NUM_STATS; ++i) asm volatile("" : "+v"(st[i]));
#endif
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
            const float *term = smem + L.term + le * L.term_stride;
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
        { const float sf = (tpn.x + tpn.y) + tpn.z; flag_nonfinite(b.nonfinite, valid && (sf - sf) != 0.0f, 2u); }
        V3 tvel1 = {0.f, 0.f, 0.f};
        if constexpr (NT == 2) {
            G.x = G.x + Fenv1.x; G.y = G.y + Fenv1.y; G.z = G.z + Fenv1.z;
            G.x = G.x + gcx; G.y = G.y + gcy; G.z = G.z + 0.0f;
            tvel1 = V3{(c.v_prey * G.x) / (__builtin_fabsf(G.x) + 1e-5f), (c.v_prey * G.y) / (__builtin_fabsf(G.y) + 1e-5f),
                       (c.v_prey * G.z) / (__builtin_fabsf(G.z) + 1e-5f)};
            const V3 tpn1 = {tp1.x + tvel1.x * c.dt, tp1.y + tvel1.y * c.dt, tp1.z + tvel1.z * c.dt};
            sTp[le * T3 + 3] = tpn1.x; sTp[le * T3 + 4] = tpn1.y; sTp[le * T3 + 5] = tpn1.z;
            { const float sf = (tpn1.x + tpn1.y) + tpn1.z; flag_nonfinite(b.nonfinite, valid && (sf - sf) != 0.0f, 2u); }
        }
        {   // [64,3 NT] slices, whole lines: the new position is already laid out in sTp
            sEnvOut[le * T3] = tvel.x; sEnvOut[le * T3 + 1] = tvel.y; sEnvOut[le * T3 + 2] = tvel.z;
            if constexpr (NT == 2) { sEnvOut[le * T3 + 3] = tvel1.x; sEnvOut[le * T3 + 4] = tvel1.y; sEnvOut[le * T3 + 5] = tvel1.z; }
            env_store_slice<GEN>(sTp, b.target_pos + (size_t)e0 * T3, kEPB * T3, lane, nv * T3);
            env_store_slice<GEN>(sEnvOut, b.target_vel + (size_t)e0 * T3, kEPB * T3, lane, nv * T3);
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
        env_store_slice<GEN>(sEnvOut, b.reward + (size_t)e0 * A, kEPB * A, lane, nv * A);
        flag_nonfinite(b.nonfinite, valid && (sum_rew - sum_rew) != 0.0f, 4u);
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
        if (valid) {
            b.done[e] = (uint8_t)done;
            if (b.detect) b.detect[e] = (uint8_t)((det_any ? 1 : 0) | (NT == 2 && det_any1 ? 2 : 0));      // bit k: evader k detected
            b.progress[e] = progress;
#pragma unroll
            for (int i = 0; i < HNS_NUM_STATS; ++i) st_f1(b.stats + (size_t)i * E + e, st[i]);
        }
        if constexpr (PROF) prof_mark(p.prof, 6);
    }
    if constexpr (PROF) prof_mark(p.prof, 7);
    if constexpr (PROF) prof_mark(p.prof, 15);
}

}  // namespace hns
