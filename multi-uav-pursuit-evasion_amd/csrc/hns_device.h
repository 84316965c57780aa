// hns_device.h — device-side scalar math of the HideAndSeek step (gfx950, wave64).
//
// Every routine states the reference lines it implements (paths relative to the reference
// repo thu-uav/Multi-UAV-pursuit-evasion).  Arithmetic is IEEE fp32 with an explicit evaluation
// order (compiled with -ffp-contract=off, correctly rounded divide/sqrt), and exp/tanh/sincos are
// the fixed polynomial forms of DESIGN.md §Numerics, so results are reproducible bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hns.h"

#define HNS_DEV static __device__ __forceinline__
// (the functions that read the configuration take it as `const Cfg &`: hns_cfg in generic memory, or the same struct behind a
//  constant-address-space reference — the step kernel reads its device-resident copy through the scalar cache that way)

namespace hns {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInf = __builtin_huge_valf();

// ---- elementary functions; explicit FMA, identical in the oracle ------------
#define HNS_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
constexpr float kInvPi = 0.31830987334251404f;   // RN(1/fp32(pi)): CUDA `tensor / python_scalar` multiplies by this

HNS_DEV float d_expf(float x) {
    // branch-free: evaluate on a clamped argument, pick the special cases at the end (same values
    // as the guarded form in the oracle: 0 for x <= -87 and -inf, +inf above 88, NaN for NaN)
    // (min / max, not compare-and-select: the compiler threads a jump around the polynomial on a compare against a constant,
    //  and a branch costs a wave more than the polynomial; NaN comes out as a clamp value here and is restored by the last select)
    float xc = __builtin_fmaxf(__builtin_fminf(x, 88.5f), -87.5f);
    float k = __builtin_rintf(xc * 1.44269504088896341f);
    float r = HNS_FMA(k, -0.693359375f, xc);
    r = HNS_FMA(k, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = HNS_FMA(p, r, 1.3981999507E-3f);
    p = HNS_FMA(p, r, 8.3334519073E-3f);
    p = HNS_FMA(p, r, 4.1665795894E-2f);
    p = HNS_FMA(p, r, 1.6666665459E-1f);
    p = HNS_FMA(p, r, 5.0000001201E-1f);
    float y = HNS_FMA(p, r * r, r) + 1.0f;
    int ki = (int)k;
    float v = y * __uint_as_float((uint32_t)(ki + 127) << 23);
    // special cases without control flow: v is finite here (|xc| <= 88.5), so v * 0 = +0 is the underflow result, exactly; a select
    // with v on one side only would let the compiler sink the whole polynomial into a branch
    v = v * ((x > -87.0f) ? 1.0f : 0.0f);
    v = (x > 88.0f) ? kInf : v;
    v = (x != x) ? x : v;
    return v;
}

// tanh(x) = sign(x) * (1 - e)/(1 + e), e = exp(-2|x|): one path for every x (|err| <= 1.2e-7)
HNS_DEV float d_tanhf(float x) {
    float e = d_expf(-2.0f * __builtin_fabsf(x));
    float r = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -r : r;
}

HNS_DEV void d_sincosf(float x, float &s_out, float &c_out) {
    float ax = __builtin_fabsf(x);
    int j = (int)(ax * 1.27323954473516f);
    if (j & 1) j += 1;
    float y = (float)j;
    j &= 7;
    float z = HNS_FMA(y, -3.77489497744594108e-8f, HNS_FMA(y, -2.4187564849853515625e-4f, HNS_FMA(y, -0.78515625f, ax)));
    float zz = z * z;
    float ps = -1.9515295891E-4f;
    ps = HNS_FMA(ps, zz, 8.3321608736E-3f);
    ps = HNS_FMA(ps, zz, -1.6666654611E-1f);
    float sp = HNS_FMA(ps * zz, z, z);
    float pc = 2.443315711809948E-005f;
    pc = HNS_FMA(pc, zz, -1.388731625493765E-003f);
    pc = HNS_FMA(pc, zz, 4.166664568298827E-002f);
    float cp = HNS_FMA(pc * zz, zz, HNS_FMA(-0.5f, zz, 1.0f));
    float s = (j == 0) ? sp : (j == 2) ? cp : (j == 4) ? -sp : -cp;
    float c = (j == 0) ? cp : (j == 2) ? -sp : (j == 4) ? -cp : sp;
    if (x < 0.0f) s = -s;
    s_out = s;
    c_out = c;
}

HNS_DEV float d_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
HNS_DEV float d_norm3(float x, float y, float z) { return __builtin_sqrtf(HNS_FMA(z, z, HNS_FMA(y, y, x * x))); }
HNS_DEV float d_norm2(float x, float y) { return __builtin_sqrtf(HNS_FMA(y, y, x * x)); }

struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };

// omni_drones/utils/torch.py:183-191 (quat_rotate) / :194-202 (quat_rotate_inverse):
// a = v(2w^2-1), b = 2w (q_vec x v), c = 2 q_vec (q_vec . v); result a +- b + c, fused
template <bool INVERSE>
HNS_DEV V3 d_quat_rot(const Q4 &q, const V3 &v) {
    float w2 = 2.0f * q.w;
    float s = HNS_FMA(w2, q.w, -1.0f);
    float cx = HNS_FMA(q.y, v.z, -(q.z * v.y));
    float cy = HNS_FMA(q.z, v.x, -(q.x * v.z));
    float cz = HNS_FMA(q.x, v.y, -(q.y * v.x));
    float dot2 = 2.0f * HNS_FMA(q.z, v.z, HNS_FMA(q.y, v.y, q.x * v.x));
    float bw = INVERSE ? -w2 : w2;
    V3 o;
    o.x = HNS_FMA(q.x, dot2, HNS_FMA(cx, bw, v.x * s));
    o.y = HNS_FMA(q.y, dot2, HNS_FMA(cy, bw, v.y * s));
    o.z = HNS_FMA(q.z, dot2, HNS_FMA(cz, bw, v.z * s));
    return o;
}

// quat_rotate(q, x_hat) and quat_rotate(q, (0,0,t)): the same formula with the zero products of the
// basis vector dropped: heading/up (multirotor.py:613-614), thrust vector (:491)
HNS_DEV V3 d_quat_rot_x(const Q4 &q) {
    float s = HNS_FMA(2.0f * q.w, q.w, -1.0f);
    V3 o;
    o.x = HNS_FMA(2.0f * q.x, q.x, s);
    o.y = 2.0f * HNS_FMA(q.z, q.w, q.y * q.x);
    o.z = 2.0f * HNS_FMA(q.z, q.x, -(q.y * q.w));
    return o;
}
HNS_DEV V3 d_quat_rot_z(const Q4 &q, float t) {
    float s = HNS_FMA(2.0f * q.w, q.w, -1.0f);
    float dot = q.z * t;
    V3 o;
    o.x = 2.0f * HNS_FMA(q.y * t, q.w, q.x * dot);
    o.y = 2.0f * HNS_FMA(q.y, dot, -((q.x * t) * q.w));
    o.z = HNS_FMA(2.0f * q.z, dot, t * s);
    return o;
}

// omni_drones/utils/torch.py:110-127
HNS_DEV Q4 d_euler_to_quat(float r, float p, float y) {
    float sr, cr, sp, cp, sy, cy;
    d_sincosf(r * 0.5f, sr, cr);
    d_sincosf(p * 0.5f, sp, cp);
    d_sincosf(y * 0.5f, sy, cy);
    Q4 q;
    q.w = (cr * cp) * cy + (sr * sp) * sy;
    q.x = (sr * cp) * cy - (cr * sp) * sy;
    q.y = (cr * sp) * cy + (sr * cp) * sy;
    q.z = (cr * cp) * sy - (sr * sp) * cy;
    return q;
}

// ---- A1 + A2: action -> CTBR -> body-rate PID -> motor commands ------------------------------
// omni_drones/utils/torchrl/transforms.py:425-459,
// omni_drones/controllers/lee_position_controller.py:476-550
// the squashing of the raw policy output (transforms.py:431) on its own: it needs nothing but the action, so the
// step kernel evaluates it while the state is still on its way from HBM
HNS_DEV float4 d_action_tanh(const float4 &action) {
    return make_float4(d_tanhf(action.x), d_tanhf(action.y), d_tanhf(action.z), d_tanhf(action.w));
}
template <class Cfg>
HNS_DEV void d_ctbr_pid_squashed(const Cfg &c, const float4 &ta, const Q4 &q, const V3 &angvel,
                                 float4 &prev_action, float4 &integ4, float4 &last4, float cmd[4], float &action_error,
                                 float *ctbr_out = nullptr, float *target_out = nullptr) {
    float a0 = ta.x, a1 = ta.y, a2 = ta.z, a3 = ta.w;
    float ctbr[4] = {a0, a1, a2, d_clamp((a3 + 1.0f) / 2.0f, 0.0f, c.max_thrust_ratio)};
    if (c.fixed_yaw) ctbr[2] = 0.0f;
    float d0 = ctbr[0] - prev_action.x, d1 = ctbr[1] - prev_action.y, d2 = ctbr[2] - prev_action.z,
          d3 = ctbr[3] - prev_action.w;
    action_error = __builtin_sqrtf(((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
    prev_action = make_float4(ctbr[0], ctbr[1], ctbr[2], ctbr[3]);
    float target[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) target[i] = (ctbr[i] * 180.0f) * c.target_clip;
    float thrust = ctbr[3] * 65536.0f;
    V3 brv = d_quat_rot<true>(q, angvel);
    float br[3] = {brv.x, brv.y, brv.z};
    float integ[3] = {integ4.x, integ4.y, integ4.z};
    float last[3] = {last4.x, last4.y, last4.z};
    float out[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        br[i] = (br[i] * 180.0f) * kInvPi;       // `* 180.0 / torch.pi` (lee_position_controller.py:505), CUDA scalar-division form
        float err = target[i] - br[i];
        float P = err * c.pid_kp[i];
        float deriv = -(br[i] - last[i]) * c.inv_dt;
        if (deriv != deriv) deriv = 0.0f;
        float D = deriv * c.pid_kd[i];
        float in = integ[i] + err * c.dt;
        in = d_clamp(in, -c.pid_ilimit[i], c.pid_ilimit[i]);
        integ[i] = in;
        float I = in * c.pid_ki[i];
        float FF = target[i] * 0.0f;
        float o = ((P + D) + I) + FF;
        if (o != o) o = 0.0f;
        out[i] = d_clamp(o, -c.pid_outlimit, c.pid_outlimit);
        last[i] = br[i];
    }
    integ4 = make_float4(integ[0], integ[1], integ[2], 0.0f);
    last4 = make_float4(last[0], last[1], last[2], 0.0f);
    float r = out[0] / 2.0f, p = out[1] / 2.0f, y = out[2];
    float m[4] = {((thrust + r) - p) + y, ((thrust + r) + p) - y, ((thrust - r) + p) + y, ((thrust - r) - p) - y};
    if (ctbr_out) { ctbr_out[0] = r; ctbr_out[1] = p; ctbr_out[2] = y; ctbr_out[3] = thrust; }
    if (target_out) { target_out[0] = target[0]; target_out[1] = target[1]; target_out[2] = target[2]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = (m[i] / 65536.0f) * 2.0f - c.max_thrust_ratio;
        v = (v != v) ? 0.0f : v;                               // torch.nan_to_num_(cmds, 0.): NaN -> 0, +-inf -> +-FLT_MAX (no control flow)
        cmd[i] = __builtin_fminf(__builtin_fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
    }
}

template <class Cfg>
HNS_DEV void d_ctbr_pid(const Cfg &c, const float4 &action, const Q4 &q, const V3 &angvel,
                        float4 &prev_action, float4 &integ4, float4 &last4, float cmd[4], float &action_error,
                        float *ctbr_out = nullptr, float *target_out = nullptr) {
    d_ctbr_pid_squashed(c, d_action_tanh(action), q, angvel, prev_action, integ4, last4, cmd, action_error, ctbr_out, target_out);
}

// ---- A3: rotor lag + thrust/moment   omni_drones/actuators/rotor_group.py:55-71 --------------
template <class Cfg>
HNS_DEV void d_rotor(const Cfg &c, const float cmd[4], float4 &throttle4, float thrust[4], float moment[4],
                     float &throttle_difference) {
    float thr_in[4] = {throttle4.x, throttle4.y, throttle4.z, throttle4.w};
    float thr_out[4];
    float dd[4];
    // into values first: `cond ? c.a : c.b` on two fields is an lvalue and compiles to ONE load from a per-lane selected address —
    // a vector memory load of a configuration constant
    const float tau_up = c.tau_up, tau_down = c.tau_down;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float tgt = __builtin_sqrtf(d_clamp((cmd[i] + 1.0f) / 2.0f, 0.0f, 1.0f));
        float tau = (tgt > thr_in[i]) ? tau_up : tau_down;
        float thr = thr_in[i] + tau * (tgt - thr_in[i]);
        thr_out[i] = thr;
        float t = d_clamp(thr * thr + 0.0f, 0.0f, 1.0f);
        thrust[i] = t * c.kf[i];
        moment[i] = (t * c.km[i]) * -c.rotor_dir[i];
        dd[i] = thr - thr_in[i];
    }
    throttle4 = make_float4(thr_out[0], thr_out[1], thr_out[2], thr_out[3]);
    throttle_difference = __builtin_sqrtf(((dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]) + dd[3] * dd[3]);  // multirotor.py:507
}

// ---- A4: downwash of drone j on drone i   omni_drones/robots/drone/multirotor.py:488-494,725-753
// 1 / (|thrust vector| + 1e-6): taken ONCE by the drone that owns the thrust vector and published beside it
HNS_DEV float d_downwash_inv_norm(const V3 &t_w) { return 1.0f / (d_norm3(t_w.x, t_w.y, t_w.z) + 1e-6f); }
// (kr r / z)^2 = 4 r^2 / z^2 from the squared radial distance: no square root; z = 0 gives +inf (r > 0) or NaN (r = 0)
// exactly as kr r / z does
HNS_DEV V3 d_downwash_pair(const V3 &pi, const V3 &pj, const V3 &tj_w, float inv_nj) {
    float dx = tj_w.x * inv_nj, dy = tj_w.y * inv_nj, dz = tj_w.z * inv_nj;
    float rx = pj.x - pi.x, ry = pj.y - pi.y, rz = pj.z - pi.z;
    float zd = HNS_FMA(rz, dz, HNS_FMA(ry, dy, rx * dx));
    float ox = HNS_FMA(-zd, dx, rx), oy = HNS_FMA(-zd, dy, ry), oz = HNS_FMA(-zd, dz, rz);
    float r2 = HNS_FMA(oz, oz, HNS_FMA(oy, oy, ox * ox));
    float z = zd < 0.0f ? 0.0f : zd;
    float u2 = (4.0f * r2) / (z * z);
    float den = HNS_FMA(0.3f, z, 1.0f);
    float v = d_expf(-0.5f * u2) / (den * den);
    V3 f = {v * -tj_w.x, v * -tj_w.y, v * -tj_w.z};
    return f;
}
HNS_DEV V3 d_downwash_pair(const V3 &pi, const V3 &pj, const V3 &tj_w) { return d_downwash_pair(pi, pj, tj_w, d_downwash_inv_norm(tj_w)); }

// ---- A7: line of sight drone->target blocked by any cylinder (xy plane) ---------------------
// omni_drones/envs/hide_and_seek/hideandseek.py:47-103.  cyl: this env's [C,3] in LDS.
//
// The reference forms two quotients per (drone, cylinder) and only COMPARES them:
//     dist = num/(den+1e-5) <= size      t = numt/(dent+1e-5), 0 <= t <= 1
// Both denominators are per-drone constants > 0, so the IEEE results of the comparisons can be
// decided without dividing, except in a vanishing band next to the thresholds where the exact
// division is executed ("division-free line of sight"):
//   * RN(numt/d) <= 1  <=>  numt <= d   (a quotient in (1, 1+2^-23) needs numt = d*(1+2^-24) for a tie,
//     which is not representable, so it rounds up; numt <= d gives a quotient <= 1)
//   * RN(numt/d) >= 0  <=>  numt >= 0   (|numt| >= 1e-30 rules out underflow to -0; else exact path)
//   * RN(num/d) <= s: with p = s*d, num < p*(1-2^-19) => true, num > p*(1+2^-19) => false
//     (the margin is > 4x the accumulated error of p: 1 ulp of the approximate sqrt, the +1e-5, the product and the
//     scaling), else exact path.
// The fast tests are branch-free; a lane that meets an undecidable case redoes its whole loop in
// the exact divide-and-compare form afterwards.  The result is bit-identical to that form
// (tests/test_hip_parity.py).
struct LosLine {          // per (drone, evader) constants of the line-of-sight test
    float diffx, diffy, dx, dy, dt1, plo, phi, dpx, dpy, tpx, tpy;
};
// The filter band is built on the hardware's approximate square root (v_sqrt_f32, 1 ulp): its value only positions a
// band that is 4 x wider than its error, every decision inside the band is taken by the exact path (d_blocked_exact,
// correctly rounded sqrt + divisions), so the results do not depend on the approximation.
template <class Cfg>
HNS_DEV LosLine d_los_setup(const Cfg &c, const V3 &dp, const V3 &tp) {
    LosLine l;
    l.diffx = dp.x - tp.x; l.diffy = dp.y - tp.y;
    const float den = __builtin_amdgcn_sqrtf(HNS_FMA(l.diffy, l.diffy, l.diffx * l.diffx));
    l.dx = tp.x - dp.x; l.dy = tp.y - dp.y;
    float dent = HNS_FMA(l.dy, l.dy, l.dx * l.dx);
    l.dt1 = dent + 1e-5f;
    float p = c.cylinder_size * (den + 1e-5f);
    l.plo = p * 0.99999809265137f; l.phi = p * 1.00000190734863f;   // 1 -+ 2^-19
    l.dpx = dp.x; l.dpy = dp.y; l.tpx = tp.x; l.tpy = tp.y;
    return l;
}
// Branch-free fast test.  Returns the decision when it is certain; sets `uncertain` when either
// quotient lies in the band where only the exact division can decide (or an input is NaN).
HNS_DEV bool d_los_cylinder_fast(const LosLine &l, float ccx, float ccy, float ccz, bool &uncertain) {
    float d2x = ccx - l.tpx, d2y = ccy - l.tpy;
    float num = __builtin_fabsf(HNS_FMA(l.diffx, d2y, -(l.diffy * d2x)));
    float numt = HNS_FMA(ccy - l.dpy, l.dy, (ccx - l.dpx) * l.dx);
    bool lo = num < l.plo, hi = num > l.phi;
    bool tpos = numt >= 0.0f, tneg = numt < -1e-30f;
    uncertain = uncertain || !(lo || hi) || !(tpos || tneg);
    return lo && tpos && (numt <= l.dt1) && (ccz > 0.0f);
}
// The same test given the pursuer-relative offsets ex = dp.x - ccx, ey = dp.y - ccy, which the k-nearest key of the same cylinder computes anyway
// (cylinder_pass): ccx - dp.x = -ex and (ccx - dp.x) dx = -(ex dx) exactly, so numt = fma(-ey, dy, -(ex dx)) is the SAME value bit for bit — two
// subtractions fewer per cylinder and evader (negations ride on the operands).
HNS_DEV bool d_los_cylinder_fast_rel(const LosLine &l, float ccx, float ccy, float ccz, float ex, float ey, bool &uncertain) {
    float d2x = ccx - l.tpx, d2y = ccy - l.tpy;
    float num = __builtin_fabsf(HNS_FMA(l.diffx, d2y, -(l.diffy * d2x)));
    float numt = HNS_FMA(-ey, l.dy, -(ex * l.dx));
    bool lo = num < l.plo, hi = num > l.phi;
    // (the projection's sign is uncertain only in a sliver around zero — ONE compare on |numt|; the exact path decides there.  A NaN needs no flag: neither
    //  path calls a NaN test blocked.  Round 6: one compare and two scalar operations fewer per cylinder and evader than `!(numt >= 0 || numt < -1e-30)`.)
    bool tpos = numt >= 0.0f;
    uncertain = uncertain || !(lo || hi) || (__builtin_fabsf(numt) < 1e-30f);
    return lo && tpos && (numt <= l.dt1) && (ccz > 0.0f);
}
// The same decision with the UNCERTAINTY bookkeeping taken out of the loop (round 6): instead of two more compares and three scalar operations per cylinder and
// evader, two running minima — the distance of `num` from the band's centre and |numt| — are kept with three vector instructions and tested ONCE behind the
// loop (d_los_uncertain).  The band tested there is twice as wide as [plo, phi] (num - p is exact that close to p), so every case the per-cylinder flags caught
// is caught; a wider band only sends a few more lanes to the exact path, whose result is the same by construction.
HNS_DEV bool d_los_cylinder_fast_acc(const LosLine &l, float ccx, float ccy, float ccz, float ex, float ey, float &min_band, float &min_numt) {
    float d2x = ccx - l.tpx, d2y = ccy - l.tpy;
    float num = __builtin_fabsf(HNS_FMA(l.diffx, d2y, -(l.diffy * d2x)));
    float numt = HNS_FMA(-ey, l.dy, -(ex * l.dx));
    min_band = __builtin_fminf(min_band, __builtin_fabsf(num - l.phi));
    min_numt = __builtin_fminf(min_numt, __builtin_fabsf(numt));
    return (num < l.plo) && (numt >= 0.0f) && (numt <= l.dt1) && (ccz > 0.0f);
}
// phi - plo = p 2^-18: |num - phi| <= that covers [plo, phi] (and up to phi + p 2^-18); a NaN minimum (NaN inputs) compares false on `>` below -> uncertain
HNS_DEV bool d_los_uncertain(const LosLine &l, float min_band, float min_numt) {
    return !(min_band > (l.phi - l.plo)) || !(min_numt >= 1e-30f);
}
// Exact form: divide and compare, as the reference does (hideandseek.py:47-103)
template <class Cfg>
HNS_DEV bool d_los_cylinder(const Cfg &c, const LosLine &l, float d1, float ccx, float ccy, float ccz) {
    float d2x = ccx - l.tpx, d2y = ccy - l.tpy;
    float num = __builtin_fabsf(HNS_FMA(l.diffx, d2y, -(l.diffy * d2x)));
    float numt = HNS_FMA(ccy - l.dpy, l.dy, (ccx - l.dpx) * l.dx);
    bool blocked = (num / d1) <= c.cylinder_size;
    float t = numt / l.dt1;
    bool on = (t >= 0.0f) && (t <= 1.0f);
    return blocked && on && (ccz > 0.0f);
}
template <class Cfg>
HNS_DEV bool d_blocked_exact(const Cfg &c, int C, const LosLine &l, const float *cyl) {
    const float d1 = d_norm2(l.diffx, l.diffy) + 1e-5f;       // the correctly rounded denominator of :63
    bool any = false;
    for (int k = 0; k < C; ++k) any = d_los_cylinder(c, l, d1, cyl[3 * k], cyl[3 * k + 1], cyl[3 * k + 2]) || any;
    return any;
}
template <class Cfg>
HNS_DEV bool d_blocked(const Cfg &c, int C, const V3 &dp, const V3 &tp, const float *cyl) {
    const LosLine l = d_los_setup(c, dp, tp);
    bool any = false, uncertain = false;
#pragma unroll 4
    for (int k = 0; k < C; ++k) any = d_los_cylinder_fast(l, cyl[3 * k], cyl[3 * k + 1], cyl[3 * k + 2], uncertain) || any;
    if (uncertain) any = d_blocked_exact(c, C, l, cyl);      // rare (~1e-6 per test): redo the env exactly
    return any;
}

// ---- A6 pieces: evader potential field   hideandseek.py:1067-1141 -----------------------------
// pursuer term of one drone (:1074-1088)
template <class Cfg>
HNS_DEV V3 d_prey_pursuer_term(const Cfg &c, const V3 &dp, const V3 &tp, bool blocked) {
    float rx = dp.x - tp.x, ry = dp.y - tp.y, rz = dp.z - tp.z;
    float dist = d_norm3(rx, ry, rz);
    float active = ((dist < c.target_detect_radius) && !blocked) ? 1.0f : 0.0f;
    float rec = 1.0f / (dist + 1e-5f);           // one reciprocal: direction (:1084) and magnitude (:1085) both scale by it
    V3 f;
    f.x = ((-rx * rec) * rec) * active;
    f.y = ((-ry * rec) * rec) * active;
    f.z = ((-rz * rec) * rec) * active;
    return f;
}
// arena walls/ceiling/floor (:1090-1112); also reports the out-of-arena flag (:1096-1098)
template <class Cfg>
HNS_DEV V3 d_prey_arena_term(const Cfg &c, const V3 &tp, bool &out_of_arena) {
    float od = d_norm2(tp.x, tp.y);
    float ro = 1.0f / (od + 1e-5f);
    float dirx = -tp.x * ro, diry = -tp.y * ro;
    bool out = HNS_FMA(tp.y, tp.y, tp.x * tp.x) > c.arena_sq;
    out_of_arena = out;
    float outf = out ? 1.0f : 0.0f, nout = out ? 0.0f : 1.0f;
    float rin = 1.0f / ((c.arena_size - od) + 1e-5f);
    V3 f;
    f.x = (outf * dirx) * 1e5f + (nout * dirx) * rin;
    f.y = (outf * diry) * 1e5f + (nout * diry) * rin;
    float H = c.max_height;
    bool hi = tp.z > H;
    float hif = hi ? 1.0f : 0.0f, nhi = hi ? 0.0f : 1.0f;
    float hz = H - tp.z;
    float frz = hif * -1e5f + (nhi * -hz) / (hz * hz + 1e-5f);
    bool lo = tp.z < 0.0f;
    float lof = lo ? 1.0f : 0.0f, nlo = lo ? 0.0f : 1.0f;
    float lz = 0.0f - tp.z;
    f.z = frz + (lof * 1e5f + (nlo * -lz) / (lz * lz + 1e-5f));
    return f;
}
// repulsion of one cylinder (:1129-1136)
template <class Cfg>
HNS_DEV void d_prey_cylinder_term(const Cfg &c, const V3 &tp, float ccx, float ccy, float ccz, float &tx, float &ty) {
    float rx = tp.x - ccx, ry = tp.y - ccy;
    float dc = d_norm2(rx, ry);
    float db = dc - c.cylinder_size;
    float act = (!(ccz < 0.0f) && (dc < c.target_detect_radius)) ? 1.0f : 0.0f;
    float w = 1.0f / ((dc + 1e-5f) * (db + 1e-5f));    // direction / (dc + eps) times magnitude 1 / (db + eps): one reciprocal
    tx = (act * rx) * w;
    ty = (act * ry) * w;
}

// ---- A5: rigid-body integration — the build's own spec -----------------------
struct Rigid { V3 pos; Q4 q; V3 lin; V3 ang; };

template <class Cfg>
HNS_DEV void d_integrate(const Cfg &c, Rigid &s, const V3 &force_w, const V3 &torque_b) {
    const float dt = c.dt;
    float ax = force_w.x * c.inv_mass, ay = force_w.y * c.inv_mass, az = HNS_FMA(force_w.z, c.inv_mass, -c.gravity);
    float vx = HNS_FMA(ax, dt, s.lin.x) * c.lin_damp_factor;
    float vy = HNS_FMA(ay, dt, s.lin.y) * c.lin_damp_factor;
    float vz = HNS_FMA(az, dt, s.lin.z) * c.lin_damp_factor;
    float sp = __builtin_sqrtf(HNS_FMA(vz, vz, HNS_FMA(vy, vy, vx * vx)));
    if (sp > c.max_lin_vel) {
        float sc = c.max_lin_vel / sp;
        vx *= sc; vy *= sc; vz *= sc;
    }
    V3 wb = d_quat_rot<true>(s.q, s.ang);
    float Iwx = wb.x * c.inertia[0], Iwy = wb.y * c.inertia[1], Iwz = wb.z * c.inertia[2];
    float gx = HNS_FMA(wb.y, Iwz, -(wb.z * Iwy)), gy = HNS_FMA(wb.z, Iwx, -(wb.x * Iwz)), gz = HNS_FMA(wb.x, Iwy, -(wb.y * Iwx));
    V3 w2;
    w2.x = HNS_FMA((torque_b.x - gx) * c.inv_inertia[0], dt, wb.x) * c.ang_damp_factor;
    w2.y = HNS_FMA((torque_b.y - gy) * c.inv_inertia[1], dt, wb.y) * c.ang_damp_factor;
    w2.z = HNS_FMA((torque_b.z - gz) * c.inv_inertia[2], dt, wb.z) * c.ang_damp_factor;
    // |w| > max_ang_vel (1000 rad/s: practically never): decided on the squares unless within 2^-20 of the limit —
    // RN(sqrt(x)) > m is then certain either way — the correctly rounded square root only on that path
    const float wn2 = HNS_FMA(w2.z, w2.z, HNS_FMA(w2.y, w2.y, w2.x * w2.x));
    const float m2 = c.max_ang_vel * c.max_ang_vel;
    if (!(wn2 < m2 * 0.99999904632568f)) {
        float wn = __builtin_sqrtf(wn2);
        if (wn > c.max_ang_vel) {
            float sc = c.max_ang_vel / wn;
            w2.x *= sc; w2.y *= sc; w2.z *= sc;
        }
    }
    V3 ww = d_quat_rot<false>(s.q, w2);
    float px = HNS_FMA(vx, dt, s.pos.x), py = HNS_FMA(vy, dt, s.pos.y), pz = HNS_FMA(vz, dt, s.pos.z);
    if (c.ground_clamp && pz < 0.0f) {
        pz = 0.0f;
        if (vz < 0.0f) vz = 0.0f;
    }
    float wwn = __builtin_sqrtf(HNS_FMA(ww.z, ww.z, HNS_FMA(ww.y, ww.y, ww.x * ww.x)));
    float half = (wwn * dt) * 0.5f;
    float sn, co;
    d_sincosf(half, sn, co);
    float so = (wwn > 1e-8f) ? sn / wwn : 0.5f * dt;
    float w1 = co, x1 = ww.x * so, y1 = ww.y * so, z1 = ww.z * so;
    float w2q = s.q.w, x2 = s.q.x, y2 = s.q.y, z2 = s.q.z;
    float nw = HNS_FMA(-z1, z2, HNS_FMA(-y1, y2, HNS_FMA(-x1, x2, w1 * w2q)));
    float nx = HNS_FMA(-z1, y2, HNS_FMA(y1, z2, HNS_FMA(x1, w2q, w1 * x2)));
    float ny = HNS_FMA(z1, x2, HNS_FMA(y1, w2q, HNS_FMA(-x1, z2, w1 * y2)));
    float nz = HNS_FMA(z1, w2q, HNS_FMA(-y1, x2, HNS_FMA(x1, y2, w1 * z2)));
    float iq = 1.0f / __builtin_sqrtf(HNS_FMA(nz, nz, HNS_FMA(ny, ny, HNS_FMA(nx, nx, nw * nw))));
    s.pos = {px, py, pz};
    s.lin = {vx, vy, vz};
    s.ang = ww;
    s.q = {nw * iq, nx * iq, ny * iq, nz * iq};
}

// ---- Philox4x32-10 (reset RNG) ---------------------------------------------
HNS_DEV void d_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
    uint32_t k0, k1, env, epoch, block;
    uint32_t buf[4];
    int have;
    __device__ float uniform() {
        if (have == 0) { d_philox(k0, k1, env, epoch, block++, 0u, buf); have = 4; }
        // the next word by selects: `buf[4 - have]` with a run-time index put the whole generator (44 bytes) in scratch memory
        const uint32_t u = have == 4 ? buf[0] : have == 3 ? buf[1] : have == 2 ? buf[2] : buf[3];
        have--;
        return (float)(u >> 8) * 5.9604644775390625e-8f;
    }
};

// continuous_to_grid, hideandseek.py:143-164
HNS_DEV int d_cell(const hns_cfg &c, float x) {
    int g = (int)__builtin_rintf(x / c.grid_size) + c.grid_num / 2;
    return g < 0 ? 0 : (g > c.grid_num - 1 ? c.grid_num - 1 : g);
}

}  // namespace hns
