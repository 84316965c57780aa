/*
 * hns_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (plain C, fp32, one env at a time) of the reference's HideAndSeek step
 * so that the HIP path can be checked against it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  Every function cites the reference lines
 * (relative to the reference repo) it follows.  Pinned against golden vectors produced by
 * executing the reference's own torch code (tests/golden/make_golden.py writes the .npz fixtures under tests/golden),
 * see tests/test_oracle_golden.py.
 *
 * PARITY NOTE: the rigid-body integrator (o_integrate) has no reference source — the reference
 * delegates it to closed-source PhysX (omni_drones/envs/isaac_env.py:233-234).  It follows the
 * build's own spec; for that one stage parity is UNPINNED by the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * All arithmetic is IEEE fp32 with explicit evaluation order; exp/tanh/sin/cos are the fixed
 * polynomial forms specified in DESIGN.md §Numerics (the HIP kernels implement the same
 * forms, which is what makes HIP-vs-oracle comparisons exact rather than approximate).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hns.h"

#define O_PI 3.14159265358979323846f

/* ------------------------------------------------------------------------------------------
 * elementary functions
 * ---------------------------------------------------------------------------------------- */
static inline float o_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define O_FMA(a, b, c) fmaf((a), (b), (c))
#define O_INV_PI 0.31830987334251404f   /* RN(1/fp32(pi)) */

/* expf: Cody-Waite reduction by ln2 (hi 0.693359375, lo -2.12194440e-4), degree-5 Cephes poly, FMA */
static float o_expf(float x) {
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return INFINITY;
    float k = rintf(x * 1.44269504088896341f);
    float r = O_FMA(k, -0.693359375f, x);
    r = O_FMA(k, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = O_FMA(p, r, 1.3981999507E-3f);
    p = O_FMA(p, r, 8.3334519073E-3f);
    p = O_FMA(p, r, 4.1665795894E-2f);
    p = O_FMA(p, r, 1.6666665459E-1f);
    p = O_FMA(p, r, 5.0000001201E-1f);
    float y = O_FMA(p, r * r, r) + 1.0f;
    int ki = (int)k;
    return y * o_asfloat((uint32_t)(ki + 127) << 23);
}

/* tanh(x) = sign(x) * (1 - e)/(1 + e), e = exp(-2|x|) */
static float o_tanhf(float x) {
    float e = o_expf(-2.0f * fabsf(x));
    float r = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -r : r;
}

/* sincosf: Cephes octant reduction with the 3-part pi/4, |x| < 8192, FMA */
static void o_sincosf(float x, float *s_out, float *c_out) {
    float ax = fabsf(x);
    int j = (int)(ax * 1.27323954473516f); /* 4/pi */
    if (j & 1) j += 1;
    float y = (float)j;
    j &= 7;
    float z = O_FMA(y, -3.77489497744594108e-8f, O_FMA(y, -2.4187564849853515625e-4f, O_FMA(y, -0.78515625f, ax)));
    float zz = z * z;
    float ps = -1.9515295891E-4f;
    ps = O_FMA(ps, zz, 8.3321608736E-3f);
    ps = O_FMA(ps, zz, -1.6666654611E-1f);
    float sp = O_FMA(ps * zz, z, z);
    float pc = 2.443315711809948E-005f;
    pc = O_FMA(pc, zz, -1.388731625493765E-003f);
    pc = O_FMA(pc, zz, 4.166664568298827E-002f);
    float cp = O_FMA(pc * zz, zz, O_FMA(-0.5f, zz, 1.0f));
    float s, c;
    switch (j) {
        case 0: s = sp; c = cp; break;
        case 2: s = cp; c = -sp; break;
        case 4: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break; /* 6 */
    }
    if (x < 0.0f) s = -s;
    *s_out = s;
    *c_out = c;
}

static inline float o_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float o_norm3(float x, float y, float z) { return sqrtf(O_FMA(z, z, O_FMA(y, y, x * x))); }
static inline float o_norm2(float x, float y) { return sqrtf(O_FMA(y, y, x * x)); }

/* omni_drones/utils/torch.py:183-191 (quat_rotate) and :194-202 (quat_rotate_inverse) */
static void o_quat_rot(const float q[4], const float v[3], float out[3], int inverse) {
    /* a = v(2w^2-1), b = 2w (q_vec x v), c = 2 q_vec (q_vec . v); result a +- b + c, fused */
    float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    float w2 = 2.0f * qw;
    float s = O_FMA(w2, qw, -1.0f);
    float cx = O_FMA(qy, v[2], -(qz * v[1]));
    float cy = O_FMA(qz, v[0], -(qx * v[2]));
    float cz = O_FMA(qx, v[1], -(qy * v[0]));
    float dot2 = 2.0f * O_FMA(qz, v[2], O_FMA(qy, v[1], qx * v[0]));
    float bw = inverse ? -w2 : w2;
    out[0] = O_FMA(qx, dot2, O_FMA(cx, bw, v[0] * s));
    out[1] = O_FMA(qy, dot2, O_FMA(cy, bw, v[1] * s));
    out[2] = O_FMA(qz, dot2, O_FMA(cz, bw, v[2] * s));
}

/* quat_rotate(q, x_hat) and quat_rotate(q, (0,0,t)): the same formula with the zero products of the
 * basis vector dropped.  Used for heading/up (multirotor.py:613-614) and the thrust vector (:491). */
static void o_quat_rot_x(const float q[4], float out[3]) {
    float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    float s = O_FMA(2.0f * qw, qw, -1.0f);
    out[0] = O_FMA(2.0f * qx, qx, s);
    out[1] = 2.0f * O_FMA(qz, qw, qy * qx);
    out[2] = 2.0f * O_FMA(qz, qx, -(qy * qw));
}
static void o_quat_rot_z(const float q[4], float t, float out[3]) {
    float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    float s = O_FMA(2.0f * qw, qw, -1.0f);
    float dot = qz * t;
    out[0] = 2.0f * O_FMA(qy * t, qw, qx * dot);
    out[1] = 2.0f * O_FMA(qy, dot, -((qx * t) * qw));
    out[2] = O_FMA(2.0f * qz, dot, t * s);
}

/* omni_drones/utils/torch.py:110-127 */
static void o_euler_to_quat(const float rpy[3], float q[4]) {
    float sr, cr, sp, cp, sy, cy;
    o_sincosf(rpy[0] * 0.5f, &sr, &cr);
    o_sincosf(rpy[1] * 0.5f, &sp, &cp);
    o_sincosf(rpy[2] * 0.5f, &sy, &cy);
    q[0] = (cr * cp) * cy + (sr * sp) * sy;
    q[1] = (sr * cp) * cy - (cr * sp) * sy;
    q[2] = (cr * sp) * cy + (sr * cp) * sy;
    q[3] = (cr * cp) * sy - (sr * sp) * cy;
}

/* ------------------------------------------------------------------------------------------
 * A1+A2: action -> CTBR -> body-rate PID -> motor commands
 * transforms.py:425-459, lee_position_controller.py:476-550
 * ---------------------------------------------------------------------------------------- */
static void o_ctbr_pid(const hns_cfg *c, const float action[4], const float q[4], const float angvel[3],
                       float prev_action[4], float integ[3], float last[3],
                       float cmd[4], float *action_error, float ctbr_out[4], float target_rate_out[3]) {
    float a[4];
    for (int i = 0; i < 4; ++i) a[i] = o_tanhf(action[i]);
    float ctbr[4] = {a[0], a[1], a[2], o_clamp((a[3] + 1.0f) / 2.0f, 0.0f, c->max_thrust_ratio)};
    if (c->fixed_yaw) ctbr[2] = 0.0f;
    float e2 = 0.0f;
    for (int i = 0; i < 4; ++i) {
        float d = ctbr[i] - prev_action[i];
        e2 = (i == 0) ? d * d : e2 + d * d;
    }
    *action_error = sqrtf(e2);
    for (int i = 0; i < 4; ++i) prev_action[i] = ctbr[i];
    float target[3];
    for (int i = 0; i < 3; ++i) target[i] = (ctbr[i] * 180.0f) * c->target_clip;
    float thrust = ctbr[3] * 65536.0f;
    if (target_rate_out) for (int i = 0; i < 3; ++i) target_rate_out[i] = target[i];

    float br[3];
    o_quat_rot(q, angvel, br, 1);
    float out[3];
    for (int i = 0; i < 3; ++i) {
        br[i] = (br[i] * 180.0f) * O_INV_PI;   /* CUDA `tensor / python_scalar` = multiply by the fp32 reciprocal */
        float err = target[i] - br[i];
        float P = err * c->pid_kp[i];
        float deriv = -(br[i] - last[i]) * c->inv_dt;   /* `/ self.dt` (:509): CUDA multiplies by the fp32 reciprocal of the scalar */
        if (deriv != deriv) deriv = 0.0f;
        float D = deriv * c->pid_kd[i];
        float in = integ[i] + err * c->dt;
        in = o_clamp(in, -c->pid_ilimit[i], c->pid_ilimit[i]);
        integ[i] = in;
        float I = in * c->pid_ki[i];
        float FF = target[i] * 0.0f;
        float o = ((P + D) + I) + FF;
        if (o != o) o = 0.0f;
        out[i] = o_clamp(o, -c->pid_outlimit, c->pid_outlimit);
        last[i] = br[i];
    }
    float r = out[0] / 2.0f, p = out[1] / 2.0f, y = out[2];
    float m[4] = {((thrust + r) - p) + y, ((thrust + r) + p) - y, ((thrust - r) + p) + y, ((thrust - r) - p) - y};
    if (ctbr_out) { ctbr_out[0] = r; ctbr_out[1] = p; ctbr_out[2] = y; ctbr_out[3] = thrust; }
    for (int i = 0; i < 4; ++i) {
        float v = (m[i] / 65536.0f) * 2.0f - c->max_thrust_ratio;
        if (v != v) v = 0.0f;                               /* torch.nan_to_num_(cmds, 0.) */
        else if (v == INFINITY) v = 3.4028234663852886e38f;
        else if (v == -INFINITY) v = -3.4028234663852886e38f;
        cmd[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * A3: rotor lag + thrust/moment   rotor_group.py:55-71
 * ---------------------------------------------------------------------------------------- */
static void o_rotor(const hns_cfg *c, const float cmd[4], float throttle[4], float thrust[4], float moment[4],
                    float *throttle_difference) {
    float d2 = 0.0f;
    for (int i = 0; i < 4; ++i) {
        float tgt = sqrtf(o_clamp((cmd[i] + 1.0f) / 2.0f, 0.0f, 1.0f));
        float tau = (tgt > throttle[i]) ? c->tau_up : c->tau_down;
        float old = throttle[i];
        float thr = old + tau * (tgt - old);
        throttle[i] = thr;
        float t = o_clamp(thr * thr + 0.0f, 0.0f, 1.0f);
        thrust[i] = t * c->kf[i];
        moment[i] = (t * c->km[i]) * -c->rotor_dir[i];
        float d = thr - old;
        d2 = (i == 0) ? d * d : d2 + d * d;
    }
    *throttle_difference = sqrtf(d2);     /* multirotor.py:507 */
}

/* ------------------------------------------------------------------------------------------
 * A4: downwash of drone j on drone i   multirotor.py:488-494, 725-753
 * tj_w = quat_rotate(q_j, [0,0,sum thrust_j])
 * ---------------------------------------------------------------------------------------- */
/* Evaluation form (same quantities as :725-753, fewer roundings): the unit thrust direction is tj_w times ONE reciprocal
 * of (|tj_w| + 1e-6); (kr r / z)^2 is formed from the squared radial distance, 4 r^2 / z^2, without taking r itself
 * (z = 0: +inf for r > 0, NaN for r = 0, as kr r / z gives). */
static void o_downwash_pair(const float pi[3], const float pj[3], const float tj_w[3], float f[3]) {
    float n = o_norm3(tj_w[0], tj_w[1], tj_w[2]);
    float inv = 1.0f / (n + 1e-6f);
    float d[3] = {tj_w[0] * inv, tj_w[1] * inv, tj_w[2] * inv};
    float rel[3] = {pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2]};
    float zd = O_FMA(rel[2], d[2], O_FMA(rel[1], d[1], rel[0] * d[0]));
    float rx = O_FMA(-zd, d[0], rel[0]), ry = O_FMA(-zd, d[1], rel[1]), rz = O_FMA(-zd, d[2], rel[2]);
    float r2 = O_FMA(rz, rz, O_FMA(ry, ry, rx * rx));
    float z = zd < 0.0f ? 0.0f : zd;
    float u2 = (4.0f * r2) / (z * z);
    float den = O_FMA(0.3f, z, 1.0f);
    float v = o_expf(-0.5f * u2) / (den * den);
    f[0] = v * -tj_w[0]; f[1] = v * -tj_w[1]; f[2] = v * -tj_w[2];
}

/* ------------------------------------------------------------------------------------------
 * A7: line of sight drone->target blocked by a cylinder (xy plane)   hideandseek.py:47-103
 * ---------------------------------------------------------------------------------------- */
static int o_blocked(const hns_cfg *c, int C, const float dp[3], const float tp[3], const float *cyl) {
    float diffx = dp[0] - tp[0], diffy = dp[1] - tp[1];
    float den = o_norm2(diffx, diffy);
    float dx = tp[0] - dp[0], dy = tp[1] - dp[1];
    float dent = O_FMA(dy, dy, dx * dx);
    int any = 0;
    for (int k = 0; k < C; ++k) {
        const float *cc = cyl + 3 * k;
        float d2x = cc[0] - tp[0], d2y = cc[1] - tp[1];
        float num = fabsf(O_FMA(diffx, d2y, -(diffy * d2x)));
        float dist = num / (den + 1e-5f);
        int blocked = dist <= c->cylinder_size;
        float numt = O_FMA(cc[1] - dp[1], dy, (cc[0] - dp[0]) * dx);
        float t = numt / (dent + 1e-5f);
        int on = (t >= 0.0f) && (t <= 1.0f);
        int ground = cc[2] > 0.0f;
        any |= (blocked && on && ground);
    }
    return any;
}

/* ------------------------------------------------------------------------------------------
 * A6: evader potential field + per-axis velocity   hideandseek.py:737-744, 1067-1141
 * drone_pos [A,3]; cyl [C,3]; returns force[3], vel[3]; out_of_arena stat is OR-ed in place
 * ---------------------------------------------------------------------------------------- */
static void o_prey(const hns_cfg *c, int A, int C, const float *drone_pos, const float tp[3], const float *cyl,
                   float force[3], float vel[3], float *out_of_arena_stat) {
    float F[3] = {0.0f, 0.0f, 0.0f};
    for (int a = 0; a < A; ++a) {
        const float *dp = drone_pos + 3 * a;
        float rp[3] = {dp[0] - tp[0], dp[1] - tp[1], dp[2] - tp[2]};
        float dist = o_norm3(rp[0], rp[1], rp[2]);
        int blocked = o_blocked(c, C, dp, tp, cyl);
        float active = ((dist < c->target_detect_radius) && !blocked) ? 1.0f : 0.0f;
        float rec = 1.0f / (dist + 1e-5f);           /* one reciprocal: direction (:1084) and magnitude (:1085) both scale by it */
        for (int i = 0; i < 3; ++i) {
            float dir = -rp[i] * rec;
            float fp = (dir * rec) * active;
            F[i] = (a == 0) ? fp : F[i] + fp;
        }
    }
    /* arena  :1094-1112 */
    float od = o_norm2(tp[0], tp[1]);
    float ro = 1.0f / (od + 1e-5f);
    float dirx = -tp[0] * ro, diry = -tp[1] * ro;
    int out = O_FMA(tp[1], tp[1], tp[0] * tp[0]) > c->arena_sq;
    if (out_of_arena_stat) *out_of_arena_stat = ((*out_of_arena_stat != 0.0f) || out) ? 1.0f : 0.0f;
    float outf = out ? 1.0f : 0.0f, nout = out ? 0.0f : 1.0f;
    float rin = 1.0f / ((c->arena_size - od) + 1e-5f);
    float frx = (outf * dirx) * 1e5f + (nout * dirx) * rin;
    float fry = (outf * diry) * 1e5f + (nout * diry) * rin;
    float H = c->max_height;
    int hi = tp[2] > H;
    float hif = hi ? 1.0f : 0.0f, nhi = hi ? 0.0f : 1.0f;
    float hz = H - tp[2];
    float frz = hif * -1e5f + (nhi * -hz) / (hz * hz + 1e-5f);
    int lo = tp[2] < 0.0f;
    float lof = lo ? 1.0f : 0.0f, nlo = lo ? 0.0f : 1.0f;
    float lz = 0.0f - tp[2];
    frz = frz + (lof * 1e5f + (nlo * -lz) / (lz * lz + 1e-5f));
    F[0] = F[0] + frx; F[1] = F[1] + fry; F[2] = F[2] + frz;
    /* cylinders  :1129-1136 (mask = cylinder z < 0, written by the previous obs pass :759) */
    float fcx = 0.0f, fcy = 0.0f;
    for (int k = 0; k < C; ++k) {
        const float *cc = cyl + 3 * k;
        float rx = tp[0] - cc[0], ry = tp[1] - cc[1];
        float dc = o_norm2(rx, ry);
        float db = dc - c->cylinder_size;
        float act = (!(cc[2] < 0.0f) && (dc < c->target_detect_radius)) ? 1.0f : 0.0f;
        float w = 1.0f / ((dc + 1e-5f) * (db + 1e-5f));    /* direction / (dc + eps) times magnitude 1 / (db + eps): one reciprocal */
        float tx = (act * rx) * w;
        float ty = (act * ry) * w;
        fcx += tx;
        fcy += ty;
    }
    F[0] = F[0] + fcx; F[1] = F[1] + fcy; F[2] = F[2] + 0.0f;
    for (int i = 0; i < 3; ++i) {
        force[i] = F[i];
        vel[i] = (c->v_prey * F[i]) / (fabsf(F[i]) + 1e-5f);   /* :741 norm over a size-1 dim */
    }
}

/* ------------------------------------------------------------------------------------------
 * A5: rigid-body integration — the build's own spec, no reference source
 * ds = [pos3 quat4 linvel3 angvel3]; force_w world force, torque_b body torque
 * ---------------------------------------------------------------------------------------- */
static void o_integrate(const hns_cfg *c, float ds[13], const float force_w[3], const float torque_b[3]) {
    float *pos = ds, *q = ds + 3, *lin = ds + 7, *ang = ds + 10;
    const float dt = c->dt;
    float acc[3] = {force_w[0] * c->inv_mass, force_w[1] * c->inv_mass, O_FMA(force_w[2], c->inv_mass, -c->gravity)};
    float v[3];
    for (int i = 0; i < 3; ++i) v[i] = O_FMA(acc[i], dt, lin[i]) * c->lin_damp_factor;
    float sp = sqrtf(O_FMA(v[2], v[2], O_FMA(v[1], v[1], v[0] * v[0])));
    if (sp > c->max_lin_vel) {
        float sc = c->max_lin_vel / sp;
        v[0] *= sc; v[1] *= sc; v[2] *= sc;
    }
    float wb[3];
    o_quat_rot(q, ang, wb, 1);
    float Iw[3] = {wb[0] * c->inertia[0], wb[1] * c->inertia[1], wb[2] * c->inertia[2]};
    float gy[3] = {O_FMA(wb[1], Iw[2], -(wb[2] * Iw[1])), O_FMA(wb[2], Iw[0], -(wb[0] * Iw[2])), O_FMA(wb[0], Iw[1], -(wb[1] * Iw[0]))};
    float w2[3];
    for (int i = 0; i < 3; ++i) w2[i] = O_FMA((torque_b[i] - gy[i]) * c->inv_inertia[i], dt, wb[i]) * c->ang_damp_factor;
    float wn = sqrtf(O_FMA(w2[2], w2[2], O_FMA(w2[1], w2[1], w2[0] * w2[0])));
    if (wn > c->max_ang_vel) {
        float sc = c->max_ang_vel / wn;
        w2[0] *= sc; w2[1] *= sc; w2[2] *= sc;
    }
    float ww[3];
    o_quat_rot(q, w2, ww, 0);
    float p[3] = {O_FMA(v[0], dt, pos[0]), O_FMA(v[1], dt, pos[1]), O_FMA(v[2], dt, pos[2])};
    if (c->ground_clamp && p[2] < 0.0f) {
        p[2] = 0.0f;
        if (v[2] < 0.0f) v[2] = 0.0f;
    }
    float wwn = sqrtf(O_FMA(ww[2], ww[2], O_FMA(ww[1], ww[1], ww[0] * ww[0])));
    float half = (wwn * dt) * 0.5f;
    float s, co;
    o_sincosf(half, &s, &co);
    float so = (wwn > 1e-8f) ? s / wwn : 0.5f * dt;
    float w1 = co, x1 = ww[0] * so, y1 = ww[1] * so, z1 = ww[2] * so;
    float w2q = q[0], x2 = q[1], y2 = q[2], z2 = q[3];
    float nq[4] = {O_FMA(-z1, z2, O_FMA(-y1, y2, O_FMA(-x1, x2, w1 * w2q))),
                   O_FMA(-z1, y2, O_FMA(y1, z2, O_FMA(x1, w2q, w1 * x2))),
                   O_FMA(z1, x2, O_FMA(y1, w2q, O_FMA(-x1, z2, w1 * y2))),
                   O_FMA(z1, w2q, O_FMA(-y1, x2, O_FMA(x1, y2, w1 * z2)))};
    float iq = 1.0f / sqrtf(O_FMA(nq[3], nq[3], O_FMA(nq[2], nq[2], O_FMA(nq[1], nq[1], nq[0] * nq[0]))));
    for (int i = 0; i < 3; ++i) { pos[i] = p[i]; lin[i] = v[i]; ang[i] = ww[i]; }
    for (int i = 0; i < 4; ++i) q[i] = nq[i] * iq;
}

/* ------------------------------------------------------------------------------------------
 * A8: observation pass for one env   multirotor.py:599-633, hideandseek.py:746-917
 * Side products kept for the reward pass: blocked[A], bdetect, knn index/mask.
 * ---------------------------------------------------------------------------------------- */
/* Two-evader extension (hns_cfg.num_targets = 2, NOT in the reference; include/hns.h): `tp` then points to
 * [2][3], rows have 24 values, line of sight / detection per evader. */
static inline int o_nt(const hns_cfg *c) { return c->num_targets == 2 ? 2 : 1; }
static inline int o_self_dim(const hns_cfg *c) { return c->num_targets == 2 ? 24 : HNS_SELF_DIM; }

typedef struct o_obs_side {
    int blocked[HNS_MAX_AGENTS];
    int bdetect;
    int blocked1[HNS_MAX_AGENTS];   /* second evader */
    int bdetect1;
    int knn_idx[HNS_MAX_AGENTS][HNS_MAX_CYLINDERS];
    int knn_masked[HNS_MAX_AGENTS][HNS_MAX_CYLINDERS];
} o_obs_side;

static void o_obs(const hns_cfg *c, int A, int C, int K, const float *drone_state /*[A,13]*/, const float *tp /*[NT,3]*/,
                  const float *cyl, float progress, float *obs_self /*[A,20]*/, float *obs_others /*[A,A-1,3]*/,
                  float *obs_cyl /*[A,K,5]*/, float *state_drones /*[A,20] or NULL*/, o_obs_side *side) {
    int det_any = 0, det_any1 = 0;
    const int NT = o_nt(c), SD = o_self_dim(c);
    float rt[HNS_MAX_AGENTS][3], rt1[HNS_MAX_AGENTS][3];
    for (int a = 0; a < A; ++a) {
        const float *ds = drone_state + 13 * a;
        for (int i = 0; i < 3; ++i) rt[a][i] = ds[i] - tp[i];
        float dist = o_norm3(rt[a][0], rt[a][1], rt[a][2]);
        side->blocked[a] = o_blocked(c, C, ds, tp, cyl);
        int det = (dist < c->drone_detect_radius) && !side->blocked[a];
        det_any |= det;
        side->blocked1[a] = 0;
        if (NT == 2) {
            for (int i = 0; i < 3; ++i) rt1[a][i] = ds[i] - tp[3 + i];
            float dist1 = o_norm3(rt1[a][0], rt1[a][1], rt1[a][2]);
            side->blocked1[a] = o_blocked(c, C, ds, tp + 3, cyl);
            det_any1 |= (dist1 < c->drone_detect_radius) && !side->blocked1[a];
        }
    }
    side->bdetect = det_any;
    side->bdetect1 = det_any1;
    float t = progress * c->inv_max_episode_length;   /* :796, CUDA scalar-division form */
    for (int a = 0; a < A; ++a) {
        const float *ds = drone_state + 13 * a;
        float heading[3], up[3];
        o_quat_rot_x(ds + 3, heading);
        o_quat_rot_z(ds + 3, 1.0f, up);
        float *o = obs_self + SD * a;
        for (int i = 0; i < 3; ++i) o[i] = det_any ? rt[a][i] : c->mask_value;
        if (NT == 2) {
            for (int i = 0; i < 3; ++i) o[HNS_SELF_DIM + i] = det_any1 ? rt1[a][i] : c->mask_value;
            o[HNS_SELF_DIM + 3] = 0.0f;
        }
        for (int i = 0; i < 7; ++i) o[3 + i] = ds[3 + i];
        for (int i = 0; i < 3; ++i) { o[10 + i] = heading[i]; o[13 + i] = up[i]; }
        for (int i = 0; i < 4; ++i) o[16 + i] = t;
        if (state_drones) {
            float *s = state_drones + SD * a;
            for (int i = 0; i < SD; ++i) s[i] = o[i];
            for (int i = 0; i < 3; ++i) s[i] = rt[a][i];
            if (NT == 2) for (int i = 0; i < 3; ++i) s[HNS_SELF_DIM + i] = rt1[a][i];
        }
        /* state_others: p_i - p_j, j != i ascending (utils/torch.py:41-53) */
        int w = 0;
        for (int j = 0; j < A; ++j) {
            if (j == a) continue;
            const float *dj = drone_state + 13 * j;
            for (int i = 0; i < 3; ++i) obs_others[((a * (A - 1)) + w) * 3 + i] = ds[i] - dj[i];
            ++w;
        }
        /* k nearest cylinders by (3-D distance - size), ascending, ties -> lower index (:767-778) */
        float md[HNS_MAX_CYLINDERS];
        int taken[HNS_MAX_CYLINDERS];
        for (int k = 0; k < C; ++k) {
            const float *cc = cyl + 3 * k;
            md[k] = o_norm3(ds[0] - cc[0], ds[1] - cc[1], ds[2] - cc[2]) - c->cylinder_size;
            taken[k] = 0;
        }
        for (int s = 0; s < K; ++s) {
            int best = -1;
            for (int k = 0; k < C; ++k) {
                if (taken[k]) continue;
                if (best < 0 || md[k] < md[best]) best = k;
            }
            taken[best] = 1;
            const float *cc = cyl + 3 * best;
            int masked = cc[2] < 0.0f;
            side->knn_idx[a][s] = best;
            side->knn_masked[a][s] = masked;
            float *oc = obs_cyl + ((a * K) + s) * 5;
            if (masked) {
                for (int i = 0; i < 5; ++i) oc[i] = c->mask_value;
            } else {
                for (int i = 0; i < 3; ++i) oc[i] = ds[i] - cc[i];
                oc[3] = c->cylinder_height;
                oc[4] = c->cylinder_size;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * A9: reward / done / stats for one env   hideandseek.py:919-1065
 * stats: pointer to this env's column, stride = E floats between rows
 * ---------------------------------------------------------------------------------------- */
static void o_reward(const hns_cfg *c, int A, int C, int K, const float *drone_state, const float *tp /*[NT,3]*/,
                     const float *cyl, float progress, const o_obs_side *side, const float *action_error,
                     const float *thr_diff, float *stats, size_t sstride, float *reward, uint8_t *done_out) {
    (void)C;
#define ST(i) stats[(size_t)(i) * sstride]
    float iA = c->inv_num_agents;   /* mean over agents = sum * (1/A), as torch's CUDA mean */
    float dist_rew[HNS_MAX_AGENTS], speed_rew[HNS_MAX_AGENTS], coll_rew[HNS_MAX_AGENTS], smooth_rew[HNS_MAX_AGENTS];
    float sum_dist = 0, sum_speed = 0, sum_cc = 0, sum_cd = 0, sum_cw = 0, sum_coll = 0, sum_smooth = 0, sum_td = 0;
    float max_td = 0;
    int any_cap = 0, all_blocked = 1, any_coll = 0;
    for (int a = 0; a < A; ++a) {
        const float *ds = drone_state + 13 * a;
        float d = o_norm3(tp[0] - ds[0], tp[1] - ds[1], tp[2] - ds[2]);
        int cap_ok = (d < c->catch_radius) && !side->blocked[a];
        int blk = side->blocked[a];
        if (o_nt(c) == 2) {   /* nearest evader for the distance term, any evader for the capture */
            float d1 = o_norm3(tp[3] - ds[0], tp[4] - ds[1], tp[5] - ds[2]);
            cap_ok = cap_ok || ((d1 < c->catch_radius) && !side->blocked1[a]);
            blk = blk && side->blocked1[a];
            d = d1 < d ? d1 : d;
        }
        float act = (d > c->catch_radius) ? 1.0f : 0.0f;
        dist_rew[a] = (-c->dist_reward_coef * d) * act;
        any_cap |= cap_ok;
        all_blocked &= blk;
        float sp = o_norm3(ds[7], ds[8], ds[9]);
        speed_rew[a] = -c->speed_coef * ((sp > c->v_drone) ? 1.0f : 0.0f);
        float cc = 0.0f;
        for (int s = 0; s < K; ++s) {
            const float *cy = cyl + 3 * side->knn_idx[a][s];
            float rx = ds[0] - cy[0], ry = ds[1] - cy[1];
            float dxy = o_norm2(rx, ry);
            float hit = ((dxy - c->cylinder_size) < c->collision_radius) ? 1.0f : 0.0f;
            if (side->knn_masked[a][s]) hit = 0.0f;
            cc = (s == 0) ? hit : cc + hit;
        }
        float cr = -c->collision_coef * cc;
        float cd = 0.0f;
        int first = 1;
        for (int j = 0; j < A; ++j) {
            if (j == a) continue;
            const float *dj = drone_state + 13 * j;
            float dd = o_norm3(ds[0] - dj[0], ds[1] - dj[1], ds[2] - dj[2]);
            float hit = (dd < c->coll_drone_dist) ? 1.0f : 0.0f;
            cd = first ? hit : cd + hit;
            first = 0;
        }
        cr = cr + -c->collision_coef * cd;
        float cw = ((ds[2] > c->max_height) ? 1.0f : 0.0f)
                   + ((O_FMA(ds[1], ds[1], ds[0] * ds[0]) > c->arena_sq) ? 1.0f : 0.0f);
        cr = cr + -c->collision_coef * cw;
        coll_rew[a] = cr;
        any_coll |= (cr < 0.0f);
        float sm = c->smoothness_coef * o_expf(-action_error[a]);
        if (!c->use_deployment) sm = 0.0f;
        smooth_rew[a] = sm;
        if (a == 0) {
            sum_dist = dist_rew[a]; sum_speed = speed_rew[a]; sum_cc = cc; sum_cd = cd; sum_cw = cw;
            sum_coll = cr; sum_smooth = sm; sum_td = thr_diff[a]; max_td = thr_diff[a];
        } else {
            sum_dist += dist_rew[a]; sum_speed += speed_rew[a]; sum_cc += cc; sum_cd += cd; sum_cw += cw;
            sum_coll += cr; sum_smooth += sm; sum_td += thr_diff[a];
            if (thr_diff[a] > max_td) max_td = thr_diff[a];
        }
    }
    float detf = (side->bdetect || side->bdetect1) ? 1.0f : 0.0f;
    float detect_rew = c->detect_reward_coef * detf;
    float catch_rew = c->catch_reward_coef * (any_cap ? 1.0f : 0.0f);
    int capture_flag = catch_rew != 0.0f;             /* torch.any(catch_reward, dim=1) :945 */

    ST(HNS_ST_DISTANCE_REWARD) += sum_dist * iA;
    ST(HNS_ST_SUM_DETECT_STEP) += 1.0f * detf;
    {   /* mean over A identical values (:933), summed the same way */
        float s = detect_rew;
        for (int a = 1; a < A; ++a) s += detect_rew;
        ST(HNS_ST_DETECT_REWARD) += s * iA;
        s = catch_rew;
        for (int a = 1; a < A; ++a) s += catch_rew;
        ST(HNS_ST_BLOCKED) += all_blocked ? 1.0f : 0.0f;
        ST(HNS_ST_SUCCESS) = (capture_flag || ST(HNS_ST_SUCCESS) != 0.0f) ? 1.0f : 0.0f;
        float cur = (capture_flag ? 1.0f : 0.0f) * progress + (capture_flag ? 0.0f : 1.0f) * (float)c->max_episode_length;
        if (cur < ST(HNS_ST_FIRST_CAPTURE_STEP)) ST(HNS_ST_FIRST_CAPTURE_STEP) = cur;
        ST(HNS_ST_CATCH_REWARD) += s * iA;
    }
    ST(HNS_ST_SPEED_REWARD) += sum_speed * iA;
    ST(HNS_ST_COLLISION_CYLINDER) += sum_cc * iA;
    ST(HNS_ST_COLLISION_DRONE) += sum_cd * iA;
    ST(HNS_ST_COLLISION) += any_coll ? 1.0f : 0.0f;
    ST(HNS_ST_COLLISION_WALL) += sum_cw * iA;
    ST(HNS_ST_COLLISION_REWARD) += sum_coll * iA;
    ST(HNS_ST_SMOOTHNESS_COEF) = c->smoothness_coef;
    ST(HNS_ST_SMOOTHNESS_REWARD) += sum_smooth * iA;
    ST(HNS_ST_SMOOTHNESS_MEAN) += sum_td * iA;
    if (max_td > ST(HNS_ST_SMOOTHNESS_MAX)) ST(HNS_ST_SMOOTHNESS_MAX) = max_td;

    float sum_rew = 0.0f;
    for (int a = 0; a < A; ++a) {
        float r = ((((dist_rew[a] + detect_rew) + catch_rew) + coll_rew[a]) + speed_rew[a]) + smooth_rew[a];
        reward[a] = r;
        sum_rew = (a == 0) ? r : sum_rew + r;
    }
    int done = progress >= (float)c->max_episode_length;
    *done_out = (uint8_t)done;
    if (done) {
        static const int div[] = {HNS_ST_COLLISION, HNS_ST_ACTION_ERROR_ORDER1_MEAN, HNS_ST_TARGET_PREDICTED_ERROR,
                                  HNS_ST_SMOOTHNESS_MEAN, HNS_ST_SMOOTHNESS_REWARD, HNS_ST_DISTANCE_REWARD,
                                  HNS_ST_DETECT_REWARD, HNS_ST_CATCH_REWARD, HNS_ST_COLLISION_REWARD,
                                  HNS_ST_COLLISION_WALL, HNS_ST_COLLISION_DRONE, HNS_ST_COLLISION_CYLINDER,
                                  HNS_ST_SPEED_REWARD};
        for (size_t i = 0; i < sizeof(div) / sizeof(div[0]); ++i) ST(div[i]) = ST(div[i]) / progress;
    }
    ST(HNS_ST_RETURN) += sum_rew * iA;
#undef ST
}

/* ------------------------------------------------------------------------------------------
 * Full step over all envs, on the ABI's buffers (host pointers here).  Order = SURVEY App. C.
 * ---------------------------------------------------------------------------------------- */
static int g_step_threads = 1;
/* envs are independent: the step may be spread over host threads (bench.py's multi-core cpu_baseline leg);
 * the results do not depend on the thread count */
void hns_oracle_set_threads(int n) { g_step_threads = n < 1 ? 1 : n; }

int hns_oracle_step(const hns_cfg *c, const hns_buffers *b, const float *action) {
    const int E = c->num_envs, A = c->num_agents, C = c->num_cylinders, K = c->obs_max_cylinder;
    const size_t S = c->stats_stride ? (size_t)c->stats_stride : (size_t)E;       /* row stride of `stats` (include/hns.h) */
    if (A < 1 || A > HNS_MAX_AGENTS || C > HNS_MAX_CYLINDERS || K > C) return HNS_ERR_INVALID_ARG;
#pragma omp parallel for schedule(static) num_threads(g_step_threads)
    for (int e = 0; e < E; ++e) {
        float *ds = b->drone_state + (size_t)e * A * 13;
        const int NT = o_nt(c), SD = o_self_dim(c);
        float *tp = b->target_pos + (size_t)e * 3 * NT;
        const float *cyl = b->cylinders + (size_t)e * C * 3;
        float *stats = b->stats + e;
        float thrust[HNS_MAX_AGENTS][4], moment[HNS_MAX_AGENTS][4], thr_diff[HNS_MAX_AGENTS];
        float tw[HNS_MAX_AGENTS][3];
        float sum_ae = 0.0f;
        /* reset_pid = tensordict['done'] (transforms.py:449-454 -> lee_position_controller.py:497-502): read before anything of this env is
         * written — it may alias `done` (include/hns.h) */
        const int reset_pid = b->reset_pid && b->reset_pid[e];
        for (int a = 0; a < A; ++a) {
            size_t ia = (size_t)e * A + a;
            float cmd[4], ctbr[4], trate[3];
            if (c->action_input == HNS_ACTION_MOTOR) {
                /* include/hns.h: the caller's PIDRateController transform ran in front (scripts/train.py:165-171) — `action` holds its rotor
                 * commands (transforms.py:455-456), b->action_error / b->prev_action what it left under ("stats","action_error_order1") /
                 * ("info","prev_action"); the step starts at HideAndSeek._pre_sim_step (hideandseek.py:725-735) */
                for (int i = 0; i < 4; ++i) cmd[i] = action[ia * 4 + i];
            } else {
            if (reset_pid)
                for (int i = 0; i < 3; ++i) { b->pid_integ[ia * 4 + i] = 0.0f; b->pid_last_rate[ia * 4 + i] = 0.0f; }
            o_ctbr_pid(c, action + ia * 4, ds + 13 * a + 3, ds + 13 * a + 10, b->prev_action + ia * 4,
                       b->pid_integ + ia * 4, b->pid_last_rate + ia * 4, cmd, b->action_error + ia, ctbr, trate);
            if (b->ctbr) for (int i = 0; i < 4; ++i) b->ctbr[ia * 4 + i] = ctbr[i];                 /* transforms.py:456 */
            if (b->target_rate) {                                                                     /* transforms.py:457 */
                for (int i = 0; i < 3; ++i) b->target_rate[ia * 4 + i] = trate[i];
                b->target_rate[ia * 4 + 3] = 0.0f;
            }
            }
            sum_ae = (a == 0) ? b->action_error[ia] : sum_ae + b->action_error[ia];
            o_rotor(c, cmd, b->throttle + ia * 4, thrust[a], moment[a], &thr_diff[a]);
            float ts = ((thrust[a][0] + thrust[a][1]) + thrust[a][2]) + thrust[a][3];
            o_quat_rot_z(ds + 13 * a + 3, ts, tw[a]);
        }
        /* A10  hideandseek.py:731-733 */
        float mae = sum_ae * c->inv_num_agents;
        stats[(size_t)HNS_ST_ACTION_ERROR_ORDER1_MEAN * S] += mae;
        if (mae > stats[(size_t)HNS_ST_ACTION_ERROR_ORDER1_MAX * S]) stats[(size_t)HNS_ST_ACTION_ERROR_ORDER1_MAX * S] = mae;
        /* A6 on S_t */
        float dpos[HNS_MAX_AGENTS * 3];
        for (int a = 0; a < A; ++a) for (int i = 0; i < 3; ++i) dpos[3 * a + i] = ds[13 * a + i];
        float force[3], tvel[3];
        o_prey(c, A, C, dpos, tp, cyl, force, tvel, &stats[(size_t)HNS_ST_OUT_OF_ARENA * S]);
        float force1[3], tvel1[3] = {0.0f, 0.0f, 0.0f};
        if (NT == 2) o_prey(c, A, C, dpos, tp + 3, cyl, force1, tvel1, &stats[(size_t)HNS_ST_OUT_OF_ARENA * S]);
        /* A4 forces/torques on S_t, then A5 */
        float fw[HNS_MAX_AGENTS][3], tb[HNS_MAX_AGENTS][3];
        for (int a = 0; a < A; ++a) {
            float f[3] = {0.0f, 0.0f, 0.0f};
            int first = 1;
            for (int j = 0; j < A; ++j) {
                if (j == a) continue;
                float fj[3];
                o_downwash_pair(dpos + 3 * a, dpos + 3 * j, tw[j], fj);
                for (int i = 0; i < 3; ++i) f[i] = first ? fj[i] : f[i] + fj[i];
                first = 0;
            }
            for (int i = 0; i < 3; ++i) fw[a][i] = tw[a][i] + f[i];
            const float *T = thrust[a];
            tb[a][0] = ((c->rotor_py[0] * T[0] + c->rotor_py[1] * T[1]) + c->rotor_py[2] * T[2]) + c->rotor_py[3] * T[3];
            tb[a][1] = -(((c->rotor_px[0] * T[0] + c->rotor_px[1] * T[1]) + c->rotor_px[2] * T[2]) + c->rotor_px[3] * T[3]);
            tb[a][2] = ((moment[a][0] + moment[a][1]) + moment[a][2]) + moment[a][3];
        }
        for (int a = 0; a < A; ++a) o_integrate(c, ds + 13 * a, fw[a], tb[a]);
        uint32_t bad = 0;                                   /* failure-detection word (include/hns.h: hns_buffers.nonfinite) */
        for (int a = 0; a < A; ++a) {
            float s = ds[13 * a];
            for (int i = 1; i < 13; ++i) s = s + ds[13 * a + i];
            if ((s - s) != 0.0f) bad |= 1u;
        }
        for (int i = 0; i < 3; ++i) {
            b->target_vel[(size_t)e * 3 * NT + i] = tvel[i];
            tp[i] = tp[i] + tvel[i] * c->dt;
            if (NT == 2) {
                b->target_vel[(size_t)e * 3 * NT + 3 + i] = tvel1[i];
                tp[3 + i] = tp[3 + i] + tvel1[i] * c->dt;
            }
        }
        b->progress[e] += 1.0f;
        o_obs_side side;
        o_obs(c, A, C, K, ds, tp, cyl, b->progress[e], b->obs_self + (size_t)e * A * SD,
              b->obs_others + (size_t)e * A * (A - 1) * 3, b->obs_cylinders + (size_t)e * A * K * 5,
              (c->write_critic_state && b->state_drones) ? b->state_drones + (size_t)e * A * SD : NULL, &side);
        /* include/hns.h: the line-of-sight flags of S_{t+1} ride in the spare column of the controller record.  The oracle only WRITES
         * them — its evader policy above tests the line of sight afresh every step, as the reference does (hideandseek.py:1080) — so the
         * parity tests check the kernel's carried-over flag against an independent evaluation. */
        for (int a = 0; a < A; ++a) b->pid_last_rate[((size_t)e * A + a) * 4 + 3] = (float)(side.blocked[a] + 2 * side.blocked1[a]);
        o_reward(c, A, C, K, ds, tp, cyl, b->progress[e], &side, b->action_error + (size_t)e * A, thr_diff,
                 stats, S, b->reward + (size_t)e * A, b->done + e);
        if (b->detect) b->detect[e] = (uint8_t)(side.bdetect | (side.bdetect1 << 1));
        for (int k = 0; k < NT; ++k) {
            float s = (tp[3 * k] + tp[3 * k + 1]) + tp[3 * k + 2];
            if ((s - s) != 0.0f) bad |= 2u;
        }
        {
            const float *r = b->reward + (size_t)e * A;
            float s = r[0];
            for (int a = 1; a < A; ++a) s = s + r[a];
            if ((s - s) != 0.0f) bad |= 4u;
        }
        if (bad && b->nonfinite) {
#pragma omp atomic
            *b->nonfinite |= bad;
        }
    }
    return HNS_OK;
}

/* ------------------------------------------------------------------------------------------
 * Reset (A11)   hideandseek.py:576-723, multirotor.py:635-650
 * Random numbers: Philox4x32-10, key = seed, counter = (global env, epoch, draw block, 0);
 * uniform = (u32 >> 8) * 2^-24.  torch's RNG stream cannot be reproduced, so the
 * reference is followed in distribution and in every deterministic step.
 * ---------------------------------------------------------------------------------------- */
static void o_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct o_rng { uint32_t k0, k1, env, epoch, block; uint32_t buf[4]; int have; } o_rng;
static float o_uniform(o_rng *r) {
    if (r->have == 0) { o_philox(r->k0, r->k1, r->env, r->epoch, r->block++, 0u, r->buf); r->have = 4; }
    uint32_t u = r->buf[4 - r->have];
    r->have--;
    return (float)(u >> 8) * 5.9604644775390625e-8f;
}

/* continuous_to_grid :143-164 : round-half-even(offset/grid) + center, clamped */
static int o_cell(const hns_cfg *c, float x) {
    int g = (int)rintf(x / c->grid_size) + c->grid_num / 2;
    return g < 0 ? 0 : (g > c->grid_num - 1 ? c->grid_num - 1 : g);
}

static int o_reset_impl(const hns_cfg *c, const hns_buffers *b, const uint8_t *mask, uint64_t seed, uint32_t epoch,
                        float *tasks, int task_first);
int hns_oracle_reset(const hns_cfg *c, const hns_buffers *b, const uint8_t *mask, uint64_t seed, uint32_t epoch) {
    return o_reset_impl(c, b, mask, seed, epoch, NULL, 0);
}
/* envgen: placement of envs >= task_first from task vectors (hideandseek_envgen.py:896-898) */
int hns_oracle_reset_tasks(const hns_cfg *c, const hns_buffers *b, const uint8_t *mask, uint64_t seed, uint32_t epoch,
                           float *tasks, int task_first) {
    return o_reset_impl(c, b, mask, seed, epoch, tasks, task_first);
}
static int o_reset_impl(const hns_cfg *c, const hns_buffers *b, const uint8_t *mask, uint64_t seed, uint32_t epoch,
                        float *tasks, int task_first) {
    const int E = c->num_envs, A = c->num_agents, C = c->num_cylinders, K = c->obs_max_cylinder, G = c->grid_num;
    const size_t S = c->stats_stride ? (size_t)c->stats_stride : (size_t)E;
    if (G > 16) return HNS_ERR_INVALID_ARG;
    for (int e = 0; e < E; ++e) {
        /* :712 sets first_capture_step for ALL envs on any reset call */
        b->stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * S + e] = (float)c->max_episode_length;
        const int masked = !(mask && !mask[e]);
        if (!masked && !c->reset_extra_step) continue;
        o_rng rng = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(e + c->env_index_offset), epoch, 0u, {0, 0, 0, 0}, 0};
        float *ds = b->drone_state + (size_t)e * A * 13;
        const int NT = o_nt(c), SD = o_self_dim(c);
        float *tp = b->target_pos + (size_t)e * 3 * NT;
        float *cyl = b->cylinders + (size_t)e * C * 3;
        const float *task = (tasks && e >= task_first) ? tasks + (size_t)e * (3 * A + 3 * NT + 3 * C) : NULL;
        if (masked) {
        for (int a = 0; a < A; ++a) {
            float *d = ds + 13 * a;
            if (task) {
                d[0] = task[3 * a]; d[1] = task[3 * a + 1];
            } else if (c->init_mode == HNS_INIT_RANDOM) {
                d[0] = c->drone_xy_lo[0] + o_uniform(&rng) * (c->drone_xy_hi[0] - c->drone_xy_lo[0]);
                d[1] = c->drone_xy_lo[1] + o_uniform(&rng) * (c->drone_xy_hi[1] - c->drone_xy_lo[1]);
            } else {
                d[0] = c->fixed_drone_pos[a][0]; d[1] = c->fixed_drone_pos[a][1];
            }
            if (task) d[2] = task[3 * a + 2];
            else if (c->init_mode == HNS_INIT_SCENARIO) d[2] = c->fixed_drone_pos[a][2];
            else d[2] = c->z_lo + o_uniform(&rng) * (c->z_hi - c->z_lo);
            float rpy[3];
            for (int i = 0; i < 3; ++i) rpy[i] = c->rpy_lo[i] + o_uniform(&rng) * (c->rpy_hi[i] - c->rpy_lo[i]);
            o_euler_to_quat(rpy, d + 3);
            for (int i = 7; i < 13; ++i) d[i] = 0.0f;
            size_t ia = (size_t)e * A + a;
            float pa = 0.0f;
            for (int i = 0; i < 4; ++i) {
                b->throttle[ia * 4 + i] = c->hover_throttle;
                if (c->pid_reset_on_reset) {      /* 0 = the reference: `_reset_idx` never touches the controller (cleared through reset_pid) */
                    b->pid_integ[ia * 4 + i] = 0.0f;
                    b->pid_last_rate[ia * 4 + i] = 0.0f;
                }
                float thr = c->hover_throttle;
                float ci = 0.5f * (c->max_thrust_ratio + (2.0f * (thr * thr) - 1.0f));   /* :714-716 */
                pa = (i == 0) ? ci : pa + ci;
            }
            b->prev_action[ia * 4 + 3] = pa / 4.0f;       /* components 0..2 are NOT reset (:716) */
        }
        if (task) {
            for (int i = 0; i < 3; ++i) tp[i] = task[3 * A + i];
        } else {
            if (c->init_mode == HNS_INIT_RANDOM) {
                tp[0] = c->target_xy_lo[0] + o_uniform(&rng) * (c->target_xy_hi[0] - c->target_xy_lo[0]);
                tp[1] = c->target_xy_lo[1] + o_uniform(&rng) * (c->target_xy_hi[1] - c->target_xy_lo[1]);
            } else {
                tp[0] = c->fixed_target_pos[0]; tp[1] = c->fixed_target_pos[1];
            }
            if (c->init_mode == HNS_INIT_SCENARIO) tp[2] = c->fixed_target_pos[2];
            else tp[2] = c->z_lo + o_uniform(&rng) * (c->z_hi - c->z_lo);
        }
        if (NT == 2) {   /* second evader: same box, its draws follow the first evader's */
            if (task) {
                for (int i = 0; i < 3; ++i) tp[3 + i] = task[3 * A + 3 + i];
            } else {
                if (c->init_mode == HNS_INIT_RANDOM) {
                    tp[3] = c->target_xy_lo[0] + o_uniform(&rng) * (c->target_xy_hi[0] - c->target_xy_lo[0]);
                    tp[4] = c->target_xy_lo[1] + o_uniform(&rng) * (c->target_xy_hi[1] - c->target_xy_lo[1]);
                } else {
                    tp[3] = c->fixed_target_pos[0]; tp[4] = -c->fixed_target_pos[1];
                }
                if (c->init_mode == HNS_INIT_SCENARIO) tp[5] = c->fixed_target_pos[2];
                else tp[5] = c->z_lo + o_uniform(&rng) * (c->z_hi - c->z_lo);
            }
        }
        if (task) {
            for (int k = 0; k < 3 * C; ++k) cyl[k] = task[3 * A + 3 * NT + k];
        } else if (c->init_mode == HNS_INIT_SCENARIO) {
            for (int k = 0; k < C; ++k) {
                for (int i = 0; i < 3; ++i) cyl[3 * k + i] = c->fixed_cyl_pos[k][i];
                if (k >= c->fixed_cyl_active) cyl[3 * k + 2] = c->invalid_z;
            }
        } else {
            /* rejection_sampling_random_cylinder :576-607 */
            uint8_t occ[16 * 16];
            int half = G / 2;
            for (int i = 0; i < G; ++i)
                for (int j = 0; j < G; ++j) {
                    float dd = sqrtf((float)((i - half) * (i - half) + (j - half) * (j - half)));
                    occ[i * G + j] = dd >= (float)half;       /* set_outside_circle_to_one :168-181 */
                }
            for (int a = 0; a < A; ++a) occ[o_cell(c, ds[13 * a]) * G + o_cell(c, ds[13 * a + 1])] = 1;
            occ[o_cell(c, tp[0]) * G + o_cell(c, tp[1])] = 1;
            if (NT == 2) occ[o_cell(c, tp[3]) * G + o_cell(c, tp[4])] = 1;
            int n_active;
            if (c->cyl_fixed_num >= 0) n_active = c->cyl_fixed_num;
            else {
                int span = C + 1 - c->cyl_min_num;
                int r = (int)(o_uniform(&rng) * (float)span);
                if (r > span - 1) r = span - 1;
                n_active = c->cyl_min_num + r;
            }
            uint8_t freec[256];
            int nfree = 0;
            for (int i = 0; i < G * G; ++i) if (!occ[i]) freec[nfree++] = (uint8_t)i;   /* torch.nonzero order */
            if (nfree < C) return HNS_ERR_CONFIG;
            for (int k = 0; k < C; ++k) {   /* randperm[:C] as a partial Fisher-Yates */
                int span = nfree - k;
                int j = (int)(o_uniform(&rng) * (float)span);
                if (j > span - 1) j = span - 1;
                j += k;
                uint8_t t = freec[k]; freec[k] = freec[j]; freec[j] = t;
                int gx = freec[k] / G, gy = freec[k] % G;
                float x = 0.0f + (float)(gx - half) * c->grid_size, y = 0.0f + (float)(gy - half) * c->grid_size;
                cyl[3 * k + 0] = o_clamp(x, -c->boundary, c->boundary);   /* grid_to_continuous :121-141 */
                cyl[3 * k + 1] = o_clamp(y, -c->boundary, c->boundary);
                cyl[3 * k + 2] = (k >= n_active) ? c->invalid_z : 0.5f * c->cylinder_height;
            }
        }
        if (tasks && !task) {
            /* a uniformly sampled task is archived as sampled: hideandseek_envgen.py:883-895 inserts `tasks_unif` (the sampled placement)
             * into the generator, the scene's sim.step comes after (:1013) */
            float *row = tasks + (size_t)e * (3 * A + 3 * NT + 3 * C);
            for (int a = 0; a < A; ++a) for (int i = 0; i < 3; ++i) row[3 * a + i] = ds[13 * a + i];
            for (int i = 0; i < 3 * NT; ++i) row[3 * A + i] = tp[i];
            for (int k = 0; k < 3 * C; ++k) row[3 * A + 3 * NT + k] = cyl[k];
        }
        for (int s = 0; s < HNS_NUM_STATS; ++s) b->stats[(size_t)s * S + e] = 0.0f;
        b->stats[(size_t)HNS_ST_FIRST_CAPTURE_STEP * S + e] = (float)c->max_episode_length;
        b->progress[e] = 0.0f;
        b->done[e] = 0;
        }   /* masked */
        if (c->reset_extra_step) {
            /* hideandseek.py:722-723: `_reset_idx` ends with one sim.step() of the WHOLE scene — no rotor force (apply_action runs in
             * _pre_sim_step only): every drone integrates one dt under gravity and damping, every evader (gravity disabled, :544-565) moves
             * one dt with the velocity it holds; envs that are not being reset included */
            const float zero[3] = {0.0f, 0.0f, 0.0f};
            for (int a = 0; a < A; ++a) o_integrate(c, ds + 13 * a, zero, zero);
            for (int i = 0; i < 3 * NT; ++i) tp[i] = tp[i] + b->target_vel[(size_t)e * 3 * NT + i] * c->dt;
        }
        o_obs_side side;
        o_obs(c, A, C, K, ds, tp, cyl, b->progress[e], b->obs_self + (size_t)e * A * SD,
              b->obs_others + (size_t)e * A * (A - 1) * 3, b->obs_cylinders + (size_t)e * A * K * 5,
              (c->write_critic_state && b->state_drones) ? b->state_drones + (size_t)e * A * SD : NULL, &side);
        for (int a = 0; a < A; ++a) b->pid_last_rate[((size_t)e * A + a) * 4 + 3] = (float)(side.blocked[a] + 2 * side.blocked1[a]);
        if (b->detect) b->detect[e] = (uint8_t)(side.bdetect | (side.bdetect1 << 1));
    }
    return HNS_OK;
}

/* ------------------------------------------------------------------------------------------
 * Stage-level entry points for the golden-vector tests (tests/test_oracle_golden.py)
 * ---------------------------------------------------------------------------------------- */
void hns_oracle_quat_rotate(int n, const float *q, const float *v, float *out, int inverse) {
    for (int i = 0; i < n; ++i) o_quat_rot(q + 4 * i, v + 3 * i, out + 3 * i, inverse);
}
void hns_oracle_euler_to_quat(int n, const float *rpy, float *q) {
    for (int i = 0; i < n; ++i) o_euler_to_quat(rpy + 3 * i, q + 4 * i);
}
void hns_oracle_elementary(int n, const float *x, float *e, float *t, float *s, float *co) {
    for (int i = 0; i < n; ++i) { e[i] = o_expf(x[i]); t[i] = o_tanhf(x[i]); o_sincosf(x[i], s + i, co + i); }
}
void hns_oracle_rotor(const hns_cfg *c, int n, const float *cmd, float *throttle, float *thrust, float *moment, float *thr_diff) {
    for (int i = 0; i < n; ++i) o_rotor(c, cmd + 4 * i, throttle + 4 * i, thrust + 4 * i, moment + 4 * i, thr_diff + i);
}
/* integ/last are [n,3] here (reference layout); reset_mask zeroes PID state first (lee_position_controller.py:497-502) */
void hns_oracle_ctbr_pid(const hns_cfg *c, int n, const float *action, const float *rot, const float *angvel,
                         const uint8_t *reset_mask, float *prev_action, float *integ, float *last, float *cmd,
                         float *aerr, float *ctbr, float *target_rate) {
    for (int i = 0; i < n; ++i) {
        if (reset_mask && reset_mask[i]) for (int k = 0; k < 3; ++k) { integ[3 * i + k] = 0.0f; last[3 * i + k] = 0.0f; }
        o_ctbr_pid(c, action + 4 * i, rot + 4 * i, angvel + 3 * i, prev_action + 4 * i, integ + 3 * i, last + 3 * i,
                   cmd + 4 * i, aerr + i, ctbr + 4 * i, target_rate + 3 * i);
    }
}
/* total downwash force on each drone: pos [E,A,3], rot [E,A,4], tsum [E,A] -> f [E,A,3] */
void hns_oracle_downwash(int E, int A, const float *pos, const float *rot, const float *tsum, float *f) {
    for (int e = 0; e < E; ++e) {
        float tw[HNS_MAX_AGENTS][3];
        for (int a = 0; a < A; ++a) o_quat_rot_z(rot + (e * A + a) * 4, tsum[e * A + a], tw[a]);
        for (int a = 0; a < A; ++a) {
            float acc[3] = {0, 0, 0};
            int first = 1;
            for (int j = 0; j < A; ++j) {
                if (j == a) continue;
                float fj[3];
                o_downwash_pair(pos + (e * A + a) * 3, pos + (e * A + j) * 3, tw[j], fj);
                for (int i = 0; i < 3; ++i) acc[i] = first ? fj[i] : acc[i] + fj[i];
                first = 0;
            }
            for (int i = 0; i < 3; ++i) f[(e * A + a) * 3 + i] = acc[i];
        }
    }
}
void hns_oracle_blocked(const hns_cfg *c, int E, int A, int C, const float *dpos, const float *tpos, const float *cyl, uint8_t *out) {
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < A; ++a)
            out[e * A + a] = (uint8_t)o_blocked(c, C, dpos + (e * A + a) * 3, tpos + e * 3, cyl + e * C * 3);
}
void hns_oracle_prey(const hns_cfg *c, int E, int A, int C, const float *dpos, const float *tpos, const float *cyl,
                     float *force, float *vel, float *out_of_arena) {
    for (int e = 0; e < E; ++e)
        o_prey(c, A, C, dpos + e * A * 3, tpos + e * 3, cyl + e * C * 3, force + e * 3, vel + e * 3, out_of_arena + e);
}
void hns_oracle_integrate(const hns_cfg *c, int n, float *ds, const float *force_w, const float *torque_b) {
    for (int i = 0; i < n; ++i) o_integrate(c, ds + 13 * i, force_w + 3 * i, torque_b + 3 * i);
}
/* obs + reward on a given post-physics state (no dynamics): fills outputs and side masks */
void hns_oracle_obs_reward(const hns_cfg *c, const hns_buffers *b, const float *thr_diff, int do_reward,
                           uint8_t *blocked /*[E,A]*/, uint8_t *bdetect /*[E]*/, uint8_t *knn_mask /*[E,A,K]*/) {
    const int E = c->num_envs, A = c->num_agents, C = c->num_cylinders, K = c->obs_max_cylinder;
    for (int e = 0; e < E; ++e) {
        o_obs_side side;
        const float *ds = b->drone_state + (size_t)e * A * 13;
        const float *tp = b->target_pos + (size_t)e * 3;
        const float *cyl = b->cylinders + (size_t)e * C * 3;
        o_obs(c, A, C, K, ds, tp, cyl, b->progress[e], b->obs_self + (size_t)e * A * HNS_SELF_DIM,
              b->obs_others + (size_t)e * A * (A - 1) * 3, b->obs_cylinders + (size_t)e * A * K * 5,
              b->state_drones ? b->state_drones + (size_t)e * A * HNS_SELF_DIM : NULL, &side);
        for (int a = 0; a < A; ++a) {
            blocked[e * A + a] = (uint8_t)side.blocked[a];
            for (int s = 0; s < K; ++s) knn_mask[(e * A + a) * K + s] = (uint8_t)side.knn_masked[a][s];
        }
        bdetect[e] = (uint8_t)side.bdetect;
        if (do_reward)
            o_reward(c, A, C, K, ds, tp, cyl, b->progress[e], &side, b->action_error + (size_t)e * A,
                     thr_diff + (size_t)e * A, b->stats + e, c->stats_stride ? (size_t)c->stats_stride : (size_t)E, b->reward + (size_t)e * A, b->done + e);
    }
}
int hns_oracle_cell(const hns_cfg *c, float x) { return o_cell(c, x); }
void hns_oracle_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out) {
    o_philox(k0, k1, c0, c1, c2, c3, out);
}
size_t hns_oracle_cfg_size(void) { return sizeof(hns_cfg); }

/* ------------------------------------------------------------------------------------------
 * Hover task (BASELINE config 1)   omni_drones/envs/single/hover.py:322-523
 * stats rows: hover.py:238-278 order; acc rows: linear_v/angular_v/linear_a/angular_a/
 * linear_jerk/angular_jerk _episode, then last_ of the same six (hover.py:150-155,313-320)
 * ---------------------------------------------------------------------------------------- */
enum { HS_RETURN = 0, HS_POS_BONUS, HS_HEAD_BONUS, HS_REWARD_POS, HS_REWARD_UP, HS_REWARD_VEL, HS_REWARD_ACC, HS_REWARD_JERK,
       HS_EPISODE_LEN, HS_POS_ERROR, HS_HEADING_ALIGNMENT, HS_UPRIGHTNESS, HS_ACTION_SMOOTHNESS, HS_LINEAR_V_MAX,
       HS_ANGULAR_V_MAX, HS_LINEAR_A_MAX, HS_ANGULAR_A_MAX, HS_LINEAR_JERK_MAX, HS_ANGULAR_JERK_MAX, HS_LINEAR_V_MEAN,
       HS_ANGULAR_V_MEAN, HS_LINEAR_A_MEAN, HS_ANGULAR_A_MEAN, HS_LINEAR_JERK_MEAN, HS_ANGULAR_JERK_MEAN, HS_MOTOR1,
       HS_MOTOR2, HS_MOTOR3, HS_MOTOR4, HS_CMD_R, HS_CMD_P, HS_CMD_Y, HS_CMD_THRUST, HS_TARGET_R_RATE, HS_TARGET_P_RATE,
       HS_TARGET_Y_RATE, HS_REAL_R_RATE, HS_REAL_P_RATE, HS_REAL_Y_RATE };
enum { HA_LV_EP = 0, HA_AV_EP, HA_LA_EP, HA_AA_EP, HA_LJ_EP, HA_AJ_EP, HA_LAST_LV, HA_LAST_AV, HA_LAST_LA, HA_LAST_AA,
       HA_LAST_LJ, HA_LAST_AJ };

/* _compute_state_and_obs (hover.py:361-437) for one env; returns linear_v/linear_a/linear_jerk for the reward */
static void o_hover_obs(const hns_cfg *c, const hns_hover_cfg *h, const float ds[13], float progress, float *st, float *ac,
                        size_t E, float obs[20], float heading[3], float up[3], float *lin_v, float *lin_a, float *lin_j) {
#define ST(i) st[(size_t)(i) * E]
#define AC(i) ac[(size_t)(i) * E]
    float br[3];
    o_quat_rot(ds + 3, ds + 10, br, 1);
    ST(HS_REAL_R_RATE) = (br[0] * 180.0f) * O_INV_PI;
    ST(HS_REAL_P_RATE) = (br[1] * 180.0f) * O_INV_PI;
    ST(HS_REAL_Y_RATE) = (br[2] * 180.0f) * O_INV_PI;
    o_quat_rot_x(ds + 3, heading);
    o_quat_rot_z(ds + 3, 1.0f, up);
    for (int i = 0; i < 3; ++i) obs[i] = h->target_pos[i] - ds[i];
    for (int i = 0; i < 7; ++i) obs[3 + i] = ds[3 + i];
    for (int i = 0; i < 3; ++i) { obs[10 + i] = heading[i]; obs[13 + i] = up[i]; }
    float t = progress * c->inv_max_episode_length;
    for (int i = 0; i < 4; ++i) obs[16 + i] = t;
    float lv = o_norm3(ds[7], ds[8], ds[9]), av = o_norm3(ds[10], ds[11], ds[12]);
    float n = progress + 1.0f;
    if (fabsf(lv) > ST(HS_LINEAR_V_MAX)) ST(HS_LINEAR_V_MAX) = fabsf(lv);
    AC(HA_LV_EP) += fabsf(lv); ST(HS_LINEAR_V_MEAN) = AC(HA_LV_EP) / n;
    if (fabsf(av) > ST(HS_ANGULAR_V_MAX)) ST(HS_ANGULAR_V_MAX) = fabsf(av);
    AC(HA_AV_EP) += fabsf(av); ST(HS_ANGULAR_V_MEAN) = AC(HA_AV_EP) / n;
    float la = fabsf(lv - AC(HA_LAST_LV)) / c->dt, aa = fabsf(av - AC(HA_LAST_AV)) / c->dt;
    if (fabsf(la) > ST(HS_LINEAR_A_MAX)) ST(HS_LINEAR_A_MAX) = fabsf(la);
    AC(HA_LA_EP) += fabsf(la); ST(HS_LINEAR_A_MEAN) = AC(HA_LA_EP) / n;
    if (fabsf(aa) > ST(HS_ANGULAR_A_MAX)) ST(HS_ANGULAR_A_MAX) = fabsf(aa);
    AC(HA_AA_EP) += fabsf(aa); ST(HS_ANGULAR_A_MEAN) = AC(HA_AA_EP) / n;
    float lj = fabsf(la - AC(HA_LAST_LA)) / c->dt, aj = fabsf(aa - AC(HA_LAST_AA)) / c->dt;
    if (fabsf(lj) > ST(HS_LINEAR_JERK_MAX)) ST(HS_LINEAR_JERK_MAX) = fabsf(lj);
    AC(HA_LJ_EP) += fabsf(lj); ST(HS_LINEAR_JERK_MEAN) = AC(HA_LJ_EP) / n;
    if (fabsf(aj) > ST(HS_ANGULAR_JERK_MAX)) ST(HS_ANGULAR_JERK_MAX) = fabsf(aj);
    AC(HA_AJ_EP) += fabsf(aj); ST(HS_ANGULAR_JERK_MEAN) = AC(HA_AJ_EP) / n;
    AC(HA_LAST_LV) = lv; AC(HA_LAST_AV) = av; AC(HA_LAST_LA) = la; AC(HA_LAST_AA) = aa; AC(HA_LAST_LJ) = lj; AC(HA_LAST_AJ) = aj;
    *lin_v = lv; *lin_a = la; *lin_j = lj;
#undef ST
#undef AC
}

int hns_oracle_hover_step(const hns_cfg *c, const hns_hover_cfg *h, const hns_hover_buffers *b, const float *action) {
    const size_t E = (size_t)c->num_envs;
    for (size_t e = 0; e < E; ++e) {
        float *ds = b->drone_state + e * 13, *st = b->stats + e, *ac = b->acc + e;
#define ST(i) st[(size_t)(i) * E]
        float cmd[4], aerr, ctbr[4], trate[3], thrust[4], moment[4], td;
        o_ctbr_pid(c, action + e * 4, ds + 3, ds + 10, b->prev_action + e * 4, b->pid_integ + e * 4, b->pid_last_rate + e * 4,
                   cmd, &aerr, ctbr, trate);
        ST(HS_MOTOR1) = cmd[0]; ST(HS_MOTOR2) = cmd[1]; ST(HS_MOTOR3) = cmd[2]; ST(HS_MOTOR4) = cmd[3];   /* hover.py:326-329 */
        o_rotor(c, cmd, b->throttle + e * 4, thrust, moment, &td);
        ST(HS_CMD_R) = ctbr[0]; ST(HS_CMD_P) = ctbr[1]; ST(HS_CMD_Y) = ctbr[2]; ST(HS_CMD_THRUST) = ctbr[3];   /* :334-338 */
        ST(HS_TARGET_R_RATE) = trate[0]; ST(HS_TARGET_P_RATE) = trate[1]; ST(HS_TARGET_Y_RATE) = trate[2];   /* :341-344 */
        float ts = ((thrust[0] + thrust[1]) + thrust[2]) + thrust[3], fw[3], tb[3];
        o_quat_rot_z(ds + 3, ts, fw);
        tb[0] = ((c->rotor_py[0] * thrust[0] + c->rotor_py[1] * thrust[1]) + c->rotor_py[2] * thrust[2]) + c->rotor_py[3] * thrust[3];
        tb[1] = -(((c->rotor_px[0] * thrust[0] + c->rotor_px[1] * thrust[1]) + c->rotor_px[2] * thrust[2]) + c->rotor_px[3] * thrust[3]);
        tb[2] = ((moment[0] + moment[1]) + moment[2]) + moment[3];
        o_integrate(c, ds, fw, tb);
        b->progress[e] += 1.0f;
        float progress = b->progress[e], heading[3], up[3], lv, la, lj;
        o_hover_obs(c, h, ds, progress, st, ac, E, b->obs + e * 20, heading, up, &lv, &la, &lj);
        /* _compute_reward_and_done  hover.py:439-523 */
        const float *obs = b->obs + e * 20;
        float pos_error = o_norm3(obs[0], obs[1], obs[2]);
        float rh[3] = {h->target_heading[0] - heading[0], h->target_heading[1] - heading[1], h->target_heading[2] - heading[2]};
        float head_error = o_norm3(rh[0], rh[1], rh[2]);
        float align = (heading[0] * h->target_heading[0] + heading[1] * h->target_heading[1]) + heading[2] * h->target_heading[2];
        float reward_pos = -pos_error * h->reward_distance_scale;
        float bonus = (pos_error <= 0.02f) ? 10.0f : 0.0f;
        float bpos = bonus > 0.0f ? 1.0f : 0.0f;
        float reward_head = -head_error * bpos;
        float head_bonus = ((head_error <= 0.02f) ? 10.0f : 0.0f) * bpos;
        float u = (up[2] + 1.0f) / 2.0f;
        float reward_up = u * u;
        float reward_v = (h->reward_v_scale * bpos) * ((lv < h->linear_vel_max) ? 1.0f : 0.0f);
        float reward_acc = (h->reward_acc_scale * bpos) * ((la < h->linear_acc_max) ? 1.0f : 0.0f);
        float reward_jerk = (h->reward_jerk_scale * bpos) * -lj;
        float reward = ((((((reward_pos + bonus) + reward_head) + head_bonus) + reward_up) + reward_v) + reward_acc) + reward_jerk;
        b->reward[e] = reward;
        b->done[e] = (uint8_t)(progress >= (float)c->max_episode_length);
        float w = 1.0f - h->alpha;
        ST(HS_POS_ERROR) += w * (pos_error - ST(HS_POS_ERROR));                 /* lerp_  :506-509 */
        ST(HS_HEADING_ALIGNMENT) += w * (align - ST(HS_HEADING_ALIGNMENT));
        ST(HS_UPRIGHTNESS) += w * (up[2] - ST(HS_UPRIGHTNESS));
        ST(HS_ACTION_SMOOTHNESS) += w * (-td - ST(HS_ACTION_SMOOTHNESS));
        ST(HS_RETURN) += reward;
        ST(HS_REWARD_POS) = reward_pos; ST(HS_POS_BONUS) = bonus; ST(HS_HEAD_BONUS) = head_bonus;
        ST(HS_REWARD_VEL) = reward_v; ST(HS_REWARD_ACC) = reward_acc; ST(HS_REWARD_JERK) = reward_jerk;
        ST(HS_EPISODE_LEN) = progress;
#undef ST
    }
    return HNS_OK;
}

int hns_oracle_hover_reset(const hns_cfg *c, const hns_hover_cfg *h, const hns_hover_buffers *b, const uint8_t *mask,
                           uint64_t seed, uint32_t epoch) {
    const size_t E = (size_t)c->num_envs;
    for (size_t e = 0; e < E; ++e) {
        /* hover.py:313-320 re-creates the six *_episode accumulators for ALL envs on any reset call */
        for (int i = HA_LV_EP; i <= HA_AJ_EP; ++i) b->acc[(size_t)i * E + e] = 0.0f;
        if (mask && !mask[e]) continue;
        o_rng rng = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(e + c->env_index_offset), epoch, 0u, {0, 0, 0, 0}, 0};
        float *ds = b->drone_state + e * 13;
        for (int i = 0; i < 3; ++i) ds[i] = h->pos_lo[i] + o_uniform(&rng) * (h->pos_hi[i] - h->pos_lo[i]);
        float rpy[3];
        for (int i = 0; i < 3; ++i) rpy[i] = h->rpy_lo[i] + o_uniform(&rng) * (h->rpy_hi[i] - h->rpy_lo[i]);
        o_euler_to_quat(rpy, ds + 3);
        for (int i = 7; i < 13; ++i) ds[i] = 0.0f;
        for (int i = 0; i < 4; ++i) {
            b->throttle[e * 4 + i] = c->hover_throttle;       /* multirotor.py:647-648 */
            b->pid_integ[e * 4 + i] = 0.0f;
            b->pid_last_rate[e * 4 + i] = 0.0f;
        }
        for (int i = 0; i < HNS_HOVER_NUM_STATS; ++i) b->stats[(size_t)i * E + e] = 0.0f;
        for (int i = HA_LAST_LV; i <= HA_LAST_AJ; ++i) b->acc[(size_t)i * E + e] = 0.0f;
        b->progress[e] = 0.0f;
        b->done[e] = 0;
        float heading[3], up[3], lv, la, lj;
        o_hover_obs(c, h, ds, 0.0f, b->stats + e, b->acc + e, E, b->obs + e * 20, heading, up, &lv, &la, &lj);
    }
    return HNS_OK;
}

/* ------------------------------------------------------------------------------------------
 * Extension (NOT in the reference, which only has the line-of-sight test and the k-nearest
 * selection): planar ray-fan range sensor.  For pursuer (e,a), ray r points along
 * the drone's horizontal heading rotated by 2*pi*r/N in the world xy-plane; the range is the
 * distance to the first active cylinder (radius cylinder_size) or to the arena wall (radius
 * arena_size, seen from inside), clamped to [0, max_range].
 * ---------------------------------------------------------------------------------------- */
void hns_oracle_raycast(const hns_cfg *c, const hns_buffers *b, int N, float max_range, float *out) {
    const int E = c->num_envs, A = c->num_agents, C = c->num_cylinders;
    const float step = 6.283185307179586f / (float)N;
    for (int e = 0; e < E; ++e) {
        const float *cyl = b->cylinders + (size_t)e * C * 3;
        for (int a = 0; a < A; ++a) {
            const float *ds = b->drone_state + ((size_t)e * A + a) * 13;
            float h[3];
            o_quat_rot_x(ds + 3, h);
            float hn = o_norm2(h[0], h[1]);
            float ux0 = hn > 1e-6f ? h[0] / hn : 1.0f, uy0 = hn > 1e-6f ? h[1] / hn : 0.0f;
            float ox = ds[0], oy = ds[1];
            float oo = O_FMA(oy, oy, ox * ox);
            for (int r = 0; r < N; ++r) {
                float sn, cs;
                o_sincosf(step * (float)r, &sn, &cs);
                float ux = O_FMA(ux0, cs, -(uy0 * sn)), uy = O_FMA(ux0, sn, uy0 * cs);
                /* arena wall from inside: t = -o.u + sqrt((o.u)^2 - (|o|^2 - R^2)) */
                float ou = O_FMA(oy, uy, ox * ux);
                float dw = O_FMA(ou, ou, -(oo - c->arena_sq));
                float best = dw >= 0.0f ? sqrtf(dw) - ou : 0.0f;
                if (!(best >= 0.0f)) best = 0.0f;
                for (int k = 0; k < C; ++k) {
                    const float *cc = cyl + 3 * k;
                    if (!(cc[2] > 0.0f)) continue;
                    float mx = cc[0] - ox, my = cc[1] - oy;
                    float bq = O_FMA(my, uy, mx * ux);
                    float cq = O_FMA(my, my, mx * mx) - c->cylinder_size * c->cylinder_size;
                    float disc = O_FMA(bq, bq, -cq);
                    if (disc >= 0.0f) {
                        float t = bq - sqrtf(disc);
                        if (cq <= 0.0f) t = 0.0f;               /* origin inside the cylinder */
                        if (t >= 0.0f && t < best) best = t;
                    }
                }
                out[((size_t)e * A + a) * N + r] = best > max_range ? max_range : best;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Trajectory predictor in the observation: the `use_TP_net` branch of
 * HideAndSeek._compute_state_and_obs (hideandseek.py:805-854, 871-880) with
 * TP_net.forward (learning/mappo.py:572-589): LSTM(I->64, zero initial state) over the T-frame
 * window, last hidden state -> Linear(64->3F) -> tanh.  torch.nn.LSTM gate order i,f,g,o:
 *   z = W_ih x + b_ih + W_hh h + b_hh;  c' = sig(z_f) c + sig(z_i) tanh(z_g);  h' = sig(z_o) tanh(c')
 * Plain fp32 with libm nonlinearities.  The HIP kernel evaluates the same products on the matrix
 * cores with two-term fp16 splits (fp32-class accuracy, hns_tp.hip) and hardware exp/rcp — compared
 * at the north star's 1e-5.
 * ---------------------------------------------------------------------------------------- */
static inline float o_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

int hns_oracle_tp_observe(const hns_cfg *c, const hns_buffers *b, const hns_tp_buffers *tp, int T, int F, int fill) {
    const int E = c->num_envs, A = c->num_agents, Cn = c->tp_use_obstacles ? c->num_cylinders : 0;
    const int NT = c->num_targets == 2 ? 2 : 1, SD = NT == 2 ? 24 : HNS_SELF_DIM;
    const int I = 7 + 3 * A + 3 * Cn, H = HNS_TP_HIDDEN, R = 3 * F, D = SD + NT * R;
    if (T < 1 || T > 16 || R > 32 || !b->detect) return HNS_ERR_INVALID_ARG;
    if (A > HNS_MAX_AGENTS || Cn > HNS_MAX_CYLINDERS) return HNS_ERR_INVALID_ARG;
#pragma omp parallel for schedule(static) num_threads(g_step_threads)
    for (int e = 0; e < E; ++e) {
      /* two evaders (extension, not in the reference): the same network once per evader — unit u = 2 e + j sees evader j under its
       * own detection bit; window, prediction, ground truth and done flag per unit */
      for (int jt = 0; jt < NT; ++jt) {
        const size_t u = (size_t)e * NT + jt;
        /* frame (:815-820) and window (:825-831) */
        float frame[7 + 3 * HNS_MAX_AGENTS + 3 * HNS_MAX_CYLINDERS];
        const int det = NT == 2 ? (b->detect[e] >> jt) & 1 : b->detect[e] != 0;
        frame[0] = b->progress[e];
        for (int j = 0; j < 3; ++j) {
            frame[1 + j] = det ? b->target_pos[u * 3 + j] : c->mask_value;
            frame[4 + j] = det ? b->target_vel[u * 3 + j] : c->mask_value;
        }
        for (int a = 0; a < A; ++a)
            for (int j = 0; j < 3; ++j) frame[7 + 3 * a + j] = b->drone_state[((size_t)e * A + a) * 13 + j];
        for (int k = 0; k < Cn; ++k) {          /* task.use_obstacles (:808-816): [x, y, cylinder_size] of every slot */
            frame[7 + 3 * A + 3 * k] = b->cylinders[((size_t)e * c->num_cylinders + k) * 3];
            frame[7 + 3 * A + 3 * k + 1] = b->cylinders[((size_t)e * c->num_cylinders + k) * 3 + 1];
            frame[7 + 3 * A + 3 * k + 2] = c->cylinder_size;
        }
        float *hist = tp->history + u * T * I;
        if (fill) {
            for (int t = 0; t < T; ++t) memcpy(hist + (size_t)t * I, frame, sizeof(float) * (size_t)I);
        } else {
            memmove(hist, hist + I, sizeof(float) * (size_t)(T - 1) * I);
            memcpy(hist + (size_t)(T - 1) * I, frame, sizeof(float) * (size_t)I);
        }
        /* LSTM */
        float h[HNS_TP_HIDDEN], cs[HNS_TP_HIDDEN], hn[HNS_TP_HIDDEN];
        for (int v = 0; v < H; ++v) { h[v] = 0.0f; cs[v] = 0.0f; }
        for (int t = 0; t < T; ++t) {
            const float *x = hist + (size_t)t * I;
            for (int v = 0; v < H; ++v) {
                float z[4];
                for (int q = 0; q < 4; ++q) {
                    const int g = q * H + v;
                    float acc = tp->b_ih[g] + tp->b_hh[g];
                    for (int k = 0; k < I; ++k) acc = O_FMA(tp->w_ih[(size_t)g * I + k], x[k], acc);
                    for (int k = 0; k < H; ++k) acc = O_FMA(tp->w_hh[(size_t)g * H + k], h[k], acc);
                    z[q] = acc;
                }
                const float ig = o_sigmoid(z[0]), fg = o_sigmoid(z[1]), gg = tanhf(z[2]), og = o_sigmoid(z[3]);
                const float cn = O_FMA(fg, cs[v], ig * gg);
                cs[v] = cn;
                hn[v] = og * tanhf(cn);
            }
            memcpy(h, hn, sizeof(h));
        }
        /* output layer + rescale (:834-836) */
        float *pr = tp->pred + u * R;
        for (int r = 0; r < R; ++r) {
            float acc = tp->b_fc[r];
            for (int k = 0; k < H; ++k) acc = O_FMA(tp->w_fc[(size_t)r * H + k], h[k], acc);
            const float v = tanhf(acc);
            pr[r] = (r % 3 < 2) ? (v * 0.5f) * c->arena_size : ((v + 1.0f) * 0.5f) * c->max_height;
        }
        /* TP_groundtruth / TP_done (:838-842) */
        const float *tpos = b->target_pos + u * 3;
        tp->groundtruth[u * 3] = tpos[0] * (1.0f / (0.5f * c->arena_size));
        tp->groundtruth[u * 3 + 1] = tpos[1] * (1.0f / (0.5f * c->arena_size));
        tp->groundtruth[u * 3 + 2] = (tpos[2] * (1.0f / c->max_height)) * 2.0f - 1.0f;
        tp->tp_done[u] = (uint8_t)(b->progress[e] <= (float)(c->max_episode_length - F));
      }
        /* rows (:844-854, :873-880); two evaders: [the reference's row for evader 0 | rpos of evader 1, 0 | drone - predicted evader 1] */
        const float *tpos = b->target_pos + (size_t)e * NT * 3, *pr = tp->pred + (size_t)e * NT * R;
        for (int a = 0; a < A; ++a) {
            const size_t ia = (size_t)e * A + a;
            const float *o20 = b->obs_self + ia * SD, *ds = b->drone_state + ia * 13;
            for (int pass = 0; pass < 2; ++pass) {
                float *dst = pass == 0 ? tp->obs_self : tp->state_drones;
                if (!dst) continue;
                float *row = dst + ia * D;
                for (int j = 0; j < 3; ++j) row[j] = pass == 0 ? o20[j] : ds[j] - tpos[j];
                for (int f = 0; f < F; ++f)
                    for (int j = 0; j < 3; ++j) row[3 + 3 * f + j] = ds[j] - pr[3 * f + j];
                for (int j = 3; j < HNS_SELF_DIM; ++j) row[R + j] = o20[j];
                if (NT == 2) {
                    for (int j = 0; j < 3; ++j) row[R + 20 + j] = pass == 0 ? o20[20 + j] : ds[j] - tpos[3 + j];
                    row[R + 23] = o20[23];
                    for (int f = 0; f < F; ++f)
                        for (int j = 0; j < 3; ++j) row[R + 24 + 3 * f + j] = ds[j] - pr[R + 3 * f + j];
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Adaptive Environment Generator: farthest-point sampling as GenBuffer uses it
 * (hideandseek_envgen.py:291-304; DGL's sampler is not vendored and its start point / tie order are
 * unpinned — this is the build's definition: start given, ties -> lower index) and samplenearby
 * (:316-370) with the grid sanity check (:187-207), on the Philox stream of hns_perturb_tasks.
 * ---------------------------------------------------------------------------------------- */
int hns_oracle_fps(const float *points, int n, int d, int k, int start, int32_t *out_idx) {
    if (n < 1 || k < 1 || k > n || start < 0 || start >= n) return HNS_ERR_INVALID_ARG;
    float *dist = (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) dist[i] = INFINITY;
    int cur = start;
    for (int r = 0; r < k; ++r) {
        out_idx[r] = cur;
        if (r == k - 1) break;
        const float *q = points + (size_t)cur * d;
        float bd = -1.0f;
        int bi = 0x7fffffff;
        for (int i = 0; i < n; ++i) {
            const float *x = points + (size_t)i * d;
            float acc = 0.0f;
            for (int c = 0; c < d; ++c) {
                const float df = x[c] - q[c];
                acc = O_FMA(df, df, acc);
            }
            if (acc < dist[i]) dist[i] = acc;
            if (i == cur) dist[i] = -1.0f;                   /* chosen points leave the pool */
            if (dist[i] > bd) { bd = dist[i]; bi = i; }       /* ascending i: ties keep the lower index */
        }
        cur = bi;
    }
    free(dist);
    return 0;
}

/* the reference divides by the Python double 2*cylinder_size (0.2), not by its fp32 rounding: recover the decimal
 * the YAML holds (6 places) so that bodies sitting exactly on a cell edge fall into the same cell */
static double o_envgen_grid_size(const hns_cfg *c) { return rint((double)c->grid_size * 1e6) / 1e6; }
static int o_envgen_cell(double x, double grid_size, int num_grid) {
    int g = (int)rint(x / grid_size) + num_grid / 2;
    return g < 0 ? 0 : (g > num_grid - 1 ? num_grid - 1 : g);
}

/* the grid sanity check alone (hideandseek_envgen.py:187-207): every body in its own free cell of the disc */
void hns_oracle_tasks_sane(const hns_cfg *c, const float *tasks, int n, uint8_t *out) {
    const int A = c->num_agents, Cn = c->num_cylinders, NM = A + (c->num_targets == 2 ? 2 : 1), nb = NM + Cn, TD = 3 * nb, GN = c->grid_num, half = GN / 2;   /* [pursuers | evader(s) | cylinder slots] */
    int cells[HNS_MAX_AGENTS + 2 + HNS_MAX_CYLINDERS];
    for (int t = 0; t < n; ++t) {
        const float *v = tasks + (size_t)t * TD;
        int ok = 1;
        for (int b = 0; b < nb; ++b) {
            const int gx = o_envgen_cell((double)v[3 * b], o_envgen_grid_size(c), GN), gy = o_envgen_cell((double)v[3 * b + 1], o_envgen_grid_size(c), GN);
            const int dx = gx - half, dy = gy - half;
            if (dx * dx + dy * dy >= half * half) ok = 0;
            cells[b] = gx * GN + gy;
        }
        for (int b = 1; b < nb && ok; ++b)
            for (int b2 = 0; b2 < b; ++b2)
                if (cells[b] == cells[b2]) { ok = 0; break; }
        out[t] = (uint8_t)ok;
    }
}

int hns_oracle_perturb_tasks(const hns_cfg *c, const float *history, int n_hist, float *tasks_out, int n_tasks,
                             int expand_cylinders, float expand_step, uint64_t seed) {
    const int A = c->num_agents, Cn = c->num_cylinders, NM = A + (c->num_targets == 2 ? 2 : 1), nb = NM + Cn, TD = 3 * nb, GN = c->grid_num, half = GN / 2;   /* [pursuers | evader(s) | cylinder slots] */
    const double gs = o_envgen_grid_size(c);
    const float cb = (float)((int)(c->arena_size / c->grid_size)) * c->grid_size;
    const float bxy = c->arena_size / 1.41421356237309515f - 0.1f;
    int cells[HNS_MAX_AGENTS + 2 + HNS_MAX_CYLINDERS];
    for (int t = 0; t < n_tasks; ++t) {
        o_rng rng = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)t, 0x9E3779B9u, 0u, {0, 0, 0, 0}, 0};
        float *out = tasks_out + (size_t)t * TD;
        const float *origin = NULL;
        int done = 0;
        for (int attempt = 0; attempt < 10 && !done; ++attempt) {
            int h = (int)(o_uniform(&rng) * (float)n_hist);
            if (h > n_hist - 1) h = n_hist - 1;
            origin = history + (size_t)h * TD;
            int ok = 1;
            for (int b = 0; b < nb; ++b) {
                float v[3] = {origin[3 * b], origin[3 * b + 1], origin[3 * b + 2]};
                if (b < NM) {
                    for (int j = 0; j < 3; ++j) v[j] += (o_uniform(&rng) * 2.0f - 1.0f) * expand_step;
                    v[0] = o_clamp(v[0], -bxy, bxy); v[1] = o_clamp(v[1], -bxy, bxy);
                    v[2] = o_clamp(v[2], c->max_height - 0.1f, c->max_height + 0.1f);
                } else {
                    if (expand_cylinders)
                        for (int j = 0; j < 2; ++j) {
                            int s = (int)(o_uniform(&rng) * 3.0f);
                            v[j] += (float)((s > 2 ? 2 : s) - 1) * c->grid_size;
                        }
                    v[0] = o_clamp(v[0], -cb, cb); v[1] = o_clamp(v[1], -cb, cb);
                    v[2] = o_clamp(v[2], -20.0f, c->max_height * 0.5f);
                }
                out[3 * b] = v[0]; out[3 * b + 1] = v[1]; out[3 * b + 2] = v[2];
                const int gx = o_envgen_cell((double)v[0], gs, GN), gy = o_envgen_cell((double)v[1], gs, GN);
                const int dx = gx - half, dy = gy - half;
                if (dx * dx + dy * dy >= half * half) ok = 0;
                cells[b] = gx * GN + gy;
            }
            for (int b = 1; b < nb && ok; ++b)
                for (int b2 = 0; b2 < b; ++b2)
                    if (cells[b] == cells[b2]) { ok = 0; break; }
            done = ok;
        }
        if (!done)
            for (int j = 0; j < TD; ++j) out[j] = origin[j];
    }
    return 0;
}
