"""A5 — the rigid-body integrator is the build's own spec (PhysX is closed source, nothing in the reference pins it), so
it gets evidence of its own: against an independent fp64 solution of the CONTINUOUS rigid-body ODE it discretises
(scipy solve_ivp, rtol 1e-11) the integrator (oracle restatement = the HIP kernel, bit for bit) converges at first order
and its one-step defect is O(dt^2).

    v' = F_w/m + g - c_lin v          p' = v
    w_b' = I^-1 (tau_b - w_b x I w_b) - c_ang w_b          q' = 1/2 (0, w_w) (x) q,  w_w = R(q) w_b

with constant world force F_w and body torque tau_b (what one env step applies), damping c = 0.2 (robots/config.py:32-34)."""
import numpy as np
import pytest
from scipy.integrate import solve_ivp

import hns_oracle as O
from hns_amd import config

M, G = 0.0321, 9.81
I = np.array([1.4e-5, 1.4e-5, 2.17e-5])
C_LIN = C_ANG = 0.2


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def rot(q, v):
    return qmul(qmul(q, np.array([0.0, *v])), q * np.array([1, -1, -1, -1]))[1:]


def rhs(t, y, F, tau):
    p, q, v, ww = y[0:3], y[3:7] / np.linalg.norm(y[3:7]), y[7:10], y[10:13]
    wb = rot(q * np.array([1, -1, -1, -1]), ww)
    dwb = (tau - np.cross(wb, I * wb)) / I - C_ANG * wb
    dq = 0.5 * qmul(np.array([0.0, *ww]), q)
    # d/dt (R w_b) = R (w_b' + w_b x w_b) = R w_b'
    return np.concatenate([v, dq, F / M + np.array([0, 0, -G]) - C_LIN * v, rot(q, dwb)])


def cfg_for(dt):
    c = config.resolve_hns_cfg(config.make_cfg({"num_agents": 1, "v_drone": 50.0, "env": {"num_envs": 1}, "sim": {"dt": dt}}))
    c.ground_clamp = 0
    return c


def run_oracle(y0, F, tau, dt, T):
    c = cfg_for(dt)
    ds = y0.astype(np.float32)[None]
    for _ in range(int(round(T / dt))):
        ds = O.integrate(c, ds, F.astype(np.float32), tau.astype(np.float32))
    return ds[0].astype(np.float64)


CASES = [
    # hovering thrust with a tilt torque; a thrown, spinning drone; pure spin about a non-principal axis
    (np.array([0, 0, 0.6, 1, 0, 0, 0, 0.2, -0.1, 0.0, 0.0, 0.0, 0.0]), np.array([0.02, -0.01, M * G * 1.05]), np.array([2e-6, -1e-6, 5e-7])),
    (np.array([0.1, -0.2, 0.8, 0.9238795, 0.0, 0.3826834, 0.0, 0.5, 0.3, 0.4, 1.0, -2.0, 0.5]), np.array([0.05, 0.02, 0.2]), np.array([-3e-6, 2e-6, 1e-6])),
    (np.array([0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 3.0, 2.0, -4.0]), np.array([0.0, 0.0, M * G]), np.array([0.0, 0.0, 0.0])),
]


@pytest.mark.parametrize("y0,F,tau", CASES)
def test_first_order_convergence_to_the_continuous_ode(y0, F, tau):
    T = 0.08
    ref = solve_ivp(rhs, (0, T), y0, args=(F, tau), rtol=1e-11, atol=1e-13, method="DOP853").y[:, -1]
    ref[3:7] /= np.linalg.norm(ref[3:7])
    errs = []
    for dt in (0.02, 0.01, 0.005, 0.0025):
        got = run_oracle(y0, F, tau, dt, T)
        if np.dot(got[3:7], ref[3:7]) < 0:
            got[3:7] = -got[3:7]
        errs.append(np.abs(got - ref).max())
    ratios = [errs[i] / errs[i + 1] for i in range(3)]
    assert all(1.6 < r < 2.6 for r in ratios), (errs, ratios)          # error halves with dt: first order
    assert errs[1] < 3e-2                                               # at the reference's dt = 0.01 over 8 steps (worst case: the 5 rad/s tumble)


@pytest.mark.parametrize("y0,F,tau", CASES)
def test_one_step_defect_is_second_order(y0, F, tau):
    d = []
    for dt in (0.01, 0.005, 0.0025):
        ref = solve_ivp(rhs, (0, dt), y0, args=(F, tau), rtol=1e-12, atol=1e-14, method="DOP853").y[:, -1]
        ref[3:7] /= np.linalg.norm(ref[3:7])
        got = run_oracle(y0, F, tau, dt, dt)
        d.append(np.abs(got - ref).max())
    assert d[0] < 4e-3 and all(3.0 < d[i] / d[i + 1] < 5.0 for i in range(2)), d   # O(dt^2) per step
    # the quaternion stays a unit quaternion, exactly as far as fp32 allows
    assert abs(np.linalg.norm(run_oracle(y0, F, tau, 0.01, 0.01)[3:7]) - 1.0) < 2e-7
