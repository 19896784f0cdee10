#!/usr/bin/env python3
"""Generate tests/golden/*.json -- independent pins for the oracle (TEST INFRASTRUCTURE).

The reference (Rust) cannot be built or imported in this image and its own tests pin only UCM
at a zero-residual point (tests/optimization_test.rs:36-80).  These vectors pin the *published math*
of the path independently of both the oracle's C++ and the HIP kernels:

  factor_golden.json   50-digit mpmath evaluation of  r = pi(theta, R(rvec) X + t) - x_obs  and of
                       the other-camera chain T_i_0 * T_0_b, with Jacobians by mpmath's high-order
                       numerical differentiation at 50 digits (good to ~1e-25), for UCM / EUCM / KB4 /
                       OPENCV5, with and without xy_same_focal.  Rotation uses the matrix Rodrigues
                       formula (NOT the quaternion path the oracle restates).
  reference_tests.json the known answers of the reference's own tests on this path.
  converged_golden.json an independent converged optimum: scipy.optimize.least_squares (TRF, linear
                       loss, analytic-free 2-point Jacobian replaced by mp-free numpy model) on a small
                       inlier-only synthetic set, where Huber(1.0) is inactive at the optimum.

Run from the repo root:  python oracle/gen_golden.py
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
mp.mp.dps = 50

UCM, EUCM, KB4, OCV5 = 0, 1, 2, 3
NP = {UCM: 5, EUCM: 6, KB4: 8, OCV5: 9}


def rodrigues(w):
    th = mp.sqrt(w[0] ** 2 + w[1] ** 2 + w[2] ** 2)
    K = mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th == 0:
        return mp.eye(3)
    return mp.eye(3) + (mp.sin(th) / th) * K + ((1 - mp.cos(th)) / th ** 2) * (K * K)


def project(model, p, pt):
    x, y, z = pt
    fx, fy, cx, cy = p[0], p[1], p[2], p[3]
    if model in (UCM, EUCM):
        beta = p[5] if model == EUCM else mp.mpf(1)
        rho = mp.sqrt(beta * (x * x + y * y) + z * z)
        den = p[4] * rho + (1 - p[4]) * z
        mx, my = x / den, y / den
    elif model == KB4:
        r = mp.sqrt(x * x + y * y)
        th = mp.atan2(r, z)
        thd = th + p[4] * th ** 3 + p[5] * th ** 5 + p[6] * th ** 7 + p[7] * th ** 9
        mx, my = thd * x / r, thd * y / r
    else:
        xn, yn = x / z, y / z
        r2 = xn * xn + yn * yn
        rad = 1 + p[4] * r2 + p[5] * r2 ** 2 + p[8] * r2 ** 3
        mx = xn * rad + 2 * p[6] * xn * yn + p[7] * (r2 + 2 * xn * xn)
        my = yn * rad + p[6] * (r2 + 2 * yn * yn) + 2 * p[7] * xn * yn
    return fx * mx + cx, fy * my + cy


def residual(model, one_focal, other, vec, X, obs):
    """vec = [theta_eff..., rvec0, tvec0(, rvec1, tvec1)] as mp numbers."""
    P = NP[model]
    pe = P - (1 if one_focal else 0)
    th = list(vec[:pe])
    if one_focal:
        th = [th[0], th[0]] + th[1:]
    w0, t0 = vec[pe:pe + 3], vec[pe + 3:pe + 6]
    p = rodrigues(w0) * mp.matrix(X) + mp.matrix(t0)
    if other:
        w1, t1 = vec[pe + 6:pe + 9], vec[pe + 9:pe + 12]
        p = rodrigues(w1) * p + mp.matrix(t1)
    u, v = project(model, th, (p[0], p[1], p[2]))
    return u - obs[0], v - obs[1]


def jacobian(model, one_focal, other, vec, X, obs):
    n = len(vec)
    J = [[None] * n for _ in range(2)]
    for k in range(n):
        for row in range(2):
            def f(t, k=k, row=row):
                vv = list(vec)
                vv[k] = t
                return residual(model, one_focal, other, vv, X, obs)[row]
            J[row][k] = mp.diff(f, vec[k])
    return J


def f32(v):
    return float(np.float32(v))


def gen_factor_golden():
    rng = np.random.default_rng(20240914)
    params = {
        UCM: [190.89618687183938, 190.87022285882367, 254.9375370481962, 256.86414483060787, 0.6283550447635853],
        EUCM: [190.89618687183938, 190.87022285882367, 254.9375370481962, 256.86414483060787, 0.6283550447635853,
               1.0458678747533083],
        KB4: [190.9, 190.9, 255.0, 257.0, 0.003, 0.0007, -0.002, 0.0002],
        OCV5: [380.0, 380.0, 255.0, 257.0, -0.28, 0.07, 0.0002, 0.00002, 0.001],
    }
    cases = []
    for model in (UCM, EUCM, KB4, OCV5):
        for one_focal in (False, True):
            for other in (False, True):
                for rep in range(3):
                    th = list(params[model])
                    if one_focal:
                        th = [th[0]] + th[2:]
                    th = [t * (1 + 0.02 * rng.uniform(-1, 1)) for t in th]
                    # three pose regimes: generic, small angle (series branch on the GPU), near pi (real boards)
                    if rep == 0:
                        w0 = rng.uniform(-0.5, 0.5, 3)
                    elif rep == 1:
                        w0 = rng.uniform(-0.05, 0.05, 3)
                    else:
                        w0 = np.array([3.0, 0.1, -0.2]) + rng.uniform(-0.1, 0.1, 3)
                    X = [f32(rng.uniform(0, 0.66)), f32(rng.uniform(-0.66, 0)), f32(rng.uniform(-0.02, 0.02) if rep else 0.0)]
                    R0 = np.array(rodrigues([mp.mpf(float(a)) for a in w0]).tolist(), dtype=float)
                    centre = R0 @ np.array([0.33, -0.33, 0.0])
                    t0 = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.7, 1.1)]) - centre
                    vec = th + list(w0) + list(t0)
                    if other:
                        vec += list(np.array([0.01, -0.02, 0.005]) + rng.uniform(-0.01, 0.01, 3))
                        vec += list(np.array([-0.101, 0.002, 0.001]) + rng.uniform(-0.01, 0.01, 3))
                    vec = [float(a) for a in vec]
                    mvec = [mp.mpf(a) for a in vec]
                    mX = [mp.mpf(a) for a in X]
                    u, v = residual(model, one_focal, other, mvec, mX, (mp.mpf(0), mp.mpf(0)))
                    obs = [f32(float(u) + rng.normal(0, 0.3)), f32(float(v) + rng.normal(0, 0.3))]
                    mobs = [mp.mpf(a) for a in obs]
                    r = residual(model, one_focal, other, mvec, mX, mobs)
                    J = jacobian(model, one_focal, other, mvec, mX, mobs)
                    cases.append(dict(model=model, one_focal=one_focal, other=other, vec=vec, p3d=X, p2d=obs,
                                      r=[float(a) for a in r], J=[[float(a) for a in row] for row in J]))
                    print(f"model {model} of {one_focal} other {other} rep {rep}: r = {float(r[0]):.6f}, {float(r[1]):.6f}")
    with open(os.path.join(OUT, "factor_golden.json"), "w") as f:
        json.dump(dict(note="mpmath 50-digit r and J; vec=[theta_eff, rvec_0_b, tvec_0_b(, rvec_i_0, tvec_i_0)]",
                       cases=cases), f, indent=1)


def gen_reference_tests():
    """Known answers of the reference's own tests on this path."""
    p = [mp.mpf(500), mp.mpf(500), mp.mpf(320), mp.mpf(240), mp.mpf("0.5")]
    u, v = project(UCM, p, (mp.mpf(1), mp.mpf(2), mp.mpf(10)))
    data = dict(
        test_reprojection_factor=dict(  # tests/optimization_test.rs:36-80
            model=UCM, params=[500.0, 500.0, 320.0, 240.0, 0.5], width=640, height=480, p3d=[1.0, 2.0, 10.0],
            project_mp=[float(u), float(v)], zero_pose_residual_norm_lt=1e-4, tvec_bad=[0.1, 0.0, 0.0],
            bad_residual_norm_gt=1e-3),
        test_rvec_tvec_conversion=dict(rvec=[0.1, 0.2, 0.3], tvec=[1.0, 2.0, 3.0], tol=1e-6),  # tests/types_test.rs:5-20
        test_convert_model=dict(ucm=[500.0, 500.0, 320.0, 240.0, 0.5], eucm_expected=[500.0, 500.0, 320.0, 240.0, 0.5, 1.0]),
        test_board_init=dict(n_ids=144, tag_size=0.088, tag0=[[0, 0, 0], [0.088, 0, 0], [0.088, -0.088, 0], [0, -0.088, 0]]),
        # the reference's sample model file data/eucm.json (a data file: examples/convert_model.rs:13 loads it), as data
        data_eucm_json=json.load(open("/root/reference/data/eucm.json")),
    )
    with open(os.path.join(OUT, "reference_tests.json"), "w") as f:
        json.dump(data, f, indent=1)


def gen_converged():
    """Independent converged optimum (scipy TRF) on a 12-frame inlier-only EUCM set."""
    from scipy.optimize import least_squares
    from camera_intrinsic_calibration_rs_amd import synth
    sp = synth.make_problem(12, "eucm", seed=0xBEEF, noise_px=0.1)
    X = sp.p3d.astype(np.float64); obs = sp.p2d.astype(np.float64)
    slot = np.repeat(sp.obs_slot, np.diff(sp.obs_offsets))

    def fun(v):
        th = v[:6]; poses = v[6:].reshape(-1, 6)
        R = synth.rodrigues(poses[:, :3])
        pc = np.einsum("nij,nj->ni", R[slot], X) + poses[slot, 3:]
        return (synth.project(synth.MODEL_EUCM, th, pc) - obs).ravel()

    v0 = np.concatenate([sp.intr0[0, :6], sp.poses0.ravel()])
    sol = least_squares(fun, v0, method="trf", loss="linear", xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale="jac", max_nfev=200)
    res = fun(sol.x).reshape(-1, 2)
    assert (np.sum(res ** 2, axis=1) < 1.0).all(), "Huber must be inactive at the optimum"
    data = dict(note="scipy least_squares TRF linear loss; make_problem(12,'eucm',seed=0xBEEF,noise_px=0.1)",
                n_frames=12, model="eucm", seed=0xBEEF, noise_px=0.1,
                intr=sol.x[:6].tolist(), poses=sol.x[6:].reshape(-1, 6).tolist(),
                cost=float(np.sum(res ** 2)), p2d_head=sp.p2d[:4].tolist(), p3d_head=sp.p3d[:4].tolist())
    with open(os.path.join(OUT, "converged_golden.json"), "w") as f:
        json.dump(data, f, indent=1)
    print("converged intr", sol.x[:6], "cost", data["cost"], "nfev", sol.nfev)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_reference_tests()
    gen_converged()
    gen_factor_golden()
