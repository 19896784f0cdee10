"""CPU: host-side pieces of the API mirror that need no GPU (wire formats, pose algebra)."""
import json

import numpy as np
import pytest

from camera_intrinsic_calibration_rs_amd import api


def test_model_json_matches_reference_sample(tmp_path):
    """data/eucm.json of the reference: {"EUCM": {fx, fy, cx, cy, alpha, beta, width, height}}."""
    ref = {"EUCM": {"fx": 190.89618687183938, "fy": 190.87022285882367, "cx": 254.9375370481962,
                    "cy": 256.86414483060787, "alpha": 0.6283550447635853, "beta": 1.0458678747533083,
                    "width": 512, "height": 512}}
    m = api.GenericModel.from_json_obj(ref)
    assert m.kind == "eucm" and m.width() == 512 and m.params()[5] == ref["EUCM"]["beta"]
    p = tmp_path / "cam0.json"
    api.model_to_json(str(p), m)
    assert json.load(open(p)) == ref
    assert np.array_equal(api.model_from_json(str(p)).params(), m.params())


def test_pose_and_extrinsics_json_roundtrip(tmp_path):
    poses = {7: api.RvecTvec((0.1, 0.2, 0.3), (1.0, 2.0, 3.0)), 2: api.RvecTvec((0.0, 0.0, 0.1), (0.0, 0.0, 1.0))}
    p = tmp_path / "cam0_poses.json"
    api.poses_to_json(str(p), poses)
    raw = json.load(open(p))
    assert list(raw.keys()) == ["2", "7"] and raw["7"] == {"rvec": [0.1, 0.2, 0.3], "tvec": [1.0, 2.0, 3.0]}
    assert api.poses_from_json(str(p)) == poses
    e = tmp_path / "extrinsics.json"
    api.extrinsics_to_json(str(e), list(poses.values()))
    assert list(json.load(open(e)).keys()) == ["rtvecs"]
    assert api.extrinsics_from_json(str(e)) == list(poses.values())


def test_write_report_format(tmp_path):
    """src/io.rs:21-31."""
    p = tmp_path / "report.txt"
    api.write_report(str(p), True, [(0.123456789, 0.1), (0.2, 0.098765)])
    assert open(p).read() == ("Calibrate with extrinsics: true\n\n"
                              "cam0:\n    average reprojection error: 0.12346 px\n    median  reprojection error: 0.10000 px\n\n"
                              "cam1:\n    average reprojection error: 0.20000 px\n    median  reprojection error: 0.09877 px\n\n")


def test_rvec_tvec_algebra_matches_oracle(oracle):
    """tests/types_test.rs:5-20 round trip + compose/inverse against the oracle's nalgebra restatement."""
    a = api.RvecTvec((0.1, 0.2, 0.3), (1.0, 2.0, 3.0)); b = api.RvecTvec((-0.4, 0.05, 0.2), (0.3, -0.1, 0.9))
    np.testing.assert_allclose(a.compose(b).as6(), oracle.pose_compose(a.as6(), b.as6()), atol=1e-13)
    np.testing.assert_allclose(a.inverse().as6(), oracle.pose_inverse(a.as6()), atol=1e-13)
    np.testing.assert_allclose(a.inverse().inverse().as6(), a.as6(), atol=1e-13)


def test_se3_factor_analytic_jacobian_matches_dual_numbers(oracle):
    """SE3Factor (src/optimization/factors.rs:248-271): the library's closed-form 6 x 6 block Jacobian (inverse left
    Jacobian of SO(3) at the residual rotation, left Jacobian at rvec) against the oracle's forward-mode duals through the
    quaternion path nalgebra takes - random poses, small and large rotations, residual rotations up to ~2.5 rad."""
    import ctypes as C
    from camera_intrinsic_calibration_rs_amd import _ffi
    lib = _ffi.load()
    fn = lib.ccal_se3_factor
    dp = C.POINTER(C.c_double)
    fn.restype = C.c_int; fn.argtypes = [dp, dp, dp, dp, dp]
    rng = np.random.default_rng(11)
    worst = 0.0
    for k in range(200):
        scale = [1e-4, 0.3, 1.2][k % 3]
        a = np.concatenate([rng.normal(0, 1.0, 3), rng.normal(0, 1.0, 3)])
        x = np.concatenate([rng.normal(0, scale, 3), rng.normal(0, 0.5, 3)])
        b = np.asarray(api.RvecTvec.from6(x).compose(api.RvecTvec.from6(a)).as6()) + np.concatenate([rng.normal(0, scale, 3), rng.normal(0, 0.1, 3)])
        r = np.empty(6); J = np.empty((6, 6))
        assert fn(a.ctypes.data_as(dp), b.ctypes.data_as(dp), x.ctypes.data_as(dp), r.ctypes.data_as(dp), J.ctypes.data_as(dp)) == 0
        ro, Jo = oracle.se3_factor(a, b, x)
        np.testing.assert_allclose(r, ro, rtol=0, atol=1e-12)
        worst = max(worst, np.abs(J - Jo).max())
    assert worst <= 1e-9, worst


def test_init_camera_extrinsic_matches_oracle_and_truth(oracle):
    """util::init_camera_extrinsic (src/util.rs:511-561) -- SE3Factor + HuberLoss(0.5) + GN.  The library's
    host implementation (analytic Jacobian) against the oracle's dual-number one and against the
    transform the poses were generated with; one gross outlier frame is absorbed by the Huber loss."""
    rng = np.random.default_rng(3)
    t_i_0 = api.RvecTvec((0.01, -0.02, 0.005), (-0.101, 0.002, 0.001))
    cam0, cam1 = {}, {}
    for k in range(40):
        t_0_b = api.RvecTvec(tuple(np.array([3.0, 0.1, -0.2]) + rng.normal(0, 0.2, 3)), tuple(np.array([-0.3, 0.3, 0.9]) + rng.normal(0, 0.1, 3)))
        noisy = api.RvecTvec.from6(t_i_0.compose(t_0_b).as6() + rng.normal(0, 1e-3, 6))
        cam0[k * 3] = t_0_b
        cam1[k * 3] = noisy
    cam1[6] = api.RvecTvec.from6(cam1[6].as6() + np.array([0.4, 0.0, 0.0, 0.3, 0.0, 0.0]))     # outlier frame
    cam0[1000] = api.RvecTvec((0.0, 0.0, 0.0), (0.0, 0.0, 1.0))                                   # not shared
    out = api.init_camera_extrinsic([cam0, cam1])
    assert out[0].as6().tolist() == [0.0] * 6
    keys = sorted(set(cam0) & set(cam1))
    ref, iters = oracle.init_camera_extrinsic([cam0[k].as6() for k in keys], [cam1[k].as6() for k in keys])
    np.testing.assert_allclose(out[1].as6(), ref, rtol=0, atol=1e-10)      # same Jacobian to rounding: the same iterates
    # Huber bounds the outlier's pull to ~delta / n = 0.5 / 40 per component
    np.testing.assert_allclose(out[1].as6(), t_i_0.as6(), rtol=0, atol=1.5e-2)
    assert 1 <= iters <= 20


def test_oracle_stop_rule_under_both_error_metrics(oracle):
    """ccal_solver_opts.error_metric in the checker: on a problem whose cost is >> 1 the norm reading stops no later than the
    squared-norm reading, and the reports carry the squared norm either way."""
    from camera_intrinsic_calibration_rs_amd import synth
    from camera_intrinsic_calibration_rs_amd.engine import default_opts
    sp = synth.make_problem(60, "eucm", seed=9, outlier_frac=0.1, noise_px=0.5)
    op = oracle.OracleProblem.from_synth(sp)
    for method in (0, 1):
        reps = {}
        for em in (0, 1):
            _, _, _, reps[em] = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method, error_metric=em))
            assert reps[em].status == 0
        # d sqrt(c) / sqrt(c) = dc / 2c: the norm reading meets the relative threshold no later (Gauss-Newton's quadratic
        # convergence usually crosses both in the same iteration; this LM run stops one iteration earlier)
        assert reps[1].iterations <= reps[0].iterations
        if method == 1:
            assert reps[1].iterations < reps[0].iterations
        assert reps[0].initial_cost == reps[1].initial_cost > 100.0
        assert reps[1].final_cost == pytest.approx(reps[0].final_cost, rel=1e-4)


def test_init_camera_extrinsic_opts_entry_point():
    """ccal_init_camera_extrinsic_opts: NULL options = the plain call; max_iterations = 1 stops after one step."""
    import ctypes as C
    from camera_intrinsic_calibration_rs_amd import _ffi
    from camera_intrinsic_calibration_rs_amd.engine import default_opts
    rng = np.random.default_rng(5)
    n = 12
    p0 = np.concatenate([0.2 * rng.standard_normal((n, 3)), rng.standard_normal((n, 3))], axis=1)
    pi = p0 + 0.01 * rng.standard_normal((n, 6)); pi[:, 3] += 0.1
    lib = _ffi.load()
    dp = C.POINTER(C.c_double)
    def run(opts):
        x = np.zeros(6); rep = _ffi.Report()
        rc = lib.ccal_init_camera_extrinsic_opts(p0.ctypes.data_as(dp), pi.ctypes.data_as(dp), n, x.ctypes.data_as(dp), 0,
                                                 C.byref(opts) if opts is not None else None, C.byref(rep))
        assert rc == 0
        return x, rep
    x_a, r_a = run(None)
    x_b = np.zeros(6); r_b = _ffi.Report()
    assert lib.ccal_init_camera_extrinsic(p0.ctypes.data_as(dp), pi.ctypes.data_as(dp), n, x_b.ctypes.data_as(dp), 0, C.byref(r_b)) == 0
    np.testing.assert_array_equal(x_a, x_b)
    x_c, r_c = run(default_opts(0))
    np.testing.assert_array_equal(x_a, x_c)
    _, r_1 = run(default_opts(0, max_iterations=1))
    assert r_1.iterations == 1 <= r_a.iterations
    x_n, r_n = run(default_opts(0, error_metric=1))       # a cost below 1: the norm's decreases are LARGER - it may stop later
    assert r_n.status == 0 and r_n.iterations >= r_a.iterations
    np.testing.assert_allclose(x_n, x_a, atol=1e-6)


def test_convert_model_ucm_to_eucmt_closed_form():
    """src/util.rs:236-243: UCM -> EUCMT inserts beta = 1 and two zero tangential terms; through the Python mirror and
    through the C ABI (a host-only path of ccal_convert_model, no GPU needed); everything else with EUCMT is refused."""
    import ctypes as C
    from camera_intrinsic_calibration_rs_amd import _ffi
    from camera_intrinsic_calibration_rs_amd.engine import CcalError
    ucm = api.GenericModel("ucm", [500.0, 510.0, 320.0, 240.0, 0.55], 640, 480)
    t = api.GenericModel("eucmt", [0.0] * 8, 640, 480)
    assert api.convert_model(ucm, t).params().tolist() == [500.0, 510.0, 320.0, 240.0, 0.55, 1.0, 0.0, 0.0]
    assert t.model_id == 4 and _ffi.load().ccal_model_num_params(4) == 8
    with pytest.raises(NotImplementedError):
        t.to_json_obj()
    with pytest.raises(CcalError):
        api.convert_model(api.GenericModel("eucm", [1, 1, 1, 1, 0.5, 1.0], 640, 480), api.GenericModel("eucmt", [0.0] * 8, 640, 480))


def test_convert_model_closed_form():
    """tests/util_test.rs:77-110: UCM -> EUCM copies the parameters and sets beta = 1."""
    ucm = api.GenericModel("ucm", [500.0, 500.0, 320.0, 240.0, 0.5], 640, 480)
    eucm = api.GenericModel("eucm", [400.0, 400.0, 320.0, 240.0, 0.0, 1.0], 640, 480)
    p = api.convert_model(ucm, eucm).params()
    assert abs(p[0] - 500.0) < 1e-6 and abs(p[4] - 0.5) < 1e-6 and abs(p[5] - 1.0) < 1e-6


def test_bench_mode_n_roofline_inputs_present():
    """bench.py's extra.mode_N_roofline reads the ISA operation counts of the kernels it times from the latest
    profiles/rNN/flops.json (tools/count_flops.py): every key the default workloads ask for must be there."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    path = bench._latest_profile_file("flops.json")
    assert path is not None
    kernels = json.load(open(path))["kernels"]
    for model, K in (("ucm", 5), ("eucm", 6), ("kb4", 8), ("opencv5", 9)):
        for frames in (1000, 10000):
            gk = bench._gram_kernel_key(model, False, frames)
            assert kernels[gk]["per_corner"]["flops"] > 200, gk
        assert kernels[f"k_schur1m<K={K}>"]["per_lane_whole_kernel"]["per_frame_flops_16_lanes"] > 1000
