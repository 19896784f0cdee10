"""GPU parity, mode N: fused normal equations + per-frame Schur complement and the optimizer loop,
through the C ABI, against the CPU oracle (dual-number Jacobians, dense host algebra).

Tolerances (fp64): reduced system S, b: |d| <= 1e-9 * max|S| (resp. max|b|) -- the Schur terms cancel
~3 digits of the raw sums; cost: 1e-12 relative; converged intrinsics: <= 1e-6 relative (north_star);
final cost: 1e-9 relative."""
import numpy as np
import pytest

from camera_intrinsic_calibration_rs_amd import _ffi, synth
from camera_intrinsic_calibration_rs_amd.engine import CcalError, Problem, default_opts

pytestmark = pytest.mark.gpu


def _pair(gpu_ctx, oracle, sp):
    return Problem.from_synth(gpu_ctx, sp), oracle.OracleProblem.from_synth(sp)


@pytest.mark.parametrize("model", ["ucm", "eucm", "kb4", "opencv5"])
@pytest.mark.parametrize("one_focal", [False, True])
@pytest.mark.parametrize("n_cams", [1, 2])
@pytest.mark.parametrize("lam", [0.0, 1e-3])
def test_build_normal_matches_oracle(gpu_ctx, oracle, model, one_focal, n_cams, lam):
    sp = synth.make_problem(23, model, n_cams=n_cams, xy_same_focal=one_focal, ragged=True, outlier_frac=0.02)
    gp, op = _pair(gpu_ctx, oracle, sp)
    S, b, cost = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
    So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
    assert np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    assert np.abs(S - S.T).max() <= 1e-12 * np.abs(S).max()
    # the GN step itself (what the reference's Cholesky would return for the camera block)
    dc, dco = np.linalg.solve(S, -b), np.linalg.solve(So, -bo)
    assert np.abs(dc - dco).max() <= 1e-6 * np.abs(dco).max()


@pytest.mark.parametrize("model,n_cams,one_focal", [("eucm", 1, False), ("kb4", 1, True), ("opencv5", 1, False),
                                                    ("eucm", 2, False), ("ucm", 3, True)])
@pytest.mark.parametrize("method", [_ffi.METHOD_GN, _ffi.METHOD_LM])
@pytest.mark.parametrize("error_metric", [_ffi.ERROR_SQUARED_NORM, _ffi.ERROR_NORM])
def test_solve_matches_oracle(gpu_ctx, oracle, model, n_cams, one_focal, method, error_metric):
    """Both readings of tiny-solver's compute_error (ccal_solver_opts.error_metric; call sites /root/reference/src/util.rs:443,455):
    the stop rules on the squared norm and on the norm - GPU and oracle stop after the same number of iterations under each."""
    sp = synth.make_problem(30, model, n_cams=n_cams, xy_same_focal=one_focal, outlier_frac=0.01)
    gp, op = _pair(gpu_ctx, oracle, sp)
    gp.apply_reference_bounds(); op.apply_reference_bounds()
    intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method, error_metric=error_metric))
    intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method, error_metric=error_metric))
    assert rep.status == rep_o.status == 0
    assert rep.iterations == rep_o.iterations
    assert abs(rep.initial_cost - rep_o.initial_cost) <= 1e-12 * rep_o.initial_cost
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    scale = np.maximum(np.abs(intr_o[:, :P]), 1e-3)          # distortion terms near 0: absolute 1e-9
    assert (np.abs(intr[:, :P] - intr_o[:, :P]) / scale).max() <= 1e-6
    np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
    np.testing.assert_allclose(extr, extr_o, rtol=0, atol=1e-7)
    if one_focal:
        assert (intr[:, 0] == intr[:, 1]).all()              # fy re-inserted (src/util.rs:467-470)
    # and the answer is the right one: close to ground truth
    assert np.abs(intr[:, :4] / sp.intr_gt[:, :4] - 1).max() < 5e-3


@pytest.mark.parametrize("frames", [60, 3000])
def test_error_metric_moves_the_stop_and_both_sides_follow(gpu_ctx, oracle, frames):
    """The norm reading stops no later than the squared-norm reading (d sqrt(c) / sqrt(c) = dc / 2 c) - one LM iteration earlier on
    the 60-frame problem: the option is live in the single-launch form (60 frames) and in the three-launch form (3 000), and the
    oracle follows."""
    sp = synth.make_problem(frames, "eucm", seed=9, outlier_frac=0.1, noise_px=0.5)
    gp, op = _pair(gpu_ctx, oracle, sp)
    its = {}
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        for em in (_ffi.ERROR_SQUARED_NORM, _ffi.ERROR_NORM):
            _, _, _, r = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method, error_metric=em))
            rd = None
            if frames <= 100:
                _, _, _, ro = op.solve(sp.intr0, sp.poses0, opts=default_opts(method, error_metric=em))
                assert (r.status, r.iterations) == (ro.status, ro.iterations), (method, em)
                assert abs(r.final_cost - ro.final_cost) <= 1e-9 * ro.final_cost
            gp.upload_params(sp.intr0, sp.poses0, sp.extr0)
            rd = gp.solve_dev(default_opts(method, error_metric=em))
            assert (rd.status, rd.iterations) == (r.status, r.iterations)
            its[(method, em)] = r.iterations
    assert its[(0, 1)] <= its[(0, 0)] and its[(1, 1)] <= its[(1, 0)], its
    if frames == 60:
        assert its[(1, 1)] < its[(1, 0)], its       # LM: one iteration less under the norm (the oracle's counts: 4 against 5)


def test_gn_and_lm_converge_to_same_intrinsics(gpu_ctx):
    sp = synth.make_problem(200, "eucm")
    gp = Problem.from_synth(gpu_ctx, sp)
    tight = dict(min_abs_error_decrease=1e-11, min_rel_error_decrease=1e-13)
    i_gn, _, _, r_gn = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN, **tight))
    i_lm, _, _, r_lm = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM, **tight))
    assert np.abs(i_gn[0, :6] / i_lm[0, :6] - 1).max() < 1e-6
    assert abs(r_gn.final_cost / r_lm.final_cost - 1) < 1e-9


def test_fixed_focal_disabled_distortion_and_bounds(gpu_ctx, oracle):
    """set_problem_parameter_disabled (src/util.rs:50-71), fix_variable("params", 0) (src/util.rs:461),
    bounds clamp after every step."""
    sp = synth.make_problem(12, "opencv5", xy_same_focal=True)
    gp, op = _pair(gpu_ctx, oracle, sp)
    i0 = sp.intr0.copy(); i0o = sp.intr0.copy()
    gp.disable_distortions(1, i0); op.disable_distortions(1, i0o)
    np.testing.assert_array_equal(i0, i0o)
    gp.fix_param(0, 0); op.fix_param(0, 0)
    intr, poses, _, rep = gp.solve(i0, sp.poses0)
    intr_o, poses_o, _, rep_o = op.solve(i0o, sp.poses0)
    assert intr[0, 8] == 0.0 and intr[0, 0] == i0[0, 0] == intr[0, 1]
    assert rep.iterations == rep_o.iterations
    np.testing.assert_allclose(intr[0, :9], intr_o[0, :9], rtol=1e-6, atol=1e-9)

    sp = synth.make_problem(8, "eucm")
    gp, op = _pair(gpu_ctx, oracle, sp)
    gp.set_bounds(0, 4, 0.0, 0.5); op.set_bounds(0, 4, 0.0, 0.5)
    i0 = sp.intr0.copy(); i0[0, 4] = 0.45
    intr, _, _, rep = gp.solve(i0, sp.poses0)
    intr_o, _, _, rep_o = op.solve(i0, sp.poses0)
    assert intr[0, 4] <= 0.5
    np.testing.assert_allclose(intr[0, :6], intr_o[0, :6], rtol=1e-6)


def test_failure_is_a_status_not_a_crash(gpu_ctx):
    """A degenerate problem (one corner per frame: rank-deficient pose blocks) returns CCAL_ERR_NOT_PD
    (tiny-solver: None, src/util.rs:455-457), never aborts."""
    from camera_intrinsic_calibration_rs_amd.engine import make_desc
    sp = synth.make_problem(4, "eucm")
    offs = np.arange(5, dtype=np.int64)
    idx = sp.obs_offsets[:-1]
    d, keep = make_desc(1, [1], [512.0], [512.0], False, 4, [0] * 4, [0, 1, 2, 3], offs,
                        sp.p3d[idx, 0], sp.p3d[idx, 1], sp.p3d[idx, 2], sp.p2d[idx, 0], sp.p2d[idx, 1], 1.0)
    gp = Problem(gpu_ctx, d, keep)
    with pytest.raises(CcalError) as ei:
        gp.solve(sp.intr0, sp.poses0)
    assert ei.value.code in (_ffi.ERR_NOT_PD, _ffi.ERR_NONFINITE)
    # the context is still usable afterwards
    sp2 = synth.make_problem(6, "eucm")
    _, _, _, rep = Problem.from_synth(gpu_ctx, sp2).solve(sp2.intr0, sp2.poses0)
    assert rep.status == 0


def test_slot_without_observations_keeps_its_pose(gpu_ctx, oracle):
    sp = synth.make_problem(6, "eucm")
    keep = sp.obs_slot != 2
    counts = np.diff(sp.obs_offsets)[keep]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rows = np.concatenate([np.arange(sp.obs_offsets[o], sp.obs_offsets[o + 1]) for o in np.nonzero(keep)[0]])
    import dataclasses
    sp2 = dataclasses.replace(sp, obs_cam=sp.obs_cam[keep], obs_slot=sp.obs_slot[keep], obs_offsets=offs,
                              p3d=sp.p3d[rows], p2d=sp.p2d[rows])
    gp, op = _pair(gpu_ctx, oracle, sp2)
    intr, poses, _, rep = gp.solve(sp2.intr0, sp2.poses0)
    intr_o, poses_o, _, _ = op.solve(sp2.intr0, sp2.poses0)
    np.testing.assert_array_equal(poses[2], sp2.poses0[2])
    np.testing.assert_allclose(intr[0, :6], intr_o[0, :6], rtol=1e-6)


def test_full_size_solve_recovers_ground_truth(gpu_ctx):
    """North-star size, 10 000 frames x 144 corners: GN and LM agree to 1e-6 and sit at the noise floor."""
    sp = synth.make_problem(10000, "eucm")
    gp = Problem.from_synth(gpu_ctx, sp)
    i_gn, p_gn, _, r_gn = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN))
    i_lm, _, _, r_lm = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    assert r_gn.status == 0 and r_lm.status == 0
    assert np.abs(i_gn[0, :6] / i_lm[0, :6] - 1).max() < 1e-6
    assert np.abs(i_gn[0, :6] / sp.intr_gt[0, :6] - 1).max() < 5e-4       # 1.44 M observations at 0.1 px + f32 rounding
    # cost at the optimum ~ 2 sigma^2 per corner (sigma = 0.1 px + f32 rounding)
    assert 0.015 < r_gn.final_cost / gp.n_corners < 0.025
    a, m = gp.validation(0, i_gn, p_gn)
    assert 0.08 < m < 0.16


def test_lm_with_rejected_steps_matches_oracle(gpu_ctx, dev_ctx, oracle):
    """A poor starting point (intrinsics off by up to 80 %, 5 % gross outliers): the LM path takes a rejected
    step (radius shrink, re-elimination of the pose blocks from the stored records with the new damping, no
    Jacobian re-evaluation) and must walk the same accept / reject sequence as the oracle -- in the
    device-resident loop and in the general loop."""
    import os
    sp = synth.make_problem(8, "eucm", init_perturb=0.8, outlier_frac=0.05, seed=1)
    gp, op = _pair(gpu_ctx, oracle, sp)
    gd = Problem.from_synth(dev_ctx, sp)                # the general loop on a single camera: a switch of the second library
    gp.apply_reference_bounds(); op.apply_reference_bounds(); gd.apply_reference_bounds()
    intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    assert rep_o.lm_rejected >= 1 and rep_o.status == 0
    for disable_fused in ("", "1"):
        if disable_fused:
            os.environ["CCAL_DISABLE_FUSED"] = "1"
            gp = gd
        try:
            intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
        finally:
            os.environ.pop("CCAL_DISABLE_FUSED", None)
        assert (rep.status, rep.iterations, rep.lm_accepted, rep.lm_rejected) == \
               (rep_o.status, rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected), disable_fused
        assert abs(rep.final_cost - rep_o.final_cost) <= 1e-8 * rep_o.final_cost
        assert (np.abs(intr[0, :6] - intr_o[0, :6]) / np.abs(intr_o[0, :6])).max() <= 1e-6


def test_lm_hard_start_never_crashes(gpu_ctx):
    """Intrinsics off by up to 200 %: whatever happens (convergence, max_iterations, not-PD) is a status."""
    sp = synth.make_problem(8, "kb4", init_perturb=2.0, outlier_frac=0.05, seed=2)
    gp = Problem.from_synth(gpu_ctx, sp)
    gp.apply_reference_bounds()
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM), raise_on_error=False)
    assert rep.status in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE, _ffi.ERR_NOT_PD, _ffi.ERR_NONFINITE)
    assert rep.lm_accepted + rep.lm_rejected == rep.iterations or rep.status != _ffi.OK


@pytest.mark.parametrize("n_frames", [700, 1500, 2500, 5000])
def test_solve_all_lane_mappings(gpu_ctx, oracle, n_frames):
    """The register-Gram kernels map 64 / 32 / 16 lanes to a frame depending on the problem size (<= 1024, <= 4800,
    more) and keep all accumulators in registers below 2000 frames (k_gram1v) or part of them in LDS above
    (k_gram1w): every combination against the oracle, ragged frames included."""
    sp = synth.make_problem(n_frames, "eucm", ragged=True, outlier_frac=0.01)
    gp, op = _pair(gpu_ctx, oracle, sp)
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN))
    intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN))
    assert (rep.status, rep.iterations) == (rep_o.status, rep_o.iterations)
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    assert (np.abs(intr[0, :6] - intr_o[0, :6]) / np.abs(intr_o[0, :6])).max() <= 1e-6
    np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)


@pytest.mark.parametrize("model,n_cams", [("eucm", 1), ("opencv5", 1), ("eucm", 2)])
@pytest.mark.parametrize("method", [_ffi.METHOD_GN, _ffi.METHOD_LM])
def test_solve_dev_equals_solve(gpu_ctx, model, n_cams, method):
    """ccal_solve_dev (parameters uploaded once, result left on the device) is the same loop as ccal_solve without the
    host staging: bit-identical parameters and report; a second solve_dev continues from the first one's result."""
    sp = synth.make_problem(300, model, n_cams=n_cams, ragged=True, outlier_frac=0.01)
    gp = Problem.from_synth(gpu_ctx, sp)
    gp.apply_reference_bounds()
    intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
    gp.upload_params(sp.intr0, sp.poses0, sp.extr0)
    rep_d = gp.solve_dev(default_opts(method))
    intr_d, poses_d, extr_d = gp.download_params()
    assert (rep_d.status, rep_d.iterations, rep_d.lm_accepted, rep_d.lm_rejected) == (rep.status, rep.iterations, rep.lm_accepted, rep.lm_rejected)
    assert rep_d.final_cost == rep.final_cost and rep_d.initial_cost == rep.initial_cost
    np.testing.assert_array_equal(intr_d, intr); np.testing.assert_array_equal(poses_d, poses)
    if n_cams > 1:
        np.testing.assert_array_equal(extr_d, extr)
    # continuing from the optimum: converges at once, the cost does not go up
    rep2 = gp.solve_dev(default_opts(method))
    assert rep2.status == 0 and rep2.iterations <= 2
    assert rep2.initial_cost == rep.final_cost and rep2.final_cost <= rep.final_cost * (1 + 1e-12)


def test_lm_speculative_elimination_bookkeeping(gpu_ctx, oracle):
    """LM eliminates the candidate's pose blocks with the damping an accepted step of gain ratio ~1 gets (radius x 3).
    Well-conditioned start: every accepted step but the last is a hit (one group per step, like GN).  Poor start: misses
    and rejections are re-eliminated - and the accept / reject sequence is the oracle's either way."""
    sp = synth.make_problem(400, "eucm", outlier_frac=0.01)
    gp, op = _pair(gpu_ctx, oracle, sp)
    _, _, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    _, _, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    assert (rep.status, rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.status, rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)
    assert rep.lm_spec_hits >= 1 and rep.lm_spec_hits + rep.lm_spec_misses <= rep.lm_accepted
    sp = synth.make_problem(8, "eucm", init_perturb=0.8, outlier_frac=0.05, seed=1)
    gp, op = _pair(gpu_ctx, oracle, sp)
    gp.apply_reference_bounds(); op.apply_reference_bounds()
    _, _, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    _, _, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    assert rep.lm_rejected >= 1
    assert (rep.status, rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.status, rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)


@pytest.mark.parametrize("fused", [True, False])
def test_one_degenerate_frame_gn_not_pd_lm_recovers(gpu_ctx, dev_ctx, fused, monkeypatch):
    """One frame with a single corner (rank-deficient 6 x 6 pose block) among good frames: Gauss-Newton reports
    CCAL_ERR_NOT_PD on both loops (the failed block also poisons the all-reduced cost, so every rank of a sharded
    solve stops in the same iteration); Levenberg-Marquardt damps the block and converges."""
    from camera_intrinsic_calibration_rs_amd.engine import make_desc
    if not fused:
        monkeypatch.setenv("CCAL_DISABLE_FUSED", "1")    # (a switch of the second library)
        gpu_ctx = dev_ctx
    sp = synth.make_problem(12, "eucm")
    offs = sp.obs_offsets.copy()
    keep_idx = np.concatenate([np.arange(offs[0], offs[5]), [offs[5]], np.arange(offs[6], offs[-1])])
    new_offs = np.concatenate([offs[:6], offs[6:] - (offs[6] - offs[5] - 1)]).astype(np.int64)
    d, keep = make_desc(1, [1], [512.0], [512.0], False, 12, [0] * 12, list(range(12)), new_offs,
                        sp.p3d[keep_idx, 0], sp.p3d[keep_idx, 1], sp.p3d[keep_idx, 2], sp.p2d[keep_idx, 0], sp.p2d[keep_idx, 1], 1.0)
    gp = Problem(gpu_ctx, d, keep)
    with pytest.raises(CcalError) as ei:
        gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN))
    assert ei.value.code == _ffi.ERR_NOT_PD
    intr, _, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    assert rep.status == _ffi.OK
    assert (np.abs(intr[0, :4] / sp.intr_gt[0, :4] - 1)).max() < 5e-3


def test_six_cameras_matches_oracle(gpu_ctx, oracle):
    """Six one-focal UCM cameras: reduced system K = 6 * 4 + 5 * 6 = 54, the largest that the one-wavefront camera solve and
    the four-wavefront elimination serve (K + 1 <= 64)."""
    sp = synth.make_problem(12, "ucm", n_cams=6, xy_same_focal=True)
    gp, op = _pair(gpu_ctx, oracle, sp)
    intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0)
    intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0)
    assert (rep.status, rep.iterations) == (rep_o.status, rep_o.iterations) and rep.status == 0
    assert (np.abs(intr[:, :5] / intr_o[:, :5] - 1)).max() <= 1e-9
    np.testing.assert_allclose(extr, extr_o, rtol=0, atol=1e-9)


@pytest.mark.parametrize("model,one_focal,K", [("ucm", True, 74), ("eucm", False, 90), ("opencv5", False, 114)])
def test_eight_cameras_matches_oracle(gpu_ctx, oracle, model, one_focal, K):
    """The reference has no cap on --cam-num (src/bin/camera_calibration.rs:25-68); this ABI stops at CCAL_MAX_CAMS = 8
    cameras, whose reduced system has up to 8 * 9 + 7 * 6 = 114 columns.  Systems of 64 .. 127 columns take the large forms of
    the elimination (one wavefront per workgroup, (K + 1)^2 accumulators in LDS) and of the camera solve (two wavefronts, the
    matrix in dynamic LDS): normal equations, GN and LM against the oracle."""
    if model == "opencv5":          # narrow field of view: a rig whose cameras see different subsets of the slots
        rng = np.random.default_rng(8)
        extr = [[0.0] * 6] + [list(np.concatenate([rng.uniform(-0.04, 0.04, 3), rng.uniform(-0.03, 0.03, 3)])) for _ in range(7)]
        sp = synth.make_rig(16, (model,) * 8, extr, xy_same_focal=one_focal, seed=0xE16)
    else:
        sp = synth.make_problem(10, model, n_cams=8, xy_same_focal=one_focal, ragged=True)
    gp, op = _pair(gpu_ctx, oracle, sp)
    assert gp.K == K
    for lam in (0.0, 1e-3):
        S, b, cost = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
        assert np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        gp.apply_reference_bounds(); op.apply_reference_bounds()
        intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        assert (rep.status, rep.iterations) == (rep_o.status, rep_o.iterations) and rep.status == 0
        assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
        assert (np.abs(intr - intr_o) / np.maximum(np.abs(intr_o), 1e-3)).max() <= 1e-6
        np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
        np.testing.assert_allclose(extr, extr_o, rtol=0, atol=1e-7)


@pytest.mark.parametrize("models,one_focal,K", [(("opencv5",) * 4 + ("kb4",), True, 63), (("opencv5", "kb4", "ucm", "kb4", "kb4"), False, 62),
                                                 (("opencv5", "ucm", "eucm", "opencv5", "kb4"), False, 61)])
def test_five_camera_rigs_at_the_lds_boundary(gpu_ctx, oracle, models, one_focal, K):
    """61 .. 63 columns: four wavefronts' (K + 1)^2 accumulators plus OPENCV5-sized record staging exceed a CU's 160 KB of LDS -
    the elimination has to take its one-wavefront form below 64 columns too (found by tools/fuzz_parity.py: the launch was
    refused with 'invalid argument', and the runtime's sticky error then surfaced in the next problem's first launch)."""
    rng = np.random.default_rng(5)
    extr = [[0.0] * 6] + [list(np.concatenate([rng.uniform(-0.04, 0.04, 3), rng.uniform(-0.05, 0.05, 3)])) for _ in range(4)]
    sp = synth.make_rig(24, models, extr, xy_same_focal=one_focal, seed=0x5CA)
    gp, op = _pair(gpu_ctx, oracle, sp)
    assert gp.K == K
    S, b, cost = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=1e-3)
    So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=1e-3)
    assert abs(cost - costo) <= 1e-12 * costo and np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    gp.apply_reference_bounds(); op.apply_reference_bounds()
    intr, poses, extr_s, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(_ffi.METHOD_LM))
    intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(_ffi.METHOD_LM))
    assert (rep.status, rep.iterations) == (rep_o.status, rep_o.iterations) and rep.status == 0
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    np.testing.assert_allclose(extr_s, extr_o, rtol=0, atol=1e-7)
    # and a failed launch does not leak into the next problem
    r, J = Problem.from_synth(gpu_ctx, synth.make_problem(5, "kb4")).eval(synth.make_problem(5, "kb4").intr0, synth.make_problem(5, "kb4").poses0)
    assert np.isfinite(r).all()


@pytest.mark.parametrize("n_corners,n_frames", [(400, 6), (700, 3), (24, 30), (6, 40)])
def test_frames_of_any_size(gpu_ctx, oracle, n_corners, n_frames):
    """Frames far larger than the 144-corner board (several 64-corner tiles per wavefront, lanes per frame that do not
    divide the corner count) and down to the reference's minimum of 24 corners - and 6, twice the fewest that give a
    full-rank pose block: mode E and the solve against the oracle."""
    from camera_intrinsic_calibration_rs_amd.engine import make_desc
    base = synth.make_problem(n_frames, "eucm")
    rng = np.random.default_rng(n_corners)
    off, X, U = [0], [], []
    for f in range(n_frames):
        idx = rng.choice(np.arange(base.obs_offsets[f], base.obs_offsets[f + 1]), size=n_corners, replace=n_corners > 144)
        X.append(base.p3d[idx]); U.append(base.p2d[idx]); off.append(off[-1] + n_corners)
    X = np.concatenate(X).astype(np.float32); U = np.concatenate(U).astype(np.float32)
    d, keep = make_desc(1, [1], [512.0], [512.0], False, n_frames, [0] * n_frames, list(range(n_frames)),
                        np.array(off, dtype=np.int64), X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    gp = Problem(gpu_ctx, d, keep)
    op = oracle.OracleProblem(d, keep)
    r, J = gp.eval(base.intr0, base.poses0)
    ro, Jo = op.eval(base.intr0, base.poses0)
    assert np.abs(r - ro).max() < 1e-10 and (np.abs(J - Jo) / np.maximum(1.0, np.abs(Jo))).max() < 1e-11
    opts = default_opts(_ffi.METHOD_LM if n_corners < 24 else _ffi.METHOD_GN)
    intr, poses, _, rep = gp.solve(base.intr0, base.poses0, opts=opts)
    intr_o, poses_o, _, rep_o = op.solve(base.intr0, base.poses0, opts=opts)
    assert (rep.status, rep.iterations) == (rep_o.status, rep_o.iterations)
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * max(rep_o.final_cost, 1e-6)
    assert (np.abs(intr[0, :6] - intr_o[0, :6]) / np.abs(intr_o[0, :6])).max() <= 1e-6


_LPF_SCRIPT = """
import numpy as np, sys
sys.path.insert(0, {root!r})
from camera_intrinsic_calibration_rs_amd import _ffi, synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
from oracle import binding as ob
ctx = Context(0, lib=_ffi.load_for_switches())          # CCAL_GRAMV_LPF: a switch of the second library
for n_frames, model in ((300, "eucm"), (2600, "eucm"), (2600, "ucm"), (700, "kb4")):
    sp = synth.make_problem(n_frames, model, ragged=True, outlier_frac=0.01)
    gp = Problem.from_synth(ctx, sp); op = ob.OracleProblem.from_synth(sp)
    S, b, c = gp.build_normal(sp.intr0, sp.poses0, lam=1e-3); So, bo, co = op.build_normal(sp.intr0, sp.poses0, lam=1e-3)
    assert abs(c - co) <= 1e-12 * co and np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    for m in (0, 1):
        i, p, _, r = gp.solve(sp.intr0, sp.poses0, opts=default_opts(m)); io, po, _, ro = op.solve(sp.intr0, sp.poses0, opts=default_opts(m))
        assert (r.status, r.iterations) == (ro.status, ro.iterations), (r.status, r.iterations, ro.status, ro.iterations)
        P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
        assert (np.abs(i[0, :P] - io[0, :P]) / np.maximum(np.abs(io[0, :P]), 1e-3)).max() <= 1e-6
        assert np.abs(p - po).max() <= 1e-7
print("LPF-OK")
"""


@pytest.mark.parametrize("lpf", [6, 8, 12, 16, 32, 64])
def test_gram_lanes_per_frame_every_mapping(lpf):
    """Every lanes-per-frame mapping of the register Gram kernels (6 and 12 - ten / five frames per wavefront, four lanes
    idle -, 8, 16, 32, 64) forced through CCAL_GRAMV_LPF (read once per process, hence the subprocess), ragged frames, with damping
    (the per-frame phi -> rvec map only shows under damping: the undamped Schur complement is basis invariant)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CCAL_GRAMV_LPF=str(lpf))
    out = subprocess.run([sys.executable, "-c", _LPF_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
    assert "LPF-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _tile(sp, rep):
    """`rep` copies of a single-camera synthetic problem, frames renumbered (a large problem without regenerating it)."""
    import dataclasses
    n = sp.obs_offsets[-1]
    offs = np.concatenate([[0]] + [sp.obs_offsets[1:] + k * n for k in range(rep)]).astype(np.int64)
    return dataclasses.replace(
        sp, n_slots=sp.n_slots * rep, obs_cam=np.tile(sp.obs_cam, rep),
        obs_slot=np.concatenate([sp.obs_slot + k * sp.n_slots for k in range(rep)]).astype(np.int32), obs_offsets=offs,
        p3d=np.tile(sp.p3d, (rep, 1)), p2d=np.tile(sp.p2d, (rep, 1)),
        poses_gt=np.tile(sp.poses_gt, (rep, 1)), poses0=np.tile(sp.poses0, (rep, 1)))


def test_config4_size_properties(gpu_ctx, oracle):
    """BASELINE configs[3] size on ONE GPU: 50 000 frames x 144 corners (7.2 M blocks, 1.5 GB of r and J).  Against the
    oracle: sampled frames of the mode-E pass (every part of the launch: first, last, the chunk boundaries of the tiling)
    and the normal equations of ALL 50 000 frames (a second of CPU).  Size-independent properties: the squared norm of
    the loss-weighted residual (mode E) is the cost the normal-equation builder reports (mode N); the cost is additive
    over frame shards; five copies of a 10 000-frame problem have five times its cost and the same optimum."""
    base = synth.make_problem(10000, "eucm")
    sp = _tile(base, 5)
    gp = Problem.from_synth(gpu_ctx, sp)
    assert gp.n_corners == 50000 * 144
    r, J = gp.eval(sp.intr0, sp.poses0)
    Jb = J.reshape(-1, 2, 12)
    for f in (0, 9999, 10000, 25000, 39999, 40001, 49999):
        sub = base.shard(f % 10000, 10000)
        ro, Jo = oracle.OracleProblem.from_synth(sub).eval(sp.intr0, sub.poses0)
        sl = slice(f * 144, (f + 1) * 144)
        assert np.abs(r[sl] - ro).max() <= 1e-10
        assert (np.abs(Jb[sl].ravel() - Jo) / np.maximum(1.0, np.abs(Jo))).max() <= 1e-11
    del r, J, Jb
    r, J = gp.eval(sp.intr0, sp.poses0, apply_loss=True)
    assert np.isfinite(r).all() and np.isfinite(J).all()
    S, b, cost = gp.build_normal(sp.intr0, sp.poses0)
    assert abs(float((r * r).sum()) - cost) <= 1e-11 * cost
    del r, J
    So, bo, costo = oracle.OracleProblem.from_synth(sp).build_normal(sp.intr0, sp.poses0)
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    gb = Problem.from_synth(gpu_ctx, base)
    Sb, bb, cost_b = gb.build_normal(base.intr0, base.poses0)
    assert abs(cost - 5.0 * cost_b) <= 1e-11 * cost
    assert np.abs(S - 5.0 * Sb).max() <= 1e-9 * np.abs(S).max() and np.abs(b - 5.0 * bb).max() <= 1e-9 * np.abs(b).max()
    shard_cost = 0.0
    for k in range(4):
        sh = sp.shard(k, 4)
        shard_cost += Problem.from_synth(gpu_ctx, sh).build_normal(sh.intr0, sh.poses0)[2]
    assert abs(shard_cost - cost) <= 1e-11 * cost
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0)
    intr_b, poses_b, _, rep_b = gb.solve(base.intr0, base.poses0)
    assert rep.status == 0 and rep.iterations == rep_b.iterations
    assert np.abs(intr[0, :6] / intr_b[0, :6] - 1).max() <= 1e-9
    np.testing.assert_allclose(poses[:10000], poses_b, rtol=0, atol=1e-9)
    np.testing.assert_allclose(poses[40000:], poses_b, rtol=0, atol=1e-9)


def test_two_camera_rig_at_scale(gpu_ctx):
    """BASELINE configs[4] shape at scale (two EUCM cameras x 4 000 frames, 1.15 M blocks): GN and LM reach the same
    optimum, intrinsics and the inter-camera transform sit at the noise floor around ground truth."""
    sp = synth.make_problem(4000, "eucm", n_cams=2)
    gp = Problem.from_synth(gpu_ctx, sp)
    gp.apply_reference_bounds()
    i_gn, p_gn, e_gn, r_gn = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(_ffi.METHOD_GN))
    i_lm, p_lm, e_lm, r_lm = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(_ffi.METHOD_LM))
    assert r_gn.status == 0 and r_lm.status == 0
    assert np.abs(i_gn[:, :6] / i_lm[:, :6] - 1).max() < 1e-6
    np.testing.assert_allclose(e_gn, e_lm, rtol=0, atol=1e-7)
    assert np.abs(i_gn[:, :6] / sp.intr_gt[:, :6] - 1).max() < 1e-3
    assert np.abs(e_gn[1, 3:] - sp.extr_gt[1, 3:]).max() < 2e-4            # metres
    assert np.abs(e_gn[1, :3] - sp.extr_gt[1, :3]).max() < 5e-4            # radians
    assert 0.015 < r_gn.final_cost / gp.n_corners < 0.025


def test_two_camera_rig_at_headline_size_vs_oracle(gpu_ctx, oracle):
    """BASELINE configs[4] shape against the oracle AT SCALE: two EUCM cameras x 10 000 frames (2.88 M blocks) - the reduced
    normal equations S / b / cost of the general loop's size-dependent choices (lanes per frame of the Gram kernels, the
    persistent wavefronts of k_schur, the wide reduction) - and GN + LM to convergence at 2 x 1 000 frames, same iteration
    counts and accept / reject sequences as the oracle (src/util.rs:567-715)."""
    sp = synth.make_problem(10000, "eucm", n_cams=2)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    for lam in (0.0, 1e-3):
        S, b, cost = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        assert S.shape == (18, 18)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
        assert np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
        dc, dco = np.linalg.solve(S, -b), np.linalg.solve(So, -bo)
        assert np.abs(dc - dco).max() <= 1e-6 * np.abs(dco).max()
    gp.close()
    sp = synth.make_problem(1000, "eucm", n_cams=2, outlier_frac=0.01)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    gp.apply_reference_bounds(); op.apply_reference_bounds()
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        assert rep.status == rep_o.status == 0
        assert (rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)
        assert abs(rep.initial_cost - rep_o.initial_cost) <= 1e-12 * rep_o.initial_cost
        assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
        assert (np.abs(intr[:, :6] - intr_o[:, :6]) / np.abs(intr_o[:, :6])).max() <= 1e-6
        np.testing.assert_allclose(extr, extr_o, rtol=0, atol=1e-8)
        np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
    gp.close()


RIG_EXTR = {3: [[0.0] * 6, [0.3, -0.25, 0.2, -0.1, 0.02, 0.01], [-0.2, 0.35, -0.15, 0.1, -0.03, 0.02]],
            2: [[0.0] * 6, [0.12, -0.1, 0.3, -0.1, 0.02, 0.01]]}


_LEGACY_CTX = None


def _legacy_ctx():
    """A context of the SECOND build of the library (-DCCAL_LEGACY_KERNELS): the superseded matrix-core kernels are not in the
    product library; the tests that hold the product kernels against them load both builds side by side."""
    global _LEGACY_CTX
    if _LEGACY_CTX is None:
        from camera_intrinsic_calibration_rs_amd.engine import Context
        _LEGACY_CTX = Context(0, lib=_ffi.load_legacy())
    return _LEGACY_CTX


@pytest.mark.parametrize("models", [("eucm", "kb4", "ucm"), ("opencv5", "opencv5"), ("kb4", "eucm")])
@pytest.mark.parametrize("one_focal", [False, True])
def test_rig_of_different_cameras(gpu_ctx, oracle, models, one_focal, monkeypatch):
    """calib_all_camera_with_extrinsics' general input (src/util.rs:567-651): a different model per camera, LARGE extrinsic
    rotations (0.3-0.5 rad: R_c0, J_l(rvec_c0) and -[R_c0 t_0b]x of the record expansion are far from I / 0), cameras
    that see different subsets of the frame slots (slots without camera 0, slots seen by one camera only) and ragged
    corner sets.  Normal equations, GN and LM against the oracle; and the matrix-core implementation (19-column Gram
    per corner, CCAL_GENERAL_GRAM=mfma) against the register one (13 columns at the composed pose + expansion)."""
    sp = synth.make_rig(48, models, RIG_EXTR[len(models)], xy_same_focal=one_focal, seed=0xBEEF + len(models))
    seen = np.zeros((sp.n_slots, sp.n_cams), bool)
    seen[sp.obs_slot, sp.obs_cam] = True
    assert (~seen[:, 0]).any() and (seen.sum(1) == 1).any()            # the cases the test is about are present
    gp, op = _pair(gpu_ctx, oracle, sp)
    monkeypatch.setenv("CCAL_GENERAL_GRAM", "mfma")
    gm = Problem.from_synth(_legacy_ctx(), sp)                          # the matrix-core pair lives in libccal_hip_legacy.so only
    gm.build_normal(sp.intr0, sp.poses0, sp.extr0)                      # workspace (and the choice) made under the switch
    monkeypatch.delenv("CCAL_GENERAL_GRAM")
    for lam in (0.0, 1e-3):
        S, b, cost = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        Sm, bm, costm = gm.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        for (S_, b_, c_) in ((S, b, cost), (Sm, bm, costm)):
            assert abs(c_ - costo) <= 1e-12 * costo
            assert np.abs(S_ - So).max() <= 1e-9 * np.abs(So).max()
            assert np.abs(b_ - bo).max() <= 1e-9 * np.abs(bo).max()
        assert np.abs(S - S.T).max() == 0.0                             # mirrored from the lower triangle the engine keeps
        dc, dco = np.linalg.solve(S, -b), np.linalg.solve(So, -bo)
        assert np.abs(dc - dco).max() <= 1e-6 * np.abs(dco).max()
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        gp.apply_reference_bounds(); op.apply_reference_bounds(); gm.apply_reference_bounds()
        intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        intr_m, poses_m, extr_m, rep_m = gm.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        assert rep.status == rep_m.status == rep_o.status == 0
        assert rep.iterations == rep_m.iterations == rep_o.iterations
        assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
        assert abs(rep_m.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
        scale = np.maximum(np.abs(intr_o), 1e-3)
        assert (np.abs(intr - intr_o) / scale).max() <= 1e-6
        np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
        np.testing.assert_allclose(extr, extr_o, rtol=0, atol=1e-7)
        np.testing.assert_allclose(extr_m, extr_o, rtol=0, atol=1e-7)
        assert np.abs(extr - sp.extr_gt).max() < 5e-3                   # and the rig is recovered
    gm.close()


@pytest.mark.parametrize("model", ["ucm", "eucm", "kb4", "opencv5"])
@pytest.mark.parametrize("one_focal", [False, True])
def test_two_equal_cameras_elimination_kernels(dev_ctx, oracle, model, one_focal, monkeypatch):
    """A rig of two cameras of one model takes, above 1 000 slots, the compile-time structured elimination k_schurq (four or eight lanes
    per slot; UCM and EUCM: P_eff = 4 .. 6) and ONE launch of the register Gram kernel for both cameras.  Forced here on a small rig - 0.4 rad
    extrinsic rotation, slots seen by one camera only, ragged corner sets - against the oracle and against
    the generic pair (k_schur, one Gram launch per camera): normal equations, GN, LM."""
    sp = synth.make_rig(37, (model, model), RIG_EXTR[2], xy_same_focal=one_focal, seed=0xC0DE)
    seen = np.zeros((sp.n_slots, sp.n_cams), bool)
    seen[sp.obs_slot, sp.obs_cam] = True
    assert (seen.sum(1) == 1).any() and (~seen[:, 0]).any() and (~seen[:, 1]).any()
    op = oracle.OracleProblem.from_synth(sp)
    probs = {}
    # k_schurq in both forms: 16 slots per wavefront with four lanes each, 8 slots with eight lanes each
    for name, env in (("fast", {"CCAL_SCHURQ": "1", "CCAL_MERGE_GRAM": "1", "CCAL_SCHURQ_SLOTS": "16"}),
                      ("fast8", {"CCAL_SCHURQ": "1", "CCAL_MERGE_GRAM": "1", "CCAL_SCHURQ_SLOTS": "8"}),
                      ("generic", {"CCAL_SCHURQ": "0", "CCAL_MERGE_GRAM": "0"}),
                      ("merged_only", {"CCAL_SCHURQ": "0", "CCAL_MERGE_GRAM": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = Problem.from_synth(dev_ctx, sp)                             # the forcing switches belong to the second library
        g.build_normal(sp.intr0, sp.poses0, sp.extr0)                   # the workspace (and the choice of kernels) is made here
        probs[name] = g
    monkeypatch.delenv("CCAL_SCHURQ"); monkeypatch.delenv("CCAL_MERGE_GRAM"); monkeypatch.delenv("CCAL_SCHURQ_SLOTS")
    for lam in (0.0, 1e-3):
        So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        for g in probs.values():
            S, b, cost = g.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
            assert abs(cost - costo) <= 1e-12 * costo
            assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
            assert np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
        # the merged launch forms the same records as one launch per camera: bit-identical systems
        np.testing.assert_array_equal(probs["merged_only"].build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)[0],
                                      probs["generic"].build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)[0])
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        op.apply_reference_bounds()
        intr_o, poses_o, extr_o, rep_o = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        for g in probs.values():
            g.apply_reference_bounds()
            intr, poses, extr, rep = g.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
            assert rep.status == rep_o.status == 0 and rep.iterations == rep_o.iterations
            assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
            assert (np.abs(intr - intr_o) / np.maximum(np.abs(intr_o), 1e-3)).max() <= 1e-6
            np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
            np.testing.assert_allclose(extr, extr_o, rtol=0, atol=1e-7)
    # k_schurq's record buffers keep HOLES (slots one camera did not see) and never-written rows of the partial sums that must
    # stay zero for good: after GN and LM solves on both parameter sets the normal equations still match the oracle's
    So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=1e-3)
    for g in probs.values():
        S, b, cost = g.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=1e-3)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    for g in probs.values():
        g.close()


@pytest.mark.parametrize("model,frames", [("eucm", 2600), ("kb4", 2100), ("eucm", 300)])
def test_fused_elimination_equals_separate_launch(gpu_ctx, dev_ctx, oracle, model, frames, monkeypatch):
    """The Gram kernels eliminate their frames' pose blocks in their own tail (k_gram1w from 2 000 frames up for UCM / EUCM,
    k_gram1v otherwise; re-elimination groups of LM run there too).  Against the separate elimination launch
    (CCAL_FUSE_ELIM=0: k_schur1m) and the oracle, from a poor start that makes LM reject steps and miss speculations."""
    sp = synth.make_problem(frames, model, init_perturb=0.6, outlier_frac=0.03, seed=5, ragged=True)
    gp, op = _pair(gpu_ctx, oracle, sp)
    monkeypatch.setenv("CCAL_FUSE_ELIM", "0")
    gs = Problem.from_synth(dev_ctx, sp)                    # the separate launch (k_schur1m) lives in the second library only
    gs.build_normal(sp.intr0, sp.poses0)                    # workspace (and the choice) made under the switch
    monkeypatch.delenv("CCAL_FUSE_ELIM")
    for p_ in (gp, gs, op):
        p_.apply_reference_bounds()
    for lam in (0.0, 1e-2):
        S, b, c = gp.build_normal(sp.intr0, sp.poses0, lam=lam)
        S2, b2, c2 = gs.build_normal(sp.intr0, sp.poses0, lam=lam)
        So, bo, co = op.build_normal(sp.intr0, sp.poses0, lam=lam)
        for (S_, b_, c_) in ((S, b, c), (S2, b2, c2)):
            assert abs(c_ - co) <= 1e-12 * co
            assert np.abs(S_ - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b_ - bo).max() <= 1e-9 * np.abs(bo).max()
    rejected = 0
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        o = default_opts(method)
        intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=o, raise_on_error=False)
        intr2, poses2, _, rep2 = gs.solve(sp.intr0, sp.poses0, opts=o, raise_on_error=False)
        intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=o)
        sig = lambda r: (r.status, r.iterations, r.lm_accepted, r.lm_rejected)
        assert sig(rep) == sig(rep2) == sig(rep_o)
        if rep_o.status == 0:
            assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
            assert abs(rep2.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
            np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
            np.testing.assert_allclose(poses2, poses_o, rtol=0, atol=1e-7)
        if method == _ffi.METHOD_LM:
            assert (rep.lm_spec_hits, rep.lm_spec_misses) == (rep2.lm_spec_hits, rep2.lm_spec_misses)
            rejected = rep.lm_rejected + rep.lm_spec_misses
    if (model, frames) == ("eucm", 2600):
        assert rejected >= 1                                 # the re-elimination path of k_gram1w did run
    gs.close()


@pytest.mark.parametrize("model,one_focal,frames", [("eucm", False, 150), ("kb4", True, 150), ("opencv5", False, 2300)])
def test_matrix_core_gram_switch(gpu_ctx, oracle, model, one_focal, frames, monkeypatch):
    """CCAL_GRAM=mfma: the single-camera loop through the matrix-core Gram kernel (k_gram1: v_mfma_f64_16x16x4_f64 on LDS-staged
    rows) and the separate elimination launch - the second implementation behind the developer switch stays correct."""
    sp = synth.make_problem(frames, model, xy_same_focal=one_focal, ragged=True, outlier_frac=0.02)
    gp, op = _pair(_legacy_ctx(), oracle, sp)                           # k_gram1 is compiled into libccal_hip_legacy.so only
    monkeypatch.setenv("CCAL_GRAM", "mfma")
    S, b, c = gp.build_normal(sp.intr0, sp.poses0, lam=1e-3)
    So, bo, co = op.build_normal(sp.intr0, sp.poses0, lam=1e-3)
    assert abs(c - co) <= 1e-12 * co
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
        intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(method))
        assert (rep.status, rep.iterations) == (rep_o.status, rep_o.iterations) and rep.status == 0
        assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
        np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
