"""GPU parity at the sizes BASELINE.json names.

configs[1]  "Synthetic 1 000 frames x 144 corners, EUCM" and
configs[2]  "Same synthetic set [1 000 frames x 144 corners], KB4 and OPENCV5 models": mode E on EVERY corner,
            the reduced normal equations, and the GN / LM solves against the oracle, both focal modes; plus sampled
            frames of a 10 000-frame mode-E pass.
configs[0]  "TUM-VI dataset-calib-cam1 cam0, EUCM, end to end": the dataset is not in the image, so this is a
            STAND-IN -- a TUM-VI-shaped single-camera session (512 x 512 EUCM of data/eucm.json, 600 frames, 24..144
            detected corners per frame in HashMap order) driven through the steps of src/bin/camera_calibration.rs:
            calib_camera (:262-263, src/util.rs:384-490) -> calib_all_camera_with_extrinsics with ONE camera (always
            run, :267) -> validation (:299, src/util.rs:721-795), on the GPU through the API mirror and on the oracle
            through the same steps.

Tolerances (fp64) as in test_gpu_eval.py / test_gpu_normal.py: |dr| <= 1e-10 px, |dJ| <= 1e-11 max(1,|J|), S / b 1e-9
of their largest entry, cost 1e-12 relative, same iteration count, final cost 1e-9, converged intrinsics 1e-6 relative
(north_star), validation statistics 1e-9 px."""
import dataclasses

import numpy as np
import pytest

from camera_intrinsic_calibration_rs_amd import _ffi, api, synth
from camera_intrinsic_calibration_rs_amd.engine import Problem, default_opts

pytestmark = pytest.mark.gpu

# configs[1] as written (1 000 x 144, EUCM: every corner + normal equations + solves) and configs[2] (KB4, OPENCV5)
_CFG2 = [("eucm", False), ("eucm", True), ("kb4", False), ("kb4", True), ("opencv5", False), ("opencv5", True)]


@pytest.fixture(scope="module")
def cfg2_problems():
    cache = {}

    def get(model, one_focal):
        key = (model, one_focal)
        if key not in cache:
            cache[key] = synth.make_problem(1000, model, xy_same_focal=one_focal, outlier_frac=0.01)
        return cache[key]
    return get


@pytest.mark.parametrize("model,one_focal", _CFG2)
def test_config2_mode_e_every_corner(gpu_ctx, oracle, cfg2_problems, model, one_focal):
    sp = cfg2_problems(model, one_focal)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    assert gp.n_corners == 144000
    for apply_loss in (False, True):
        r, J = gp.eval(sp.intr0, sp.poses0, apply_loss=apply_loss)
        ro, Jo = op.eval(sp.intr0, sp.poses0, apply_loss=apply_loss, threads=8)
        assert np.isfinite(r).all() and np.isfinite(J).all()
        assert np.abs(r - ro).max() <= 1e-10
        assert (np.abs(J - Jo) / np.maximum(1.0, np.abs(Jo))).max() <= 1e-11
    gp.close()


@pytest.mark.parametrize("model,one_focal", _CFG2)
@pytest.mark.parametrize("lam", [0.0, 1e-3])
def test_config2_build_normal(gpu_ctx, oracle, cfg2_problems, model, one_focal, lam):
    """The register Gram kernels (the default for every model) with the elimination fused into their tail, 1 000 frames."""
    sp = cfg2_problems(model, one_focal)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    S, b, cost = gp.build_normal(sp.intr0, sp.poses0, lam=lam)
    So, bo, costo = op.build_normal(sp.intr0, sp.poses0, lam=lam)
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
    assert np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    dc, dco = np.linalg.solve(S, -b), np.linalg.solve(So, -bo)
    assert np.abs(dc - dco).max() <= 1e-6 * np.abs(dco).max()
    gp.close()


@pytest.mark.parametrize("model,one_focal", _CFG2)
@pytest.mark.parametrize("method", [_ffi.METHOD_GN, _ffi.METHOD_LM])
def test_config2_solve(gpu_ctx, oracle, cfg2_problems, model, one_focal, method):
    sp = cfg2_problems(model, one_focal)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    gp.apply_reference_bounds(); op.apply_reference_bounds()
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
    intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(method))
    assert rep.status == rep_o.status == 0
    assert (rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)
    assert abs(rep.initial_cost - rep_o.initial_cost) <= 1e-12 * rep_o.initial_cost
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    scale = np.maximum(np.abs(intr_o[0, :P]), 1e-3)
    assert (np.abs(intr[0, :P] - intr_o[0, :P]) / scale).max() <= 1e-6
    np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
    assert np.abs(intr[0, :4] / sp.intr_gt[0, :4] - 1).max() < 2e-3
    gp.close()


@pytest.mark.parametrize("model,one_focal", [("kb4", False), ("opencv5", False), ("kb4", True), ("eucm", False), ("eucm", True), ("ucm", False),
                                             ("opencv5", True)])
def test_config2_models_at_headline_size_sampled(gpu_ctx, oracle, model, one_focal):
    """10 000 frames x 144 corners, every model: sampled frames of the GPU pass equal the oracle on those frames, and the reduced
    normal equations equal the oracle's on all 10 000 frames - the size at which UCM / EUCM / OPENCV5 take k_gram2 (rows traded
    between the halves of the wavefront, 12 lanes per frame) and KB4 k_gram1v with 6 lanes per frame."""
    sp = synth.make_problem(10000, model, xy_same_focal=one_focal)
    gp = Problem.from_synth(gpu_ctx, sp)
    D = gp.block_dim(0)
    r, J = gp.eval(sp.intr0, sp.poses0)
    assert np.isfinite(r).all() and np.isfinite(J).all()
    J = J.reshape(-1, 2, D)
    for f in (0, 3, 5000, 8191, 9999):
        sub = sp.shard(f, 10000)
        ro, Jo = oracle.OracleProblem.from_synth(sub).eval(sp.intr0, sub.poses0)
        sl = slice(f * 144, (f + 1) * 144)
        assert np.abs(r[sl] - ro).max() <= 1e-10
        assert (np.abs(J[sl].ravel() - Jo) / np.maximum(1.0, np.abs(Jo))).max() <= 1e-11
    # mode N at this size against the oracle on the same 10 000 frames (0.2 s of CPU)
    S, b, cost = gp.build_normal(sp.intr0, sp.poses0)
    So, bo, costo = oracle.OracleProblem.from_synth(sp).build_normal(sp.intr0, sp.poses0)
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    gp.close()


@pytest.mark.parametrize("model,one_focal", [("eucm", False), ("eucm", True), ("kb4", False), ("opencv5", False), ("ucm", True)])
def test_ragged_frames_at_headline_size(gpu_ctx, oracle, model, one_focal):
    """SURVEY 8(d)'s ragged variant at the headline frame count: 10 000 frames of 24 .. 144 corners (src/data_loader.rs:15,61-62), rows
    in random order.  The Gram launch bins the frames by corner count (its geometry differs from the full-frame case: its own parity
    case): sampled frames of mode E, the reduced normal equations on all frames against the oracle, and the same bits from a second
    build (the order of summation is fixed per problem)."""
    sp = synth.make_problem(10000, model, xy_same_focal=one_focal, ragged=True, seed=0xC0FFEE + 77)
    n = np.diff(sp.obs_offsets)
    assert n.min() >= 24 and n.max() <= 144 and 70 < n.mean() < 100
    gp = Problem.from_synth(gpu_ctx, sp)
    D = gp.block_dim(0)
    r, J = gp.eval(sp.intr0, sp.poses0)
    assert np.isfinite(r).all() and np.isfinite(J).all()
    J = J.reshape(-1, 2, D)
    for f in (0, 7, 4999, 8191, 9999, int(n.argmin()), int(n.argmax())):
        sub = sp.shard(f, 10000)
        ro, Jo = oracle.OracleProblem.from_synth(sub).eval(sp.intr0, sub.poses0)
        sl = slice(int(sp.obs_offsets[f]), int(sp.obs_offsets[f + 1]))
        assert np.abs(r[sl] - ro).max() <= 1e-10
        assert (np.abs(J[sl].ravel() - Jo) / np.maximum(1.0, np.abs(Jo))).max() <= 1e-11
    for lam in (0.0, 1e-3):
        S, b, cost = gp.build_normal(sp.intr0, sp.poses0, lam=lam)
        So, bo, costo = oracle.OracleProblem.from_synth(sp).build_normal(sp.intr0, sp.poses0, lam=lam)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
        S2, b2, cost2 = gp.build_normal(sp.intr0, sp.poses0, lam=lam)
        np.testing.assert_array_equal(S, S2); np.testing.assert_array_equal(b, b2); assert cost == cost2
    # the solve on the ragged set: Gauss-Newton and LM recover the ground truth, twice the same bits
    for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
        i1, p1, _, r1 = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
        i2, p2, _, r2 = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
        assert r1.status == r2.status == 0 and r1.iterations == r2.iterations
        np.testing.assert_array_equal(i1, i2); np.testing.assert_array_equal(p1, p2)
        assert np.abs(i1[0, :4] / sp.intr_gt[0, :4] - 1).max() < 2e-3
    gp.close()


@pytest.mark.parametrize("frames,model", [(8000, "eucm"), (16000, "ucm"), (12000, "eucm"), (7300, "ucm"), (4200, "eucm"), (2600, "ucm")])
def test_ragged_plans_of_other_sizes(gpu_ctx, oracle, frames, model):
    """The planner's other shapes (csrc/ccal_kernels_gram2.hip: gram2_bin_plan): 4 200 .. 8 000 ragged frames take ONE folded bin of 16 lanes, 16 000 one
    of 8 lanes, 2 600 one of 32 (larger half of the frames largest first, smaller half smallest first), 12 000 equalised bins: the reduced normal equations
    against the oracle, the same bits twice, and a Gauss-Newton solve that ends where the oracle's does."""
    sp = synth.make_problem(frames, model, ragged=True, seed=0xF01D + frames)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    for lam in (0.0, 1e-3):
        S, b, cost = gp.build_normal(sp.intr0, sp.poses0, lam=lam)
        So, bo, costo = op.build_normal(sp.intr0, sp.poses0, lam=lam)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
        S2, b2, cost2 = gp.build_normal(sp.intr0, sp.poses0, lam=lam)
        np.testing.assert_array_equal(S, S2); np.testing.assert_array_equal(b, b2); assert cost == cost2
    oracle.set_solve_threads(8)
    try:
        io, po, _, ro = op.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN))
    finally:
        oracle.set_solve_threads(1)
    ig, pg, _, rg = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_GN))
    assert rg.status == ro.status == 0 and rg.iterations == ro.iterations
    assert abs(rg.final_cost - ro.final_cost) <= 1e-9 * ro.final_cost
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    assert (np.abs(ig[0, :P] - io[0, :P]) / np.maximum(np.abs(io[0, :P]), 1e-3)).max() <= 1e-6
    np.testing.assert_allclose(pg, po, rtol=0, atol=1e-7)
    gp.close()


@pytest.mark.parametrize("model", ["eucm", "kb4"])
@pytest.mark.parametrize("method", [_ffi.METHOD_GN, _ffi.METHOD_LM])
def test_ragged_solve_against_the_oracle(gpu_ctx, oracle, model, method):
    """2 500 ragged frames (the three-launch form with the binned Gram launch): same iteration count, accept / reject sequence and
    optimum as the oracle."""
    sp = synth.make_problem(2500, model, ragged=True, seed=0xAB5, outlier_frac=0.01)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    oracle.set_solve_threads(8)
    try:
        intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(method))
    finally:
        oracle.set_solve_threads(1)
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
    assert rep.status == rep_o.status == 0
    assert (rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    assert (np.abs(intr[0, :P] - intr_o[0, :P]) / np.maximum(np.abs(intr_o[0, :P]), 1e-3)).max() <= 1e-6
    np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
    gp.close()


@pytest.mark.parametrize("model,one_focal", [("eucm", False), ("ucm", True)])
@pytest.mark.parametrize("method", [_ffi.METHOD_GN, _ffi.METHOD_LM])
def test_headline_size_solve_against_the_oracle(gpu_ctx, oracle, model, one_focal, method):
    """9 800 frames x 144 corners: the size at which the single-camera solve runs single-launch groups of the two-wavefronts-per-SIMD
    kernel (k_gram2i: 245 workgroups of eight wavefronts) - same iteration count, accept / reject sequence and optimum as the oracle,
    host pointers and device-resident."""
    sp = synth.make_problem(9800, model, xy_same_focal=one_focal, outlier_frac=0.01, seed=0x9800)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    oracle.set_solve_threads(8)
    try:
        intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(method))
    finally:
        oracle.set_solve_threads(1)
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
    assert rep.status == rep_o.status == 0
    assert (rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    assert (np.abs(intr[0, :P] - intr_o[0, :P]) / np.maximum(np.abs(intr_o[0, :P]), 1e-3)).max() <= 1e-6
    np.testing.assert_allclose(poses, poses_o, rtol=0, atol=1e-7)
    gp.upload_params(sp.intr0, sp.poses0, sp.extr0)
    rd = gp.solve_dev(default_opts(method))
    i_d, p_d, _ = gp.download_params()
    assert (rd.status, rd.iterations) == (rep.status, rep.iterations)
    np.testing.assert_array_equal(i_d[0, :P], intr[0, :P]); np.testing.assert_array_equal(p_d, poses)
    gp.close()


@pytest.mark.parametrize("models", [("eucm", "eucm"), ("kb4", "eucm")])
def test_ragged_rig_binned_gram_launches(gpu_ctx, oracle, models):
    """A two-camera rig of 2 x 2 600 ragged observation frames: cameras of one model share ONE Gram launch whose list is sorted by corner
    count and cut into bins (k_gram2g), cameras of different models get a binned launch each: the reduced normal equations against
    the oracle, the same bits twice, and a Gauss-Newton / LM solve with the oracle's iteration counts."""
    ext = np.zeros((2, 6)); ext[1] = [0.05, -0.2, 0.1, 0.1, -0.02, 0.03]
    if models[0] == models[1]:
        sp = synth.make_problem(2600, models[0], n_cams=2, ragged=True, seed=0xA11)
    else:
        sp = synth.make_rig(3200, list(models), ext, seed=0xA12, drop_frac=0.15)
    n = np.diff(sp.obs_offsets)
    assert n.max() > 2 * n.min() and len(n) >= 4000
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    for lam in (0.0, 1e-3):
        S, b, cost = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        So, bo, costo = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
        S2, b2, _ = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        np.testing.assert_array_equal(S, S2); np.testing.assert_array_equal(b, b2)
    oracle.set_solve_threads(8)
    try:
        for method in (_ffi.METHOD_GN, _ffi.METHOD_LM):
            io_, po_, eo_, ro = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
            ig, pg, eg, rg = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
            assert (rg.status, rg.iterations) == (ro.status, ro.iterations) == (0, ro.iterations)
            assert abs(rg.final_cost - ro.final_cost) <= 1e-9 * ro.final_cost
            np.testing.assert_allclose(eg, eo_, rtol=0, atol=1e-7)
    finally:
        oracle.set_solve_threads(1)
    gp.close()


def test_ragged_lm_with_rejected_steps_in_the_binned_launch(gpu_ctx, oracle):
    """LM from a bad start on 3 000 ragged frames: rejected steps and missed speculations run the re-elimination groups of the binned
    Gram kernel (records indexed by position in the sorted table) - same accept / reject sequence as the oracle."""
    sp = synth.make_problem(3000, "eucm", ragged=True, outlier_frac=0.05, init_perturb=0.8, seed=1)
    gp = Problem.from_synth(gpu_ctx, sp)
    op = oracle.OracleProblem.from_synth(sp)
    oracle.set_solve_threads(8)
    try:
        intr_o, poses_o, _, rep_o = op.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    finally:
        oracle.set_solve_threads(1)
    intr, poses, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(_ffi.METHOD_LM))
    assert rep.status == rep_o.status
    assert (rep.iterations, rep.lm_accepted, rep.lm_rejected) == (rep_o.iterations, rep_o.lm_accepted, rep_o.lm_rejected)
    assert rep.lm_rejected + rep.lm_spec_misses > 0          # the re-elimination path ran
    assert abs(rep.final_cost - rep_o.final_cost) <= 1e-9 * rep_o.final_cost
    np.testing.assert_allclose(intr[0, :6], intr_o[0, :6], rtol=1e-6)
    gp.close()


# ---------------------------------------------------------------------------------------------------------------
# configs[0] stand-in
# ---------------------------------------------------------------------------------------------------------------
def _session(n_frames=600, seed=0x7A11):
    """TUM-VI-shaped session: ragged frames (24..144 corners, src/data_loader.rs:15), corner rows in random order."""
    return synth.make_problem(n_frames, "eucm", seed=seed, ragged=True, noise_px=0.1)


def _board_ids(sp):
    """Board corner id of every row (tag_id * 4 + corner, src/data_loader.rs:50) recovered from the f32 board points."""
    board = synth.default_board()
    key = {(float(p[0]), float(p[1])): i for i, p in enumerate(board)}
    return np.array([key[(float(x), float(y))] for x, y in sp.p3d[:, :2]], dtype=np.int64)


def _frames_with_ids(sp, ids, order="hash"):
    """`Vec<Option<FrameFeature>>` with the real corner ids; frames 7 and 300 have no detections (None)."""
    out = [None] * sp.n_slots
    for o in range(sp.n_obs):
        a, b = int(sp.obs_offsets[o]), int(sp.obs_offsets[o + 1])
        feats = {int(ids[k]): api.FeaturePoint(tuple(sp.p2d[k]), tuple(sp.p3d[k])) for k in range(a, b)}
        out[int(sp.obs_slot[o])] = api.FrameFeature(o * 50_000_000, (512, 512), feats)
    out[7] = None
    out[300] = None
    return out


def _sorted_copy(sp, ids):
    """The same session with every frame's rows sorted by corner id (what api._flatten hands to the engine)."""
    perm = np.concatenate([a + np.argsort(ids[a:b], kind="stable")
                           for a, b in zip(sp.obs_offsets[:-1], sp.obs_offsets[1:])])
    return dataclasses.replace(sp, p3d=sp.p3d[perm], p2d=sp.p2d[perm])


def test_config0_standin_order_invariance(gpu_ctx, oracle):
    """The reference adds residual blocks in HashMap order (src/util.rs:407): the optimum must not depend on it.
    Engine on the rows as generated (random order) vs the rows sorted by corner id: same iteration count, optimum equal
    to rounding; and the random-order run equals the oracle on the same rows."""
    sp = _session()
    ids = _board_ids(sp)
    assert len(set(np.diff(sp.obs_offsets))) > 50                     # really ragged
    assert (np.diff(ids[:int(sp.obs_offsets[1])]) < 0).any()          # really unordered
    sps = _sorted_copy(sp, ids)
    g1, g2 = Problem.from_synth(gpu_ctx, sp), Problem.from_synth(gpu_ctx, sps)
    for g in (g1, g2):
        g.apply_reference_bounds()
    i1, p1, _, r1 = g1.solve(sp.intr0, sp.poses0)
    i2, p2, _, r2 = g2.solve(sps.intr0, sps.poses0)
    assert r1.status == r2.status == 0 and r1.iterations == r2.iterations
    assert np.abs(i1[0, :6] / i2[0, :6] - 1).max() <= 1e-10
    np.testing.assert_allclose(p1, p2, rtol=0, atol=1e-10)
    assert abs(r1.final_cost / r2.final_cost - 1) <= 1e-11
    op = oracle.OracleProblem.from_synth(sp)
    op.apply_reference_bounds()
    io, po, _, ro = op.solve(sp.intr0, sp.poses0)
    assert (r1.status, r1.iterations) == (ro.status, ro.iterations)
    assert np.abs(i1[0, :6] / io[0, :6] - 1).max() <= 1e-6
    assert abs(r1.final_cost - ro.final_cost) <= 1e-9 * ro.final_cost
    g1.close(); g2.close()


@pytest.mark.parametrize("one_focal", [False, True])
def test_config0_standin_session_pipeline(gpu_ctx, oracle, one_focal):
    """calib_camera (poses initialised inside, src/util.rs:418-436) -> calib_all_camera_with_extrinsics with one camera
    -> validation, GPU API vs the oracle driven through the same steps from the same initial poses."""
    sp = _session()
    ids = _board_ids(sp)
    frames = _frames_with_ids(sp, ids)
    cam0 = api.GenericModel("eucm", sp.intr0[0, :6] if not one_focal else
                            np.concatenate([[sp.intr0[0, 0]] * 2, sp.intr0[0, 2:6]]), 512, 512)
    init = api.init_frame_poses(frames, cam0, ctx=gpu_ctx)
    valid = sorted(init.keys())
    assert len(valid) == 598 and 7 not in init and 300 not in init
    # the initialisation sits in the basin: a few centimetres / a few hundredths of a radian from ground truth
    p_init = np.stack([init[i].as6() for i in valid])
    assert np.abs(p_init[:, 3:] - sp.poses_gt[valid, 3:]).max() < 0.12

    # --- GPU, the steps of src/bin/camera_calibration.rs:262-299
    res = api.calib_camera(frames, cam0, one_focal, 0, False, init, ctx=gpu_ctx)
    assert res is not None
    cam1, rt1 = res
    res2 = api.calib_all_camera_with_extrinsics([cam1], [api.RvecTvec.from6(np.zeros(6))], [rt1], [frames],
                                                one_focal, 0, False, ctx=gpu_ctx)
    assert res2 is not None
    (cam2,), t_i_0, board = res2
    a_gpu, m_gpu = api.validation(0, cam2, board, frames, ctx=gpu_ctx)

    # --- oracle, the same steps on the same rows (api._flatten sorts a frame's rows by corner id)
    sps = _sorted_copy(sp, ids)
    keep = np.isin(sps.obs_slot, valid)
    rows = np.concatenate([np.arange(sps.obs_offsets[o], sps.obs_offsets[o + 1]) for o in np.nonzero(keep)[0]])
    cnt = np.diff(sps.obs_offsets)[keep]
    sub = dataclasses.replace(sps, n_slots=len(valid), obs_cam=sps.obs_cam[keep], xy_same_focal=one_focal,
                              obs_slot=np.arange(len(valid), dtype=np.int32),
                              obs_offsets=np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64),
                              p3d=sps.p3d[rows], p2d=sps.p2d[rows])
    op = oracle.OracleProblem.from_synth(sub)
    op.apply_reference_bounds()
    intr0 = np.zeros((1, synth.PMAX)); intr0[0, :6] = cam0.params()
    i1, p1, _, rep1 = op.solve(intr0, p_init)
    assert rep1.status == 0
    i2, p2, _, rep2 = op.solve(i1, p1)                                     # calib_all_camera_with_extrinsics, one camera
    assert rep2.status == 0
    a_o, m_o = op.validation(0, i2, p2)

    got1, got2 = cam1.params(), cam2.params()
    assert np.abs(got1 / i1[0, :6] - 1).max() <= 1e-6
    assert np.abs(got2 / i2[0, :6] - 1).max() <= 1e-6
    np.testing.assert_allclose(np.stack([board[i].as6() for i in valid]), p2, rtol=0, atol=1e-7)
    assert abs(a_gpu - a_o) <= 1e-9 and abs(m_gpu - m_o) <= 1e-9
    if one_focal:
        assert got2[0] == got2[1]
    # and it is the right answer: the session's ground truth within noise, errors at the TUM-VI level the reference
    # shows (data/rerun_logs.jpg: ~0.09-0.10 px)
    assert np.abs(got2[:4] / sp.intr_gt[0, :4] - 1).max() < 2e-3
    assert 0.08 < m_gpu < 0.16 and 0.08 < a_gpu < 0.16
