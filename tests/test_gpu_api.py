"""GPU: the reference-shaped API (api.py) end to end through the C ABI -- these read like the
reference's own tests (tests/optimization_test.rs) and its call sites (src/util.rs)."""
import numpy as np
import pytest

from camera_intrinsic_calibration_rs_amd import api, synth

pytestmark = pytest.mark.gpu


def test_reprojection_factor(gpu_ctx):
    """tests/optimization_test.rs:36-80, statement for statement."""
    w, h = 640, 480
    cam_params = np.array([500.0, 500.0, 320.0, 240.0, 0.5])
    model = api.GenericModel("ucm", cam_params, w, h)
    p3d = np.array([1.0, 2.0, 10.0], dtype=np.float32)
    # project using the model: residual against p2d = 0 is the projection itself
    probe = api.ReprojectionFactor.new(model, p3d, np.zeros(2, dtype=np.float32), False, gpu_ctx)
    p2d_f64 = probe.residual_func([cam_params, np.zeros(3), np.zeros(3)])
    np.testing.assert_allclose(p2d_f64, [369.3901531919198, 338.7803063838396], rtol=1e-14)
    p2d = p2d_f64.astype(np.float32)
    factor = api.ReprojectionFactor.new(model, p3d, p2d, False, gpu_ctx)
    residual = factor.residual_func([cam_params, np.zeros(3), np.zeros(3)])
    assert np.linalg.norm(residual) < 1e-4, "Residual should be zero at GT"
    residual_bad = factor.residual_func([cam_params, np.zeros(3), np.array([0.1, 0.0, 0.0])])
    assert np.linalg.norm(residual_bad) > 1e-3, "Residual should be non-zero for bad params"


def test_factor_jacobian_matches_dual_numbers(gpu_ctx, oracle):
    model = api.GenericModel("eucm", synth.GT_PARAMS[synth.MODEL_EUCM], 512, 512)
    th = np.array(synth.GT_PARAMS[synth.MODEL_EUCM]); th_of = np.delete(th, 1)
    pose = np.array([2.9, 0.1, -0.2, -0.3, 0.35, 0.9]); ext = np.array([0.01, -0.02, 0.005, -0.101, 0.002, 0.001])
    p3d = [0.2, -0.3, 0.0]; p2d = [200.0, 260.0]
    for of, p in ((False, th), (True, th_of)):
        r, J = api.ReprojectionFactor.new(model, p3d, p2d, of, gpu_ctx).residual_func([p, pose[:3], pose[3:]], jacobian=True)
        ro, Jo = oracle.factor(1, of, p, pose, p3d, p2d)
        np.testing.assert_allclose(r, ro, atol=1e-10); np.testing.assert_allclose(J, Jo, rtol=1e-11, atol=1e-11)
        r, J = api.OtherCamReprojectionFactor.new(model, p3d, p2d, of, gpu_ctx).residual_func(
            [p, pose[:3], pose[3:], ext[:3], ext[3:]], jacobian=True)
        ro, Jo = oracle.factor(1, of, p, pose, p3d, p2d, pose1=ext)
        np.testing.assert_allclose(r, ro, atol=1e-10); np.testing.assert_allclose(J, Jo, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("model,one_focal,disabled,fixed_focal",
                         [("eucm", False, 0, False), ("eucm", True, 0, False), ("opencv5", True, 1, True)])
def test_calib_camera(gpu_ctx, oracle, model, one_focal, disabled, fixed_focal):
    """util::calib_camera on synthetic frames vs the oracle driven through the same steps
    (bounds, disabled distortion, GN, optional fixed-focal second solve)."""
    sp = synth.make_problem(24, model, xy_same_focal=one_focal)
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    frames = api.frames_from_synth(sp)
    frames[5] = None                                             # a frame without detections
    cam0 = api.GenericModel(model, sp.intr0[0, :P], 512, 512)
    init = {i: api.RvecTvec.from6(sp.poses0[i]) for i in range(sp.n_slots) if i != 5}
    res = api.calib_camera(frames, cam0, one_focal, disabled, fixed_focal, init, ctx=gpu_ctx)
    assert res is not None
    model, poses = res
    assert sorted(poses.keys()) == [i for i in range(24) if i != 5]
    # oracle, same recipe
    import dataclasses
    keep = sp.obs_slot != 5
    rows = np.concatenate([np.arange(sp.obs_offsets[o], sp.obs_offsets[o + 1]) for o in np.nonzero(keep)[0]])
    sub = dataclasses.replace(sp, n_slots=23, obs_cam=sp.obs_cam[keep], obs_slot=np.arange(23, dtype=np.int32),
                              obs_offsets=np.arange(24, dtype=np.int64) * 144, p3d=sp.p3d[rows], p2d=sp.p2d[rows])
    op = oracle.OracleProblem.from_synth(sub)
    op.apply_reference_bounds()
    intr = sp.intr0.copy(); op.disable_distortions(disabled, intr)
    p0 = np.delete(sp.poses0, 5, axis=0)
    intr, p1, _, rep = op.solve(intr, p0)
    if fixed_focal:
        op.fix_param(0, 0); intr[0, 0] = sp.intr0[0, 0]; intr[0, 1] = intr[0, 0]
        intr, p1, _, rep = op.solve(intr, p1)
    got = model.params()
    assert (np.abs(got - intr[0, :P]) / np.maximum(np.abs(intr[0, :P]), 1e-3)).max() < 1e-6
    if one_focal:
        assert got[0] == got[1]
    if fixed_focal:
        assert got[0] == sp.intr0[0, 0]
    if disabled:
        assert got[P - 1] == 0.0
    np.testing.assert_allclose(np.stack([poses[i].as6() for i in sorted(poses)]), p1, atol=1e-7)


def test_calib_all_camera_with_extrinsics(gpu_ctx, oracle):
    """util::calib_all_camera_with_extrinsics, 2 cameras; camera 1 misses some frames, camera 0 others."""
    sp = synth.make_problem(16, "eucm", n_cams=2)
    f0 = api.frames_from_synth(sp, 0); f1 = api.frames_from_synth(sp, 1)
    cams = [api.GenericModel("eucm", sp.intr0[c, :6], 512, 512) for c in range(2)]
    t_i_0 = [api.RvecTvec.from6(np.zeros(6)), api.RvecTvec.from6(sp.extr0[1])]
    rt0 = {i: api.RvecTvec.from6(sp.poses0[i]) for i in range(16) if i not in (3, 4)}
    rt1 = {i: t_i_0[1].compose(api.RvecTvec.from6(sp.poses0[i])) for i in range(16) if i not in (9,)}
    res = api.calib_all_camera_with_extrinsics(cams, t_i_0, [rt0, rt1], [f0, f1], False, 0, False, ctx=gpu_ctx)
    assert res is not None
    models, t_out, board = res
    assert sorted(board.keys()) == list(range(16))
    assert t_out[0].as6().tolist() == [0.0] * 6
    # the oracle through the same steps: same problem description and starting point (src/util.rs:576-651), the
    # reference's bounds, Gauss-Newton
    d, keep, slots, intr0, poses0, extr0 = api._joint_problem_inputs(cams, t_i_0, [rt0, rt1], [f0, f1], False)
    op = oracle.OracleProblem(d, keep)
    op.apply_reference_bounds()
    intr_o, poses_o, extr_o, rep_o = op.solve(intr0, poses0, extr0)
    assert rep_o.status == 0
    for c in range(2):
        assert np.abs(models[c].params() / intr_o[c, :6] - 1).max() <= 1e-6
    np.testing.assert_allclose(t_out[1].as6(), extr_o[1], rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.stack([board[i].as6() for i in slots]), poses_o, rtol=0, atol=1e-7)
    # ground truth is recovered within noise: baseline 101 mm to < 0.5 mm, focal to 0.5 %
    assert np.abs(np.array(t_out[1].tvec) - sp.extr_gt[1, 3:]).max() < 5e-4
    for c in range(2):
        assert abs(models[c].params()[0] / sp.intr_gt[c, 0] - 1) < 5e-3
    # reprojection statistics (validation) per camera with the saved-pose convention T_i_0 * T_0_b
    for c, fr in ((0, f0), (1, f1)):
        saved = {k: t_out[c].compose(v) for k, v in board.items() if fr[k] is not None and k in (rt0, rt1)[c]}
        a, m = api.validation(c, models[c], saved, fr, ctx=gpu_ctx)
        assert 0.05 < m < 0.25 and 0.05 < a < 0.25


def test_validation_matches_oracle(gpu_ctx, oracle):
    sp = synth.make_problem(10, "kb4")
    frames = api.frames_from_synth(sp)
    model = api.GenericModel("kb4", sp.intr_gt[0, :8], 512, 512)
    poses = {i: api.RvecTvec.from6(sp.poses_gt[i]) for i in range(10)}
    a, m = api.validation(0, model, poses, frames, ctx=gpu_ctx)
    ao, mo = oracle.OracleProblem.from_synth(sp).validation(0, sp.intr_gt, sp.poses_gt)
    assert abs(a - ao) < 1e-11 and abs(m - mo) < 1e-11


def test_init_ucm_two_frames(gpu_ctx, oracle):
    """util::init_ucm (src/util.rs:287-378): [f, alpha] + two poses from two frames, principal point pinned at
    the image centre, then calib_camera(xy_same_focal = true) on the same two frames - against ground truth and against
    the oracle driven through the same two solves."""
    sp = synth.make_problem(2, "ucm", seed=77)
    frames = api.frames_from_synth(sp)
    gt = sp.intr_gt[0]
    m = api.init_ucm(frames[0], frames[1], api.RvecTvec.from6(sp.poses0[0]), api.RvecTvec.from6(sp.poses0[1]),
                     init_f=gt[0] * 1.3, init_alpha=0.4, fixed_focal=False, ctx=gpu_ctx)
    assert m is not None and m.kind == "ucm"
    p = m.params()
    assert p[0] == p[1]
    assert abs(p[0] / gt[0] - 1) < 0.03 and abs(p[4] - gt[4]) < 0.05         # two frames only: a few percent
    assert abs(p[2] - gt[2]) < 5 and abs(p[3] - gt[3]) < 5
    # oracle, the same recipe: (1) UCM, one focal, cx / cy fixed at the image centre, f in [f0/3, 3 f0], alpha in [1e-6, 1]
    # (src/util.rs:305-348); (2) calib_camera(xy_same_focal = true) on the two frames with the reference's bounds (:365-371)
    import dataclasses
    f0 = gt[0] * 1.3
    sub = dataclasses.replace(sp, xy_same_focal=True)
    op = oracle.OracleProblem.from_synth(sub)
    op.fix_param(0, 1); op.fix_param(0, 2)
    op.set_bounds(0, 0, f0 / 3.0, f0 * 3.0); op.set_bounds(0, 3, 1e-6, 1.0)
    i0 = np.zeros((1, synth.PMAX)); i0[0, :5] = [f0, f0, 256.0, 256.0, 0.4]
    i1, p1, _, r1 = op.solve(i0, sp.poses0[:2])
    assert r1.status == 0
    op2 = oracle.OracleProblem.from_synth(sub)
    op2.apply_reference_bounds()
    i1[0, 1] = i1[0, 0]; i1[0, 2:4] = 256.0
    # calib_camera initialises the poses itself (src/util.rs:418-436): the same device initialisation for both sides
    init = api.init_frame_poses(frames, api.GenericModel("ucm", i1[0, :5], 512, 512), ctx=gpu_ctx)
    i2, _, _, r2 = op2.solve(i1, np.stack([init[0].as6(), init[1].as6()]))
    assert r2.status == 0
    assert np.abs(p / i2[0, :5] - 1).max() <= 1e-6


def test_multi_camera_pipeline_from_detections(gpu_ctx):
    """The reference's flow for two cameras, end to end on the GPU engine: per-camera calib_camera (poses
    initialised inside), init_camera_extrinsic, calib_all_camera_with_extrinsics, validation."""
    sp = synth.make_problem(20, "eucm", n_cams=2)
    frames = [api.frames_from_synth(sp, c) for c in range(2)]
    cams0 = [api.GenericModel("eucm", sp.intr0[c, :6], 512, 512) for c in range(2)]
    per_cam = [api.calib_camera(frames[c], cams0[c], False, 0, False, None, ctx=gpu_ctx) for c in range(2)]
    assert all(r is not None for r in per_cam)
    t_i_0 = api.init_camera_extrinsic([r[1] for r in per_cam])
    assert np.abs(np.array(t_i_0[1].tvec) - sp.extr_gt[1, 3:]).max() < 3e-3
    res = api.calib_all_camera_with_extrinsics([r[0] for r in per_cam], t_i_0, [r[1] for r in per_cam], frames,
                                               False, 0, False, ctx=gpu_ctx)
    assert res is not None
    models, t_out, board = res
    assert np.abs(np.array(t_out[1].tvec) - sp.extr_gt[1, 3:]).max() < 5e-4
    assert np.abs(np.array(t_out[1].rvec) - sp.extr_gt[1, :3]).max() < 2e-3
    for c in range(2):
        saved = {k: t_out[c].compose(v) for k, v in board.items()}
        a, m = api.validation(c, models[c], saved, frames[c], ctx=gpu_ctx)
        assert 0.05 < m < 0.25


# ---- convert_model (src/util.rs:224-282) ------------------------------------------------------------------
_EUCM_GT = [190.89618687183938, 190.87022285882367, 254.9375370481962, 256.86414483060787, 0.6283550447635853, 1.0458678747533083]
_KB4_GT = [190.9, 190.9, 255.0, 257.0, 0.003, 0.0007, -0.002, 0.0002]
_CV5_GT = [380.0, 380.0, 255.0, 257.0, -0.28, 0.07, 0.0002, 0.00002, 0.0]
_REF_BOUNDS = {
    "ucm": ([0, 0, 0, 0, 1e-6], [1e4, 1e4, 512, 512, 1.0]),
    "eucm": ([0, 0, 0, 0, 1e-6, 1e-6], [1e4, 1e4, 512, 512, 1.0, 100.0]),
    "kb4": ([0, 0, 0, 0, -1, -1, -1, -1], [1e4, 1e4, 512, 512, 1, 1, 1, 1]),
    "opencv5": ([0, 0, 0, 0, -1, -1, -1, -1, -1], [1e4, 1e4, 512, 512, 1, 1, 1, 1, 1]),
}


@pytest.mark.parametrize("src,src_p,tgt,tgt_p,disabled", [
    ("eucm", _EUCM_GT, "kb4", [0.0] * 8, 0),
    ("eucm", _EUCM_GT, "kb4", [0.0] * 8, 2),
    ("eucm", _EUCM_GT, "ucm", [0, 0, 0, 0, 0.6], 0),
    ("kb4", _KB4_GT, "eucm", [0, 0, 0, 0, 0.5, 1.0], 0),
    ("opencv5", _CV5_GT, "kb4", [0.0] * 8, 0),
    ("kb4", _KB4_GT, "opencv5", [0.0] * 9, 0),
    ("ucm", _EUCM_GT[:5], "kb4", [0.0] * 8, 0),
])
def test_convert_model_vs_oracle(gpu_ctx, oracle, src, src_p, tgt, tgt_p, disabled):
    """Device ModelConvertFactor fit against the oracle's dual-number restatement: same iteration count, same
    parameters (f64 tolerance 1e-8 relative / 1e-10 absolute: the sums run in a different order)."""
    s = api.GenericModel(src, src_p, 512, 512)
    t = api.GenericModel(tgt, tgt_p, 512, 512)
    lo, hi = _REF_BOUNDS[tgt]
    p_o, n, rc = oracle.convert_model(s.model_id, src_p, t.model_id, tgt_p, 512, 512, disabled, lo, hi)
    assert rc == 0 and n > 0
    out = api.convert_model(s, t, disabled, ctx=gpu_ctx)
    np.testing.assert_allclose(out.params(), p_o, rtol=1e-8, atol=1e-10)
    if disabled:
        assert (np.asarray(out.params())[-disabled:] == 0.0).all()


def test_convert_model_roundtrip_identity(gpu_ctx):
    """EUCM -> EUCM from a different start recovers the source exactly (size-independent property)."""
    s = api.GenericModel("eucm", _EUCM_GT, 512, 512)
    t = api.GenericModel("eucm", [1, 1, 1, 1, 0.5, 1.0], 512, 512)
    np.testing.assert_allclose(api.convert_model(s, t, ctx=gpu_ctx).params(), _EUCM_GT, rtol=1e-9)


def test_convert_model_size_mismatch():
    with pytest.raises(ValueError):
        api.convert_model(api.GenericModel("eucm", _EUCM_GT, 512, 512), api.GenericModel("kb4", [0.0] * 8, 640, 480))


def test_pinned_caller_buffers_give_the_same_bits(gpu_ctx):
    """ccal_pin_buffer: a pose array the caller pinned is read and written in place by ccal_solve (no staging copy) - large problems
    (the three-launch form stages 480 KB both ways otherwise), session sizes, a rig (the general loop's copies become true DMAs) -
    and nothing changes in the result; pinning twice is refused, unpinning something never pinned is a no-op, a pinned range that
    the caller allocated pinned itself needs no registration (recognised through the pointer's attributes)."""
    import ctypes as C
    from camera_intrinsic_calibration_rs_amd import _ffi, synth
    from camera_intrinsic_calibration_rs_amd.engine import CcalError, Problem, default_opts
    for frames, model, n_cams in ((3000, "eucm", 1), (400, "kb4", 1), (300, "eucm", 2)):
        sp = synth.make_problem(frames, model, n_cams=n_cams, seed=5 + frames)
        gp = Problem.from_synth(gpu_ctx, sp)
        for method in (0, 1):
            i_a, p_a, e_a, r_a = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
            i_b, p_b, e_b, r_b = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method), pinned=True)
            assert (r_a.status, r_a.iterations) == (r_b.status, r_b.iterations) == (0, r_a.iterations)
            np.testing.assert_array_equal(i_a, i_b); np.testing.assert_array_equal(p_a, np.array(p_b)); np.testing.assert_array_equal(e_a, e_b)
        with pytest.raises(CcalError):
            gpu_ctx.pin(gp._pin_poses)                           # the same address twice
        gp.close()                                               # unpins
    a = np.zeros(1024)
    gpu_ctx.unpin(a)                                             # never pinned: nothing to undo
    gpu_ctx.pin(a); gpu_ctx.unpin(a)
    lib = _ffi.load()
    assert lib.ccal_pin_buffer(gpu_ctx.handle, None, 8) == _ffi.ERR_INVALID_ARG
    assert lib.ccal_pin_buffer(gpu_ctx.handle, C.c_void_p(a.ctypes.data), 0) == _ffi.ERR_INVALID_ARG
