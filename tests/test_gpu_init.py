"""GPU: per-frame pose initialisation (src/util.rs:418-436) -- unproject with the
current model, keep valid points, planar PnP -- against ground truth and against an independent numpy
restatement (DLT homography + SVD orthonormalisation).  The reference's sqpnp is a different estimator, so
the bar is "same basin": the joint solve from these poses lands on the same optimum."""
import numpy as np
import pytest

from camera_intrinsic_calibration_rs_amd import api, synth
from camera_intrinsic_calibration_rs_amd.engine import Problem, default_opts

pytestmark = pytest.mark.gpu


def _numpy_planar_pnp(xn, X):
    """Homography DLT (SVD null vector) + polar orthonormalisation; returns R, t."""
    A = []
    for (a, b), (x, y, _) in zip(xn, X):
        A.append([x, y, 1, 0, 0, 0, -a * x, -a * y, -a]); A.append([0, 0, 0, x, y, 1, -b * x, -b * y, -b])
    H = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
    H = H / H[2, 2] if H[2, 2] != 0 else H
    lam = 2.0 / (np.linalg.norm(H[:, 0]) + np.linalg.norm(H[:, 1]))
    if lam * H[2, 2] < 0:
        lam = -lam
    M = np.stack([lam * H[:, 0], lam * H[:, 1], np.cross(lam * H[:, 0], lam * H[:, 1])], axis=1)
    U, _, Vt = np.linalg.svd(M)
    R = U @ np.diag([1, 1, np.linalg.det(U @ Vt)]) @ Vt
    return R, lam * H[:, 2]


@pytest.mark.parametrize("model", ["ucm", "eucm", "kb4", "opencv5"])
def test_init_poses_close_to_ground_truth(gpu_ctx, model):
    sp = synth.make_problem(40, model, ragged=True)
    gp = Problem.from_synth(gpu_ctx, sp)
    poses, used = gp.init_poses(sp.intr_gt)
    assert (used == np.diff(sp.obs_offsets)).all()          # every corner unprojects at the true intrinsics
    R = synth.rodrigues(poses[:, :3]); Rg = synth.rodrigues(sp.poses_gt[:, :3])
    ang = np.arccos(np.clip((np.einsum("nij,nij->n", R, Rg) - 1) / 2, -1, 1))
    assert ang.max() < 0.02, ang.max()                      # 0.1 px noise: a few mrad
    assert np.abs(poses[:, 3:] - sp.poses_gt[:, 3:]).max() < 0.01
    # independent estimator on the same normalised points (EUCM / UCM closed-form unprojection in numpy)
    if model in ("ucm", "eucm"):
        th = sp.intr_gt[0]; beta = th[5] if model == "eucm" else 1.0
        for f in (0, 7, 39):
            a, b = sp.obs_offsets[f], sp.obs_offsets[f + 1]
            mx = (sp.p2d[a:b, 0].astype(float) - th[2]) / th[0]; my = (sp.p2d[a:b, 1].astype(float) - th[3]) / th[1]
            r2 = mx * mx + my * my
            k = (1 - th[4] ** 2 * beta * r2) / (th[4] * np.sqrt(1 - (2 * th[4] - 1) * beta * r2) + 1 - th[4])
            Rn, tn = _numpy_planar_pnp(np.stack([mx / k, my / k], 1), sp.p3d[a:b].astype(float))
            assert np.abs(synth.rodrigues(poses[f, :3]) - Rn).max() < 5e-3
            assert np.abs(poses[f, 3:] - tn).max() < 5e-3


def test_too_few_points_and_bad_model_give_no_pose(gpu_ctx):
    """Frames with < 10 valid unprojections are skipped (src/util.rs:431-433)."""
    from camera_intrinsic_calibration_rs_amd.engine import make_desc
    sp = synth.make_problem(3, "eucm")
    offs = [0, 9, 9 + 144, 9 + 144 + 30]
    idx = np.concatenate([np.arange(0, 9), np.arange(144, 288), np.arange(288, 318)])
    d, keep = make_desc(1, [1], [512.0], [512.0], False, 3, [0, 0, 0], [0, 1, 2], offs,
                        sp.p3d[idx, 0], sp.p3d[idx, 1], sp.p3d[idx, 2], sp.p2d[idx, 0], sp.p2d[idx, 1], 1.0)
    gp = Problem(gpu_ctx, d, keep)
    poses, used = gp.init_poses(sp.intr_gt)
    assert used.tolist() == [0, 144, 30] and (poses[0] == 0).all()
    # alpha = 0.95 shrinks the valid disc of the EUCM inverse: most detections fall outside -> None
    bad = sp.intr_gt.copy(); bad[0, 4] = 0.95; bad[0, 5] = 4.0
    _, used_bad = gp.init_poses(bad)
    assert (used_bad <= used).all() and used_bad.sum() < used.sum()


@pytest.mark.parametrize("model", ["eucm", "kb4"])
def test_calib_camera_without_initial_poses(gpu_ctx, model):
    """calib_camera as the reference calls it (no poses passed in) reaches the optimum found from the
    perturbed ground-truth poses."""
    sp = synth.make_problem(30, model)
    P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
    frames = api.frames_from_synth(sp)
    cam0 = api.GenericModel(model, sp.intr0[0, :P], 512, 512)
    tight = default_opts(0, min_abs_error_decrease=1e-10, min_rel_error_decrease=1e-12)
    a = api.calib_camera(frames, cam0, False, 0, False, None, ctx=gpu_ctx, opts=tight)
    b = api.calib_camera(frames, cam0, False, 0, False, {i: api.RvecTvec.from6(sp.poses0[i]) for i in range(30)},
                         ctx=gpu_ctx, opts=tight)
    assert a is not None and b is not None
    assert np.abs(a[0].params() / b[0].params() - 1)[:4].max() < 1e-6
    pa = np.stack([a[1][i].as6() for i in range(30)]); pb = np.stack([b[1][i].as6() for i in range(30)])
    # board poses sit near a rotation angle of pi, where rvec and rvec (1 - 2 pi / |rvec|) are the same
    # rotation: compare rotations, not their axis-angle coordinates
    np.testing.assert_allclose(synth.rodrigues(pa[:, :3]), synth.rodrigues(pb[:, :3]), atol=1e-6)
    np.testing.assert_allclose(pa[:, 3:], pb[:, 3:], atol=1e-6)
