"""Synthetic calib-frame generator.

Produces the inputs the reference's hot path consumes -- per frame a set of
``FeaturePoint {p2d: f32x2, p3d: f32x3}`` (src/detected_points.rs:6-17) on the default 6x6
AprilGrid (src/board.rs:46-99, 144 corners) -- flattened into the CSR + SoA layout of
``ccal_problem_desc`` (include/ccal.h).  Everything is deterministic: a counter-based
splitmix64 stream, so the same arrays can be regenerated on any machine.

This module is data generation only (numpy); it is not an implementation of the hot path.
"""
from __future__ import annotations

import dataclasses
import numpy as np

MODEL_UCM, MODEL_EUCM, MODEL_KB4, MODEL_OPENCV5 = 0, 1, 2, 3
MODEL_NAMES = {"ucm": MODEL_UCM, "eucm": MODEL_EUCM, "kb4": MODEL_KB4, "opencv5": MODEL_OPENCV5}
MODEL_NPARAMS = {MODEL_UCM: 5, MODEL_EUCM: 6, MODEL_KB4: 8, MODEL_OPENCV5: 9}
PMAX = 10

# Ground-truth intrinsics; EUCM is the reference's data/eucm.json:3-10.
GT_PARAMS = {
    MODEL_EUCM: [190.89618687183938, 190.87022285882367, 254.9375370481962, 256.86414483060787,
                 0.6283550447635853, 1.0458678747533083],
    MODEL_UCM: [190.89618687183938, 190.87022285882367, 254.9375370481962, 256.86414483060787,
                0.6283550447635853],
    MODEL_KB4: [190.9, 190.9, 255.0, 257.0, 0.003, 0.0007, -0.002, 0.0002],
    MODEL_OPENCV5: [380.0, 380.0, 255.0, 257.0, -0.28, 0.07, 0.0002, 0.00002, 0.0],
}
GT_SIZE = (512.0, 512.0)


# ----------------------------------------------------------------------------- PRNG
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (+ a stream offset), as uint64."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed + stream * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """Box-Muller on the same stream."""
    m = (n + 1) // 2
    u = uniform01(seed, 2 * m, stream)
    r = np.sqrt(-2.0 * np.log(1.0 - u[:m]))
    t = 2.0 * np.pi * u[m:]
    return np.concatenate([r * np.cos(t), r * np.sin(t)])[:n]


# ----------------------------------------------------------------------------- board
def default_board() -> np.ndarray:
    """Board::init_aprilgrid(0.088, 0.3, 6, 6, 0) in f32 arithmetic (src/board.rs:46-99).
    Returns [144, 3] float32, row id = tag_id * 4 + corner (src/data_loader.rs:50)."""
    return aprilgrid_board(0.088, 0.3, 6, 6)


def aprilgrid_board(tag_size: float, tag_spacing: float, rows: int, cols: int) -> np.ndarray:
    ts = np.float32(tag_size)
    pitch = ts * (np.float32(1.0) + np.float32(tag_spacing))
    pts = []
    for r in range(rows):
        for c in range(cols):
            sx = np.float32(c) * pitch
            sy = -np.float32(r) * pitch
            pts += [(sx, sy, 0.0), (sx + ts, sy, 0.0), (sx + ts, sy - ts, 0.0), (sx, sy - ts, 0.0)]
    return np.asarray(pts, dtype=np.float32)


# ----------------------------------------------------------------------------- geometry (numpy, f64)
def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """[..., 3] axis-angle -> [..., 3, 3] rotation matrices."""
    rvec = np.asarray(rvec, dtype=np.float64)
    th = np.linalg.norm(rvec, axis=-1)[..., None, None]
    small = th < 1e-12
    ths = np.where(small, 1.0, th)
    k = rvec / ths[..., 0]
    K = np.zeros(rvec.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    R = np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)
    return np.where(small, np.eye(3), R)


def rotmat_to_rvec(R: np.ndarray) -> np.ndarray:
    """[..., 3, 3] -> [..., 3] axis-angle with angle in [0, pi] (via quaternion, w >= 0)."""
    R = np.asarray(R, dtype=np.float64)
    r00, r11, r22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    tr = r00 + r11 + r22
    cand = np.stack([tr, r00, r11, r22], axis=-1)
    idx = np.argmax(cand, axis=-1)
    sq = lambda v: np.sqrt(np.maximum(v, 1e-300)) * 2.0
    s0, s1, s2, s3 = sq(tr + 1.0), sq(1.0 + r00 - r11 - r22), sq(1.0 + r11 - r00 - r22), sq(1.0 + r22 - r00 - r11)
    a, b, c = R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]
    d, e, f = R[..., 0, 1] + R[..., 1, 0], R[..., 0, 2] + R[..., 2, 0], R[..., 1, 2] + R[..., 2, 1]
    q0 = np.stack([0.25 * s0, a / s0, b / s0, c / s0], axis=-1)
    q1 = np.stack([a / s1, 0.25 * s1, d / s1, e / s1], axis=-1)
    q2 = np.stack([b / s2, d / s2, 0.25 * s2, f / s2], axis=-1)
    q3 = np.stack([c / s3, e / s3, f / s3, 0.25 * s3], axis=-1)
    qs = np.stack([q0, q1, q2, q3], axis=-2)
    q = np.take_along_axis(qs, idx[..., None, None], axis=-2)[..., 0, :]
    q = np.where(q[..., :1] < 0, -q, q)
    vn = np.linalg.norm(q[..., 1:], axis=-1, keepdims=True)
    ang = 2.0 * np.arctan2(vn, q[..., :1])
    return np.where(vn > 1e-15, q[..., 1:] / np.where(vn > 1e-15, vn, 1.0) * ang, 0.0)


def project(model: int, params, pts: np.ndarray) -> np.ndarray:
    """Standard camera-model projection, vectorised ([..., 3] -> [..., 2]); data generation only."""
    p = np.asarray(params, dtype=np.float64)
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    fx, fy, cx, cy = p[0], p[1], p[2], p[3]
    if model in (MODEL_UCM, MODEL_EUCM):
        beta = p[5] if model == MODEL_EUCM else 1.0
        rho = np.sqrt(beta * (x * x + y * y) + z * z)
        den = p[4] * rho + (1.0 - p[4]) * z
        mx, my = x / den, y / den
    elif model == MODEL_KB4:
        r = np.sqrt(x * x + y * y)
        th = np.arctan2(r, z)
        th2 = th * th
        thd = th * (1.0 + th2 * (p[4] + th2 * (p[5] + th2 * (p[6] + th2 * p[7]))))
        s = np.where(r > 1e-8, thd / np.where(r > 1e-8, r, 1.0), 1.0 / z)
        mx, my = x * s, y * s
    elif model == MODEL_OPENCV5:
        xn, yn = x / z, y / z
        r2 = xn * xn + yn * yn
        rad = 1.0 + r2 * (p[4] + r2 * (p[5] + r2 * p[8]))
        mx = xn * rad + 2.0 * p[6] * xn * yn + p[7] * (r2 + 2.0 * xn * xn)
        my = yn * rad + p[6] * (r2 + 2.0 * yn * yn) + 2.0 * p[7] * xn * yn
    else:
        raise ValueError(f"unknown model {model}")
    return np.stack([fx * mx + cx, fy * my + cy], axis=-1)


# ----------------------------------------------------------------------------- problem container
@dataclasses.dataclass
class SynthProblem:
    """Flattened calib-frame inputs + ground truth + an initial guess."""
    n_cams: int
    model: np.ndarray          # [n_cams] int32
    width: np.ndarray          # [n_cams] f64
    height: np.ndarray
    xy_same_focal: bool
    n_slots: int
    obs_cam: np.ndarray        # [n_obs] int32
    obs_slot: np.ndarray       # [n_obs] int32
    obs_offsets: np.ndarray    # [n_obs+1] int64
    p3d: np.ndarray            # [n_corners, 3] f32   (SoA views below)
    p2d: np.ndarray            # [n_corners, 2] f32
    huber_delta: float
    intr_gt: np.ndarray        # [n_cams, PMAX] f64
    poses_gt: np.ndarray       # [n_slots, 6] f64
    extr_gt: np.ndarray        # [n_cams, 6] f64
    intr0: np.ndarray
    poses0: np.ndarray
    extr0: np.ndarray

    @property
    def n_obs(self) -> int:
        return int(self.obs_cam.shape[0])

    @property
    def n_corners(self) -> int:
        return int(self.obs_offsets[-1])

    def soa(self):
        """Contiguous f32 SoA arrays (x, y, z, u, v) as ccal_problem_desc wants them."""
        return (np.ascontiguousarray(self.p3d[:, 0]), np.ascontiguousarray(self.p3d[:, 1]),
                np.ascontiguousarray(self.p3d[:, 2]), np.ascontiguousarray(self.p2d[:, 0]),
                np.ascontiguousarray(self.p2d[:, 1]))

    def shard(self, rank: int, world: int) -> "SynthProblem":
        """Contiguous frame-slot range of this problem for one rank: every camera's
        observations of a slot stay together; slots are renumbered from 0."""
        return self.slot_slice(self.n_slots * rank // world, self.n_slots * (rank + 1) // world)

    def slot_slice(self, lo: int, hi: int) -> "SynthProblem":
        """The frame slots lo .. hi - 1 as a problem of their own (every camera's observations of them; slots renumbered from 0):
        one shard of a cut made elsewhere, e.g. by the library's ccal_partition_slots (balanced by corner count)."""
        keep = np.nonzero((self.obs_slot >= lo) & (self.obs_slot < hi))[0]
        counts = (self.obs_offsets[1:] - self.obs_offsets[:-1])[keep]
        offs = np.zeros(len(keep) + 1, dtype=np.int64)
        np.cumsum(counts, out=offs[1:])
        idx = np.concatenate([np.arange(self.obs_offsets[o], self.obs_offsets[o + 1]) for o in keep]) \
            if len(keep) else np.zeros(0, dtype=np.int64)
        return dataclasses.replace(
            self, n_slots=hi - lo, obs_cam=self.obs_cam[keep].copy(),
            obs_slot=(self.obs_slot[keep] - lo).astype(np.int32), obs_offsets=offs,
            p3d=self.p3d[idx].copy(), p2d=self.p2d[idx].copy(),
            poses_gt=self.poses_gt[lo:hi].copy(), poses0=self.poses0[lo:hi].copy())


def _gen_poses(seed: int, n: int, dist_range, lateral: float):
    """n candidate board poses T_cam_board: the board roughly centred, facing the camera."""
    board_c = np.array([0.33, -0.33, 0.0])
    u = uniform01(seed, n * 8, stream=1).reshape(n, 8)
    dist = dist_range[0] + (dist_range[1] - dist_range[0]) * u[:, 0]
    off = (2.0 * u[:, 1:3] - 1.0) * lateral
    zc = 2.0 * u[:, 3] - 1.0                                  # uniform axis on the sphere
    ph = 2.0 * np.pi * u[:, 4]
    s = np.sqrt(1.0 - zc * zc)
    axis = np.stack([s * np.cos(ph), s * np.sin(ph), zc], axis=-1)
    ang = 0.05 + 0.55 * u[:, 5]
    Rs = rodrigues(axis * ang[:, None])
    Rx = np.diag([1.0, -1.0, -1.0])                           # pi about x: board faces the camera
    R = Rs @ Rx
    target = np.stack([off[:, 0], off[:, 1], dist], axis=-1)
    t = target - np.einsum("nij,j->ni", R, board_c)
    return R, t


def make_problem(n_frames: int, model: str | int = "eucm", n_cams: int = 1, seed: int = 0xC0FFEE,
                 noise_px: float = 0.1, ragged: bool = False, xy_same_focal: bool = False,
                 outlier_frac: float = 0.0, huber_delta: float = 1.0, init_perturb: float = 0.05,
                 shuffle_corners: bool = False) -> SynthProblem:
    """Synthetic single- or multi-camera problem, `n_frames` frame slots x 144 corners (or 24..144
    when `ragged`).  Camera c>0 sits at T_c0 = rvec (0.01,-0.02,0.005)*c, tvec (-0.101,0.002,0.001)*c."""
    m = MODEL_NAMES[model] if isinstance(model, str) else int(model)
    P = MODEL_NPARAMS[m]
    board = default_board()
    nb = board.shape[0]
    W, H = GT_SIZE
    gt = np.asarray(GT_PARAMS[m], dtype=np.float64)
    dist_range = (0.75, 1.3) if m == MODEL_OPENCV5 else (0.4, 1.2)
    lateral = 0.12 if m == MODEL_OPENCV5 else 0.25

    extr_gt = np.zeros((n_cams, 6))
    for c in range(1, n_cams):
        extr_gt[c] = np.array([0.01, -0.02, 0.005, -0.101, 0.002, 0.001]) * c
    Rc = rodrigues(extr_gt[:, :3])

    # rejection sampling of poses: every corner inside every camera's image, z > 0.05
    Rk, tk = [], []
    need, attempt = n_frames, 0
    while need > 0:
        ncand = max(64, int(need * 2.5))
        R, t = _gen_poses(seed + 7919 * attempt, ncand, dist_range, lateral)
        pc0 = np.einsum("nij,kj->nki", R, board.astype(np.float64)) + t[:, None, :]
        ok = np.ones(ncand, dtype=bool)
        for c in range(n_cams):
            pc = np.einsum("ij,nkj->nki", Rc[c], pc0) + extr_gt[c, 3:]
            uv = project(m, gt, pc)
            ok &= (pc[..., 2] > 0.05).all(axis=1)
            ok &= ((uv[..., 0] >= 0) & (uv[..., 0] <= W) & (uv[..., 1] >= 0) & (uv[..., 1] <= H)).all(axis=1)
        sel = np.nonzero(ok)[0][:need]
        Rk.append(R[sel]); tk.append(t[sel])
        need -= len(sel); attempt += 1
        if attempt > 200:
            raise RuntimeError("pose rejection sampling did not converge")
    R = np.concatenate(Rk); t = np.concatenate(tk)
    poses_gt = np.concatenate([rotmat_to_rvec(R), t], axis=-1)

    # per (cam, slot) observation frames, cameras interleaved per slot so a slot's observations are adjacent
    if ragged:
        cnt = 24 + (splitmix64(seed, n_frames * n_cams, stream=2) % np.uint64(nb - 24 + 1)).astype(np.int64)
    else:
        cnt = np.full(n_frames * n_cams, nb, dtype=np.int64)
    obs_cam = np.tile(np.arange(n_cams, dtype=np.int32), n_frames)
    obs_slot = np.repeat(np.arange(n_frames, dtype=np.int32), n_cams)
    offs = np.zeros(n_frames * n_cams + 1, dtype=np.int64)
    np.cumsum(cnt, out=offs[1:])
    ntot = int(offs[-1])

    # which board corner each observation row is
    if ragged or shuffle_corners:
        keys = uniform01(seed, n_frames * n_cams * nb, stream=3).reshape(-1, nb)
        order = np.argsort(keys, axis=1)                       # random order, like HashMap iteration
        ids = np.concatenate([order[i, :cnt[i]] for i in range(order.shape[0])])
    else:
        ids = np.tile(np.arange(nb), n_frames * n_cams)
    row_obs = np.repeat(np.arange(n_frames * n_cams), cnt)
    row_slot = obs_slot[row_obs]; row_cam = obs_cam[row_obs]

    Xb = board[ids].astype(np.float64)
    pc = np.einsum("nij,nj->ni", R[row_slot], Xb) + t[row_slot]
    pc = np.einsum("nij,nj->ni", Rc[row_cam], pc) + extr_gt[row_cam, 3:]
    uv = project(m, gt, pc)
    uv = uv + noise_px * normal01(seed, 2 * ntot, stream=4).reshape(ntot, 2)
    if outlier_frac > 0.0:
        u = uniform01(seed, ntot, stream=5)
        bad = u < outlier_frac
        uv[bad] += 8.0 * (2.0 * uniform01(seed, 2 * ntot, stream=6).reshape(ntot, 2)[bad] - 1.0)

    intr_gt = np.zeros((n_cams, PMAX)); intr_gt[:, :P] = gt
    # initial guess: intrinsics x (1 + U[-p, p]), poses + 0.02 rad / 0.01 m, extrinsics + small
    pert = (2.0 * uniform01(seed, n_cams * PMAX, stream=7).reshape(n_cams, PMAX) - 1.0) * init_perturb
    intr0 = intr_gt * (1.0 + pert)
    if xy_same_focal:
        intr0[:, 1] = intr0[:, 0]
    dp = (2.0 * uniform01(seed, n_frames * 6, stream=8).reshape(n_frames, 6) - 1.0)
    scale = init_perturb / 0.05
    poses0 = poses_gt + dp * np.array([0.02, 0.02, 0.02, 0.01, 0.01, 0.01]) * scale
    de = (2.0 * uniform01(seed, n_cams * 6, stream=9).reshape(n_cams, 6) - 1.0)
    extr0 = extr_gt + de * np.array([0.005, 0.005, 0.005, 0.003, 0.003, 0.003]) * scale
    extr0[0] = 0.0

    return SynthProblem(
        n_cams=n_cams, model=np.full(n_cams, m, dtype=np.int32), width=np.full(n_cams, W),
        height=np.full(n_cams, H), xy_same_focal=xy_same_focal, n_slots=n_frames,
        obs_cam=obs_cam, obs_slot=obs_slot, obs_offsets=offs,
        p3d=board[ids].astype(np.float32), p2d=uv.astype(np.float32), huber_delta=huber_delta,
        intr_gt=intr_gt, poses_gt=poses_gt, extr_gt=extr_gt, intr0=intr0, poses0=poses0, extr0=extr0)


def make_rig(n_frames: int, models, extr_gt, seed: int = 0xC0FFEE, noise_px: float = 0.1, xy_same_focal: bool = False,
             min_corners: int = 24, drop_frac: float = 0.2, init_perturb: float = 0.02, ragged: bool = True) -> SynthProblem:
    """A rig of DIFFERENT cameras: one model per camera, arbitrary (large) extrinsic rotations, and every camera seeing its
    own subset of the frame slots with only the corners that fall inside its image (`ragged`: a random subset of those,
    in random order; >= `min_corners`, the reference's minimum: src/data_loader.rs:15) - the general form of calib_all_camera_with_extrinsics' input (src/util.rs:567-651:
    cameras paired by frame index, frames seen by one camera only).  Observation frames are ordered by slot, cameras
    within a slot in index order; a slot that no camera keeps still exists (a pose without residual blocks)."""
    ms = [MODEL_NAMES[m] if isinstance(m, str) else int(m) for m in models]
    n_cams = len(ms)
    extr_gt = np.asarray(extr_gt, dtype=np.float64).reshape(n_cams, 6)
    assert not extr_gt[0].any()
    board = default_board()
    nb = board.shape[0]
    W, H = GT_SIZE
    Rc = rodrigues(extr_gt[:, :3])
    narrow = any(m == MODEL_OPENCV5 for m in ms)
    R, t = _gen_poses(seed, n_frames, (0.75, 1.3) if narrow else (0.4, 1.2), 0.12 if narrow else 0.25)
    poses_gt = np.concatenate([rotmat_to_rvec(R), t], axis=-1)
    pc0 = np.einsum("nij,kj->nki", R, board.astype(np.float64)) + t[:, None, :]
    drop = uniform01(seed, n_frames * n_cams, stream=11).reshape(n_frames, n_cams) < drop_frac
    obs_cam, obs_slot, offs, ids, uvs = [], [], [0], [], []
    for s in range(n_frames):
        for c in range(n_cams):
            pc = pc0[s] @ Rc[c].T + extr_gt[c, 3:]
            uv = project(ms[c], GT_PARAMS[ms[c]], pc)
            vis = (pc[:, 2] > 0.05) & (uv[:, 0] >= 0) & (uv[:, 0] <= W) & (uv[:, 1] >= 0) & (uv[:, 1] <= H)
            if drop[s, c] or vis.sum() < min_corners:
                continue
            k = np.nonzero(vis)[0]
            if ragged:
                key = uniform01(seed + 31 * s + c, nb + 1, stream=12)
                k = k[np.argsort(key[:len(k)])][:min_corners + int(key[nb] * (len(k) - min_corners + 1))]
            obs_cam.append(c); obs_slot.append(s); offs.append(offs[-1] + len(k)); ids.append(k); uvs.append(uv[k])
    ids = np.concatenate(ids); uv = np.concatenate(uvs)
    ntot = len(ids)
    uv = uv + noise_px * normal01(seed, 2 * ntot, stream=4).reshape(ntot, 2)
    intr_gt = np.zeros((n_cams, PMAX))
    for c, m in enumerate(ms):
        intr_gt[c, :MODEL_NPARAMS[m]] = GT_PARAMS[m]
    pert = (2.0 * uniform01(seed, n_cams * PMAX, stream=7).reshape(n_cams, PMAX) - 1.0) * init_perturb
    intr0 = intr_gt * (1.0 + pert)
    if xy_same_focal:
        intr0[:, 1] = intr0[:, 0]
    scale = init_perturb / 0.05
    dp = 2.0 * uniform01(seed, n_frames * 6, stream=8).reshape(n_frames, 6) - 1.0
    poses0 = poses_gt + dp * np.array([0.02, 0.02, 0.02, 0.01, 0.01, 0.01]) * scale
    de = 2.0 * uniform01(seed, n_cams * 6, stream=9).reshape(n_cams, 6) - 1.0
    extr0 = extr_gt + de * np.array([0.005, 0.005, 0.005, 0.003, 0.003, 0.003]) * scale
    extr0[0] = 0.0
    return SynthProblem(
        n_cams=n_cams, model=np.asarray(ms, dtype=np.int32), width=np.full(n_cams, W), height=np.full(n_cams, H),
        xy_same_focal=xy_same_focal, n_slots=n_frames, obs_cam=np.asarray(obs_cam, dtype=np.int32),
        obs_slot=np.asarray(obs_slot, dtype=np.int32), obs_offsets=np.asarray(offs, dtype=np.int64),
        p3d=board[ids].astype(np.float32), p2d=uv.astype(np.float32), huber_delta=1.0,
        intr_gt=intr_gt, poses_gt=poses_gt, extr_gt=extr_gt, intr0=intr0, poses0=poses0, extr0=extr0)
