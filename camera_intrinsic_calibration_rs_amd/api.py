"""Host-side mirror of the reference's calib-frame API for the hot path (same names, argument
meaning and error behaviour), driving the HIP engine through the C ABI.

  reference (Rust)                                             here
  ---------------------------------------------------------   ------------------------------------------
  detected_points::FeaturePoint / FrameFeature                 FeaturePoint / FrameFeature
  types::RvecTvec (+ to_na_isometry3 / to_rvec_tvec)            RvecTvec
  camera_intrinsic_model::GenericModel<f64>                     GenericModel
  optimization::factors::ReprojectionFactor::residual_func      ReprojectionFactor.residual_func
  optimization::factors::OtherCamReprojectionFactor             OtherCamReprojectionFactor.residual_func
  util::calib_camera                (src/util.rs:384-490)       calib_camera
  util::calib_all_camera_with_extrinsics (src/util.rs:567-715)  calib_all_camera_with_extrinsics
  util::validation                  (src/util.rs:721-795)       validation
  io::write_report / object_to_json (src/io.rs)                 write_report / model_to_json / poses_to_json / ...

Differences, on purpose: (1) `calib_camera` takes the per-frame initial poses as an argument (the
reference computes them inside with sqpnp, src/util.rs:418-436, which is outside the hot path); frames
without an initial pose are rejected up front instead of reproducing the reference's latent bug of
adding residual blocks that have no initial value (src/util.rs:431-433 vs 407-414).  (2) corner order
inside a frame is by corner id, not HashMap order (the reference is not reproducible with itself,
SURVEY 0.5); results agree at the converged optimum.  `None` is returned where the reference returns
`None` (solver failure); nothing here falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import json
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .engine import CcalError, Context, MultiContext, MultiProblem, Problem, default_opts, make_desc
from .synth import MODEL_NAMES, MODEL_NPARAMS, PMAX, rodrigues, rotmat_to_rvec

_MODEL_KEYS = {
    "ucm": ["fx", "fy", "cx", "cy", "alpha"],
    "eucm": ["fx", "fy", "cx", "cy", "alpha", "beta"],
    "kb4": ["fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4"],
    "opencv5": ["fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"],
}
# serde tags of GenericModel's variants.  UCM / EUCM / KannalaBrandt4 appear in the reference (data/eucm.json,
# examples/convert_model.rs:14,19); "OpenCVModel5" is this build's ASSUMPTION about the absent crate (one string to flip).
_JSON_NAMES = {"ucm": "UCM", "eucm": "EUCM", "kb4": "KannalaBrandt4", "opencv5": "OpenCVModel5"}
# EUCMT ("Extended Unified with Tangential", README.md:78): a PARAMETER CONTAINER here - target of the closed-form
# UCM -> EUCMT conversion (src/util.rs:236-243).  Its projection and its JSON field names live only in the absent crate.
MODEL_EUCMT = 4
_CONTAINER_KINDS = {"eucmt": (MODEL_EUCMT, 8)}


@dataclasses.dataclass
class FeaturePoint:                      # src/detected_points.rs:6-9 (f32 in the reference)
    p2d: Tuple[float, float]
    p3d: Tuple[float, float, float]


@dataclasses.dataclass
class FrameFeature:                      # src/detected_points.rs:13-17
    time_ns: int
    img_w_h: Tuple[int, int]
    features: Dict[int, FeaturePoint]


@dataclasses.dataclass
class RvecTvec:                          # src/types.rs:13-36
    rvec: Tuple[float, float, float]
    tvec: Tuple[float, float, float]

    def as6(self) -> np.ndarray:
        return np.array(list(self.rvec) + list(self.tvec), dtype=np.float64)

    @staticmethod
    def from6(v) -> "RvecTvec":
        v = [float(x) for x in v]
        return RvecTvec((v[0], v[1], v[2]), (v[3], v[4], v[5]))

    # Isometry algebra used by the problem set-up / result mapping (host side, a handful of poses)
    def matrix(self) -> Tuple[np.ndarray, np.ndarray]:
        return rodrigues(np.array(self.rvec)), np.array(self.tvec, dtype=np.float64)

    def inverse(self) -> "RvecTvec":
        R, t = self.matrix()
        return RvecTvec.from6(np.concatenate([rotmat_to_rvec(R.T), -R.T @ t]))

    def compose(self, other: "RvecTvec") -> "RvecTvec":      # self * other
        R1, t1 = self.matrix(); R2, t2 = other.matrix()
        return RvecTvec.from6(np.concatenate([rotmat_to_rvec(R1 @ R2), R1 @ t2 + t1]))


class GenericModel:
    """camera_intrinsic_model::GenericModel<f64> for the four models on the hot path."""

    def __init__(self, kind: str, params: Sequence[float], width: float, height: float):
        kind = kind.lower()
        if kind not in MODEL_NAMES and kind not in _CONTAINER_KINDS:
            raise ValueError(f"unsupported model {kind}")
        if len(params) != (_CONTAINER_KINDS[kind][1] if kind in _CONTAINER_KINDS else MODEL_NPARAMS[MODEL_NAMES[kind]]):
            raise ValueError("wrong number of parameters")
        self.kind = kind
        self._params = np.asarray(params, dtype=np.float64).copy()
        self._w, self._h = float(width), float(height)

    @property
    def model_id(self) -> int:
        return _CONTAINER_KINDS[self.kind][0] if self.kind in _CONTAINER_KINDS else MODEL_NAMES[self.kind]

    def params(self) -> np.ndarray:
        return self._params.copy()

    def set_params(self, p) -> None:
        self._params = np.asarray(p, dtype=np.float64).copy()

    def width(self) -> float:
        return self._w

    def height(self) -> float:
        return self._h

    def copy(self) -> "GenericModel":
        return GenericModel(self.kind, self._params, self._w, self._h)

    # cam{i}.json: {"EUCM": {"fx":..,"fy":..,"cx":..,"cy":..,"alpha":..,"beta":..,"width":..,"height":..}} (data/eucm.json)
    def to_json_obj(self) -> dict:
        if self.kind in _CONTAINER_KINDS:
            raise NotImplementedError(f"{self.kind}: the JSON field names are defined only in the absent camera-intrinsic-model crate")
        d = {k: float(v) for k, v in zip(_MODEL_KEYS[self.kind], self._params)}
        d["width"] = int(round(self._w)); d["height"] = int(round(self._h))
        return {_JSON_NAMES[self.kind]: d}

    @staticmethod
    def from_json_obj(obj: dict) -> "GenericModel":
        (name, d), = obj.items()
        kind = {v: k for k, v in _JSON_NAMES.items()}[name]
        return GenericModel(kind, [d[k] for k in _MODEL_KEYS[kind]], d["width"], d["height"])


def model_to_json(path: str, model: GenericModel) -> None:
    with open(path, "w") as f:
        json.dump(model.to_json_obj(), f, indent=2)


def model_from_json(path: str) -> GenericModel:
    with open(path) as f:
        return GenericModel.from_json_obj(json.load(f))


def poses_to_json(path: str, poses: Dict[int, RvecTvec]) -> None:
    """cam{i}_poses.json: BTreeMap frame-index -> {"rvec":[3],"tvec":[3]} (src/bin/camera_calibration.rs:288-293)."""
    with open(path, "w") as f:
        json.dump({str(k): {"rvec": list(v.rvec), "tvec": list(v.tvec)} for k, v in sorted(poses.items())}, f, indent=2)


def poses_from_json(path: str) -> Dict[int, RvecTvec]:
    with open(path) as f:
        return {int(k): RvecTvec(tuple(v["rvec"]), tuple(v["tvec"])) for k, v in json.load(f).items()}


def extrinsics_to_json(path: str, rtvecs: Sequence[RvecTvec]) -> None:
    """extrinsics.json: {"rtvecs":[{"rvec":..,"tvec":..},..]} (src/types.rs:41-44)."""
    with open(path, "w") as f:
        json.dump({"rtvecs": [{"rvec": list(r.rvec), "tvec": list(r.tvec)} for r in rtvecs]}, f, indent=2)


def extrinsics_from_json(path: str) -> List[RvecTvec]:
    with open(path) as f:
        return [RvecTvec(tuple(v["rvec"]), tuple(v["tvec"])) for v in json.load(f)["rtvecs"]]


def write_report(path: str, with_extrinsic: bool, rep_rms: Sequence[Tuple[float, float]]) -> None:
    """src/io.rs:21-31, byte for byte."""
    s = f"Calibrate with extrinsics: {'true' if with_extrinsic else 'false'}\n\n"
    for i, (avg, med) in enumerate(rep_rms):
        s += f"cam{i}:\n    average reprojection error: {avg:.5f} px\n    median  reprojection error: {med:.5f} px\n\n"
    with open(path, "w") as f:
        f.write(s)


_default_ctx: Optional[Context] = None


def _ctx(ctx: Optional[Context]) -> Context:
    global _default_ctx
    if ctx is not None:
        return ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class _Opened:
    """A problem on one context, or - `devices` given - sharded by the library over a device set of this process
    (ccal_multi_*: one call, every listed GPU; a device listed twice = two shards on it)."""

    def __init__(self, ctx: Optional[Context], devices: Optional[Sequence[int]], d, keep):
        self.mctx = None
        if devices is not None:
            self.mctx = MultiContext(list(devices))
            try:
                self.prob = MultiProblem(self.mctx, d, keep)
            except Exception:
                self.mctx.close()
                raise
        else:
            self.prob = Problem(_ctx(ctx), d, keep)

    def __enter__(self):
        return self.prob

    def __exit__(self, *exc):
        self.prob.close()
        if self.mctx is not None:
            self.mctx.close()
        return False


def _flatten(cams_frames: Sequence[Sequence[Optional[FrameFeature]]], use: Sequence[Sequence[int]]):
    """(cam, frame index) observation frames -> CSR + SoA arrays; slots = sorted union of frame indices."""
    slots = sorted({i for idxs in use for i in idxs})
    slot_of = {fi: s for s, fi in enumerate(slots)}
    obs_cam, obs_slot, offs, X, U = [], [], [0], [], []
    for fi in slots:                                     # a slot's observations stay adjacent
        for c, idxs in enumerate(use):
            if fi not in idxs:
                continue
            ff = cams_frames[c][fi]
            ids = sorted(ff.features.keys())
            X += [ff.features[k].p3d for k in ids]
            U += [ff.features[k].p2d for k in ids]
            obs_cam.append(c); obs_slot.append(slot_of[fi]); offs.append(offs[-1] + len(ids))
    X = np.asarray(X, dtype=np.float32).reshape(-1, 3); U = np.asarray(U, dtype=np.float32).reshape(-1, 2)
    return slots, obs_cam, obs_slot, offs, X, U


def _intr_matrix(cameras: Sequence[GenericModel]) -> np.ndarray:
    intr = np.zeros((len(cameras), PMAX))
    for c, m in enumerate(cameras):
        intr[c, :len(m._params)] = m._params
    return intr


def init_frame_poses(frame_feature_list: Sequence[Optional[FrameFeature]], generic_camera: GenericModel,
                     min_points: int = 10, ctx: Optional[Context] = None, devices: Optional[Sequence[int]] = None) -> Dict[int, RvecTvec]:
    """The pose initialisation inside calib_camera (src/util.rs:418-436): `unproject` the detections with
    the current model, keep the valid ones, normalise by z, planar PnP -- one wavefront per frame."""
    valid = [i for i, f in enumerate(frame_feature_list) if f is not None]
    if not valid:
        return {}
    slots, obs_cam, obs_slot, offs, X, U = _flatten([frame_feature_list], [valid])
    d, keep = make_desc(1, [generic_camera.model_id], [generic_camera.width()], [generic_camera.height()], False,
                        len(slots), obs_cam, obs_slot, offs, X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    with _Opened(ctx, devices, d, keep) as prob:
        poses, used = prob.init_poses(_intr_matrix([generic_camera]), min_points)
    return {fi: RvecTvec.from6(poses[s]) for s, fi in enumerate(slots) if used[s] > 0}


def calib_camera(frame_feature_list: Sequence[Optional[FrameFeature]], generic_camera: GenericModel,
                 xy_same_focal: bool, disabled_distortions: int, fixed_focal: bool,
                 initial_poses: Optional[Dict[int, RvecTvec]] = None, ctx: Optional[Context] = None,
                 opts: Optional[_ffi.SolverOpts] = None, devices: Optional[Sequence[int]] = None
                 ) -> Optional[Tuple[GenericModel, Dict[int, RvecTvec]]]:
    """util::calib_camera (src/util.rs:384-490): single-camera bundle adjustment, Gauss-Newton.
    `initial_poses=None` reproduces the reference's in-function initialisation (unproject + planar PnP
    per frame on the GPU, frames with fewer than 10 valid points are skipped, src/util.rs:418-436).
    `devices`: still ONE call of ONE process, like the reference's - the library shards the frames over the listed GPUs
    (ccal_multi_*), one all-reduce per Gauss-Newton step."""
    if initial_poses is None:
        initial_poses = init_frame_poses(frame_feature_list, generic_camera, ctx=ctx, devices=devices)
    valid = [i for i, f in enumerate(frame_feature_list) if f is not None and i in initial_poses]
    if not valid:
        return None
    slots, obs_cam, obs_slot, offs, X, U = _flatten([frame_feature_list], [valid])
    d, keep = make_desc(1, [generic_camera.model_id], [generic_camera.width()], [generic_camera.height()],
                        xy_same_focal, len(slots), obs_cam, obs_slot, offs, X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    with _Opened(ctx, devices, d, keep) as prob:
        intr = _intr_matrix([generic_camera])
        poses = np.stack([initial_poses[i].as6() for i in slots])
        prob.apply_reference_bounds()                                  # src/util.rs:446
        prob.disable_distortions(disabled_distortions, intr)           # src/util.rs:447-454
        try:
            intr, poses, _, rep = prob.solve(intr, poses, None, opts or default_opts())      # :455
        except CcalError:
            return None                                                # result_option.as_ref()?  -> None
        if fixed_focal:                                                # :459-464 "set focal and opt again."
            prob.fix_param(0, 0)
            intr[0, 0] = generic_camera.params()[0]
            if xy_same_focal:
                intr[0, 1] = intr[0, 0]
            intr, poses, _, rep = prob.solve(intr, poses, None, opts or default_opts())      # .unwrap()
        out = generic_camera.copy()
        out.set_params(intr[0, :len(generic_camera._params)])          # fy = f re-inserted by the engine (:467-470)
        return out, {fi: RvecTvec.from6(poses[s]) for s, fi in enumerate(slots)}


def calib_cameras(cams_frame_feature_lists: Sequence[Sequence[Optional[FrameFeature]]], generic_cameras: Sequence[GenericModel],
                  xy_same_focal: bool, disabled_distortions: int, fixed_focal: bool, device: int = 0,
                  opts: Optional[_ffi.SolverOpts] = None, devices: Optional[Sequence[int]] = None
                  ) -> List[Optional[Tuple[GenericModel, Dict[int, RvecTvec]]]]:
    """The per-camera loop of the tool - `for cam in 0..cam_num { calib_camera(...) }` (src/bin/camera_calibration.rs:255-265) -
    as ONE ccal_solve_batch: every camera's single-camera problem on a context of its own, solved side by side (a session-
    sized problem leaves the GPU almost idle).  `devices`: the cameras' contexts are placed round-robin on the listed GPUs
    (independent sessions are the path's most natural multi-GPU split at session size: no collective at all); default: all on
    `device`.  Entry i equals calib_camera(cams_frame_feature_lists[i], generic_cameras[i], ...) - same verdict and iteration
    count, results to the order of summation (ccal_solve_batch sizes every problem's launches for its share of its GPU)."""
    n = len(generic_cameras)
    devs = [int(d) for d in devices] if devices else [int(device)]
    ctxs = [Context(devs[c % len(devs)]) for c in range(n)]
    out: List[Optional[Tuple[GenericModel, Dict[int, RvecTvec]]]] = [None] * n
    jobs = []                                       # (camera, problem, slots, intr, poses)
    try:
        for c in range(n):
            frames, cam = cams_frame_feature_lists[c], generic_cameras[c]
            init = init_frame_poses(frames, cam, ctx=ctxs[c])
            valid = [i for i, f in enumerate(frames) if f is not None and i in init]
            if not valid:
                continue
            slots, obs_cam, obs_slot, offs, X, U = _flatten([frames], [valid])
            d, keep = make_desc(1, [cam.model_id], [cam.width()], [cam.height()], xy_same_focal, len(slots), obs_cam, obs_slot, offs,
                                X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
            prob = Problem(ctxs[c], d, keep)
            intr = _intr_matrix([cam])
            prob.apply_reference_bounds()
            prob.disable_distortions(disabled_distortions, intr)
            jobs.append([c, prob, slots, intr, np.stack([init[i].as6() for i in slots])])
        o = opts or default_opts()
        reps, res = Problem.solve_batch([j[1] for j in jobs], o, starts=[(j[3], j[4], None) for j in jobs])
        alive = []
        for j, rep, r in zip(jobs, reps, res):
            if rep.status not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE):
                continue                                                # result_option.as_ref()?  -> None
            j[3], j[4] = r[0], r[1]
            alive.append(j)
        if fixed_focal and alive:                                       # src/util.rs:459-464 "set focal and opt again."
            for j in alive:
                j[1].fix_param(0, 0)
                j[3][0, 0] = generic_cameras[j[0]].params()[0]
                if xy_same_focal:
                    j[3][0, 1] = j[3][0, 0]
            reps, res = Problem.solve_batch([j[1] for j in alive], o, starts=[(j[3], j[4], None) for j in alive])
            for j, rep, r in zip(alive, reps, res):
                if rep.status not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE):     # calib_camera's second solve is `.unwrap()`ed (src/util.rs:463)
                    raise CcalError(rep.status, "ccal_solve_batch", f"camera {j[0]}: the fixed-focal re-optimisation failed")
                j[3], j[4] = r[0], r[1]
        for c, prob, slots, intr, poses in alive:
            m = generic_cameras[c].copy()
            m.set_params(intr[0, :len(m._params)])
            out[c] = (m, {fi: RvecTvec.from6(poses[s]) for s, fi in enumerate(slots)})
        return out
    finally:
        for j in jobs:
            j[1].close()
        for cx in ctxs:
            cx.close()


def _joint_problem_inputs(cameras, t_cam_i_0, cam_rtvecs, cams_detected_feature_frames, xy_same_focal):
    """Problem description and starting point of the joint problem exactly as src/util.rs:576-651 lays it out (the tests
    hand the same arrays to the oracle): slots = sorted union of the frame indices with a pose, T_0_b per slot = cam0's
    pose when it saw the frame, else T_c0^-1 * T_cb of the first camera that did (`.entry().or_insert()` in camera order)."""
    n_cams = len(cameras)
    use = [sorted(cam_rtvecs[c].keys()) for c in range(n_cams)]
    slots, obs_cam, obs_slot, offs, X, U = _flatten(cams_detected_feature_frames, use)
    if not slots:
        return None
    d, keep = make_desc(n_cams, [m.model_id for m in cameras], [m.width() for m in cameras],
                        [m.height() for m in cameras], xy_same_focal, len(slots), obs_cam, obs_slot, offs,
                        X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    intr = _intr_matrix(cameras)
    extr = np.zeros((n_cams, 6))
    for c in range(1, n_cams):
        extr[c] = t_cam_i_0[c].as6()
    poses = np.zeros((len(slots), 6))
    for s, fi in enumerate(slots):
        for c in range(n_cams):
            if fi in cam_rtvecs[c]:
                rt = cam_rtvecs[c][fi]
                poses[s] = (rt if c == 0 else t_cam_i_0[c].inverse().compose(rt)).as6()
                break
    return d, keep, slots, intr, poses, extr


def calib_all_camera_with_extrinsics(cameras: Sequence[GenericModel], t_cam_i_0: Sequence[RvecTvec],
                                     cam_rtvecs: Sequence[Dict[int, RvecTvec]],
                                     cams_detected_feature_frames: Sequence[Sequence[Optional[FrameFeature]]],
                                     xy_same_focal: bool, disabled_distortions: int, cam0_fixed_focal: bool,
                                     ctx: Optional[Context] = None, opts: Optional[_ffi.SolverOpts] = None,
                                     devices: Optional[Sequence[int]] = None
                                     ) -> Optional[Tuple[List[GenericModel], List[RvecTvec], Dict[int, RvecTvec]]]:
    """util::calib_all_camera_with_extrinsics (src/util.rs:567-715): joint intrinsics + extrinsics.
    `devices`: one call, the frame slots sharded over the listed GPUs (ccal_multi_*)."""
    n_cams = len(cameras)
    built = _joint_problem_inputs(cameras, t_cam_i_0, cam_rtvecs, cams_detected_feature_frames, xy_same_focal)
    if built is None:
        return None
    d, keep, slots, intr, poses, extr = built
    with _Opened(ctx, devices, d, keep) as prob:
        prob.apply_reference_bounds()
        prob.disable_distortions(disabled_distortions, intr)
        if cam0_fixed_focal:
            prob.fix_param(0, 0)                                       # src/util.rs:664-667
        try:
            intr, poses, extr, rep = prob.solve(intr, poses, extr, opts or default_opts())
        except CcalError:
            return None
        out_models = []
        for c, m in enumerate(cameras):
            mm = m.copy(); mm.set_params(intr[c, :len(m._params)]); out_models.append(mm)
        t_i_0 = [RvecTvec((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))] + [RvecTvec.from6(extr[c]) for c in range(1, n_cams)]
        return out_models, t_i_0, {fi: RvecTvec.from6(poses[s]) for s, fi in enumerate(slots)}


def init_camera_extrinsic(cam_rtvecs: Sequence[Dict[int, RvecTvec]], opts=None) -> List[RvecTvec]:
    """util::init_camera_extrinsic (src/util.rs:511-561): T_i_0 of every camera from the frames both it and
    camera 0 have a board pose for (SE3Factor + HuberLoss(0.5) + Gauss-Newton, in the library's host code).
    opts: the optimizer's stop rules (None = GaussNewtonOptimizer::default(), as the reference)."""
    lib = _ffi.load()
    out = [RvecTvec((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))]
    for cam_i in range(1, len(cam_rtvecs)):
        keys = sorted(set(cam_rtvecs[0].keys()) & set(cam_rtvecs[cam_i].keys()))
        if not keys:
            raise ValueError(f"camera {cam_i} shares no frame with camera 0")       # the reference indexes [0] and panics
        p0 = np.ascontiguousarray(np.stack([cam_rtvecs[0][k].as6() for k in keys]))
        pi = np.ascontiguousarray(np.stack([cam_rtvecs[cam_i][k].as6() for k in keys]))
        x = np.zeros(6)
        rep = _ffi.Report()
        rc = lib.ccal_init_camera_extrinsic_opts(p0.ctypes.data_as(C.POINTER(C.c_double)), pi.ctypes.data_as(C.POINTER(C.c_double)),
                                                 len(keys), x.ctypes.data_as(C.POINTER(C.c_double)), 0,
                                                 C.byref(opts) if opts is not None else None, C.byref(rep))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_init_camera_extrinsic")                        # `.unwrap()` in the reference
        out.append(RvecTvec.from6(x))
    return out


def convert_model(source_model: GenericModel, target_model: GenericModel, disabled_distortions: int = 0,
                  ctx: Optional[Context] = None) -> GenericModel:
    """util::convert_model (src/util.rs:224-282).  UCM -> EUCM is the closed form beta = 1 (:229-235, pinned by
    tests/util_test.rs:77-110); everything else fits the target over the reference's pixel grid with
    ModelConvertFactor (src/optimization/factors.rs:10-76) through `ccal_convert_model` on the device.  The
    reference mutates `target_model`; here the fitted model is also returned."""
    if round(source_model.width()) != round(target_model.width()):
        raise ValueError("source width and target width are not the same.")          # factors.rs:29-33: panic!
    if round(source_model.height()) != round(target_model.height()):
        raise ValueError("source height and target height are not the same.")
    if source_model.kind == "ucm" and target_model.kind == "eucm":                   # closed form: no device needed
        target_model.set_params(list(source_model.params()) + [1.0])
        return target_model
    if source_model.kind == "ucm" and target_model.kind == "eucmt":                  # src/util.rs:236-243
        target_model.set_params(list(source_model.params()) + [1.0, 0.0, 0.0])
        return target_model
    if source_model.kind in _CONTAINER_KINDS or target_model.kind in _CONTAINER_KINDS:
        raise CcalError(_ffi.ERR_UNSUPPORTED, "convert_model", "EUCMT can only be the target of the closed-form UCM conversion")
    lib = _ffi.load()
    c = _ctx(ctx)
    src = np.ascontiguousarray(source_model.params(), dtype=np.float64)
    tgt = np.ascontiguousarray(target_model.params(), dtype=np.float64).copy()
    rep = _ffi.Report()
    rc = lib.ccal_convert_model(c.handle, source_model.model_id, src.ctypes.data_as(C.POINTER(C.c_double)),
                                target_model.model_id, tgt.ctypes.data_as(C.POINTER(C.c_double)),
                                float(source_model.width()), float(source_model.height()), int(disabled_distortions),
                                None, C.byref(rep))
    if rc not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE):                                 # max_iterations: the reference keeps the result
        raise CcalError(rc, "ccal_convert_model: " + c.last_error())                 # `.unwrap()` in the reference (:276)
    target_model.set_params(tgt)
    return target_model


def init_ucm(frame_feature0: FrameFeature, frame_feature1: FrameFeature, rtvec0: RvecTvec, rtvec1: RvecTvec,
             init_f: float, init_alpha: float, fixed_focal: bool, ctx: Optional[Context] = None) -> Optional[GenericModel]:
    """util::init_ucm (src/util.rs:287-378).  UCMInitFocalAlphaFactor (factors.rs:82-120) is the reprojection
    factor of a UCM whose only free intrinsics are f = fx = fy and alpha with the principal point pinned at the
    image centre, i.e. the engine's UCM + xy_same_focal problem with cx, cy fixed; bounds f in [f0/3, 3 f0],
    alpha in [1e-6, 1] (:345-346).  Then calib_camera on the two frames with xy_same_focal = true (:365-371)."""
    w, h = frame_feature0.img_w_h
    ucm0 = GenericModel("ucm", [init_f, init_f, w / 2.0, h / 2.0, init_alpha], w, h)
    frames = [frame_feature0, frame_feature1]
    slots, obs_cam, obs_slot, offs, X, U = _flatten([frames], [[0, 1]])
    d, keep = make_desc(1, [ucm0.model_id], [w], [h], True, 2, obs_cam, obs_slot, offs, X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    prob = Problem(_ctx(ctx), d, keep)
    try:
        prob.fix_param(0, 1); prob.fix_param(0, 2)                      # cx, cy (eff indices with fy removed)
        if fixed_focal:
            prob.fix_param(0, 0)
        prob.set_bounds(0, 0, init_f / 3.0, init_f * 3.0)
        prob.set_bounds(0, 3, 1e-6, 1.0)
        try:
            intr, _, _, _ = prob.solve(_intr_matrix([ucm0]), np.stack([rtvec0.as6(), rtvec1.as6()]), None, default_opts())
        except CcalError:
            return None
    finally:
        prob.close()
    ucm1 = GenericModel("ucm", [intr[0, 0], intr[0, 0], w / 2.0, h / 2.0, intr[0, 4]], w, h)
    res = calib_camera(frames, ucm1, True, 0, fixed_focal, None, ctx=ctx)
    if res is None:
        raise RuntimeError("The initial UCM model fitting failed. Might be wrong board configuration.")   # .expect(...)
    return res[0]


def validation(cam_idx: int, final_result: GenericModel, rtvec_list: Dict[int, RvecTvec],
               detected_feature_frames: Sequence[Optional[FrameFeature]], ctx: Optional[Context] = None,
               devices: Optional[Sequence[int]] = None) -> Tuple[float, float]:
    """util::validation (src/util.rs:721-795): (avg of the lowest 99 %, median) reprojection error in px.
    `devices`: the errors are evaluated shard by shard on the listed GPUs, the statistics are the same bits."""
    valid = [i for i in sorted(rtvec_list.keys()) if detected_feature_frames[i] is not None]
    slots, obs_cam, obs_slot, offs, X, U = _flatten([detected_feature_frames], [valid])
    d, keep = make_desc(1, [final_result.model_id], [final_result.width()], [final_result.height()], False,
                        len(slots), obs_cam, obs_slot, offs, X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    with _Opened(ctx, devices, d, keep) as prob:
        poses = np.stack([rtvec_list[i].as6() for i in slots])
        return prob.validation(0, _intr_matrix([final_result]), poses, None)


class ReprojectionFactor:
    """optimization::factors::ReprojectionFactor (src/optimization/factors.rs:126-173).
    residual_func(params) with params = [intrinsics (P_eff), rvec, tvec] evaluates the block on the GPU;
    `jacobian=True` also returns the 2 x D block Jacobian tiny-solver would obtain with dual numbers."""

    other = False

    def __init__(self, target: GenericModel, p3d, p2d, xy_same_focal: bool, ctx: Optional[Context] = None):
        self.target = target
        self.p3d = np.asarray(p3d, dtype=np.float32)          # glam::Vec3 (f32) widened later, factors.rs:141-143
        self.p2d = np.asarray(p2d, dtype=np.float32)
        self.xy_same_focal = bool(xy_same_focal)
        self._ctx = ctx

    @classmethod
    def new(cls, target, p3d, p2d, xy_same_focal, ctx=None):
        return cls(target, p3d, p2d, xy_same_focal, ctx)

    def residual_func(self, params: Sequence[np.ndarray], jacobian: bool = False):
        n_cams = 2 if self.other else 1
        m = self.target.model_id
        p0 = np.asarray(params[0], dtype=np.float64)
        full = np.insert(p0, 1, p0[0]) if self.xy_same_focal else p0          # factors.rs:155-158
        intr = np.zeros((n_cams, PMAX)); intr[:, :len(full)] = full
        pose = np.concatenate([np.asarray(params[1], float), np.asarray(params[2], float)])[None, :]
        extr = np.zeros((n_cams, 6))
        if self.other:
            extr[1] = np.concatenate([np.asarray(params[3], float), np.asarray(params[4], float)])
        d, keep = make_desc(n_cams, [m] * n_cams, [self.target.width()] * n_cams, [self.target.height()] * n_cams,
                            self.xy_same_focal, 1, [n_cams - 1], [0], [0, 1],
                            [self.p3d[0]], [self.p3d[1]], [self.p3d[2]], [self.p2d[0]], [self.p2d[1]], 1.0)
        prob = Problem(_ctx(self._ctx), d, keep)
        try:
            r, J = prob.eval(intr, pose, extr)
        finally:
            prob.close()
        return (r[0], J.reshape(2, -1)) if jacobian else r[0]


class OtherCamReprojectionFactor(ReprojectionFactor):
    """optimization::factors::OtherCamReprojectionFactor (factors.rs:179-228);
    params = [intrinsics, rvec_0_b, tvec_0_b, rvec_i_0, tvec_i_0]."""
    other = True


def frames_from_synth(sp, cam: int = 0) -> List[Optional[FrameFeature]]:
    """SynthProblem -> the reference's `Vec<Option<FrameFeature>>` for one camera (ids = row order)."""
    out: List[Optional[FrameFeature]] = [None] * sp.n_slots
    for o in range(sp.n_obs):
        if sp.obs_cam[o] != cam:
            continue
        a, b = int(sp.obs_offsets[o]), int(sp.obs_offsets[o + 1])
        feats = {k: FeaturePoint(tuple(sp.p2d[a + k]), tuple(sp.p3d[a + k])) for k in range(b - a)}
        out[int(sp.obs_slot[o])] = FrameFeature(0, (int(sp.width[cam]), int(sp.height[cam])), feats)
    return out
