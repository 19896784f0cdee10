"""ctypes binding of the CPU oracle (oracle/liboracle_ccal.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from camera_intrinsic_calibration_rs_amd import _ffi as F  # noqa: E402  (struct definitions only)
from camera_intrinsic_calibration_rs_amd.engine import desc_from_synth, make_desc  # noqa: E402,F401

LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle_ccal.so")     # ORACLE_LIB: the sanitizer build (tests/test_sanitizers_cpu.py)
PMAX = F.PMAX
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u8 = C.POINTER(C.c_uint8)
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def use_native_build(timeout: float = 300.0) -> str:
    """Switch this process to a build of the oracle for the host it runs on (-O3 -march=native, oracle/_native/): what
    bench.py's cpu_baseline leg times.  Returns "native", or "x86-64-v3" when that build is not possible here."""
    global _lib, LIB_PATH
    native = os.path.join(_HERE, "_native", "liboracle_ccal.so")
    try:
        subprocess.run(["make", "-C", _HERE, "-s", "native"], check=True, timeout=timeout,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:  # noqa: BLE001 - no compiler on this host: keep the portable build
        return "x86-64-v3"
    LIB_PATH = native
    _lib = None
    load()
    return "native"


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    D = C.POINTER(F.ProblemDesc)
    sig = {
        "oracle_model_num_params": (C.c_int, [C.c_int]),
        "oracle_project": (C.c_int, [C.c_int, _dp, C.c_int, _dp, _dp]),
        "oracle_factor": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _fp, _fp, _dp, _dp]),
        "oracle_rvec_tvec_roundtrip": (C.c_int, [_dp, _dp]),
        "oracle_pose_compose": (C.c_int, [_dp, _dp, _dp]),
        "oracle_pose_inverse": (C.c_int, [_dp, _dp]),
        "oracle_pose_apply": (C.c_int, [_dp, _dp, _dp]),
        "oracle_eval": (C.c_int, [D, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]),
        "oracle_eval_timed": (C.c_double, [D, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]),
        "oracle_reduced_dim": (C.c_int, [D]),
        "oracle_build_normal": (C.c_int, [D, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp]),
        "oracle_cost": (C.c_double, [D, _dp, _dp, _dp]),
        "oracle_solve": (C.c_int, [D, _dp, _dp, _u8, _u8, C.POINTER(F.SolverOpts), _dp, _dp, _dp, C.POINTER(F.Report),
                                  F.ALLREDUCE_FN, C.c_void_p]),
        "oracle_gn_step_dense": (C.c_int, [D, _u8, _dp, _dp, _dp, _dp]),
        "oracle_reprojection_errors": (C.c_int, [D, _dp, _dp, _dp, _dp]),
        "oracle_validation_stats": (C.c_int, [_dp, C.c_int64, _dp, _dp]),
        "oracle_init_camera_extrinsic": (C.c_int, [_dp, _dp, C.c_int, _dp, C.POINTER(C.c_int)]),
        "oracle_se3_factor": (C.c_int, [_dp, _dp, _dp, _dp, _dp]),
        "oracle_convert_model": (C.c_int, [C.c_int, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_int, _dp, _dp, _u8,
                                           C.POINTER(C.c_int), C.c_void_p]),
        "oracle_hardware_threads": (C.c_int, []),
        "oracle_set_solve_threads": (C.c_int, [C.c_int]),
        "oracle_usable_cpus": (C.c_int, []),
        "oracle_eval_timed_heap": (C.c_double, [D, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]),
        "oracle_eval_heap": (C.c_int, [D, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def project(model: int, params, xyz):
    xyz = _f64(xyz, (-1, 3)); params = _f64(params)
    uv = np.empty((xyz.shape[0], 2))
    assert load().oracle_project(model, _p(params), xyz.shape[0], _p(xyz), _p(uv)) == 0
    return uv


def factor(model, xy_same_focal, params_eff, pose0, p3d, p2d, pose1=None, jac=True):
    """ReprojectionFactor (pose1 None) / OtherCamReprojectionFactor residual_func; r[2], J[2, D]."""
    params_eff = _f64(params_eff); pose0 = _f64(pose0)
    other = pose1 is not None
    pose1a = _f64(pose1) if other else None
    p3 = np.ascontiguousarray(p3d, dtype=np.float32); p2 = np.ascontiguousarray(p2d, dtype=np.float32)
    D = len(params_eff) + (12 if other else 6)
    r = np.empty(2); J = np.empty((2, D)) if jac else None
    rc = load().oracle_factor(model, int(xy_same_focal), int(other), _p(params_eff), _p(pose0), _p(pose1a),
                              p3.ctypes.data_as(_fp), p2.ctypes.data_as(_fp), _p(r), _p(J))
    assert rc == 0
    return (r, J) if jac else r


def rvec_tvec_roundtrip(pose):
    pose = _f64(pose); out = np.empty(6)
    load().oracle_rvec_tvec_roundtrip(_p(pose), _p(out))
    return out


def pose_compose(a, b):
    a = _f64(a); b = _f64(b); out = np.empty(6)
    load().oracle_pose_compose(_p(a), _p(b), _p(out))
    return out


def pose_inverse(a):
    a = _f64(a); out = np.empty(6)
    load().oracle_pose_inverse(_p(a), _p(out))
    return out


class OracleProblem:
    """Same flattened calib-frame inputs as the product's Problem, evaluated on the CPU."""

    def __init__(self, desc, keep=None):
        self.lib = load()
        self.desc = desc
        self._keep = keep
        self.n_cams = desc.n_cams
        self.n_slots = desc.n_slots
        self.n_corners = int(keep["obs_offsets"][-1]) if keep is not None and len(keep["obs_offsets"]) else 0
        self.K = int(self.lib.oracle_reduced_dim(C.byref(desc)))
        one = 1 if desc.xy_same_focal else 0
        self.Peff = [int(self.lib.oracle_model_num_params(int(keep["model"][c]))) - one for c in range(self.n_cams)]
        self.D = [self.Peff[c] + (6 if c == 0 else 12) for c in range(self.n_cams)]
        cnt = np.diff(keep["obs_offsets"]) if self.n_corners else np.zeros(0, dtype=np.int64)
        self.j_len = int(sum(int(n) * 2 * self.D[int(c)] for n, c in zip(cnt, keep["obs_cam"])))
        n = self.n_cams * PMAX
        self.lo = np.zeros(n); self.hi = np.zeros(n)
        self.has_bound = np.zeros(n, dtype=np.uint8); self.fixed = np.zeros(n, dtype=np.uint8)

    @classmethod
    def from_synth(cls, sp):
        d, keep = desc_from_synth(sp)
        return cls(d, keep)

    def _params(self, intr, poses, extr):
        intr = _f64(intr, (self.n_cams, PMAX)); poses = _f64(poses, (self.n_slots, 6))
        extr = _f64(np.zeros((self.n_cams, 6)) if extr is None else extr, (self.n_cams, 6))
        return intr, poses, extr

    # constraints, same semantics as ccal_set_bounds / ccal_fix_param
    def set_bounds(self, cam, idx, lo, hi):
        i = cam * PMAX + idx
        self.lo[i], self.hi[i], self.has_bound[i] = lo, hi, 1

    def fix_param(self, cam, idx):
        self.fixed[cam * PMAX + idx] = 1

    def unfix_param(self, cam, idx):
        self.fixed[cam * PMAX + idx] = 0

    def apply_reference_bounds(self):
        shift = 1 if self.desc.xy_same_focal else 0
        for c in range(self.n_cams):
            m = int(self._keep["model"][c]); W = float(self._keep["width"][c]); H = float(self._keep["height"][c])
            self.set_bounds(c, 0, 0.0, 10000.0); self.set_bounds(c, 1 - shift, 0.0, 10000.0)
            self.set_bounds(c, 2 - shift, 0.0, W); self.set_bounds(c, 3 - shift, 0.0, H)
            if m == 0:
                self.set_bounds(c, 4 - shift, 1e-6, 1.0)
            elif m == 1:
                self.set_bounds(c, 4 - shift, 1e-6, 1.0); self.set_bounds(c, 5 - shift, 1e-6, 100.0)
            else:
                for i in range(4, 8 if m == 2 else 9):
                    self.set_bounds(c, i - shift, -1.0, 1.0)

    def disable_distortions(self, n, intr):
        shift = 1 if self.desc.xy_same_focal else 0
        for c in range(self.n_cams):
            for i in range(n):
                eff = self.Peff[c] + shift - 1 - shift - i
                self.fix_param(c, eff)
                intr[c, eff + shift] = 0.0

    def eval(self, intr, poses, extr=None, apply_loss=False, threads=1):
        intr, poses, extr = self._params(intr, poses, extr)
        r = np.empty((self.n_corners, 2)); J = np.empty(self.j_len)
        rc = self.lib.oracle_eval(C.byref(self.desc), _p(intr), _p(poses), _p(extr), int(apply_loss), threads, _p(r), _p(J))
        assert rc == 0
        return r, J

    def eval_timed(self, intr, poses, extr=None, threads=1, reps=1, heap_duals=False):
        """Wall seconds for `reps` full evaluations on `threads` threads (cpu_baseline leg of bench.py).  heap_duals: the
        per-corner evaluation with heap-backed dual numbers (the container tiny-solver uses) instead of stack arrays."""
        intr, poses, extr = self._params(intr, poses, extr)
        if not hasattr(self, "_rbuf"):
            self._rbuf = np.empty((self.n_corners, 2)); self._jbuf = np.empty(self.j_len)
        fn = self.lib.oracle_eval_timed_heap if heap_duals else self.lib.oracle_eval_timed
        return float(fn(C.byref(self.desc), _p(intr), _p(poses), _p(extr), threads, reps, _p(self._rbuf), _p(self._jbuf)))

    def eval_heap(self, intr, poses, extr=None, apply_loss=False, threads=1):
        intr, poses, extr = self._params(intr, poses, extr)
        r = np.empty((self.n_corners, 2)); J = np.empty(self.j_len)
        rc = self.lib.oracle_eval_heap(C.byref(self.desc), _p(intr), _p(poses), _p(extr), int(apply_loss), threads, _p(r), _p(J))
        assert rc == 0
        return r, J

    def build_normal(self, intr, poses, extr=None, lam=0.0, min_diag=1e-6, max_diag=1e32, full=False):
        intr, poses, extr = self._params(intr, poses, extr)
        S = np.empty((self.K, self.K)); b = np.empty(self.K); cost = C.c_double()
        hd = np.empty(self.K); gc = np.empty(self.K)
        rc = self.lib.oracle_build_normal(C.byref(self.desc), _p(intr), _p(poses), _p(extr), lam, min_diag, max_diag,
                                          _p(S), _p(b), C.byref(cost), _p(hd), _p(gc))
        assert rc == 0, rc
        return (S, b, cost.value, hd, gc) if full else (S, b, cost.value)

    def cost(self, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        return float(self.lib.oracle_cost(C.byref(self.desc), _p(intr), _p(poses), _p(extr)))

    def solve(self, intr, poses, extr=None, opts=None, allreduce=None):
        """allreduce: fn(ptr, count, stream) -> int on a HOST buffer (frame-sharded solves, gloo tests)."""
        from camera_intrinsic_calibration_rs_amd.engine import default_opts
        if allreduce is None:
            cb = C.cast(None, F.ALLREDUCE_FN)
        else:
            def tramp(user, ptr, count, stream):
                try:
                    return int(allreduce(ptr or 0, int(count), stream or 0) or 0)
                except Exception:
                    import traceback; traceback.print_exc()
                    return 1
            cb = F.ALLREDUCE_FN(tramp)
        intr, poses, extr = self._params(intr, poses, extr)
        intr, poses, extr = intr.copy(), poses.copy(), extr.copy()
        opts = opts or default_opts()
        rep = F.Report()
        rc = self.lib.oracle_solve(C.byref(self.desc), _p(self.lo), _p(self.hi), self.has_bound.ctypes.data_as(_u8),
                                   self.fixed.ctypes.data_as(_u8), C.byref(opts), _p(intr), _p(poses), _p(extr), C.byref(rep),
                                   cb, None)
        rep.status = rc
        if self.desc.xy_same_focal:
            intr[:, 1] = intr[:, 0]
        return intr, poses, extr, rep

    def gn_step_dense(self, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        dx = np.empty(self.K + 6 * self.n_slots)
        rc = self.lib.oracle_gn_step_dense(C.byref(self.desc), self.fixed.ctypes.data_as(_u8), _p(intr), _p(poses), _p(extr), _p(dx))
        assert rc == 0, rc
        return dx

    def reprojection_errors(self, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        e = np.empty(self.n_corners)
        assert self.lib.oracle_reprojection_errors(C.byref(self.desc), _p(intr), _p(poses), _p(extr), _p(e)) == 0
        return e

    def validation(self, cam, intr, poses, extr=None):
        e = self.reprojection_errors(intr, poses, extr)
        offs = self._keep["obs_offsets"]; cams = self._keep["obs_cam"]
        sel = np.concatenate([e[offs[o]:offs[o + 1]] for o in range(len(cams)) if cams[o] == cam])
        a = C.c_double(); m = C.c_double()
        sel = np.ascontiguousarray(sel)
        assert self.lib.oracle_validation_stats(_p(sel), len(sel), C.byref(a), C.byref(m)) == 0
        return a.value, m.value


def se3_factor(pose_0_b, pose_i_b, x):
    """SE3Factor::residual_func with dual numbers (src/optimization/factors.rs:248-271): r[6], J[6, 6]."""
    a = _f64(pose_0_b); b = _f64(pose_i_b); xx = _f64(x)
    r = np.empty(6); J = np.empty((6, 6))
    assert load().oracle_se3_factor(_p(a), _p(b), _p(xx), _p(r), _p(J)) == 0
    return r, J


def init_camera_extrinsic(poses0, posesi):
    """SE3Factor Gauss-Newton with dual-number Jacobians (src/util.rs:511-561); returns (T_i_0 as 6-vector, iterations)."""
    p0 = _f64(poses0, (-1, 6)); pi = _f64(posesi, (-1, 6))
    out = np.empty(6); it = C.c_int()
    rc = load().oracle_init_camera_extrinsic(_p(p0), _p(pi), p0.shape[0], _p(out), C.byref(it))
    assert rc == 0, rc
    return out, it.value


def convert_model(src_model: int, src_params, tgt_model: int, tgt_params, width: float, height: float,
                  disabled: int = 0, lo=None, hi=None):
    """util::convert_model with ModelConvertFactor (src/util.rs:224-282); returns (params, n_grid_points, status)."""
    src = _f64(src_params); tgt = _f64(tgt_params).copy()
    n = C.c_int()
    if lo is None:
        lo_a = hi_a = hb = None
    else:
        lo_a = _f64(lo); hi_a = _f64(hi); hb = np.ones(len(lo_a), dtype=np.uint8)
    rc = load().oracle_convert_model(src_model, _p(src), tgt_model, _p(tgt), float(width), float(height), int(disabled),
                                     _p(lo_a) if lo is not None else None, _p(hi_a) if lo is not None else None,
                                     hb.ctypes.data_as(_u8) if lo is not None else None, C.byref(n), None)
    return tgt, n.value, rc


def set_solve_threads(n: int) -> int:
    """Threads of OracleProblem.solve's passes over the blocks (1 = the serial restatement the parity tests use; bench.py's
    all-cores Gauss-Newton baseline sets more).  Returns the previous value."""
    return int(load().oracle_set_solve_threads(int(n)))


def hardware_threads() -> int:
    return int(load().oracle_hardware_threads())


def usable_cpus() -> int:
    """CPUs this process may run on (affinity mask), not the machine's thread count."""
    return int(load().oracle_usable_cpus())
