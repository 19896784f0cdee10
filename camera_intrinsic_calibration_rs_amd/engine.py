"""Thin object wrappers over the C ABI (include/ccal.h): Context and Problem.

Nothing here computes; every method forwards to the HIP library through ctypes.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import PMAX


class CcalError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        name = _ffi.STATUS_NAMES[code] if 0 <= code < len(_ffi.STATUS_NAMES) else str(code)
        super().__init__(f"{where}: {name}" + (f" ({detail})" if detail else ""))


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """One GPU + one HIP stream (ccal_ctx)."""

    def __init__(self, device: int = 0, stream: int | None = None, lib=None):
        self.lib = lib if lib is not None else _ffi.load()      # lib: another build of the library (tests: _ffi.load_legacy())
        h = C.c_void_p()
        rc = self.lib.ccal_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_ctx_create", "no usable HIP device" if rc == _ffi.ERR_HIP else "")
        self.handle = h
        self.device = device
        self._problems = weakref.WeakSet()        # closed before the context: a ccal_problem holds its ccal_ctx

    def last_error(self) -> str:
        return (self.lib.ccal_last_error(self.handle) or b"").decode()

    def sync(self):
        rc = self.lib.ccal_sync(self.handle)
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_sync", self.last_error())

    # -- pinned caller memory (ccal_pin_buffer): arrays the library reads and writes in place, no staging copy -----
    def pin(self, a: np.ndarray) -> np.ndarray:
        """Page-lock a C-contiguous array for the GPU (hipHostRegister through the library); it must outlive the pinning."""
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("pin: a C-contiguous array")
        rc = self.lib.ccal_pin_buffer(self.handle, C.c_void_p(a.ctypes.data), a.nbytes)
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_pin_buffer", self.last_error())
        return a

    def unpin(self, a: np.ndarray):
        rc = self.lib.ccal_unpin_buffer(self.handle, C.c_void_p(a.ctypes.data))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_unpin_buffer", self.last_error())

    # -- model conventions (what this build assumes about the absent camera-intrinsic-model crate) -----------
    def model_conventions(self) -> _ffi.ModelConventions:
        cv = _ffi.ModelConventions()
        rc = self.lib.ccal_get_model_conventions(self.handle, C.byref(cv))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_get_model_conventions")
        return cv

    def set_model_conventions(self, cv: "_ffi.ModelConventions | None"):
        rc = self.lib.ccal_set_model_conventions(self.handle, C.byref(cv) if cv is not None else None)
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_set_model_conventions")

    # -- native RCCL communicator of this context's GPU (frame-sharded solves) --------------------------------
    def rccl_comm_create(self, world: int, rank: int, unique_id: bytes) -> int:
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        buf = C.create_string_buffer(unique_id, 128)
        h = C.c_void_p()
        rc = self.lib.ccal_rccl_comm_create(self.handle, int(world), int(rank), buf, C.byref(h))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_rccl_comm_create", self.last_error())
        return h.value

    def close(self):
        if getattr(self, "handle", None):
            for p in list(getattr(self, "_problems", ())):     # whatever order the garbage collector picks: problems first
                p.close()
            self.lib.ccal_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_available() -> bool:
    return bool(_ffi.load().ccal_rccl_available())


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls this and hands the 128 bytes to every rank)."""
    buf = C.create_string_buffer(128)
    rc = _ffi.load().ccal_rccl_unique_id(buf)
    if rc != _ffi.OK:
        raise CcalError(rc, "ccal_rccl_unique_id")
    return buf.raw


def rccl_comm_destroy(comm: int) -> None:
    _ffi.load().ccal_rccl_comm_destroy(C.c_void_p(comm))


def rccl_comm_count(comm: int) -> int:
    """ncclCommCount of a communicator made by Context.rccl_comm_create (-1 on error)."""
    return int(_ffi.load().ccal_rccl_comm_count(C.c_void_p(comm)))


def partition_slots(desc: "_ffi.ProblemDesc", n_shards: int, lib=None) -> list:
    """ccal_partition_slots (host only, no GPU): the n_shards + 1 slot boundaries ccal_multi_problem_create cuts `desc` at -
    contiguous frame-slot ranges balanced by corner count."""
    first = np.zeros(n_shards + 1, dtype=np.int32)
    rc = (lib or _ffi.load()).ccal_partition_slots(C.byref(desc), int(n_shards), first.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != _ffi.OK:
        raise CcalError(rc, "ccal_partition_slots")
    return [int(x) for x in first]


def make_desc(n_cams, model, width, height, xy_same_focal, n_slots, obs_cam, obs_slot, obs_offsets,
              x, y, z, u, v, huber_delta):
    """Build a ccal_problem_desc and return (desc, keepalive-list-of-arrays)."""
    keep = dict(
        model=np.ascontiguousarray(model, dtype=np.int32), width=_f64(width), height=_f64(height),
        obs_cam=np.ascontiguousarray(obs_cam, dtype=np.int32), obs_slot=np.ascontiguousarray(obs_slot, dtype=np.int32),
        obs_offsets=np.ascontiguousarray(obs_offsets, dtype=np.int64),
        x=np.ascontiguousarray(x, dtype=np.float32), y=np.ascontiguousarray(y, dtype=np.float32),
        z=np.ascontiguousarray(z, dtype=np.float32), u=np.ascontiguousarray(u, dtype=np.float32),
        v=np.ascontiguousarray(v, dtype=np.float32))
    d = _ffi.ProblemDesc()
    d.n_cams = int(n_cams)
    d.model = keep["model"].ctypes.data_as(C.POINTER(C.c_int32))
    d.width = _dp(keep["width"]); d.height = _dp(keep["height"])
    d.xy_same_focal = 1 if xy_same_focal else 0
    d.n_slots = int(n_slots); d.n_obs = int(len(keep["obs_cam"]))
    d.obs_cam = keep["obs_cam"].ctypes.data_as(C.POINTER(C.c_int32))
    d.obs_slot = keep["obs_slot"].ctypes.data_as(C.POINTER(C.c_int32))
    d.obs_offsets = keep["obs_offsets"].ctypes.data_as(C.POINTER(C.c_int64))
    for name, key in (("p3d_x", "x"), ("p3d_y", "y"), ("p3d_z", "z"), ("p2d_u", "u"), ("p2d_v", "v")):
        setattr(d, name, keep[key].ctypes.data_as(C.POINTER(C.c_float)))
    d.huber_delta = float(huber_delta)
    return d, keep


def desc_from_synth(sp):
    x, y, z, u, v = sp.soa()
    return make_desc(sp.n_cams, sp.model, sp.width, sp.height, sp.xy_same_focal, sp.n_slots,
                     sp.obs_cam, sp.obs_slot, sp.obs_offsets, x, y, z, u, v, sp.huber_delta)


def default_opts(method: int = _ffi.METHOD_GN, **kw) -> _ffi.SolverOpts:
    o = _ffi.SolverOpts()
    _ffi.load().ccal_set_defaults(C.byref(o))
    o.method = method
    for k, val in kw.items():
        setattr(o, k, val)
    return o


class MultiContext:
    """A set of GPUs driven from THIS process (ccal_multi): one context per listed device plus the transport of the step's
    all-reduce - RCCL when the devices differ, the library's in-process transport when a device is listed more than once."""

    def __init__(self, devices, transport: int | None = None, lib=None):
        """transport: None = automatic, _ffi.TRANSPORT_RCCL / TRANSPORT_INPROC = that one or an error (ccal_multi_create_transport).
        lib: another build of the library (tests: _ffi.load_legacy(), which carries the fault-injection hook)."""
        self.lib = lib if lib is not None else _ffi.load()
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        rc = self.lib.ccal_multi_create_transport(devs.ctypes.data_as(C.POINTER(C.c_int32)), len(devs),
                                                  -1 if transport is None else int(transport), C.byref(h))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_multi_create", (self.lib.ccal_create_last_error() or b"").decode())
        self.handle = h
        self.devices = [int(d) for d in devs]
        self._problems = weakref.WeakSet()

    @property
    def transport(self) -> int:
        return int(self.lib.ccal_multi_transport(self.handle))

    @property
    def rccl_ranks(self) -> int:
        """Ranks RCCL itself counts in this set's communicators (ncclCommCount); 0 when the transport is not RCCL."""
        return int(self.lib.ccal_multi_rccl_ranks(self.handle))

    def last_error(self) -> str:
        return (self.lib.ccal_multi_last_error(self.handle) or b"").decode()

    def set_model_conventions(self, cv):
        rc = self.lib.ccal_multi_set_model_conventions(self.handle, C.byref(cv) if cv is not None else None)
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_multi_set_model_conventions")

    def sync(self):
        rc = self.lib.ccal_multi_sync(self.handle)
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_multi_sync", self.last_error())

    def close(self):
        if getattr(self, "handle", None):
            for p in list(getattr(self, "_problems", ())):
                p.close()
            self.lib.ccal_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiProblem:
    """One problem whose frame slots the library shards over the GPUs of a MultiContext (ccal_multi_problem)."""

    def __init__(self, mctx: MultiContext, desc: _ffi.ProblemDesc, keep=None):
        self.mctx = mctx
        self.lib = mctx.lib
        self._keep = keep
        h = C.c_void_p()
        rc = self.lib.ccal_multi_problem_create(mctx.handle, C.byref(desc), C.byref(h))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_multi_problem_create", mctx.last_error())
        self.handle = h
        mctx._problems.add(self)
        self.n_cams = desc.n_cams
        self.n_slots = desc.n_slots
        self.n_obs = int(desc.n_obs)
        self.n_shards = int(self.lib.ccal_multi_problem_num_shards(h))

    @classmethod
    def from_synth(cls, mctx: MultiContext, sp) -> "MultiProblem":
        d, keep = desc_from_synth(sp)
        return cls(mctx, d, keep)

    def _check(self, rc: int, where: str):
        if rc != _ffi.OK:
            raise CcalError(rc, where, self.mctx.last_error())

    def slot_range(self, i: int):
        a = C.c_int32(); b = C.c_int32()
        self._check(self.lib.ccal_multi_problem_slot_range(self.handle, i, C.byref(a), C.byref(b)), "ccal_multi_problem_slot_range")
        return a.value, b.value

    def shard_handle(self, i: int):
        return C.c_void_p(self.lib.ccal_multi_problem_shard(self.handle, i))

    def set_bounds(self, cam, idx, lo, hi):
        self._check(self.lib.ccal_multi_set_bounds(self.handle, cam, idx, lo, hi), "ccal_multi_set_bounds")

    def fix_param(self, cam, idx):
        self._check(self.lib.ccal_multi_fix_param(self.handle, cam, idx), "ccal_multi_fix_param")

    def unfix_param(self, cam, idx):
        self._check(self.lib.ccal_multi_unfix_param(self.handle, cam, idx), "ccal_multi_unfix_param")

    def apply_reference_bounds(self):
        self._check(self.lib.ccal_multi_apply_reference_bounds(self.handle), "ccal_multi_apply_reference_bounds")

    def disable_distortions(self, n: int, intr: np.ndarray):
        self._check(self.lib.ccal_multi_disable_distortions(self.handle, n, _dp(intr)), "ccal_multi_disable_distortions")

    def _params(self, intr, poses, extr):
        intr = _f64(intr, (self.n_cams, PMAX))
        poses = _f64(poses, (self.n_slots, 6))
        extr = _f64(np.zeros((self.n_cams, 6)) if extr is None else extr, (self.n_cams, 6))
        return intr, poses, extr

    def upload_params(self, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        self._check(self.lib.ccal_multi_upload_params(self.handle, _dp(intr), _dp(poses), _dp(extr)), "ccal_multi_upload_params")

    def shard_sizes(self, i: int):
        """(corners, doubles of J_out) of shard i: the sizes of its mode-E output buffers."""
        h = self.shard_handle(i)
        return int(self.lib.ccal_num_corners(h)), int(self.lib.ccal_jacobian_len(h))

    def eval_dev(self, r_dev_ptrs, J_dev_ptrs, apply_loss=False):
        """ccal_multi_eval_dev: mode E on every shard (device buffers on the shard's own GPU); enqueues only - MultiContext.sync()."""
        n = self.n_shards
        ra = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in r_dev_ptrs]); ja = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in J_dev_ptrs])
        self._check(self.lib.ccal_multi_eval_dev(self.handle, 1 if apply_loss else 0, ra, ja), "ccal_multi_eval_dev")

    def init_poses(self, intr, min_points: int = 10):
        intr = _f64(intr, (self.n_cams, PMAX))
        poses = np.zeros((max(self.n_obs, 1), 6)); used = np.zeros(max(self.n_obs, 1), dtype=np.int32)
        self._check(self.lib.ccal_multi_init_poses(self.handle, _dp(intr), int(min_points), _dp(poses),
                                                   used.ctypes.data_as(C.POINTER(C.c_int32))), "ccal_multi_init_poses")
        return poses[:self.n_obs], used[:self.n_obs]

    def validation(self, cam, intr, poses, extr=None):
        """validation() statistics (avg of the lowest 99 %, median) over the shards: bit-identical to Problem.validation."""
        intr, poses, extr = self._params(intr, poses, extr)
        a = C.c_double(); m = C.c_double()
        self._check(self.lib.ccal_multi_validation(self.handle, cam, _dp(intr), _dp(poses), _dp(extr), C.byref(a), C.byref(m)),
                    "ccal_multi_validation")
        return a.value, m.value

    def reprojection_errors(self, intr, poses, extr=None):
        """Per-corner Euclidean reprojection errors in the corner order of the description this problem was made from."""
        intr, poses, extr = self._params(intr, poses, extr)
        offs = self._keep["obs_offsets"]
        e = np.empty(int(offs[-1]))
        self._check(self.lib.ccal_multi_reprojection_errors(self.handle, _dp(intr), _dp(poses), _dp(extr), _dp(e),
                                                            offs.ctypes.data_as(C.POINTER(C.c_int64))), "ccal_multi_reprojection_errors")
        return e

    def solve(self, intr, poses, extr=None, opts: _ffi.SolverOpts | None = None, raise_on_error=True):
        """ccal_multi_solve: ONE call, every GPU of the set; poses in the caller's slot order."""
        intr, poses, extr = self._params(intr, poses, extr)
        intr, poses, extr = intr.copy(), poses.copy(), extr.copy()
        opts = opts or default_opts()
        rep = _ffi.Report()
        rc = self.lib.ccal_multi_solve(self.handle, C.byref(opts), _dp(intr), _dp(poses), _dp(extr), C.byref(rep))
        if rc not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE) and raise_on_error:
            raise CcalError(rc, "ccal_multi_solve", self.mctx.last_error())
        return intr, poses, extr, rep

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccal_multi_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_sharded(problems, intr, poses_list, extr=None, opts: _ffi.SolverOpts | None = None, raise_on_error=True):
    """ccal_solve_sharded: shards built by the caller (one Problem per context), solved as ONE problem from this thread."""
    n = len(problems)
    lib = problems[0].lib
    n_cams = problems[0].n_cams
    intr = _f64(intr, (n_cams, PMAX)).copy()
    extr = _f64(np.zeros((n_cams, 6)) if extr is None else extr, (n_cams, 6)).copy()
    poses = [_f64(p, (q.n_slots, 6)).copy() for p, q in zip(poses_list, problems)]
    opts = opts or default_opts()
    hs = (C.c_void_p * n)(*[p.handle for p in problems])
    dpp = C.POINTER(C.c_double)
    pa = (dpp * n)(*[_dp(p) for p in poses])
    rep = _ffi.Report()
    rc = lib.ccal_solve_sharded(hs, n, C.byref(opts), _dp(intr), pa, _dp(extr), C.byref(rep))
    if rc not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE) and raise_on_error:
        raise CcalError(rc, "ccal_solve_sharded", problems[0].ctx.last_error())
    return intr, poses, extr, rep


class Problem:
    """Calib-frame inputs resident in HBM (ccal_problem)."""

    def __init__(self, ctx: Context, desc: _ffi.ProblemDesc, keep=None):
        self.ctx = ctx
        self.lib = ctx.lib
        self._keep = keep
        h = C.c_void_p()
        rc = self.lib.ccal_problem_create(ctx.handle, C.byref(desc), C.byref(h))
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_problem_create", ctx.last_error())
        self.handle = h
        ctx._problems.add(self)
        self.n_cams = desc.n_cams
        self.n_slots = desc.n_slots
        self.n_obs = int(desc.n_obs)
        self.n_corners = int(self.lib.ccal_num_corners(h))
        self.K = int(self.lib.ccal_reduced_dim(h))
        self.j_len = int(self.lib.ccal_jacobian_len(h))
        self._cb = None

    @classmethod
    def from_synth(cls, ctx: Context, sp) -> "Problem":
        d, keep = desc_from_synth(sp)
        return cls(ctx, d, keep)

    def _check(self, rc: int, where: str):
        if rc != _ffi.OK:
            raise CcalError(rc, where, self.ctx.last_error())

    def block_dim(self, cam: int = 0) -> int:
        return int(self.lib.ccal_block_dim(self.handle, cam))

    def eff_num_params(self, cam: int = 0) -> int:
        return int(self.lib.ccal_eff_num_params(self.handle, cam))

    # -- constraints (tiny_solver::Problem::set_variable_bounds / fix_variable) -------------------
    def set_bounds(self, cam, idx, lo, hi):
        self._check(self.lib.ccal_set_bounds(self.handle, cam, idx, lo, hi), "ccal_set_bounds")

    def fix_param(self, cam, idx):
        self._check(self.lib.ccal_fix_param(self.handle, cam, idx), "ccal_fix_param")

    def unfix_param(self, cam, idx):
        self._check(self.lib.ccal_unfix_param(self.handle, cam, idx), "ccal_unfix_param")

    def apply_reference_bounds(self):
        self._check(self.lib.ccal_apply_reference_bounds(self.handle), "ccal_apply_reference_bounds")

    def disable_distortions(self, n: int, intr: np.ndarray):
        self._check(self.lib.ccal_disable_distortions(self.handle, n, _dp(intr)), "ccal_disable_distortions")

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, stream:int) -> int; kept alive on the object."""
        if fn is None:
            self._cb = None
            self._check(self.lib.ccal_set_allreduce(self.handle, C.cast(None, _ffi.ALLREDUCE_FN), None), "ccal_set_allreduce")
            return
        def tramp(user, ptr, count, stream):
            try:
                return int(fn(ptr or 0, int(count), stream or 0) or 0)
            except Exception:   # never unwind through C
                import traceback; traceback.print_exc()
                return 1
        self._cb = _ffi.ALLREDUCE_FN(tramp)
        self._check(self.lib.ccal_set_allreduce(self.handle, self._cb, None), "ccal_set_allreduce")

    def set_rccl_comm(self, comm: "int | None"):
        """ncclComm_t (from Context.rccl_comm_create or the host's own RCCL): the library issues the step's all-reduce itself."""
        self._check(self.lib.ccal_set_rccl_comm(self.handle, C.c_void_p(comm) if comm else None), "ccal_set_rccl_comm")

    # -- mode E -----------------------------------------------------------------------------------
    def _params(self, intr, poses, extr):
        intr = _f64(intr, (self.n_cams, PMAX))
        poses = _f64(poses, (self.n_slots, 6))
        extr = _f64(np.zeros((self.n_cams, 6)) if extr is None else extr, (self.n_cams, 6))
        return intr, poses, extr

    def eval(self, intr, poses, extr=None, apply_loss=False):
        intr, poses, extr = self._params(intr, poses, extr)
        r = np.empty((self.n_corners, 2)); J = np.empty(self.j_len)
        self._check(self.lib.ccal_eval(self.handle, _dp(intr), _dp(poses), _dp(extr), 1 if apply_loss else 0,
                                       _dp(r), _dp(J)), "ccal_eval")
        return r, J

    def upload_params(self, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        self._check(self.lib.ccal_upload_params(self.handle, _dp(intr), _dp(poses), _dp(extr)), "ccal_upload_params")

    def download_params(self):
        intr = np.zeros((self.n_cams, PMAX)); poses = np.zeros((self.n_slots, 6)); extr = np.zeros((self.n_cams, 6))
        self._check(self.lib.ccal_download_params(self.handle, _dp(intr), _dp(poses), _dp(extr)), "ccal_download_params")
        return intr, poses, extr

    def eval_dev(self, r_dev_ptr: int, J_dev_ptr: int, apply_loss=False):
        self._check(self.lib.ccal_eval_dev(self.handle, 1 if apply_loss else 0, C.c_void_p(r_dev_ptr), C.c_void_p(J_dev_ptr)),
                    "ccal_eval_dev")

    # -- mode N -----------------------------------------------------------------------------------
    def build_normal(self, intr, poses, extr=None, lam=0.0):
        intr, poses, extr = self._params(intr, poses, extr)
        S = np.empty((self.K, self.K)); b = np.empty(self.K); cost = C.c_double()
        self._check(self.lib.ccal_build_normal(self.handle, _dp(intr), _dp(poses), _dp(extr), float(lam),
                                               _dp(S), _dp(b), C.byref(cost)), "ccal_build_normal")
        return S, b, cost.value

    def build_normal_dev(self, lam=0.0):
        self._check(self.lib.ccal_build_normal_dev(self.handle, float(lam)), "ccal_build_normal_dev")

    def solve(self, intr, poses, extr=None, opts: _ffi.SolverOpts | None = None, raise_on_error=True, pinned=False):
        intr, poses, extr = self._params(intr, poses, extr)
        if pinned:
            # the host keeps its pose array pinned (ccal_pin_buffer, once per problem): the library reads the start from it and
            # writes the result into it, no staging copy either way.  The arrays returned are that buffer (valid until the next call)
            if getattr(self, "_pin_poses", None) is None:
                self._pin_poses = self.ctx.pin(np.zeros((max(self.n_slots, 1), 6)))
            np.copyto(self._pin_poses[:self.n_slots], poses)
            intr, poses, extr = intr.copy(), self._pin_poses[:self.n_slots], extr.copy()
        else:
            intr, poses, extr = intr.copy(), poses.copy(), extr.copy()
        opts = opts or default_opts()
        rep = _ffi.Report()
        rc = self.lib.ccal_solve(self.handle, C.byref(opts), _dp(intr), _dp(poses), _dp(extr), C.byref(rep))
        if rc not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE) and raise_on_error:
            raise CcalError(rc, "ccal_solve", self.ctx.last_error())
        return intr, poses, extr, rep

    def solve_dev(self, opts: _ffi.SolverOpts | None = None, raise_on_error=True) -> _ffi.Report:
        """ccal_solve_dev: start from the parameters on the device (upload_params / a previous solve), leave the result
        there (download_params fetches it)."""
        opts = opts or default_opts()
        rep = _ffi.Report()
        rc = self.lib.ccal_solve_dev(self.handle, C.byref(opts), C.byref(rep))
        if rc not in (_ffi.OK, _ffi.ERR_NO_CONVERGENCE) and raise_on_error:
            raise CcalError(rc, "ccal_solve_dev", self.ctx.last_error())
        return rep

    @staticmethod
    def solve_batch(problems, opts: _ffi.SolverOpts | None = None, starts=None):
        """ccal_solve_batch: independent problems (ideally one context each) solved side by side from this thread.
        starts = None: device-resident (every problem starts from what upload_params / a previous solve left on the device,
        download_params fetches the results); else a list of (intr, poses, extr) per problem, returned updated.
        Returns (reports, results-or-None); a problem's own verdict is in its report."""
        n = len(problems)
        lib = problems[0].lib if n else _ffi.load()
        opts = opts or default_opts()
        hs = (C.c_void_p * max(n, 1))(*[p.handle for p in problems])
        reps = (_ffi.Report * max(n, 1))()
        results = None
        dpp = C.POINTER(C.c_double)
        if starts is None:
            rc = lib.ccal_solve_batch(hs, n, C.byref(opts), None, None, None, reps)
        else:
            results = []
            for p, (intr, poses, extr) in zip(problems, starts):
                i_, p_, e_ = p._params(intr, poses, extr)
                results.append((i_.copy(), p_.copy(), e_.copy()))
            ia = (dpp * max(n, 1))(*[_dp(r[0]) for r in results]); pa = (dpp * max(n, 1))(*[_dp(r[1]) for r in results])
            ea = (dpp * max(n, 1))(*[_dp(r[2]) for r in results])
            rc = lib.ccal_solve_batch(hs, n, C.byref(opts), ia, pa, ea, reps)
        if rc != _ffi.OK:
            raise CcalError(rc, "ccal_solve_batch", problems[0].ctx.last_error() if n else "")
        return [reps[i] for i in range(n)], results

    # -- pose initialisation (src/util.rs:418-436) ---------------------------------------------------
    def init_poses(self, intr, min_points: int = 10):
        """T_cam_board per observation frame [n_obs, 6] and the number of corners used (0 = no pose)."""
        intr = _f64(intr, (self.n_cams, PMAX))
        n_obs = self.n_obs                                   # from the description the library holds, not from the keep-alive arrays
        poses = np.zeros((max(n_obs, 1), 6)); used = np.zeros(max(n_obs, 1), dtype=np.int32)
        self._check(self.lib.ccal_init_poses(self.handle, _dp(intr), int(min_points), _dp(poses),
                                             used.ctypes.data_as(C.POINTER(C.c_int32))), "ccal_init_poses")
        return poses[:n_obs], used[:n_obs]

    # -- validation ---------------------------------------------------------------------------------
    def reprojection_errors(self, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        e = np.empty(self.n_corners)
        self._check(self.lib.ccal_reprojection_errors(self.handle, _dp(intr), _dp(poses), _dp(extr), _dp(e)),
                    "ccal_reprojection_errors")
        return e

    def validation(self, cam, intr, poses, extr=None):
        intr, poses, extr = self._params(intr, poses, extr)
        a = C.c_double(); m = C.c_double()
        self._check(self.lib.ccal_validation(self.handle, cam, _dp(intr), _dp(poses), _dp(extr), C.byref(a), C.byref(m)),
                    "ccal_validation")
        return a.value, m.value

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self, "_pin_poses", None) is not None and getattr(self.ctx, "handle", None):
                try:
                    self.ctx.unpin(self._pin_poses)          # (before the array goes: the registration must not outlive the memory)
                finally:
                    self._pin_poses = None
            self.lib.ccal_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
