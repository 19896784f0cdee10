import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts, make_desc, CcalError
from oracle import binding as ob
ctx = Context(0)
def run(name, fn):
    try:
        print(name, "->", fn())
    except CcalError as e:
        print(name, "-> CcalError", e)
    except Exception as e:
        print(name, "-> EXC", type(e).__name__, e)
# 1. single frame
sp = synth.make_problem(1, "eucm")
gp = Problem.from_synth(ctx, sp)
run("1 frame GN", lambda: gp.solve(sp.intr0, sp.poses0)[3].iterations)
run("1 frame LM", lambda: gp.solve(sp.intr0, sp.poses0, opts=default_opts(1))[3].iterations)
# 2. eight cameras
sp8 = synth.make_problem(12, "ucm", n_cams=6, xy_same_focal=True)
g8 = Problem.from_synth(ctx, sp8); o8 = ob.OracleProblem.from_synth(sp8)
def c8():
    a = g8.solve(sp8.intr0, sp8.poses0, sp8.extr0); b = o8.solve(sp8.intr0, sp8.poses0, sp8.extr0)
    return a[3].iterations, b[3].iterations, float(np.abs(a[0][:, :5] / b[0][:, :5] - 1).max())
run("6 cams (K = 54)", c8)
# 3. many corners per frame (400) and few (3)
rng = np.random.default_rng(0)
def big(ncorn, frames=6):
    base = synth.make_problem(frames, "eucm")
    # resample board points on a denser grid by repeating corners with jitter on the plane
    off = [0]; X = []; U = []
    for f in range(frames):
        s, e = base.obs_offsets[f], base.obs_offsets[f + 1]
        idx = rng.integers(s, e, size=ncorn)
        X.append(base.p3d[idx]); U.append(base.p2d[idx]); off.append(off[-1] + ncorn)
    X = np.concatenate(X).astype(np.float32); U = np.concatenate(U).astype(np.float32)
    d, keep = make_desc(1, [1], [512.0], [512.0], False, frames, [0] * frames, list(range(frames)), np.array(off, dtype=np.int64),
                        X[:, 0], X[:, 1], X[:, 2], U[:, 0], U[:, 1], 1.0)
    p = Problem(ctx, d, keep)
    r, J = p.eval(base.intr0, base.poses0)
    rep = p.solve(base.intr0, base.poses0)[3]
    return r.shape, J.shape, rep.status, rep.iterations
run("400 corners/frame", lambda: big(400))
run("700 corners/frame", lambda: big(700, 3))
run("24 corners/frame", lambda: big(24, 30))
# 4. empty problem
d, keep = make_desc(1, [1], [512.0], [512.0], False, 0, [], [], np.zeros(1, dtype=np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), 1.0)
pe = Problem(ctx, d, keep)
run("empty eval", lambda: [a.shape for a in pe.eval(sp.intr0, np.zeros((0, 6)))])
run("empty solve", lambda: pe.solve(sp.intr0, np.zeros((0, 6)))[3].status)
run("empty build", lambda: pe.build_normal(sp.intr0, np.zeros((0, 6)))[2])
print("done")
