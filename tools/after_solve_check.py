"""Developer tool: ccal_build_normal at the solution right after GN / LM solves (parameter / record sets flipped by the
device loops) against the oracle - single camera and a two-camera rig.  b is ~0 there: compare S and the cost."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
from oracle import binding as ob
ctx = Context(0)
for models, ext in ((("eucm", "kb4"), [[0]*6, [0.12, -0.1, 0.3, -0.1, 0.02, 0.01]]), (("eucm",), [[0]*6])):
    sp = synth.make_rig(40, models, ext) if len(models) > 1 else synth.make_problem(40, "eucm", ragged=True)
    gp = Problem.from_synth(ctx, sp); op = ob.OracleProblem.from_synth(sp)
    for method in (0, 1, 1, 0):
        intr, poses, extr, rep = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        S, b, c = gp.build_normal(intr, poses, extr, lam=1e-3)
        So, bo, co = op.build_normal(intr, poses, extr, lam=1e-3)
        print(models, method, rep.status, rep.iterations, abs(c - co) / co, np.abs(S - So).max() / np.abs(So).max(), np.abs(b - bo).max() / max(np.abs(bo).max(), 1e-30))
