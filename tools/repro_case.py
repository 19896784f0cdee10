"""Developer tool: one fuzz case on the device against the oracle (edit the case below)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
from oracle import binding as ob
sp = synth.make_problem(3, "eucm", n_cams=1, seed=76536073, ragged=False, xy_same_focal=False, outlier_frac=0.05)
ctx = Context(0)
for bounds in (False, True):
    gp = Problem.from_synth(ctx, sp); op = ob.OracleProblem.from_synth(sp)
    if bounds: gp.apply_reference_bounds(); op.apply_reference_bounds()
    for m in (0, 1):
        o = default_opts(m)
        i, p, e, r = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=o, raise_on_error=False)
        io, po, eo, ro = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=o)
        print("bounds", bounds, "method", m, "gpu", (r.status, r.iterations, "%.12f" % r.final_cost, r.lm_accepted, r.lm_rejected), "oracle", (ro.status, ro.iterations, "%.12f" % ro.final_cost),
              "dintr_rel %.2e dposes %.2e" % ((np.abs(i[0, :6] - io[0, :6]) / np.maximum(np.abs(io[0, :6]), 1e-3)).max(), np.abs(p - po).max()), "intr", i[0, :6], io[0, :6])
