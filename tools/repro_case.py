"""Developer tool: one fuzz case step by step (max_iterations = 1, 2, ...): cost and parameters after every Gauss-Newton step."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
from oracle import binding as ob
sp = synth.make_problem(5, "eucm", n_cams=3, seed=500663308, ragged=True, xy_same_focal=False, outlier_frac=0.05)
ctx = Context(0); gp = Problem.from_synth(ctx, sp); op = ob.OracleProblem.from_synth(sp)
gp.apply_reference_bounds(); op.apply_reference_bounds()
print("slots", sp.n_slots, "obs", sp.n_obs, "obs_slot", sp.obs_slot.tolist(), "obs_cam", sp.obs_cam.tolist())
for it in (1, 2, 3, 4, 5):
    o = default_opts(0); o.max_iterations = it
    i, p, e, r = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=o, raise_on_error=False)
    io, po, eo, ro = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=o)
    print(it, "gpu", (r.status, r.iterations, "%.10f" % r.final_cost), "oracle", (ro.status, ro.iterations, "%.10f" % ro.final_cost),
          "dposes %.2e dintr %.2e dextr %.2e" % (np.abs(p - po).max(), np.abs(i - io).max(), np.abs(e - eo).max()))
