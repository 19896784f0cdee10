"""Developer tool: 625-frame solves (host pointers / device-resident, GN / LM), best and median of 30, for whatever CCAL_FUSED_DEPTH says."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
ctx = Context(0)
sp = synth.make_problem(625, "eucm")
p = Problem.from_synth(ctx, sp)
def best(fn, n=30):
    return min(fn() for _ in range(n))
def host(m):
    return p.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(m))[3].solve_ms
def dev(m):
    p.upload_params(sp.intr0, sp.poses0, sp.extr0)
    return p.solve_dev(default_opts(m)).solve_ms
for _ in range(5): host(0)
import statistics
def med(fn, n=30):
    return statistics.median(fn() for _ in range(n))
print(os.environ.get("CCAL_FUSED_DEPTH", "default"), "GN host best/median %.4f %.4f  dev %.4f %.4f   LM host %.4f %.4f dev %.4f %.4f" % (
    best(lambda: host(0)), med(lambda: host(0)), best(lambda: dev(0)), med(lambda: dev(0)), best(lambda: host(1)), med(lambda: host(1)), best(lambda: dev(1)), med(lambda: dev(1))))
