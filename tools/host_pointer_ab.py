#!/usr/bin/env python3
"""Developer tool: what a host-pointer ccal_solve costs over the device-resident solve, from pageable and from pinned caller arrays
(ccal_pin_buffer), per problem size:  tools/host_pointer_ab.py [frames[:ragged],...] [reps]
A/B of developer switches: run it twice with CCAL_LIB=<legacy library> and the switch in the environment (e.g. CCAL_RESULT_SPREAD=0)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts

cases = (sys.argv[1] if len(sys.argv) > 1 else "10000,10000:ragged,5000,2500,625").split(",")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Context(0)
tag = os.environ.get("CCAL_RESULT_SPREAD", "-")
for c in cases:
    frames, _, rg = c.partition(":")
    sp = synth.make_problem(int(frames), "eucm", ragged=bool(rg), seed=0xC0FFEE + 77)
    p = Problem.from_synth(ctx, sp)
    for method in (0, 1):
        row = []
        for pinned in (False, True):
            row.append(min(p.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method), pinned=pinned)[3].solve_ms for _ in range(reps)))
        bd = 1e9
        for _ in range(reps):
            p.upload_params(sp.intr0, sp.poses0, sp.extr0); bd = min(bd, p.solve_dev(default_opts(method)).solve_ms)
        print(f"spread={tag} {c:>14s} {'LM' if method else 'GN'}: ccal_solve pageable {row[0]:.4f} ms  pinned {row[1]:.4f} ms  ccal_solve_dev {bd:.4f} ms  gap {row[0] - bd:.4f} / {row[1] - bd:.4f}")
    p.close()
