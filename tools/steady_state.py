import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(device=dev)
ctx = Context(0, stream=stream.cuda_stream)
for frames in (600, 1000, 10000):
    sp = synth.make_problem(frames, "eucm")
    prob = Problem.from_synth(ctx, sp)
    for m in (0, 1):
        res = {}
        for iters in (10, 50):
            o = default_opts(m); o.max_iterations = iters; o.min_abs_error_decrease = -1.0; o.min_rel_error_decrease = -1.0; o.min_error = -1.0
            best = 1e9
            for _ in range(3):
                try:
                    _, _, _, rep = prob.solve(sp.intr0, sp.poses0, sp.extr0, opts=o)
                except Exception as e:
                    rep = None
                    import re
                    best = min(best, float("nan"))
                    err = str(e)
                    continue
                best = min(best, rep.solve_ms)
            res[iters] = best
        print(json.dumps({"frames": frames, "method": m, "ms": res, "us_per_group": (res[50] - res[10]) / 40 * 1e3}))
