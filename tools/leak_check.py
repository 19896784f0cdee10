import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from camera_intrinsic_calibration_rs_amd import synth, api
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
ctx = Context(0)
sp1 = synth.make_problem(300, "eucm"); sp2 = synth.make_problem(100, "kb4", n_cams=2)
def cycle():
    for sp in (sp1, sp2):
        p = Problem.from_synth(ctx, sp)
        p.solve(sp.intr0, sp.poses0, sp.extr0); p.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(1))
        p.eval(sp.intr0, sp.poses0, sp.extr0); p.build_normal(sp.intr0, sp.poses0, sp.extr0)
        p.validation(0, sp.intr0, sp.poses0, sp.extr0)
        p.close()
    s = api.GenericModel("eucm", [190.9, 190.87, 254.9, 256.9, 0.63, 1.05], 512, 512)
    api.convert_model(s, api.GenericModel("kb4", [0.0] * 8, 512, 512), ctx=ctx)
for _ in range(5): cycle()
torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
for _ in range(200): cycle()
torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]
import resource
print("free before", f0, "after", f1, "delta MB", (f0 - f1) / 1e6, "host maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3)
