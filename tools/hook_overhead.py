#!/usr/bin/env python3
"""Developer tool: cost of the all-reduce hook (torch.distributed / RCCL, 1-rank group) per solver iteration."""
import os, socket, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
from camera_intrinsic_calibration_rs_amd.dist import make_allreduce_hook
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stream = torch.cuda.Stream(device=dev)
ctx = Context(0, stream=stream.cuda_stream)
for frames in (1000, 10000):
    sp = synth.make_problem(frames, "eucm")
    gp = Problem.from_synth(ctx, sp)
    out = {"frames": frames}
    for name, hook in (("no_hook", None), ("hook", make_allreduce_hook(device=dev))):
        gp.set_allreduce(hook)
        for m, mn in ((0, "gn"), (1, "lm")):
            best = min(gp.solve(sp.intr0, sp.poses0, opts=default_opts(m))[3].solve_ms for _ in range(4))
            out[f"{name}_{mn}_ms"] = best
    print(json.dumps(out))
dist.destroy_process_group()
