#!/usr/bin/env python3
"""Developer tool: time mode E / mode N device entry points for the library in $CCAL_LIB (variant A/B runs)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=10000)
ap.add_argument("--model", default="eucm")
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--cams", type=int, default=1)
ap.add_argument("--joff-mb", type=int, default=0, help="place J this many MiB into a larger allocation")
ap.add_argument("--pad-mb", type=int, default=0, help="allocate (and keep) this much before the outputs")
ap.add_argument("--one-focal", action="store_true")
ap.add_argument("--ragged", action="store_true", help="24 .. 144 corners per frame (the real sessions' shape) instead of 144")
ap.add_argument("--outliers", type=float, default=0.0, help="fraction of corners moved 5 .. 30 px off (the Huber branch of the kernels)")
ap.add_argument("--what", default="eval,normal,solve")
ap.add_argument("--tag", default=os.environ.get("CCAL_LIB", "default"))
args = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
ctx = Context(0, stream=stream.cuda_stream)
sp = synth.make_problem(args.frames, args.model, n_cams=args.cams, xy_same_focal=args.one_focal, ragged=args.ragged, outlier_frac=args.outliers)
prob = Problem.from_synth(ctx, sp)
prob.upload_params(sp.intr0, sp.poses0, sp.extr0)
out = {"tag": os.path.basename(args.tag), "frames": args.frames, "model": args.model, "cams": args.cams}
def timeit(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): fn()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
if "eval" in args.what:
    pad = torch.empty(args.pad_mb << 20, dtype=torch.uint8, device=dev) if args.pad_mb else None
    off = (args.joff_mb << 20) // 8
    r = torch.empty(prob.n_corners * 2, dtype=torch.float64, device=dev); J = torch.empty(prob.j_len + off, dtype=torch.float64, device=dev)
    jp = J.data_ptr() + off * 8
    out["J_ptr"] = hex(jp); out["r_ptr"] = hex(r.data_ptr())
    us = min(timeit(lambda: prob.eval_dev(r.data_ptr(), jp), args.reps) for _ in range(3))
    # algorithmic bytes of one pass: 5 f32 in + r[2] per corner, the block Jacobians as laid out (every camera's own width:
    # j_len sums 2 D_cam doubles per corner of that camera), one 48-B pose per slot
    algo = prob.n_corners * 36 + prob.j_len * 8 + sp.n_slots * 48
    out["eval_us"] = us; out["eval_algorithmic_bytes"] = algo; out["eval_GBps"] = algo / us / 1e3
if "normal" in args.what:
    out["normal_us"] = min(timeit(lambda: prob.build_normal_dev(0.0), args.reps) for _ in range(3))
if "solve" in args.what:
    for name, m in (("gn", 0), ("lm", 1)):
        best = None
        for _ in range(3):
            _, _, _, rep = prob.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(m))
            if best is None or rep.solve_ms < best[0]: best = (rep.solve_ms, rep.iterations, rep.final_cost)
        out[f"{name}_ms"], out[f"{name}_iters"], out[f"{name}_cost"] = best
        out[f"{name}_it_per_s"] = best[1] / best[0] * 1e3
print(json.dumps(out))
