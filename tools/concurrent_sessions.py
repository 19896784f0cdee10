#!/usr/bin/env python3
"""Developer tool / bench helper: N independent session-sized problems solved concurrently, one host thread + one
context (own HIP stream) + one problem each - the regime of the reference's real workload (a single camera session of a
few hundred frames; the per-camera calib_camera calls of a rig, the three retries of src/bin/camera_calibration.rs:205-246).
Reports aggregate Gauss-Newton iterations/s for N = 1, 2, 4, 8 and checks every concurrent result against the sequential one
(same iteration count; results equal to the order of summation, 1e-11: a batch sizes each problem's launches for its share of the GPU).
    python tools/concurrent_sessions.py [frames] [model] [--lm]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def measure(frames=625, model="eucm", method=0, reps=200, counts=(1, 2, 4, 8), device=0):
    """ccal_solve_batch over N problems on N contexts (one host thread, host pointers in and out: every solve stages its
    starting point and fetches its result like ccal_solve does) against the same problems solved one after the other."""
    from camera_intrinsic_calibration_rs_amd import synth
    from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
    nmax = max(counts)
    sps = [synth.make_problem(frames, model, seed=0xC0FFEE + 17 * i) for i in range(nmax)]
    ctxs = [Context(device) for _ in range(nmax)]                  # own stream each
    probs = [Problem.from_synth(c, s) for c, s in zip(ctxs, sps)]
    opts = default_opts(method)
    starts = [(s.intr0, s.poses0, s.extr0) for s in sps]
    ref = []
    for p, s in zip(probs, sps):                                   # sequential reference results (and warm-up)
        p.solve(s.intr0, s.poses0, s.extr0, opts=opts)
        ref.append(p.solve(s.intr0, s.poses0, s.extr0, opts=opts))
    out = {"frames": frames, "model": model, "method": "lm" if method else "gn", "batches_timed": reps,
           "how": "ONE ccal_solve_batch call per batch, one context (stream + host thread inside the library) per problem, host pointers in / out", "by_sessions": {}}
    for n in counts:
        for _ in range(5):
            Problem.solve_batch(probs[:n], opts, starts=starts[:n])
        wall = None
        for _ in range(3):                                         # best of three timed runs (host threads: scheduling noise)
            t0 = time.perf_counter()
            for _ in range(reps):
                rp, res = Problem.solve_batch(probs[:n], opts, starts=starts[:n])
            w = time.perf_counter() - t0
            wall = w if wall is None or w < wall else wall
        same = all(rp[i].iterations == ref[i][3].iterations and abs(rp[i].final_cost - ref[i][3].final_cost) <= 1e-11 * abs(ref[i][3].final_cost) and
                   np.allclose(res[i][0], ref[i][0], rtol=1e-11, atol=1e-13) and np.allclose(res[i][1], ref[i][1], rtol=1e-11, atol=1e-13) for i in range(n))
        out["by_sessions"][str(n)] = {"wall_s": wall, "solves_per_s": n * reps / wall,
                                      "iters_per_s": sum(r.iterations for r in rp) * reps / wall,
                                      "ms_per_batch": wall / reps * 1e3, "equal_to_sequential_1e-11": bool(same)}
    b = out["by_sessions"]
    if "1" in b:
        for k in b: b[k]["speedup_vs_1"] = b[k]["iters_per_s"] / b["1"]["iters_per_s"]
    for p in probs: p.close()
    for c in ctxs: c.close()
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    frames = int(args[0]) if args else 625
    model = args[1] if len(args) > 1 else "eucm"
    print(json.dumps(measure(frames, model, 1 if "--lm" in sys.argv else 0)))
