#!/usr/bin/env python3
"""Drives the library's own host threads for a host-sanitizer build (tools/build_hosttsan.sh; also run under tools/build_hostasan.sh's
AddressSanitizer build): ccal_solve_batch over several
contexts with rigs of DIFFERENT reduced-system sizes (their first launches set the dynamic-LDS attribute of the same kernels
concurrently - the case the guard's mutex is for), then ccal_multi_solve over three shards of one GPU (one thread per shard, the
in-process transport's barrier).  Prints TSAN-DRIVE-OK; ThreadSanitizer's reports go to stderr."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, MultiContext, MultiProblem, Problem, default_opts

# rigs (per-context helper threads) next to session-sized single-camera problems of one model (the caller's thread drives their
# lockstep groups while the helpers run the rigs)
sps = [synth.make_problem(60, "eucm", n_cams=2, seed=1), synth.make_problem(40, "opencv5", n_cams=3, seed=2),
       synth.make_problem(30, "ucm", n_cams=5, seed=3), synth.make_problem(200, "kb4", seed=4), synth.make_problem(150, "eucm", seed=5),
       synth.make_problem(150, "eucm", seed=7, ragged=True), synth.make_problem(150, "eucm", seed=8), synth.make_problem(200, "kb4", seed=9),
       synth.make_problem(80, "opencv5", seed=10), synth.make_problem(80, "opencv5", seed=11)]      # equal sizes: one lockstep group per model
ctxs = [Context(0) for _ in sps]
probs = [Problem.from_synth(c, s) for c, s in zip(ctxs, sps)]
for method in (0, 1):
    for _ in range(3):
        reps, res = Problem.solve_batch(probs, default_opts(method), starts=[(s.intr0, s.poses0, s.extr0) for s in sps])
        assert all(r.status in (0, 5) for r in reps), [r.status for r in reps]
for p in probs:
    p.close()
# round 6: the context's block cache under churn (problems of several sizes created, solved and destroyed on ONE context, with a batch on
# other contexts in between), the ragged-frame plan + sorted table (from 2 000 frames), a pinned pose array read and written in place
churn = Context(0)
for rnd in range(3):
    for frames, model, ragged in ((120, "eucm", True), (2400, "eucm", True), (300, "kb4", False), (120, "eucm", True), (2100, "ucm", True)):
        sp = synth.make_problem(frames, model, seed=20 + rnd, ragged=ragged)
        p = Problem.from_synth(churn, sp)
        a = p.solve(sp.intr0, sp.poses0, opts=default_opts(rnd % 2))
        b = p.solve(sp.intr0, sp.poses0, opts=default_opts(rnd % 2), pinned=True)
        assert a[3].status == 0 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], np.array(b[1]))
        v = p.validation(0, a[0], a[1])
        po, used = p.init_poses(sp.intr0)
        assert v[0] > 0 and used.min() >= 0
        p.close()
churn.close()
mc = MultiContext([0, 0, 0])
for model, n_cams in (("eucm", 1), ("kb4", 2)):
    sp = synth.make_problem(45, model, n_cams=n_cams, seed=6, ragged=True)
    mp = MultiProblem.from_synth(mc, sp)
    mp.apply_reference_bounds()
    for method in (0, 1):
        a = mp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        b = mp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
        assert a[3].status == 0 and np.array_equal(a[0], b[0])
    # the one-shot entry points over the shards run on the contexts' persistent helpers too
    po, used = mp.init_poses(sp.intr0)
    v1 = mp.validation(0, a[0], a[1], a[2]); v2 = mp.validation(0, a[0], a[1], a[2])
    assert v1 == v2 and used.min() >= 0
    mp.close()
mc.close()
print("TSAN-DRIVE-OK", flush=True)
os._exit(0)          # (sanitizer runtimes trip over the HIP runtime's teardown order at interpreter exit: nothing left to check)
