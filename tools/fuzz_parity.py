#!/usr/bin/env python3
"""Developer tool: randomised parity sweep, device vs oracle (mode E, normal equations, GN / LM solves) over random
models, camera counts, frame counts, ragged frames, outliers, one-focal, bounds and disabled distortions."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, MultiContext, MultiProblem, Problem, default_opts, CcalError
from oracle import binding as ob

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--shards", type=int, default=0, help="also solve every case through ccal_multi_solve over this many shards of GPU 0 "
                "(in-process transport) and hold it against the unsharded device solve: verdict, iterations, intrinsics")
ap.add_argument("--batch", type=int, default=0, help="also re-solve the cases in groups of this many through ONE ccal_solve_batch call per group "
                "(lockstep groups for session-sized single-camera problems of one model, per-context drivers for the rest) and hold every member "
                "against its own ccal_solve: verdict, iterations, intrinsics, poses")
ap.add_argument("--huge", action="store_true", help="big cases also take 7 400 / 9 300 frames (ragged: the folded one-bin plans of 16 / 12 lanes; the oracle solve of one takes seconds)")
ap.add_argument("--big", type=float, default=0.04, help="probability that a single-camera case has 1 100 / 2 300 / 5 200 frames (from 2 000 ragged frames "
                "the Gram launch is binned by corner count, k_gram2b): raise it for a sweep of that launch")
args = ap.parse_args()
mctx = MultiContext([0] * args.shards) if args.shards > 1 else None
worst_sh = {"intr": 0.0, "poses": 0.0}; n_sh = 0
rng = np.random.default_rng(args.seed)
ctx = Context(0, lib=_ffi.load_for_switches())      # developer switches (CCAL_SCHURQ=1, ...) live in the second library
pending = {0: [], 1: []}; n_batched = 0; worst_b = {"intr": 0.0, "poses": 0.0}
def flush_batch(method, lst):
    """ONE ccal_solve_batch over the pending problems of one method; every member against its own earlier ccal_solve."""
    global n_batched
    probs = [e[0] for e in lst]
    try:
        reps, res = Problem.solve_batch(probs, default_opts(method), starts=[(e[1].intr0, e[1].poses0, e[1].extr0) for e in lst])
    except CcalError as ex:
        fails.append(dict(what="solve_batch raised", err=repr(ex), cases=[e[4] for e in lst][:3])); reps = None
    if reps is not None:
        for (gp_, sp_, g_, gs_, case_), rep, r in zip(lst, reps, res):
            n_batched += 1
            bs = (rep.status, rep.iterations if rep.status == 0 else -1)
            if bs[0] != gs_[0] or (gs_[0] == 0 and bs[1] != gs_[1]):
                fails.append(dict(case=case_, what="batch vs single verdict", batch=bs, single=gs_, method=method))
            elif gs_[0] == 0:
                dbi = float((np.abs(r[0] - g_[0]) / np.maximum(np.abs(g_[0]), 1e-3)).max()); dbp = float(np.abs(r[1] - g_[1]).max())
                worst_b["intr"] = max(worst_b["intr"], dbi); worst_b["poses"] = max(worst_b["poses"], dbp)
                if dbi > 1e-7 or dbp > 1e-7:
                    fails.append(dict(case=case_, what="batch vs single result", d_intr=dbi, d_poses=dbp, method=method))
    for e in lst:
        e[0].close()
t0 = time.time(); n = 0; worst = {"r": 0.0, "J": 0.0, "S": 0.0, "intr": 0.0, "poses": 0.0}; fails = []; both_none = []; both_none_sh = []
while time.time() - t0 < args.seconds:
    model = rng.choice(["ucm", "eucm", "kb4", "opencv5"])
    n_cams = int(rng.choice([1, 1, 1, 1, 2, 2, 3, 3, 5, 8]))      # 5 and 8 cameras: reduced systems of 64 .. 114 columns
    frames = int(rng.choice([3, 7, 20, 45, 130, 300])) if n_cams == 1 else int(rng.choice([5, 12, 30]))
    if n_cams == 1 and rng.random() < args.big:        # every lanes-per-frame mapping / both register-Gram kernels / k_schur1m
        frames = int(rng.choice([1100, 2300, 5200] + ([7400, 9300] if args.huge else [])))
    kw = dict(n_cams=n_cams, seed=int(rng.integers(1, 1 << 30)), ragged=bool(rng.integers(0, 2)),
              xy_same_focal=bool(rng.integers(0, 2)), outlier_frac=float(rng.choice([0.0, 0.01, 0.05])))
    force_rig = n_cams >= 5 and model == "opencv5"      # narrow field of view: no pose that every camera of a big rig sees
    if not force_rig:
        sp = synth.make_problem(frames, model, **kw)
    if n_cams > 1 and (force_rig or rng.random() < 0.5):
        # a rig of different cameras: random model per camera, extrinsic rotations up to ~0.45 rad, uneven visibility
        models = [str(rng.choice(["ucm", "eucm", "kb4", "opencv5"])) for _ in range(n_cams)]
        if rng.random() < 0.4: models = [str(model)] * n_cams          # rigs of one model: the merged Gram launch (and k_schurq under CCAL_SCHURQ=1)
        ext = np.zeros((n_cams, 6))
        narrow = "opencv5" in models
        ext[1:, :3] = rng.uniform(-0.26, 0.26, (n_cams - 1, 3)) * (0.4 if narrow else 1.0)
        ext[1:, 3:] = rng.uniform(-0.12, 0.12, (n_cams - 1, 3))
        sp = synth.make_rig(frames * 2, models, ext, seed=kw["seed"], xy_same_focal=kw["xy_same_focal"],
                            ragged=kw["ragged"], drop_frac=float(rng.choice([0.0, 0.2, 0.4])))
        if np.bincount(sp.obs_cam, minlength=n_cams).min() < 3:
            continue
        kw = dict(kw, rig=models, ext=ext[1:].round(3).tolist())
    gp = Problem.from_synth(ctx, sp); op = ob.OracleProblem.from_synth(sp)
    case = dict(model=str(model), frames=frames, **kw)
    try:
        r, J = gp.eval(sp.intr0, sp.poses0, sp.extr0); ro, Jo = op.eval(sp.intr0, sp.poses0, sp.extr0)
        worst["r"] = max(worst["r"], float(np.abs(r - ro).max()))
        worst["J"] = max(worst["J"], float((np.abs(J - Jo) / np.maximum(1.0, np.abs(Jo))).max()))
        lam = float(rng.choice([0.0, 1e-4]))
        S, b, c = gp.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam); So, bo, co = op.build_normal(sp.intr0, sp.poses0, sp.extr0, lam=lam)
        worst["S"] = max(worst["S"], float(np.abs(S - So).max() / np.abs(So).max()))
        bounds = bool(rng.integers(0, 2))
        if bounds:
            gp.apply_reference_bounds(); op.apply_reference_bounds()
        method = int(rng.integers(0, 2))
        o = default_opts(method)
        try:
            g = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=o); gs = (g[3].status, g[3].iterations)
        except CcalError as e:
            g = None; gs = (e.code, -1)
        orc = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=o); os_ = (orc[3].status, orc[3].iterations)
        if mctx is not None:
            mp = MultiProblem.from_synth(mctx, sp)
            if bounds:
                mp.apply_reference_bounds()
            try:
                m = mp.solve(sp.intr0, sp.poses0, sp.extr0, opts=o); ms = (m[3].status, m[3].iterations)
            except CcalError as e:
                m = None; ms = (e.code, -1)
            mp.close(); n_sh += 1
            if ms[0] != 0 and gs[0] != 0 and ms[0] != gs[0]:
                # NEITHER form converges, with different codes (a system singular to rounding: NOT_PD after a few steps in one order of
                # summation, iterations exhausted in the other - the oracle is a third opinion): counted apart, like gpu-vs-oracle below
                both_none_sh.append(dict(case=case, sharded=ms, unsharded=gs, oracle=os_))
            elif ms[0] != gs[0] or (gs[0] == 0 and ms[1] != gs[1]):
                # a verdict that differs between the sharded and the unsharded DEVICE solve (summation order can move a marginal
                # convergence test by one iteration on a flat optimum: reported, judged with the deviations below)
                fails.append(dict(case=case, what="sharded vs unsharded verdict", sharded=ms, unsharded=gs, method=method))
            elif gs[0] == 0:
                dsi = float((np.abs(m[0] - g[0]) / np.maximum(np.abs(g[0]), 1e-3)).max()); dsp = float(np.abs(m[1] - g[1]).max())
                worst_sh["intr"] = max(worst_sh["intr"], dsi); worst_sh["poses"] = max(worst_sh["poses"], dsp)
                if dsi > 1e-7 or dsp > 1e-7:
                    fails.append(dict(case=case, what="sharded vs unsharded result", d_intr=dsi, d_poses=dsp, method=method))
        if g is None or gs[0] != 0 or os_[0] != 0:
            if (gs[0] == 0) != (os_[0] == 0):
                fails.append(dict(case=case, what="one side solved, the other did not", gpu=gs, oracle=os_))
            elif gs[0] != os_[0]:
                both_none.append(dict(case=case, gpu=gs, oracle=os_))      # the reference's None on both sides, different code
        else:
            P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[str(model)]]
            scale = np.maximum(np.abs(orc[0][:, :P]), 1e-3)
            di = float((np.abs(g[0][:, :P] - orc[0][:, :P]) / scale).max()); dp = float(np.abs(g[1] - orc[1]).max())
            worst["intr"] = max(worst["intr"], di); worst["poses"] = max(worst["poses"], dp)
            if gs != os_ or di > 1e-6 or dp > 1e-6:
                fails.append(dict(case=case, what="solve", gpu=gs, oracle=os_, d_intr=di, d_poses=dp, method=method))
    except Exception as e:  # noqa: BLE001
        fails.append(dict(case=case, what="exception", err=repr(e)))
        g = None
    n += 1
    if args.batch > 1 and g is not None and frames <= 2300:
        pending[method].append((gp, sp, g, gs, case))
        if len(pending[method]) >= args.batch:
            flush_batch(method, pending[method]); pending[method] = []
    else:
        gp.close()
for mth, lst in pending.items():
    if lst: flush_batch(mth, lst)
print(json.dumps(dict(cases=n, worst=worst, batched_cases=n_batched, batch=args.batch, worst_batch_vs_single=worst_b, sharded_cases=n_sh, shards=args.shards, worst_sharded_vs_unsharded=worst_sh, n_fail=len(fails), fails=fails[:6], n_both_none_different_code=len(both_none),
                      both_none=both_none[:3], n_sharded_both_none_different_code=len(both_none_sh), sharded_both_none=both_none_sh[:3]), indent=1))
