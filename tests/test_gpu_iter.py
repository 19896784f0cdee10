"""GPU: single-launch groups (k_gram1v<.., ITER>) - the session-size form of the single-camera loop, in which
the Gram kernel itself sums the previous launch's rows, decides and solves the camera system - against
  * the three-launch form (Gram + reduce + head; CCAL_ITER_ROWS=0) of the same library: same verdicts, iteration counts and
    accept / reject sequences, results equal to summation order, on plain, outlier-laden, LM-rejecting, one-focal, bounded,
    non-positive-definite and empty problems, every model the form takes and every lanes-per-frame mapping;
  * the oracle.
The switch is read once per process: each form runs in a fresh child process (killed by the parent if it hangs)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (frames, model, one_focal, ragged, outlier_frac, init_perturb, bounds, seed)
CASES = [
    (3, "eucm", False, False, 0.0, None, False, 3),            # fewer frames than a workgroup holds
    (41, "eucm", False, True, 0.01, None, True, 5),
    (12, "eucm", False, True, 0.05, 0.8, False, 1),              # LM rejects several steps from here (tests/test_gpu_multi.py)
    (150, "ucm", True, True, 0.02, None, True, 14),
    (300, "eucm", False, False, 0.02, None, False, 11),
    (625, "eucm", False, False, 0.0, None, False, 7),
    (200, "kb4", True, False, 0.0, None, True, 12),
    (350, "kb4", False, True, 0.03, None, False, 21),
    (1100, "eucm", True, True, 0.01, None, False, 31),           # two frames per wavefront
    (2300, "ucm", False, False, 0.0, None, False, 32),           # narrower lane mappings
    (9600, "eucm", False, False, 0.01, None, False, 41),         # k_gram2i: the single-launch form of the two-wavefronts-per-SIMD kernel (8 960 .. 10 240 frames)
    (9100, "ucm", True, True, 0.02, None, True, 42),             #   ... one focal, ragged frames, bounds
    (9000, "eucm", False, True, 0.05, 0.8, False, 43),           #   ... LM from a bad start: rejected steps, re-elimination groups
]


def _make(case):
    from camera_intrinsic_calibration_rs_amd import synth
    frames, model, one_focal, ragged, outl, pert, bounds, seed = case
    kw = dict(seed=seed, ragged=ragged, outlier_frac=outl, xy_same_focal=one_focal)
    if pert is not None:
        kw["init_perturb"] = pert
    return synth.make_problem(frames, model, **kw), bounds


def _child(q, iter_rows):
    sys.path.insert(0, ROOT)
    if iter_rows is not None:
        os.environ["CCAL_ITER_ROWS"] = str(iter_rows)
    else:
        os.environ.pop("CCAL_ITER_ROWS", None)
    from camera_intrinsic_calibration_rs_amd import _ffi
    from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
    ctx = Context(0, lib=_ffi.load_for_switches())       # CCAL_ITER_ROWS is a switch of the second library; without it: the product
    out = []
    for case in CASES:
        sp, bounds = _make(case)
        gp = Problem.from_synth(ctx, sp)
        if bounds:
            gp.apply_reference_bounds()
        row = []
        for method in (0, 1):
            for rep_i in range(2):                       # twice: the second solve starts behind the first one's early-exit launches
                i, p, _, r = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method), raise_on_error=False)
            gp.upload_params(sp.intr0, sp.poses0, sp.extr0)
            rd = gp.solve_dev(default_opts(method), raise_on_error=False)           # parameters resident
            i_d, p_d, _ = gp.download_params()
            row.append((i, p, (r.status, r.iterations, r.lm_accepted, r.lm_rejected, r.lm_spec_misses), r.final_cost, r.initial_cost,
                        i_d, p_d, (rd.status, rd.iterations)))
        # a singular pose block (every corner of the last frame the same board point): Gauss-Newton has no step, LM freezes it
        lo, hi = int(sp.obs_offsets[-2]), int(sp.obs_offsets[-1])
        sp.p3d[lo:hi] = sp.p3d[lo]
        gs = Problem.from_synth(ctx, sp)
        sing = []
        for method in (0, 1):
            _, _, _, r = gs.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method), raise_on_error=False)
            sing.append((r.status, r.iterations))
        gs.close(); gp.close()
        out.append((row, sing))
    q.put(out)


def _run(iter_rows, timeout=300):
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    p = ctxm.Process(target=_child, args=(q, iter_rows))
    p.start()
    try:
        res = q.get(timeout=timeout)
    except Exception:
        if p.is_alive():
            p.kill()                                     # exactly the child started here
        p.join(10)
        raise AssertionError(f"solve hung or crashed (exit code {p.exitcode})")
    p.join(60)
    assert p.exitcode == 0
    return res


@pytest.fixture(scope="module")
def both_forms():
    return _run(None), _run(0)


def test_single_launch_groups_equal_three_launch_groups(both_forms):
    one, three = both_forms
    # the switch did switch: the two forms add the same terms in another order, somewhere a last bit differs
    assert any(not np.array_equal(a[0], b[0]) for (r1, _), (r3, _) in zip(one, three) for a, b in zip(r1, r3))
    for case, (row1, sing1), (row3, sing3) in zip(CASES, one, three):
        assert sing1 == sing3, case                      # NOT_PD for Gauss-Newton, LM's verdict: the same in both forms
        for m, (a, b) in enumerate(zip(row1, row3)):
            i1, p1, v1, c1, c01, id1, pd1, vd1 = a
            i3, p3, v3, c3, c03, id3, pd3, vd3 = b
            assert v1 == v3, (case, m, v1, v3)           # status, iterations, accepted / rejected / missed speculations
            assert vd1 == vd3 and vd1 == v1[:2], (case, m)
            assert c01 == pytest.approx(c03, rel=1e-13)
            if v1[0] in (0, 5):
                np.testing.assert_allclose(i1, i3, rtol=1e-9, atol=1e-12)
                # (the two forms add the frames' sums in different orders: a 9 000-frame LM solve from a bad start with 5 % outliers has
                #  ended 1.1e-9 apart in ONE of its 54 000 pose entries)
                np.testing.assert_allclose(p1, p3, rtol=0, atol=1e-9 if case[0] < 5000 else 3e-9)
                assert c1 == pytest.approx(c3, rel=1e-10)
                # host-pointer and device-resident entries of one form: the same launches, the same bits
                np.testing.assert_array_equal(i1, id1); np.testing.assert_array_equal(p1, pd1)
                np.testing.assert_array_equal(i3, id3); np.testing.assert_array_equal(p3, pd3)


def test_single_launch_groups_against_the_oracle(both_forms, oracle):
    from camera_intrinsic_calibration_rs_amd.engine import default_opts
    one, _ = both_forms
    for case, (row1, _) in zip(CASES, one):
        if case[0] > 700:
            continue                                     # (the oracle's seconds)
        sp, bounds = _make(case)
        op = oracle.OracleProblem.from_synth(sp)
        if bounds:
            op.apply_reference_bounds()
        for m in (0, 1):
            io, po, _, ro = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(m))
            i1, p1, v1, c1 = row1[m][:4]
            assert (v1[0], v1[1]) == (ro.status, ro.iterations), (case, m)
            if ro.status == 0:
                P = 6 if case[1] != "ucm" else 5
                np.testing.assert_allclose(i1[0, :P], io[0, :P], rtol=1e-6)
                np.testing.assert_allclose(p1, po, rtol=0, atol=1e-6)
                assert c1 == pytest.approx(ro.final_cost, rel=1e-9)


# ---------------------------------------------------------------------------------------------------------------------------
# General (multi-camera) loop at session size: the GEN Gram kernels form the candidate poses in their prologue
# (FusedArgs::gen_backsub) - against the form with k_backsub as a launch of its own (CCAL_GEN_BACKSUB=0), fresh child each
# ---------------------------------------------------------------------------------------------------------------------------
def _rigs():
    from camera_intrinsic_calibration_rs_amd import synth
    ext2 = np.zeros((2, 6)); ext2[1] = [0.05, -0.2, 0.1, 0.1, -0.02, 0.03]
    ext3 = np.zeros((3, 6)); ext3[1] = [0.05, -0.2, 0.1, 0.1, -0.02, 0.03]; ext3[2] = [-0.1, 0.15, -0.05, -0.08, 0.04, 0.02]
    return [synth.make_problem(40, "eucm", n_cams=2, seed=13, outlier_frac=0.02),
            synth.make_rig(60, ["eucm", "kb4"], ext2, seed=21, drop_frac=0.45),                # slots that no camera observes
            synth.make_rig(50, ["ucm", "eucm", "kb4"], ext3, seed=22, drop_frac=0.3, xy_same_focal=True),
            synth.make_problem(12, "eucm", n_cams=2, outlier_frac=0.05, ragged=True, init_perturb=0.8, seed=0xBEEF)]      # LM rejections


def _child_rig(q, separate_backsub):
    sys.path.insert(0, ROOT)
    if separate_backsub:
        os.environ["CCAL_GEN_BACKSUB"] = "0"
    else:
        os.environ.pop("CCAL_GEN_BACKSUB", None)
    from camera_intrinsic_calibration_rs_amd import _ffi
    from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
    ctx = Context(0, lib=_ffi.load_for_switches())       # CCAL_GEN_BACKSUB=0 is a switch of the second library; without it: the product
    out = []
    for sp in _rigs():
        gp = Problem.from_synth(ctx, sp)
        row = []
        for method in (0, 1):
            for _ in range(2):
                i, p, e, r = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method), raise_on_error=False)
            gp.upload_params(sp.intr0, sp.poses0, sp.extr0)
            rd = gp.solve_dev(default_opts(method), raise_on_error=False)
            i_d, p_d, e_d = gp.download_params()
            row.append((i, p, e, (r.status, r.iterations, r.lm_accepted, r.lm_rejected, r.lm_spec_misses), r.final_cost, i_d, p_d, e_d, (rd.status, rd.iterations)))
        gp.close()
        out.append(row)
    q.put(out)


def _run_rig(separate_backsub, timeout=300):
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    p = ctxm.Process(target=_child_rig, args=(q, separate_backsub))
    p.start()
    try:
        res = q.get(timeout=timeout)
    except Exception:
        if p.is_alive():
            p.kill()
        p.join(10)
        raise AssertionError(f"solve hung or crashed (exit code {p.exitcode})")
    p.join(60)
    assert p.exitcode == 0
    return res


def test_candidate_poses_formed_in_the_gram_prologue_equal_k_backsub(oracle):
    from camera_intrinsic_calibration_rs_amd.engine import default_opts
    fused, separate = _run_rig(False), _run_rig(True)
    assert any(len(np.unique(sp.obs_slot)) < sp.n_slots for sp in _rigs())          # a rig with slots that no camera observes is among them
    for sp, rf, rs in zip(_rigs(), fused, separate):
        unobserved = np.setdiff1d(np.arange(sp.n_slots), np.unique(sp.obs_slot))
        op = oracle.OracleProblem.from_synth(sp)
        for m, (a, b) in enumerate(zip(rf, rs)):
            i1, p1, e1, v1, c1, id1, pd1, ed1, vd1 = a
            i2, p2, e2, v2, c2, id2, pd2, ed2, vd2 = b
            assert v1 == v2 and vd1 == vd2 == v1[:2], (m, v1, v2)
            if v1[0] in (0, 5):
                # the same operations in the same order (every FMA spelled out, the step kept out of the pose update's addition): the same bits
                np.testing.assert_array_equal(i1, i2); np.testing.assert_array_equal(p1, p2); np.testing.assert_array_equal(e1, e2)
                assert c1 == c2
                np.testing.assert_array_equal(i1, id1); np.testing.assert_array_equal(p1, pd1); np.testing.assert_array_equal(e1, ed1)
                # a slot that no camera observes keeps the pose it came with, whichever parameter set ends up the accepted one
                np.testing.assert_array_equal(p1[unobserved], sp.poses0[unobserved])
                np.testing.assert_array_equal(pd1[unobserved], sp.poses0[unobserved])
            io, po, eo, ro = op.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(m))
            assert (v1[0], v1[1]) == (ro.status, ro.iterations)
            if ro.status == 0:
                np.testing.assert_allclose(p1, po, rtol=0, atol=1e-6); np.testing.assert_allclose(e1, eo, rtol=0, atol=1e-6)


def test_zero_copy_result_is_the_device_state(gpu_ctx):
    """Session-sized ccal_solve returns from its poll of the status word without a copy and without a synchronise: the launch that
    said `done` wrote intrinsics and poses into pinned host memory BEFORE it published the word (head_finish's publish-last
    invariant).  The host's copy must be what the device holds once the stream has drained - for the single-launch form, the
    three-launch form's sizes (2 000+ frames: staged copies) and LM, repeated (a stale word of the previous solve must not end the
    next one early)."""
    from camera_intrinsic_calibration_rs_amd import synth
    from camera_intrinsic_calibration_rs_amd.engine import Problem, default_opts
    for frames, model, ragged in ((60, "eucm", False), (625, "eucm", False), (300, "kb4", False), (2500, "eucm", False),
                                  (5000, "ucm", True), (9700, "eucm", False), (10000, "eucm", True), (7000, "kb4", True)):
        # (from 2 049 frames every workgroup of the finishing single-launch group writes its slice of the poses, the last one to
        #  count itself in publishes the word: head_finish, SPREAD - k_gram1v's form, k_gram2i, k_gram2i over the folded table)
        sp = synth.make_problem(frames, model, seed=frames, outlier_frac=0.01, ragged=ragged)
        gp = Problem.from_synth(gpu_ctx, sp)
        for method in (0, 1, 0):
            for pinned in ((False, True) if frames > 2048 else (False,)):
                i, p, _, r = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method), pinned=pinned)
                i_d, p_d, _ = gp.download_params()              # synchronises the stream, reads the accepted set
                assert r.status == 0
                np.testing.assert_array_equal(p, p_d)
                P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[model]]
                np.testing.assert_array_equal(i[0, :P], i_d[0, :P])
        gp.close()
