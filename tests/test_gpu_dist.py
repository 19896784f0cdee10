"""GPU: the sharded-solve paths of ccal_solve on the one GPU of the box.
  * native RCCL inside the library (ccal_set_rccl_comm; communicator created through ccal_rccl_comm_create) on a
    1-rank communicator: the unsharded solve's result (bit-identical on the general path), groups enqueued ahead like on a single GPU;
  * the callback path (ccal_set_allreduce) with torch.distributed's RCCL on a 1-rank group: exact collective counts
    (ONE all-reduce per group, Gauss-Newton and Levenberg-Marquardt alike);
  * two ranks on the device path (both processes on cuda:0, gloo moving the device buffers).
RCCL with more than one rank needs more than one GPU: the driver's multi-GPU bench runs that (bench.py --gpus N);
multi-rank semantics are also covered on CPU with gloo (tests/test_dist_cpu.py)."""
import os
import socket

import numpy as np
import pytest

from camera_intrinsic_calibration_rs_amd import engine, synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,n_cams", [("eucm", 1), ("kb4", 1), ("eucm", 2)])
def test_native_rccl_single_rank(model, n_cams):
    """ncclAllReduce issued by the library itself (no Python, no callback in the collective)."""
    assert engine.rccl_available()
    ctx = Context(0)
    comm = ctx.rccl_comm_create(1, 0, engine.rccl_unique_id())
    try:
        sp = synth.make_problem(60, model, n_cams=n_cams, outlier_frac=0.01, ragged=True)
        gp = Problem.from_synth(ctx, sp)
        gp.apply_reference_bounds()
        for method in (0, 1):
            gp.set_rccl_comm(None)
            ref = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
            gp.set_rccl_comm(comm)
            got = gp.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
            assert ref[3].status == got[3].status == 0
            assert (ref[3].iterations, ref[3].lm_accepted, ref[3].lm_rejected) == (got[3].iterations, got[3].lm_accepted, got[3].lm_rejected)
            for a, b in zip(ref[:3], got[:3]):
                if n_cams > 1:                          # a 1-rank sum is the identity: bit for bit
                    np.testing.assert_array_equal(a, b)
                else:
                    # single camera: the unsharded solve of a session-sized problem runs single-launch groups (k_gram1v<.., ITER>:
                    # four wavefronts' rows added per workgroup, then the workgroups' rows), the sharded one Gram -> reduce ->
                    # all-reduce -> head: the same sums in another order
                    np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13)
            assert ref[3].final_cost == pytest.approx(got[3].final_cost, rel=1e-12)
        gp.set_rccl_comm(None)
        gp.close()
    finally:
        engine.rccl_comm_destroy(comm)


def test_native_rccl_empty_shard_takes_the_same_path():
    """A rank whose shard holds no observation frame still runs the single-camera loop and its collectives (with zeros):
    with one rank that is a problem without observations - Gauss-Newton has no step (NOT_PD), never a hang or a crash."""
    from camera_intrinsic_calibration_rs_amd import _ffi
    from camera_intrinsic_calibration_rs_amd.engine import make_desc
    ctx = Context(0)
    comm = ctx.rccl_comm_create(1, 0, engine.rccl_unique_id())
    try:
        d, keep = make_desc(1, [1], [512.0], [512.0], False, 3, [], [], [0], [], [], [], [], [], 1.0)
        gp = Problem(ctx, d, keep)
        gp.set_rccl_comm(comm)
        sp = synth.make_problem(3, "eucm")
        _, _, _, rep = gp.solve(sp.intr0, sp.poses0, opts=default_opts(0), raise_on_error=False)
        assert rep.status in (_ffi.OK, _ffi.ERR_NOT_PD)
        gp.close()
    finally:
        engine.rccl_comm_destroy(comm)


def test_solve_with_rccl_hook_single_rank():
    import torch
    import torch.distributed as dist
    from camera_intrinsic_calibration_rs_amd.dist import make_allreduce_hook
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=_free_port(), rank=0, world_size=1, device_id=dev)
    try:
        stream = torch.cuda.Stream(device=dev)
        ctx = Context(0, stream=stream.cuda_stream)
        sp = synth.make_problem(50, "eucm", outlier_frac=0.01)
        gp = Problem.from_synth(ctx, sp)
        calls = []
        hook = make_allreduce_hook(device=dev)

        def counting(ptr, count, st):
            calls.append(count)
            return hook(ptr, count, st)

        for method in (0, 1):
            gp.set_allreduce(None)
            ref = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
            gp.set_allreduce(counting)
            calls.clear()
            got = gp.solve(sp.intr0, sp.poses0, opts=default_opts(method))
            # the same decisions with and without the hook; the sums meet in another order (without the hook a session-sized
            # single-camera problem runs single-launch groups: rows added per workgroup of four wavefronts)
            np.testing.assert_allclose(ref[0], got[0], rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(ref[1], got[1], rtol=1e-11, atol=1e-13)
            assert ref[3].iterations == got[3].iterations
            K = gp.K
            rep = got[3]
            # with a callback no group is enqueued ahead: the first evaluation, one group per decision and one
            # re-elimination group per rejected step / missed speculation - and ONE packed all-reduce per group
            # ([A_dir | Y^T Y | model decrease | failed blocks], 2 (K+1)^2 + 2 doubles), GN and LM alike
            groups = 1 + rep.iterations + (rep.lm_rejected + rep.lm_spec_misses if method == 1 else 0)
            assert set(calls) == {2 * (K + 1) ** 2 + 2}
            assert len(calls) == groups
            if method == 1:
                assert rep.lm_spec_hits + rep.lm_spec_misses <= rep.lm_accepted
    finally:
        dist.destroy_process_group()


def _free_port():
    """The rendezvous of a test's process group: a FILE store (init_method file://...) in a fresh temporary directory - no TCP
    port to pick and lose to another process before the group binds it (EADDRINUSE on a busy box)."""
    import tempfile
    return "file://" + os.path.join(tempfile.mkdtemp(prefix="ccal_rdv_"), "store")


def _gpu_worker(rank, world, port, model, n_cams, method, n_frames, q):
    """One rank of a sharded solve on the PRODUCT path (HIP kernels + the all-reduce hook on device buffers).  The GPU box
    has one GPU: both ranks use cuda:0 and the process group is gloo (it all-reduces CUDA tensors through the host),
    which exercises everything except RCCL itself."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    from camera_intrinsic_calibration_rs_amd.dist import gather_poses, make_allreduce_hook
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = Context(0, stream=stream.cuda_stream, lib=engine._ffi.load_for_switches())      # (`fast`: CCAL_SCHURQ=1 - a switch of the second library)
    sp = synth.make_problem(n_frames, model, n_cams=n_cams, outlier_frac=0.01, ragged=True)
    shard = sp.shard(rank, world)
    gp = Problem.from_synth(ctx, shard)
    gp.apply_reference_bounds()
    calls = []
    hook = make_allreduce_hook(device=dev)

    def counting(ptr, count, st):
        calls.append(count)
        return hook(ptr, count, st)

    gp.set_allreduce(counting)
    intr, poses, extr, rep = gp.solve(shard.intr0, shard.poses0, shard.extr0, opts=default_opts(method))
    poses_all = gather_poses(poses, sp.n_slots)
    q.put((rank, intr, extr, poses_all, rep.iterations, rep.final_cost, rep.status, list(calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model,n_cams,method,fast", [("eucm", 1, 0, False), ("eucm", 1, 1, False), ("eucm", 2, 0, False), ("kb4", 2, 1, False),
                                                      ("eucm", 2, 1, True), ("ucm", 2, 0, True)])
def test_two_ranks_on_the_device_path(model, n_cams, method, fast, monkeypatch):
    """Frame-sharded solve, two ranks, HIP kernels on both: identical camera block on both ranks (bit for bit), the same
    collective sequence on both ranks, and the single-process solve of the whole problem up to summation order.
    fast: the two-camera rig's shards through k_schurq (forced: the shards are far below its 1 000-slot threshold) and the
    merged Gram launch - their partial sums feed the same all-reduce."""
    import torch.multiprocessing as mp
    if fast:
        monkeypatch.setenv("CCAL_SCHURQ", "1")           # inherited by the spawned ranks
    n_frames = 41                                      # odd: shards of 20 and 21 slots
    sp = synth.make_problem(n_frames, model, n_cams=n_cams, outlier_frac=0.01, ragged=True)
    ctx0 = Context(0)
    full = Problem.from_synth(ctx0, sp)
    full.apply_reference_bounds()
    intr1, poses1, extr1, rep1 = full.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_gpu_worker, args=(r, 2, port, model, n_cams, method, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, i0, e0, p0, it0, c0, s0, calls0), (_, i1, e1, p1, it1, c1, s1, calls1) = res
    np.testing.assert_array_equal(i0, i1); np.testing.assert_array_equal(e0, e1)
    assert calls0 == calls1 and len(calls0) > 0
    assert it0 == it1 == rep1.iterations and s0 == s1 == rep1.status == 0
    np.testing.assert_allclose(i0, intr1, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(e0, extr1, rtol=0, atol=1e-10)
    np.testing.assert_allclose(p0, poses1, rtol=0, atol=1e-9)
    assert abs(c0 - rep1.final_cost) <= 1e-10 * rep1.final_cost


# ---------------------------------------------------------------------------------------------------------------------
# Groups enqueued AHEAD of the host in a sharded solve (what the native RCCL path does by default): the group count must
# be a function of the decisions only, early-exit groups must still issue their collective, and that on every rank alike -
# with LM rejections, with a pose block that fails on ONE rank only, with a rank whose shard is empty.  One GPU: both ranks
# on cuda:0, callback transport (gloo), CCAL_FUSED_DEPTH_HOOK = 2 makes the callback path enqueue ahead like the native one.
# A child that hangs is killed by the parent (the watchdog is the parent's queue timeout; nothing re-execs).
# ---------------------------------------------------------------------------------------------------------------------
def _ahead_worker(rank, world, port, scenario, method, n_cams, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["CCAL_FUSED_DEPTH_HOOK"] = "2"              # read once per process by the library: set before it loads
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    from camera_intrinsic_calibration_rs_amd.dist import make_allreduce_hook
    from camera_intrinsic_calibration_rs_amd.engine import make_desc
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = Context(0, stream=stream.cuda_stream, lib=engine._ffi.load_for_switches())      # CCAL_FUSED_DEPTH_HOOK lives in the second library
    if scenario == "lm_rejections":                      # poor starting points on which the (CPU oracle's) LM rejects 7 / 3 steps
        sp = synth.make_problem(12, "eucm", n_cams=n_cams, outlier_frac=0.05, ragged=True, init_perturb=0.8, seed=1 if n_cams == 1 else 0xBEEF)
    else:
        sp = synth.make_problem(36, "eucm", n_cams=n_cams, outlier_frac=0.03, ragged=True, seed=0xBEEF)
    if scenario == "empty_rank1":
        if rank == 0:
            shard = sp
            gp = Problem.from_synth(ctx, shard)
        else:                                            # three pose slots, no observation frame at all
            d, keep = make_desc(1, [1], [512.0], [512.0], False, 3, [], [], [0], [], [], [], [], [], 1.0)
            gp = Problem(ctx, d, keep)
            shard = synth.make_problem(3, "eucm")
            shard = synth.dataclasses.replace(shard, intr0=sp.intr0)
    else:
        shard = sp.shard(rank, world)
        if scenario == "notpd_rank1" and rank == 1:      # every corner of rank 1's first slot (all cameras) the same board point: its 6 x 6 block is singular
            shard.p3d[: shard.obs_offsets[n_cams]] = shard.p3d[0]
        gp = Problem.from_synth(ctx, shard)
    gp.apply_reference_bounds()
    calls = []
    hook = make_allreduce_hook(device=dev)

    def counting(ptr, count, st):
        calls.append(count)
        return hook(ptr, count, st)

    gp.set_allreduce(counting)
    intr, poses, extr, rep = gp.solve(shard.intr0, shard.poses0, shard.extr0, opts=default_opts(method), raise_on_error=False)
    gp.set_allreduce(None)                               # drains the early-exit groups (their collectives included)
    q.put((rank, intr, extr, rep.status, rep.iterations, rep.lm_accepted, rep.lm_rejected, rep.lm_spec_misses, rep.final_cost, list(calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,method,n_cams", [("lm_rejections", 1, 1), ("lm_rejections", 1, 2), ("notpd_rank1", 0, 1), ("notpd_rank1", 1, 1),
                                                    ("notpd_rank1", 0, 2), ("empty_rank1", 0, 1), ("empty_rank1", 1, 1)])
def test_two_ranks_groups_enqueued_ahead(scenario, method, n_cams):
    import torch.multiprocessing as mp
    from camera_intrinsic_calibration_rs_amd import _ffi
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_ahead_worker, args=(r, 2, port, scenario, method, n_cams, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    except Exception:
        for p in procs:                                  # a hung collective: kill exactly the children started here, fail
            if p.is_alive():
                p.kill()
        raise AssertionError(f"sharded solve with groups enqueued ahead hung ({scenario})")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, i0, e0, s0, it0, a0, r0, m0, c0, calls0), (_, i1, e1, s1, it1, a1, r1, m1, c1, calls1) = res
    # every decision is a function of all-reduced sums: status, iteration count, accept / reject sequence, cost and the
    # camera block are identical on both ranks, and so is the SEQUENCE of collectives (early-exit groups included)
    assert (s0, it0, a0, r0, m0) == (s1, it1, a1, r1, m1)
    assert calls0 == calls1 and len(calls0) >= 2
    np.testing.assert_array_equal(i0, i1); np.testing.assert_array_equal(e0, e1)
    assert c0 == c1 or (np.isnan(c0) and np.isnan(c1))
    # groups = first evaluation + one per decision + one re-elimination group per rejection / missed speculation, + the
    # group that was in flight ahead of the deciding one (depth 2), which exits early on the device but still sums
    groups = 1 + it0 + (r0 + m0 if method == 1 else 0)
    assert len(calls0) in (groups, groups + 1)
    if scenario == "lm_rejections":
        assert s0 == _ffi.OK and r0 + m0 >= 1
    if scenario == "notpd_rank1":
        # Gauss-Newton: a failed pose block anywhere is tiny-solver's None on EVERY rank; LM freezes it and goes on
        assert s0 == (_ffi.ERR_NOT_PD if method == 0 else s1)
    if scenario == "empty_rank1":
        assert s0 == _ffi.OK


def _native_worker(rank, world, uid, model, n_cams, method, n_frames, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ctx = Context(rank)
    comm = ctx.rccl_comm_create(world, rank, uid)
    sp = synth.make_problem(n_frames, model, n_cams=n_cams, outlier_frac=0.01, ragged=True)
    shard = sp.shard(rank, world)
    gp = Problem.from_synth(ctx, shard)
    gp.apply_reference_bounds()
    gp.set_rccl_comm(comm)
    intr, poses, extr, rep = gp.solve(shard.intr0, shard.poses0, shard.extr0, opts=default_opts(method))
    gp.set_rccl_comm(None)
    q.put((rank, intr, extr, poses, rep.iterations, rep.final_cost, rep.status))
    gp.close()
    engine.rccl_comm_destroy(comm)


@pytest.mark.parametrize("model,n_cams,method", [("eucm", 1, 0), ("eucm", 1, 1), ("eucm", 2, 0), ("kb4", 2, 1)])
def test_native_rccl_two_ranks(model, n_cams, method):
    """ncclAllReduce issued by the library on a 2-rank communicator, one process per GPU, groups enqueued two ahead: needs
    two GPUs (skipped on the 1-GPU box; the driver's multi-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("native multi-rank RCCL needs two GPUs")
    import torch.multiprocessing as mp
    n_frames = 41
    sp = synth.make_problem(n_frames, model, n_cams=n_cams, outlier_frac=0.01, ragged=True)
    ctx0 = Context(0)
    full = Problem.from_synth(ctx0, sp)
    full.apply_reference_bounds()
    intr1, poses1, extr1, rep1 = full.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
    uid = engine.rccl_unique_id()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_native_worker, args=(r, 2, uid, model, n_cams, method, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    except Exception:
        for p in procs:
            if p.is_alive():
                p.kill()
        raise AssertionError("native 2-rank RCCL solve hung")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, i0, e0, p0, it0, c0, s0), (_, i1, e1, p1, it1, c1, s1) = res
    np.testing.assert_array_equal(i0, i1); np.testing.assert_array_equal(e0, e1)
    assert it0 == it1 == rep1.iterations and s0 == s1 == rep1.status == 0 and c0 == c1
    np.testing.assert_allclose(i0, intr1, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([p0, p1]), poses1, rtol=0, atol=1e-9)
    assert abs(c0 - rep1.final_cost) <= 1e-10 * rep1.final_cost


def test_bench_launches_its_own_ranks_on_the_gpu():
    """`python bench.py --gpus 2` with no launcher around it, on the GPU: bench.py's own parent starts two workers (CCAL_BENCH_BACKEND=gloo
    lets both ranks share the box's one GPU - the numbers mean nothing, the code path is the multi-rank one: per-rank mode E, the
    max-over-ranks protocol, the frame-sharded solves with one collective per step through the callback transport, the split
    problem cut by ccal_partition_slots) and relays rank 0's line."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CCAL_BENCH_BACKEND="gloo", CCAL_BENCH_CONFIG3_FRAMES="2000", CCAL_BENCH_NO_CONCURRENT="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--frames", "1000",
                        "--no-cpu-baseline", "--no-rig", "--no-traffic"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launcher"]["workers_spawned"] == 2 and out["launcher"]["worker_exit_codes"] == [0, 0]
    assert out["config"]["frames_total"] == 2000 and out["value"] > 0
    sh = out["extra"]["sharded_solve"]
    assert "error" not in sh and sh["gn"]["status"] == 0 and sh["lm"]["status"] == 0 and sh["frames_total"] == 2000
    c3 = out["extra"]["config3_split"]
    assert "error" not in c3 and c3["gn"]["status"] == 0 and c3["frames_total"] == 2000
