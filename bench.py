#!/usr/bin/env python3
"""Headline benchmark: corner residual+Jacobian evals/s (mode E) on synthetic calib frames.

    python bench.py --gpus N --steps K --warmup W [--frames F | --frames-total F] [--model eucm]

One process per GPU.  Under torch.distributed.run the launcher sets RANK/LOCAL_RANK/WORLD_SIZE; started plainly with
`--gpus N` (N > 1, WORLD_SIZE unset) this file is its own launcher: the parent - which never touches the GPU - starts N fresh
worker processes of itself, relays rank 0's line and exits non-zero when a worker fails or fewer than N GPUs are visible.
A "step" is one pass
of the hot path over one batch: every rank evaluates r[2] + J[2 x D] for all corners of its frame
shard (F frames x 144 corners per GPU -- weak scaling; mode E needs no collective).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
`--frames-total F` instead splits ONE F-frame problem over the ranks by contiguous slot range (strong scaling:
`--gpus 8 --frames-total 50000` is BASELINE.json configs[3] as written); without it a multi-GPU run still carries that
split 50 000-frame solve as `extra.config3_split` next to the weak-scaling numbers.

The workload defaults to the north-star headline (10 000 frames x 144 corners, EUCM, per GPU);
`--frames 1000` is BASELINE.json configs[1].  PyTorch is used only for the process group, the
barrier and HIP events; the kernels are the hand-written HIP library behind include/ccal.h, and the
sharded solves of the `extra` section all-reduce through the library's own RCCL communicator
(ccal_set_rccl_comm) - torch.distributed only carries the 128-byte communicator id to the ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6         # FP64 vector = matrix peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
CLOCK_RAMP_S = float(os.environ.get("CCAL_BENCH_CLOCK_RAMP_S", "0.1"))      # untimed launches before the W warm-up steps (idle power state ->
                                # run clocks); the variable exists for with / without-ramp A/B tables (tools/gpu_r05_ab.sh), the default is what counts


def _latest_profile_file(name):
    import re
    try:
        dirs = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+", d))
    except OSError:
        return None
    for d in reversed(dirs):
        p = os.path.join(ROOT, "profiles", d, name)
        if os.path.exists(p):
            return p
    return None


def _gram_kernel_key(model, one_focal, frames):
    """Which Gram kernel the launchers (use_gram2 / launch_gram1v_t, ccal_kernels_fused.hip) pick for a single camera, as a key of
    profiles/*/flops.json: k_gram2 (the block's rows on neighbouring lanes) for OPENCV5 and, from 2 000 frames, every model;
    k_gram1v (all accumulators in registers / AGPRs) for smaller UCM / EUCM / KB4 problems."""
    of = "one-focal" if one_focal else "two-focal"
    if model == "opencv5" or frames >= 2000:
        return f"k_gram2<{model.upper()},{of}>"
    return f"k_gram1v<{model.upper()},{of}>"


_REAL_STDOUT = None


def _stdout_for_the_line_only():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when its first
    communicator comes up: 'RCCL version : ... Librccl path : ...'), so file descriptor 1 is pointed at stderr for the
    whole run and the line is written to the saved descriptor at the end."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _sig(x, n=4):
    """n significant digits (the summary has to stay short)."""
    try:
        return float(f"{float(x):.{n}g}")
    except (TypeError, ValueError):
        return None


def _summary(out):
    """The numbers this repo claims, in <= 1 500 characters, as the LAST key of the line: the driver keeps the tail of stdout, and
    the line is ~20 KB.  Every value is copied from the full blocks in front of it (ms unless the key says otherwise; [host
    pointers, device-resident] pairs for solves; frac = fraction of the 8 TB/s HBM / 78.6 TF FP64 peak)."""
    ex = out.get("extra") or {}
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    def pair(blk, name):
        return [_sig(g(blk, name, "solve_ms")), _sig(g(blk, name + "_device_resident", "solve_ms"))]
    s = {"E_ms": _sig(g(out, "roofline", "kernel_ms")), "E_frac_hbm": _sig(g(out, "roofline", "frac")),
         "N_build_ms": {"eucm": _sig(ex.get("mode_N_build_ms")), "kb4": _sig(g(ex, "config2", "kb4", "mode_N_build_ms")),
                        "opencv5": _sig(g(ex, "config2", "opencv5", "mode_N_build_ms"))},
         "N_frac_fp64": {"eucm": _sig(g(ex, "mode_N_roofline", "frac_fp64")), "kb4": _sig(g(ex, "config2", "kb4", "mode_N_roofline", "frac_fp64")),
                         "opencv5": _sig(g(ex, "config2", "opencv5", "mode_N_roofline", "frac_fp64"))},
         "gn_ms": {"f10000": [_sig(ex.get("gn_solve_ms")), _sig(g(ex, "gn_device_resident", "solve_ms"))],
                   "f625": pair(ex.get("frames625"), "gn")},
         "lm_ms": {"f10000": [_sig(ex.get("lm_solve_ms")), _sig(g(ex, "lm_device_resident", "solve_ms"))],
                   "f625": pair(ex.get("frames625"), "lm")},
         "iters": {"gn": ex.get("gn_iterations"), "lm": ex.get("lm_iterations")},
         "cam2": {"build_ms": _sig(g(ex, "two_cameras", "build_ms")), "gn_ms": pair(ex.get("two_cameras"), "gn")},
         "config0_total_ms": _sig(g(ex, "config0", "gpu_ms", "total")),
         "batch8_ms": _sig(g(ex, "concurrent_sessions", "by_sessions", "8", "ms_per_batch"))}
    for key in ("ragged", "ragged50k"):
        r = ex.get(key)
        if isinstance(r, dict) and "error" not in r:
            s[key] = {"corners": r.get("corners"), "E_ms": _sig(g(r, "mode_E", "kernel_ms")), "E_frac_hbm": _sig(g(r, "mode_E", "frac_hbm")),
                      "N_build_ms": _sig(g(r, "mode_N", "build_ms")), "N_frac_fp64": _sig(g(r, "mode_N", "frac_fp64")),
                      "gn_ms": pair(r, "gn"), "lm_ms": pair(r, "lm")}
            if isinstance(r.get("two_cameras"), dict):
                s[key]["cam2_build_ms"] = _sig(g(r, "two_cameras", "build_ms")); s[key]["cam2_gn_ms"] = pair(r.get("two_cameras"), "gn")
        elif r is not None:
            s[key] = "error"
    sh = g(ex, "single_process_sharded")
    if isinstance(sh, dict):
        s["allreduce_us_per_step"] = {k: _sig(g(sh, k, "allreduce_us_per_step")) for k in ("in_process", "rccl") if isinstance(sh.get(k), dict)}
    if out.get("allreduce_us_per_step") is not None:
        s.setdefault("allreduce_us_per_step", {})["rccl_ranks_%s" % out.get("rccl_ranks")] = _sig(out["allreduce_us_per_step"])
    s["gpu_over_cpu"] = _sig(out.get("gpu_over_cpu"))
    s["parity_pass"] = g(out, "parity", "pass")
    return s


def _emit(out):
    if "roofline" in out and "dry_run" not in out:
        out.pop("summary", None)
        try:
            out["summary"] = _summary(out)          # LAST key (the launcher's block, added later, goes in front of it)
        except Exception as e:  # noqa: BLE001
            out["summary"] = {"error": repr(e)}
    line = (json.dumps(out) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is not None:
        os.write(_REAL_STDOUT, line)
    else:
        os.write(1, line)


def _under_rocprofiler() -> bool:
    """rocprofv3 preloads its tool library into the process it profiles."""
    try:
        with open("/proc/self/maps") as f:
            return any("rocprofiler-sdk-tool" in line or "rocprofv3" in line for line in f)      # (librocprofiler-register.so is always there)
    except OSError:
        return False


def _eval_kernel_source_hash():
    """sha256 (16 hex digits) over the sources of the headline kernel: what a committed PMC figure was measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in ("ccal_kernels_eval.hip", "ccal_device.hpp", "ccal_models.hpp"):
        with open(os.path.join(ROOT, "camera_intrinsic_calibration_rs_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _pmc_traffic_now(args):
    """HBM bytes per k_eval launch measured by THIS run: two short child runs of this file under rocprofv3 - `--pmc FETCH_SIZE`, then
    `--pmc WRITE_SIZE` (they do not fit one pass; counters alone with --kernel-trace, as MI355X_MICROARCH.md prescribes) - on the same
    workload, the guide's corrections applied (both counters are KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes:
    x 2).  Returns (bytes per launch or None, how)."""
    import csv, shutil, subprocess, tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    means = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ccal_pmc_", dir="/tmp")
        try:
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "2", "--frames", str(args.frames),
                   "--model", args.model, "--no-cpu-baseline", "--no-extra"]
            env = dict(os.environ, CCAL_BENCH_TRAFFIC_CHILD="1", TMPDIR="/tmp")
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
            path = None
            for dp, _dn, fn in os.walk(d):
                for f in fn:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(dp, f)
            if r.returncode != 0 or path is None:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            per_dispatch = {}
            for row in csv.DictReader(open(path)):
                if "k_eval" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    per_dispatch[row["Dispatch_Id"]] = per_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])      # one row per XCD / instance
            if not per_dispatch:
                return None, f"no k_eval dispatches in the {ctr} pass"
            means[ctr] = sum(per_dispatch.values()) / len(per_dispatch)
        except Exception as e:  # noqa: BLE001
            return None, f"rocprofv3 --pmc {ctr}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = 2.0 * means["FETCH_SIZE"] * 1024.0 + means["WRITE_SIZE"] * 1024.0
    return traffic, ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over child runs of this "
                     "command on the same workload; KiB units, FETCH_SIZE x 2 (gfx950), mean over the k_eval dispatches")


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch_workers(n, argv):
    """`python bench.py --gpus N` without a launcher: be the launcher.  This process never initialises the GPU (counting devices
    does not, on this image); it starts N fresh children of this file (one rank per GPU, rendezvous on 127.0.0.1), relays rank 0's
    JSON line - with a `launcher` block saying what was started - and returns non-zero when any worker failed, when fewer than N
    GPUs are visible, or when no line came back."""
    import subprocess
    backend = os.environ.get("CCAL_BENCH_BACKEND", "nccl")
    dry = os.environ.get("CCAL_BENCH_DRYRUN") == "1"
    if backend == "nccl" and not dry:
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] cannot count GPUs: {e!r}", file=sys.stderr, flush=True)
            return 4
        if have < n:
            print(f"[bench] --gpus {n} asked for, {have} GPU(s) visible: refusing to run on fewer (no silent fallback)", file=sys.stderr, flush=True)
            return 4
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CCAL_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    # rank 0's pipe is drained by a thread (it prints its one line at the very end); this thread polls EVERY child: the first
    # worker that exits non-zero - too few GPUs for its rank, an import error, an RCCL initialisation failure - ends the run at once
    # (its peers would sit in init_process_group or a barrier with no partner until some timeout), as torchrun does
    import threading
    chunks = []
    def drain():
        try:
            for chunk in iter(lambda: procs[0].stdout.read(65536), b""):
                chunks.append(chunk)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reading rank 0's output failed: {e!r}", file=sys.stderr, flush=True)
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    limit = float(os.environ.get("CCAL_BENCH_LAUNCH_TIMEOUT", "3600"))
    t_start = time.time()
    codes = [None] * n
    failed = None                                # (rank, code) of the first worker that failed
    grace = None                                 # once rank 0 is done: the other ranks leave their last barrier together with it
    while any(c is None for c in codes):
        for i, pr in enumerate(procs):
            if codes[i] is None:
                codes[i] = pr.poll()
                if codes[i] not in (None, 0) and failed is None:
                    failed = (i, codes[i])
        now = time.time()
        if failed is not None:
            print(f"[bench] rank {failed[0]} exited with code {failed[1]}: stopping the other workers", file=sys.stderr, flush=True)
            break
        if codes[0] is not None and grace is None:
            grace = now + 120.0
        if now - t_start > limit:
            print("[bench] the workers did not finish in time: stopping them", file=sys.stderr, flush=True)
            break
        if grace is not None and now > grace:
            print(f"[bench] workers still running 120 s after rank 0 finished (exit codes so far {codes}): stopping them", file=sys.stderr, flush=True)
            break
        time.sleep(0.2)
    for i, pr in enumerate(procs):
        if codes[i] is None:
            pr.kill()                            # exactly the child started here
            pr.wait()
            codes[i] = -9
    reader.join(timeout=10.0)
    line = b"".join(chunks)
    out = None
    for ln in line.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                out = json.loads(ln)
            except ValueError:
                pass
    bad = failed[1] if failed is not None else next((c for c in codes if c != 0), 0)
    if out is None:
        print(f"[bench] no JSON line from rank 0 (worker exit codes {codes})", file=sys.stderr, flush=True)
        return bad or 5
    out["launcher"] = {"kind": "bench.py parent process (WORLD_SIZE was unset)", "workers_spawned": n, "worker_exit_codes": codes,
                       "master": f"127.0.0.1:{port}"}
    _emit(out)
    return bad


def _dry_run(args, rank, world):
    """CCAL_BENCH_DRYRUN=1 (tests/test_bench_cpu.py; no GPU needed): everything of a multi-rank run that is NOT the GPU - the
    launcher, the gloo rendezvous, the product's own partition of the problem (ccal_partition_slots: host code of libccal_hip.so),
    the barrier / max-over-ranks timing protocol and the one line - with a step that does nothing.  The line says so
    (`dry_run`, value 0): it is not a measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("CCAL_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        # (tests/test_bench_cpu.py: a worker that dies before the rendezvous - its peers wait for it there; the launcher must not)
        print(f"[bench] dry run: rank {rank} fails on request", file=sys.stderr, flush=True)
        sys.exit(7)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from camera_intrinsic_calibration_rs_amd import engine, synth
    strong = args.frames_total > 0
    total_frames = args.frames_total if strong else args.frames * world
    if strong:
        sp = synth.make_problem(args.frames_total, args.model, seed=0xC0FFEE, ragged=True)
        d, _keep = engine.desc_from_synth(sp)
        first = engine.partition_slots(d, world)
        mine = sp.slot_slice(first[rank], first[rank + 1])
    else:
        first = None
        mine = synth.make_problem(args.frames, args.model, seed=0xC0FFEE + 1000003 * rank)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pass
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(mine.n_corners), float(mine.n_slots)], dtype=torch.float64)
    mx = t.clone()
    if world > 1:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        _emit({"metric": "corner residual+Jacobian evals/sec; LM iters/sec to converge (EUCM, TUM-VI cam0)", "value": 0.0,
               "unit": "corner residual+Jacobian evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": float(mx[0]) / max(args.steps, 1) * 1e3, "higher_is_better": True,
               "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "dry_run": "CCAL_BENCH_DRYRUN=1: launcher, rendezvous (gloo), partition and timing protocol only - no GPU work, not a measurement",
               "config": {"workload": f"dry run, {total_frames} frames, {args.model.upper()}", "frames_total": total_frames,
                          "slots_all_ranks": int(t[2]), "corners_all_ranks": int(t[1]),
                          "partition": first, "partition_by": "ccal_partition_slots (libccal_hip.so, host code)" if strong else None}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    _stdout_for_the_line_only()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU (144 corners each)")
    ap.add_argument("--frames-total", type=int, default=0,
                    help="strong scaling: ONE problem of this many frames, every rank takes its slot range (overrides --frames)")
    ap.add_argument("--model", default="eucm", choices=["ucm", "eucm", "kb4", "opencv5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (mode N / solver) measurements")
    ap.add_argument("--no-rig", action="store_true", help="skip the two-camera leg of the secondary measurements")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic (two short child runs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this process: be one (N fresh workers; this parent never touches the GPU)
        sys.exit(_launch_workers(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if os.environ.get("CCAL_BENCH_DRYRUN") == "1":
        return _dry_run(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    # CCAL_BENCH_BACKEND=gloo (developer switch): exercise the multi-rank code path on a box with fewer GPUs than
    # ranks - ranks then share devices, the collective goes through the callback; the numbers mean nothing there
    backend = os.environ.get("CCAL_BENCH_BACKEND", "nccl")
    n_visible = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_visible:
        print(f"[bench] rank {rank}: local rank {local_rank} but only {n_visible} GPU(s) visible", file=sys.stderr, flush=True)
        sys.exit(4)
    dev_index = local_rank if backend == "nccl" else local_rank % max(n_visible, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from camera_intrinsic_calibration_rs_amd import engine, synth
    from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts

    # ---- synthetic calib frames for this rank (weak scaling: F frames per GPU) -------------------
    strong = args.frames_total > 0
    if strong:
        # ONE problem, generated identically on every rank (positional PRNG), of which the rank keeps its slot range
        sp = synth.make_problem(args.frames_total, args.model, seed=0xC0FFEE)
        if world > 1:
            # the library's own cut (ccal_partition_slots: contiguous slot ranges balanced by corner count - what
            # ccal_multi_problem_create does in the single-process form)
            d_all, keep_all = engine.desc_from_synth(sp)           # (keep_all: the arrays the description points into)
            first = engine.partition_slots(d_all, world)
            del d_all, keep_all
            sp = sp.slot_slice(first[rank], first[rank + 1])
        args.frames = sp.n_slots
    else:
        sp = synth.make_problem(args.frames, args.model, seed=0xC0FFEE + 1000003 * rank)
    stream = torch.cuda.Stream(device=dev)
    ctx = Context(dev_index, stream=stream.cuda_stream)
    prob = Problem.from_synth(ctx, sp)
    D = prob.block_dim(0)
    n_corners = prob.n_corners
    r_out = torch.empty(n_corners * 2, dtype=torch.float64, device=dev)
    J_out = torch.empty(prob.j_len, dtype=torch.float64, device=dev)
    prob.upload_params(sp.intr0, sp.poses0, sp.extr0)

    def step():
        prob.eval_dev(r_out.data_ptr(), J_out.data_ptr(), apply_loss=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        # clock ramp: a 53-us kernel launched a few dozen times does not bring the GPU out of its idle power state
        # (measured: 20 warm-up launches -> 56.3 us per step, 1000 -> 52.4 us); CLOCK_RAMP_S of untimed launches first,
        # whatever W is, then the W warm-up steps of the contract (disclosed in the JSON line: config.clock_ramp_s)
        t_ramp = time.perf_counter()
        ramp_launches = 0
        while time.perf_counter() - t_ramp < CLOCK_RAMP_S:
            for _ in range(50):
                step()
            ramp_launches += 50
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        barrier()
        # HIP events on the launch stream around the K launches of the timed region: average launch duration of
        # the kernel (one pair for the whole region - an event between launches would add its own gap)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
    kernel_ms = float(ev0.elapsed_time(ev1)) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(n_corners)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_corners = float(tot.item())
    else:
        total_corners = float(n_corners)

    out = None
    if rank == 0:
        # algorithmic bytes of one launch on this GPU: per corner 5 f32 in + r[2] + J[2][D] f64 out,
        # per frame one 48-B pose
        bytes_per_corner = 20 + 16 + 16 * D
        algo_bytes = n_corners * bytes_per_corner + sp.n_slots * 48
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch: MEASURED BY THIS RUN where it can be (two short child runs of this command under rocprofv3 --pmc,
        # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, after the timed region); else the committed PMC passes of the same command
        # and workload (profiles/<round>/pmc_summary.json) - accepted only while the kernel's sources still hash to what the
        # committed figure was measured on (traffic_stale_if_kernel_changed); null otherwise
        traffic, traffic_source = None, None
        kernel_hash = _eval_kernel_source_hash()
        if world == 1 and not args.no_traffic and not _under_rocprofiler() and os.environ.get("CCAL_BENCH_TRAFFIC_CHILD") != "1":
            try:
                traffic, traffic_source = _pmc_traffic_now(args)
            except Exception as e:  # noqa: BLE001
                traffic, traffic_source = None, f"in-run PMC measurement failed: {e!r}"
        traffic_note = traffic_source if traffic is None else None
        if traffic is None:
            try:
                pmc_path = _latest_profile_file("pmc_summary.json")
                with open(pmc_path) as f:
                    pmc = json.load(f)
                if pmc.get("k_eval_algorithmic_bytes_per_launch") == algo_bytes and pmc.get("k_eval_source_sha256_16") == kernel_hash:
                    traffic = pmc["k_eval_hbm_traffic_bytes_per_launch"]
                    # not measured by THIS run: the committed PMC passes of the same command, workload and kernel sources
                    traffic_source = os.path.relpath(pmc_path, ROOT) + " (rocprofv3 --pmc passes of this command, committed; kernel sources unchanged since)"
                elif pmc.get("k_eval_algorithmic_bytes_per_launch") == algo_bytes:
                    traffic_source = (os.path.relpath(pmc_path, ROOT) + " is STALE: the kernel's sources changed since it was measured (" +
                                      str(pmc.get("k_eval_source_sha256_16")) + " -> " + kernel_hash + "); traffic left null")
            except Exception:  # noqa: BLE001
                traffic = None
            if traffic_note and traffic_source:
                traffic_source += f" [in-run measurement: {traffic_note}]"
        out = {
            "metric": "corner residual+Jacobian evals/sec; LM iters/sec to converge (EUCM, TUM-VI cam0)",
            "value": total_corners * args.steps / elapsed,
            "unit": "corner residual+Jacobian evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"synthetic {args.frames_total} frames x 144 corners split over {world} GPU(s) by slot range, "
                                    if strong else f"synthetic {args.frames} frames x 144 corners per GPU, ") +
                                   f"{args.model.upper()}, mode E (r[2] + J[2x{D}] per corner materialised in HBM), 6x6 AprilGrid",
                       "frames_per_gpu": args.frames, "frames_total": args.frames_total if strong else args.frames * world,
                       "corners_per_frame": 144, "model": args.model,
                       "block_jacobian_cols": D, "sharding": "frames" if world > 1 else "none",
                       "clock_ramp_s": CLOCK_RAMP_S, "clock_ramp_untimed_launches": ramp_launches},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_stale_if_kernel_changed": kernel_hash,
                         "kernel": "k_eval", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes},
        }

    def ramp(fn, seconds=CLOCK_RAMP_S):
        """Untimed launches for `seconds` before a timed region of the secondary measurements: the same clock ramp the headline
        gets (a few dozen 40-us launches after an idle stretch of host work run at the idle power state's clocks)."""
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()

    def best_of(fn, n=3):
        best = None
        for _ in range(n):
            rep = fn()
            if best is None or rep.solve_ms < best.solve_ms:
                best = rep
        return best

    def solve_stats(p, s, method, on_device):
        """Whole-call wall time of ccal_solve (host pointers) or ccal_solve_dev (parameters resident), best of 3."""
        if on_device:
            def run():
                p.upload_params(s.intr0, s.poses0, s.extr0)
                return p.solve_dev(default_opts(method))
        else:
            def run():
                return p.solve(s.intr0, s.poses0, s.extr0, opts=default_opts(method))[3]
        rep = best_of(run)
        return {"iterations": rep.iterations, "solve_ms": rep.solve_ms, "status": rep.status, "final_cost": rep.final_cost,
                "iters_per_s": rep.iterations / (rep.solve_ms * 1e-3) if rep.solve_ms > 0 else None,
                **({"lm_spec_hits": rep.lm_spec_hits, "lm_spec_misses": rep.lm_spec_misses, "lm_rejected": rep.lm_rejected} if method == 1 else {})}

    config0_state = []          # (problem, initial poses, GPU result, GPU validation) of extra.config0 for the CPU leg behind the timed region
    # ---- secondary: fused normal equations (mode N) and solver iterations/s ----------------------
    if rank == 0 and not args.no_extra:
        extra = {}
        try:
            with torch.cuda.stream(stream):
                ramp(lambda: prob.build_normal_dev(0.0))
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                nb = 200
                a.record(stream)
                for _ in range(nb):
                    prob.build_normal_dev(0.0)
                b.record(stream)
                torch.cuda.synchronize()
            ms = a.elapsed_time(b) / nb
            extra["mode_N_build_ms"] = ms
            extra["mode_N_evals_per_s"] = n_corners / (ms * 1e-3)
            extra["secondary_clock_ramp_s"] = CLOCK_RAMP_S      # untimed launches in front of every timed build / eval loop of `extra`
            extra["mode_N_note"] = ("ccal_build_normal_dev: reduced normal equations [S | b | cost] from resident parameters "
                                   "(single camera: the register Gram kernel - k_gram2 or k_gram1v, see mode_N_roofline.kernels - "
                                   "with the per-frame elimination fused into its tail + k_reduce1)")
            # mode-N roofline (both rooflines; the FP64 one binds): exact FP64 operation counts read off the
            # kernels' ISA (tools/count_flops.py -> profiles/<round>/flops.json), time = the three launches together
            try:
                with open(_latest_profile_file("flops.json")) as f:
                    fl = json.load(f)["kernels"]
                K = prob.K
                gk = _gram_kernel_key(args.model, False, args.frames)
                per_corner = fl[gk]["per_corner"]
                schur = fl[f"k_schur1m<K={K}>"]["per_lane_whole_kernel"]["per_frame_flops_16_lanes"]
                mfma = fl[gk].get("mfma_f64_16x16x4_in_loop", 0)
                flops_corner = per_corner["flops"] + (2048.0 * 72 / 144 if mfma else 0.0)      # + issued matrix-core flops per corner
                flops = flops_corner * n_corners + schur * sp.n_slots
                K1 = K + 1
                rec = 21 + 6 * K1 + K1 * K1
                pf = 21 + 6 * K1 + 12
                hbm = n_corners * 20 + sp.n_slots * (48 + 8 * (2 * rec + pf))          # inputs + record written, read back, pose factor written
                extra["mode_N_roofline"] = {
                    "kernels": f"{gk} (+ fused elimination = the body of k_schur1m<K={K}>) + k_reduce1",
                    "flops_per_corner_gram": flops_corner, "flops_per_frame_elimination": schur,
                    "flops_per_corner": flops / n_corners,
                    "achieved_tflops": flops / (ms * 1e-3) / 1e12, "fp64_peak_tflops": FP64_PEAK_TFLOPS,
                    "frac_fp64": flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    "hbm_bytes": hbm, "achieved_GBps": hbm / (ms * 1e-3) / 1e9, "frac_hbm": hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "bound": "fp64 issue (VALU + LDS), not HBM",
                    "counted": "FP64 fma x2 + mul + add + LDS adds per corner in the Gram kernel's corner loop (inlier path) and "
                               "the elimination body (eliminate_frame, counted on k_schur1m: x 16 lanes per frame; the fused tail runs it "
                               "with the Gram kernel's lanes per frame) from the gfx950 ISA",
                }
            except Exception as e:  # noqa: BLE001
                extra["mode_N_roofline"] = {"error": repr(e)}
            for name, method in (("gn", 0), ("lm", 1)):
                st = solve_stats(prob, sp, method, False)
                extra[f"{name}_iterations"] = st["iterations"]
                extra[f"{name}_solve_ms"] = st["solve_ms"]
                extra[f"{name}_iters_per_s"] = st["iters_per_s"]
                extra[f"{name}_final_cost"] = st["final_cost"]
                extra[f"{name}_status"] = st["status"]
                extra[f"{name}_device_resident"] = solve_stats(prob, sp, method, True)
                intr = prob.download_params()[0]
                extra[f"{name}_max_rel_intrinsics_err_vs_gt"] = float(np.abs(intr[0, :4] / sp.intr_gt[0, :4] - 1).max())
            # the size of a real single-camera session (BASELINE configs[0]: TUM-VI calib-cam1 has a few hundred
            # frames): a 625-frame and a 1 000-frame (configs[1]) slice of the same synthetic set; whole ccal_solve calls
            # (host pointers) and ccal_solve_dev (parameters resident, result left on the device)
            for nf in (625, 1000):
                if args.frames < nf:
                    continue
                sub = sp.shard(0, args.frames // nf) if args.frames > nf else sp
                sprob = Problem.from_synth(ctx, sub)
                extra[f"frames{sub.n_slots}"] = {
                    "gn": solve_stats(sprob, sub, 0, False), "lm": solve_stats(sprob, sub, 1, False),
                    "gn_device_resident": solve_stats(sprob, sub, 0, True), "lm_device_resident": solve_stats(sprob, sub, 1, True)}
                sprob.close()
            # BASELINE configs[0] (the reference's own CPU-runnable case: one TUM-VI calibration session, EUCM) as a stand-in of its
            # shape - 600 ragged frames (24..144 corners, rows in random order, 0.1 px noise; the dataset itself is not here) -
            # through the steps of src/bin/camera_calibration.rs:262-299, END TO END through the C ABI with host arrays in and
            # out: problem creation (upload), pose initialisation (src/util.rs:287), calib_camera's solve (src/util.rs:384),
            # the second solve (calib_all_camera_with_extrinsics with one camera, src/util.rs:567) and validation (src/util.rs:778);
            # the CPU side (oracle: same steps from the same initial poses, all granted cores) is added after the timed region
            try:
                from camera_intrinsic_calibration_rs_amd import synth as _synth
                s0 = _synth.make_problem(600, "eucm", seed=0x7A11, ragged=True, noise_px=0.1)
                def config0_once():
                    t = [time.perf_counter()]
                    p0 = Problem.from_synth(ctx, s0); t.append(time.perf_counter())
                    poses_i, n_used = p0.init_poses(s0.intr0); t.append(time.perf_counter())
                    p0.apply_reference_bounds()
                    i_a, p_a, _, rep_a = p0.solve(s0.intr0, poses_i, s0.extr0, opts=default_opts(0)); t.append(time.perf_counter())
                    i_b, p_b, _, rep_b = p0.solve(i_a, p_a, s0.extr0, opts=default_opts(0)); t.append(time.perf_counter())
                    val = p0.validation(0, i_b, p_b, s0.extr0); t.append(time.perf_counter())
                    p0.close()
                    d = [1e3 * (t[k + 1] - t[k]) for k in range(5)]
                    return d, (poses_i, i_b, p_b, rep_a, rep_b, val)
                ramp(lambda: prob.build_normal_dev(0.0))
                runs = [config0_once() for _ in range(5)]
                best = min(runs, key=lambda r: sum(r[0]))
                d, (poses_i0, i_b0, p_b0, rep_a0, rep_b0, val0) = best
                extra["config0"] = {
                    "workload": "stand-in for BASELINE configs[0]: 600 ragged frames (24..144 corners, random row order, 0.1 px noise), EUCM, one camera",
                    "frames": int(s0.n_slots), "corners": int(s0.p3d.shape[0]),
                    "gpu_ms": {"problem_create_upload": d[0], "init_poses": d[1], "calib_camera_solve": d[2], "second_solve": d[3], "validation": d[4],
                               "total": sum(d)},
                    "gpu_solver_reports_ms": [rep_a0.solve_ms, rep_b0.solve_ms], "gpu_iterations": [rep_a0.iterations, rep_b0.iterations],
                    "gpu_validation": {"avg_99_percent": val0[0], "median": val0[1]},
                    "how": "engine.Problem (ctypes over the C ABI), host arrays in and out, best of 5 end-to-end passes; the reference's "
                           "binary does the same steps on its CPU path"}
                config0_state[:] = [(s0, poses_i0, i_b0, val0)]
            except Exception as e:  # noqa: BLE001
                extra["config0"] = {"error": repr(e)}
            # The session-size regime, side by side: N independent 625-frame problems through ONE ccal_solve_batch call
            # (a context + stream + host thread each), aggregate Gauss-Newton iterations/s against one problem at a time
            # (round 3 skipped this leg under rocprofv3: the profiler crashed twice inside launches issued from the worker threads.
            # The launchers' dynamic-LDS guard had a data race then - two threads setting hipFuncAttributeMaxDynamicSharedMemorySize
            # of one kernel around each other's launches; with the guard under a mutex (csrc/ccal_internal.hpp) twelve profiler
            # runs of the batch and sharded paths and of their stand-alone shape (tools/ubench/prof_threads.hip) went through:
            # the leg runs under the profiler again.  CCAL_BENCH_NO_CONCURRENT=1 leaves it out.)
            try:
                if os.environ.get("CCAL_BENCH_NO_CONCURRENT"):
                    raise RuntimeError("skipped (CCAL_BENCH_NO_CONCURRENT)")
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import concurrent_sessions
                extra["concurrent_sessions"] = concurrent_sessions.measure(625, args.model, 0, reps=100, counts=(1, 2, 4, 8), device=dev_index)
            except Exception as e:  # noqa: BLE001
                extra["concurrent_sessions"] = {"error": repr(e)}
            # BASELINE configs[4] shape: a two-camera rig with extrinsics (calib_all_camera_with_extrinsics, src/util.rs:567),
            # both cameras seeing every frame slot - the general loop: 13-column Gram at the composed pose (ONE launch for both
            # cameras: they share the model), per-slot expansion + elimination (k_schurq from 1 000 slots for UCM / EUCM, else
            # k_schur), K = 2 P_eff + 6
            if not args.no_rig:
                sp2 = synth.make_problem(args.frames, args.model, n_cams=2)
                p2 = Problem.from_synth(ctx, sp2)
                p2.upload_params(sp2.intr0, sp2.poses0, sp2.extr0)
                with torch.cuda.stream(stream):
                    ramp(lambda: p2.build_normal_dev(0.0))
                    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    for _ in range(100):
                        p2.build_normal_dev(0.0)
                    b.record(stream)
                    torch.cuda.synchronize()
                extra["two_cameras"] = {
                    "frames": sp2.n_slots, "corners": sp2.n_corners, "K": p2.K, "build_ms": a.elapsed_time(b) / 100,
                    "build_kernels": "register Gram kernel (GEN records, both cameras in one launch) + k_schurq (UCM / EUCM, >= 1 000 slots; "
                                     "else k_schur) + k_reduce",
                    "gn": solve_stats(p2, sp2, 0, False), "lm": solve_stats(p2, sp2, 1, False),
                    "gn_device_resident": solve_stats(p2, sp2, 0, True), "lm_device_resident": solve_stats(p2, sp2, 1, True)}
                p2.close()
            # BASELINE configs[2]: the other models of the hot path at the headline size - mode E roofline (HIP events on the
            # launch stream, same recipe as the headline), the mode-N build and a Gauss-Newton solve - so that KB4 and
            # OPENCV5 are measured by the driver's default command too (`--model kb4|opencv5` gives their full lines)
            if args.model == "eucm" and not strong:
                cfg2 = {}
                for m2 in ("kb4", "opencv5"):
                    try:
                        spm = synth.make_problem(args.frames, m2, seed=0xC0FFEE)
                        pm = Problem.from_synth(ctx, spm)
                        Dm = pm.block_dim(0)
                        pm.upload_params(spm.intr0, spm.poses0, spm.extr0)
                        Jm = torch.empty(pm.j_len, dtype=torch.float64, device=dev)
                        with torch.cuda.stream(stream):
                            ramp(lambda: pm.eval_dev(r_out.data_ptr(), Jm.data_ptr(), apply_loss=False))
                            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                            a.record(stream)
                            for _ in range(300):
                                pm.eval_dev(r_out.data_ptr(), Jm.data_ptr(), apply_loss=False)
                            b.record(stream)
                            torch.cuda.synchronize()
                            e_ms = a.elapsed_time(b) / 300
                            ramp(lambda: pm.build_normal_dev(0.0))
                            a.record(stream)
                            for _ in range(100):
                                pm.build_normal_dev(0.0)
                            b.record(stream)
                            torch.cuda.synchronize()
                            b_ms = a.elapsed_time(b) / 100
                        ab = pm.n_corners * (20 + 16 + 16 * Dm) + spm.n_slots * 48
                        cfg2[m2] = {"frames": spm.n_slots, "block_jacobian_cols": Dm,
                                    "mode_E": {"kernel_ms": e_ms, "evals_per_s": pm.n_corners / (e_ms * 1e-3), "algorithmic_bytes_per_launch": ab,
                                               "achieved_GBps": ab / (e_ms * 1e-3) / 1e9, "frac_hbm": ab / (e_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                                    "mode_N_build_ms": b_ms,
                                    "gn": solve_stats(pm, spm, 0, False), "gn_device_resident": solve_stats(pm, spm, 0, True)}
                        try:
                            with open(_latest_profile_file("flops.json")) as f:
                                flm = json.load(f)["kernels"]
                            gkm = _gram_kernel_key(m2, False, args.frames)
                            Km = pm.K
                            fl_c = flm[gkm]["per_corner"]["flops"]
                            fl_f = flm[f"k_schur1m<K={Km}>"]["per_lane_whole_kernel"]["per_frame_flops_16_lanes"]
                            tf = (fl_c * pm.n_corners + fl_f * spm.n_slots) / (b_ms * 1e-3) / 1e12
                            cfg2[m2]["mode_N_roofline"] = {"kernel": gkm, "flops_per_corner_gram": fl_c, "achieved_tflops": tf,
                                                          "frac_fp64": tf / FP64_PEAK_TFLOPS}
                        except Exception as e:  # noqa: BLE001
                            cfg2[m2]["mode_N_roofline"] = {"error": repr(e)}
                        del Jm
                        pm.close()
                    except Exception as e:  # noqa: BLE001
                        cfg2[m2] = {"error": repr(e)}
                extra["config2"] = cfg2
            # SURVEY 8(d)'s ragged variant: real frames hold 24 .. 144 corners (src/data_loader.rs:15,61-62: frames with fewer than 24
            # detections are dropped, a 6 x 6 AprilGrid has 144) - the same frame count with n ~ U{24..144} corners per frame, rows in
            # the HashMap's (random) order.  Rooflines on the bytes ACTUALLY moved / flops ACTUALLY needed for these corners.
            if args.model == "eucm" and not strong:
                for key, nfr, with_lm in (("ragged", args.frames, True), ("ragged50k", 50000, False)):
                    if key == "ragged50k" and (args.frames != 10000 or os.environ.get("CCAL_BENCH_NO_RAGGED50K")):
                        continue
                    try:
                        spr = synth.make_problem(nfr, args.model, seed=0xC0FFEE + 77, ragged=True)
                        pr_ = Problem.from_synth(ctx, spr)
                        Dr = pr_.block_dim(0)
                        pr_.upload_params(spr.intr0, spr.poses0, spr.extr0)
                        Jr = torch.empty(pr_.j_len, dtype=torch.float64, device=dev)
                        rr = torch.empty(pr_.n_corners * 2, dtype=torch.float64, device=dev)
                        with torch.cuda.stream(stream):
                            ramp(lambda: pr_.eval_dev(rr.data_ptr(), Jr.data_ptr(), apply_loss=False))
                            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                            ne_ = 300 if nfr <= 10000 else 60
                            a.record(stream)
                            for _ in range(ne_):
                                pr_.eval_dev(rr.data_ptr(), Jr.data_ptr(), apply_loss=False)
                            b.record(stream)
                            torch.cuda.synchronize()
                            e_ms = a.elapsed_time(b) / ne_
                            ramp(lambda: pr_.build_normal_dev(0.0))
                            nb_ = 100 if nfr <= 10000 else 30
                            a.record(stream)
                            for _ in range(nb_):
                                pr_.build_normal_dev(0.0)
                            b.record(stream)
                            torch.cuda.synchronize()
                            b_ms = a.elapsed_time(b) / nb_
                        ab = pr_.n_corners * (20 + 16 + 16 * Dr) + spr.n_slots * 48
                        blk = {"workload": f"{spr.n_slots} frames, n ~ U{{24..144}} corners per frame (mean {pr_.n_corners / spr.n_slots:.1f}), {args.model.upper()}, one camera",
                               "frames": spr.n_slots, "corners": int(pr_.n_corners),
                               "mode_E": {"kernel_ms": e_ms, "evals_per_s": pr_.n_corners / (e_ms * 1e-3), "algorithmic_bytes_per_launch": ab,
                                          "achieved_GBps": ab / (e_ms * 1e-3) / 1e9, "frac_hbm": ab / (e_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                               "mode_N": {"build_ms": b_ms, "evals_per_s": pr_.n_corners / (b_ms * 1e-3)}}
                        try:
                            with open(_latest_profile_file("flops.json")) as f:
                                flr = json.load(f)["kernels"]
                            gkr = _gram_kernel_key(args.model, False, nfr)
                            fl_c = flr[gkr]["per_corner"]["flops"]
                            fl_f = flr[f"k_schur1m<K={pr_.K}>"]["per_lane_whole_kernel"]["per_frame_flops_16_lanes"]
                            tf = (fl_c * pr_.n_corners + fl_f * spr.n_slots) / (b_ms * 1e-3) / 1e12
                            blk["mode_N"].update({"kernel": gkr, "flops_per_corner_gram": fl_c, "achieved_tflops": tf, "frac_fp64": tf / FP64_PEAK_TFLOPS,
                                                  "counted": "flops needed for the corners that exist (not for the padded passes of a wavefront)"})
                        except Exception as e:  # noqa: BLE001
                            blk["mode_N"]["roofline_error"] = repr(e)
                        blk["gn"] = solve_stats(pr_, spr, 0, False); blk["gn_device_resident"] = solve_stats(pr_, spr, 0, True)
                        if with_lm:
                            blk["lm"] = solve_stats(pr_, spr, 1, False); blk["lm_device_resident"] = solve_stats(pr_, spr, 1, True)
                        intr_r = pr_.download_params()[0]
                        blk["max_rel_intrinsics_err_vs_gt"] = float(np.abs(intr_r[0, :4] / spr.intr_gt[0, :4] - 1).max())
                        if key == "ragged" and not args.no_rig:
                            # the same for a two-camera rig (BASELINE configs[4] shape) of ragged observation frames: one Gram launch for both cameras,
                            # its list sorted by corner count and binned (k_gram2g)
                            sp2r = synth.make_problem(nfr, args.model, n_cams=2, ragged=True, seed=0xC0FFEE + 78)
                            p2r = Problem.from_synth(ctx, sp2r)
                            p2r.upload_params(sp2r.intr0, sp2r.poses0, sp2r.extr0)
                            with torch.cuda.stream(stream):
                                ramp(lambda: p2r.build_normal_dev(0.0))
                                a.record(stream)
                                for _ in range(100):
                                    p2r.build_normal_dev(0.0)
                                b.record(stream)
                                torch.cuda.synchronize()
                            blk["two_cameras"] = {"frames": sp2r.n_slots, "corners": sp2r.n_corners, "build_ms": a.elapsed_time(b) / 100,
                                                  "gn": solve_stats(p2r, sp2r, 0, False), "gn_device_resident": solve_stats(p2r, sp2r, 0, True)}
                            p2r.close()
                        del Jr, rr
                        pr_.close()
                        extra[key] = blk
                    except Exception as e:  # noqa: BLE001
                        extra[key] = {"error": repr(e)}
            # ONE process, several GPUs (ccal_multi_*: the reference's single-process shape of a multi-GPU solve; SURVEY 8(b)'s
            # `ccal_create(device_ids, n_dev)`): the headline problem cut by the library over EVERY visible GPU - range(device_count);
            # on a 1-GPU box two shards of the one GPU, where sharding can only cost - with both transports side by side (in-process:
            # HIP events + the deciding kernel adds the ranks' sums itself; RCCL: one ncclAllReduce per step on communicators made by
            # ncclCommInitAll), the slot ranges, what RCCL itself counts as ranks, and mode E through ccal_multi_eval_dev
            if world == 1:
                try:
                    from camera_intrinsic_calibration_rs_amd import _ffi as _f
                    from camera_intrinsic_calibration_rs_amd.engine import MultiContext, MultiProblem
                    devs_all = list(range(n_visible)) if n_visible > 1 else [dev_index, dev_index]
                    legs = [("in_process", _f.TRANSPORT_INPROC, devs_all),
                            ("rccl", _f.TRANSPORT_RCCL, devs_all if n_visible > 1 else [dev_index])]      # RCCL takes every device once: one rank on a 1-GPU box
                    sps = {"devices": devs_all, "devices_visible": n_visible}
                    i_ref = {m: prob.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(m))[0] for m in (0, 1)}
                    for leg, tr, devs in legs:
                        try:
                            mc = MultiContext(devs, transport=tr)
                            mpb = MultiProblem.from_synth(mc, sp)
                            row = {"devices": devs, "shards": [mpb.slot_range(i) for i in range(mpb.n_shards)],
                                   "transport": {0: "none (one device)", 1: "rccl (ncclCommInitAll, one ncclAllReduce per step issued by the library)",
                                                 2: "in-process (HIP events; k_head / k_solve add the ranks' sums in rank order)"}[mc.transport],
                                   "rccl_ranks": mc.rccl_ranks}
                            for name, method in (("gn", 0), ("lm", 1)):
                                best = None
                                for _ in range(3):
                                    i_m, p_m, _, rep_m = mpb.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
                                    if best is None or rep_m.solve_ms < best[0].solve_ms:
                                        best = (rep_m, i_m)
                                row[name] = {"iterations": best[0].iterations, "solve_ms": best[0].solve_ms, "status": best[0].status,
                                             "unsharded_solve_ms": extra.get(f"{name}_solve_ms"),
                                             "max_rel_intrinsics_diff_vs_unsharded": float(np.abs(best[1][0, :6] / i_ref[method][0, :6] - 1).max())}
                            # what the step's one collective costs on this transport: (sharded - unsharded) solve time over the groups of
                            # the solve (GN: iterations + 1 groups, one all-reduce each); on [0,0] the two shards also halve each other's chip
                            try:
                                row["allreduce_us_per_step"] = 1e3 * (row["gn"]["solve_ms"] - extra["gn_solve_ms"]) / (row["gn"]["iterations"] + 1)
                            except Exception:  # noqa: BLE001
                                row["allreduce_us_per_step"] = None
                            sps[leg] = row
                            mpb.close(); mc.close()
                        except Exception as e:  # noqa: BLE001
                            sps[leg] = {"error": repr(e)}
                    extra["single_process_sharded"] = sps
                except Exception as e:  # noqa: BLE001
                    extra["single_process_sharded"] = {"error": repr(e)}
                try:
                    # mode E through ccal_multi_eval_dev: one process, every visible GPU, no collective - each shard's launches on its
                    # own context's stream; wall clock around K passes over the whole problem (the per-GPU roofline is the headline's)
                    from camera_intrinsic_calibration_rs_amd.engine import MultiContext, MultiProblem
                    devs = list(range(n_visible)) if n_visible > 1 else [dev_index, dev_index]
                    mc = MultiContext(devs)
                    mpb = MultiProblem.from_synth(mc, sp)
                    mpb.upload_params(sp.intr0, sp.poses0, sp.extr0)
                    bufs = []
                    for i, d in enumerate(devs):
                        nc_i, jl_i = mpb.shard_sizes(i)
                        bufs.append((torch.empty(max(nc_i * 2, 1), dtype=torch.float64, device=f"cuda:{d}"),
                                     torch.empty(max(jl_i, 1), dtype=torch.float64, device=f"cuda:{d}")))
                    rp = [b[0].data_ptr() for b in bufs]; jp = [b[1].data_ptr() for b in bufs]
                    t_r = time.perf_counter()
                    while time.perf_counter() - t_r < CLOCK_RAMP_S:
                        for _ in range(20):
                            mpb.eval_dev(rp, jp)
                        mc.sync()
                    ne = 200
                    t_e = time.perf_counter()
                    for _ in range(ne):
                        mpb.eval_dev(rp, jp)
                    mc.sync()
                    dt = (time.perf_counter() - t_e) / ne
                    ab = n_corners * (20 + 16 + 16 * D) + sp.n_slots * 48
                    extra["multi_eval"] = {"devices": devs, "shards": [mpb.slot_range(i) for i in range(mpb.n_shards)],
                                           "ms_per_pass": dt * 1e3, "evals_per_s": n_corners / dt, "achieved_GBps_all_devices": ab / dt / 1e9,
                                           "frac_hbm_per_device": ab / dt / 1e9 / HBM_PEAK_GBPS / len(set(devs)),
                                           "how": "ccal_multi_eval_dev, wall clock around 200 passes (host enqueue of n launches per pass included)"}
                    del bufs
                    mpb.close(); mc.close()
                except Exception as e:  # noqa: BLE001
                    extra["multi_eval"] = {"error": repr(e)}
        except Exception as e:  # noqa: BLE001
            extra["error"] = repr(e)
        out["extra"] = extra

    # ---- multi-GPU only: frame-sharded GN / LM, ONE ncclAllReduce of the packed reduced system per step, issued by the
    # library itself on its own RCCL communicator.  Every rank solves its shard of a (frames x world)-frame problem.
    # A collective that never completes must not take the headline line down with it, and must not look like success:
    # a watchdog prints the line (rank 0) and exits NON-ZERO on every rank that hangs.
    force_sharded = os.environ.get("CCAL_BENCH_FORCE_SHARDED") == "1"     # developer switch: the sharded leg on a 1-rank communicator
    if (world > 1 or force_sharded) and not args.no_extra:
        import threading
        result = {}
        # one shared camera: every rank starts from rank 0's initial intrinsics (each rank's frames are its own)
        intr_shared = torch.from_numpy(np.ascontiguousarray(sp.intr0)).to(dev)
        if world > 1:
            dist.broadcast(intr_shared, src=0)
        start = synth.dataclasses.replace(sp, intr0=intr_shared.cpu().numpy())
        native = backend == "nccl" and engine.rccl_available()
        comm = None
        if native:
            # the communicator id travels through the launcher's process group; the collective itself never touches Python
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt = torch.tensor(list(engine.rccl_unique_id()), dtype=torch.uint8, device=dev)
            if world > 1:
                dist.broadcast(idt, src=0)
            uid = bytes(idt.cpu().tolist())

        # BASELINE configs[3] as written - 50 000 frames x 144 corners SPLIT over the ranks - next to the weak-scaling
        # solves of a default multi-GPU run (with --frames-total the main problem already is the split one).  Generated
        # here, outside the watchdog: pure host work, identical on every rank, of which the rank keeps its slot range
        split, split_sp, result3 = None, None, {}
        if not strong and (world > 1 or os.environ.get("CCAL_BENCH_CONFIG3") == "1"):
            split_total = int(os.environ.get("CCAL_BENCH_CONFIG3_FRAMES", "50000"))
            split_sp = synth.make_problem(split_total, args.model, seed=0xC0FFEE)
            if world > 1:
                d3, keep3 = engine.desc_from_synth(split_sp)
                first3 = engine.partition_slots(d3, world)
                del d3, keep3
                split_sp = split_sp.slot_slice(first3[rank], first3[rank + 1])
            split = Problem.from_synth(ctx, split_sp)

        def attach(pr):
            if native:
                pr.set_rccl_comm(comm)
            else:
                from camera_intrinsic_calibration_rs_amd.dist import make_allreduce_hook
                pr.set_allreduce(make_allreduce_hook(device=dev))

        def sharded_solves(pr, st_sp, res, total):
            with torch.cuda.stream(stream):
                for name, method in (("gn", 0), ("lm", 1)):
                    res[name] = solve_stats(pr, st_sp, method, False)
                    res[name + "_device_resident"] = solve_stats(pr, st_sp, method, True)
            res["frames_total"] = total
            res["frames_this_rank"] = st_sp.n_slots
            res["collective"] = ("ncclAllReduce issued by libccal_hip.so (ccal_set_rccl_comm), one per optimizer step"
                                 if native else "callback (torch.distributed, developer switch)")
            res["rccl_version"] = engine._ffi.load().ccal_rccl_version() if native else None
            res["rccl_ranks"] = engine.rccl_comm_count(comm) if (native and comm) else None      # what RCCL itself counts (ncclCommCount)

        def sharded():
            nonlocal comm
            try:
                if native:
                    comm = ctx.rccl_comm_create(world, rank, uid)
                attach(prob)
                sharded_solves(prob, start, result, args.frames_total if strong else args.frames * world)
            except Exception as e:  # noqa: BLE001
                result["error"] = repr(e)
            finally:
                prob.set_rccl_comm(None)            # drains the early-exit groups (and their collectives) still queued
                prob.set_allreduce(None)
            if split is not None and "error" not in result:
                try:
                    attach(split)
                    sharded_solves(split, split_sp, result3, split_total)
                except Exception as e:  # noqa: BLE001
                    result3["error"] = repr(e)
                finally:
                    split.set_rccl_comm(None)
                    split.set_allreduce(None)

        th = threading.Thread(target=sharded, daemon=True)
        th.start(); th.join(timeout=180.0)
        if th.is_alive():
            result = {"error": "timeout (180 s) in the sharded solve"}
        if rank == 0:
            try:
                # weak scaling: every rank holds what the unsharded solve of `extra` held - per GROUP (iterations + 1; one collective
                # each), sharded minus unsharded = what the step's all-reduce costs on this node
                if not strong and "gn_device_resident" in result and "gn_device_resident" in out.get("extra", {}):
                    sg, ug = result["gn_device_resident"], out["extra"]["gn_device_resident"]
                    result["allreduce_us_per_step"] = 1e3 * (sg["solve_ms"] / (sg["iterations"] + 1) - ug["solve_ms"] / (ug["iterations"] + 1))
                    out["allreduce_us_per_step"] = result["allreduce_us_per_step"]
            except Exception:  # noqa: BLE001
                pass
            out.setdefault("extra", {})["sharded_solve"] = result
            out["rccl_ranks"] = result.get("rccl_ranks")          # did RCCL see all N ranks: ncclCommCount of the communicator the solves used
            if split is not None:
                out["extra"]["config3_split"] = result3 or {"error": "not reached"}
        if th.is_alive():
            # the headline (mode E, no collective) was measured before this leg: the line goes out with the failure of the
            # SECONDARY leg spelled out in it (extra.sharded_solve.error, top-level "secondary_leg_failed") and on stderr.  The
            # process cannot be unwound (a thread sits inside a collective with no partner): it ends here, without a re-exec,
            # with a NON-ZERO exit code on every rank that hangs - a hung collective is never reported as success.  Rank 0's line
            # (the headline is valid) is on stdout first; bench.py's own launcher relays it and passes the exit code on.
            if rank == 0:
                out["secondary_leg_failed"] = "sharded_solve: timeout (180 s) inside the frame-sharded solve"
                _emit(out)
            print("[bench] sharded solve timed out after 180 s on rank %d" % rank, file=sys.stderr, flush=True)
            os._exit(3)
        if comm:
            engine.rccl_comm_destroy(comm)

    # ---- CPU baseline: the oracle (restatement of the reference's per-corner dual-number path) ----
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import binding as ob
        march = ob.use_native_build()                   # -O3 -march=native for THIS host (falls back to the portable build)
        sample_frames = min(args.frames, 2048)
        sub = sp.shard(0, max(1, args.frames // sample_frames)) if args.frames > sample_frames else sp
        op = ob.OracleProblem.from_synth(sub)
        hw = ob.hardware_threads()
        usable = ob.usable_cpus()                        # affinity mask: what a container / cpuset really grants
        cpu_max = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            cpu_max = None if q == "max" else float(q) / float(per)
        except Exception:  # noqa: BLE001
            pass
        cores = max(1, min(usable, int(cpu_max) if cpu_max else usable))
        t_single = op.eval_timed(sub.intr0, sub.poses0, threads=1, reps=2)
        single = 2 * op.n_corners / t_single
        # size the all-core run for ~10 s of wall time, every thread repeating its share of the sample
        op.eval_timed(sub.intr0, sub.poses0, threads=cores, reps=8)                # warm-up
        t_cal = op.eval_timed(sub.intr0, sub.poses0, threads=cores, reps=32)       # calibration
        reps = int(min(max(8, 10.0 / max(t_cal / 32, 1e-6)), 1e6))
        t_all = op.eval_timed(sub.intr0, sub.poses0, threads=cores, reps=reps)
        multi = op.n_corners * reps / t_all
        # the same per-corner evaluation with HEAP-backed dual numbers (the container tiny-solver instantiates factors
        # with): ~3 s single thread + ~5 s on all cores
        small = sub.shard(0, max(1, sub.n_slots // 256)) if sub.n_slots > 256 else sub
        oph = ob.OracleProblem.from_synth(small)
        th1 = oph.eval_timed(small.intr0, small.poses0, threads=1, reps=1, heap_duals=True)
        heap_single = oph.n_corners / th1
        reps_h = int(min(max(2, 5.0 / max(th1 / max(cores, 1), 1e-6)), 1e5))
        th_all = oph.eval_timed(small.intr0, small.poses0, threads=cores, reps=reps_h, heap_duals=True)
        heap_multi = oph.n_corners * reps_h / th_all
        out["cpu_baseline"] = {
            "value": multi, "unit": "corner residual+Jacobian evals/s", "cores": cores,
            "kind": "port",
            "sample": f"{sub.n_slots} frames x 144 corners of the same workload evaluated {reps} times "
                      f"({t_all:.1f} s wall) by the oracle's per-corner Dual<{D}> path on {cores} threads; "
                      f"single thread {single:.3e}/s",
            "single_thread_value": single,
            "scaling_efficiency": multi / (single * cores),
            "host": {"hardware_threads": hw, "usable_cpus_affinity": usable, "cgroup_cpu_max": cpu_max, "threads_used": cores,
                     "oracle_build": f"-O3 -march={march}"},
            "port_heap": {"kind": "port-heap", "value": heap_multi, "single_thread_value": heap_single, "cores": cores,
                          "sample": f"{small.n_slots} frames x 144 corners, {reps_h} repetitions ({th_all:.1f} s wall), heap-backed "
                                    f"dual numbers (std::vector tangent per arithmetic result, like num-dual's DualDVec64)"},
            "note": "the reference itself cannot be built here (Rust); `value` is the C++ restatement with stack duals the "
                    "compiler vectorises (optimistic for the reference), port_heap the same arithmetic with the heap-backed "
                    "container tiny-solver really uses (faithful): the reference's CPU rate lies between the two",
        }
        out["gpu_over_cpu"] = out["value"] / world / multi
        out["gpu_over_cpu_heap"] = out["value"] / world / heap_multi
        if not args.no_extra and "extra" in out:
            # M2 baseline, like for like: the oracle's Gauss-Newton - the reference's algorithm on the host - on the SAME
            # 625- and 1 000-frame problems the GPU solves in extra.frames625 / frames1000, one thread and all granted cores
            try:
                cg = {}
                for nf in (625, 1000):
                    if args.frames < nf:
                        continue
                    sm = sp.shard(0, args.frames // nf) if args.frames > nf else sp
                    ops = ob.OracleProblem.from_synth(sm)
                    row = {"frames": sm.n_slots}
                    for label, nt in (("threads_1", 1), ("all_cores", cores)):
                        ob.set_solve_threads(nt)
                        try:
                            _, _, _, orep = ops.solve(sm.intr0, sm.poses0, sm.extr0, opts=default_opts(0))
                        finally:
                            ob.set_solve_threads(1)
                        row[label] = {"threads": nt, "iterations": orep.iterations, "solve_ms": orep.solve_ms,
                                      "iters_per_s": orep.iterations / (orep.solve_ms * 1e-3), "final_cost": orep.final_cost}
                    g = out["extra"].get(f"frames{sm.n_slots}", {}).get("gn")
                    if g:
                        row["gpu_solve_ms"] = g["solve_ms"]
                        row["gpu_over_cpu_all_cores"] = row["all_cores"]["solve_ms"] / g["solve_ms"]
                        row["gpu_over_cpu_one_thread"] = row["threads_1"]["solve_ms"] / g["solve_ms"]
                    cg[f"frames{sm.n_slots}"] = row
                cg["note"] = ("oracle = the C++ restatement of tiny-solver's Gauss-Newton with dual-number Jacobians, dense Schur on the host; "
                              "threads split the frame slots, partial normal equations added in thread order")
                out["extra"]["cpu_oracle_gn"] = cg
            except Exception as e:  # noqa: BLE001
                out["extra"]["cpu_oracle_gn"] = {"error": repr(e)}
        # configs[0] stand-in on the host: the oracle through the same steps from the same initial poses, all granted cores
        st0 = config0_state[0] if config0_state else None
        if st0 is not None and isinstance(out.get("extra"), dict) and "gpu_ms" in out["extra"].get("config0", {}):
            try:
                s0, poses_i0, i_b0, val0 = st0
                op0 = ob.OracleProblem.from_synth(s0)
                op0.apply_reference_bounds()
                ob.set_solve_threads(cores)
                try:
                    t0c = time.perf_counter()
                    i_oa, p_oa, _, r_oa = op0.solve(s0.intr0, poses_i0, s0.extr0, opts=default_opts(0)); t1c = time.perf_counter()
                    i_ob, p_ob, _, r_ob = op0.solve(i_oa, p_oa, s0.extr0, opts=default_opts(0)); t2c = time.perf_counter()
                    v_o = op0.validation(0, i_ob, p_ob, s0.extr0); t3c = time.perf_counter()
                finally:
                    ob.set_solve_threads(1)
                c0 = out["extra"]["config0"]
                c0["cpu_oracle_ms"] = {"threads": cores, "calib_camera_solve": 1e3 * (t1c - t0c), "second_solve": 1e3 * (t2c - t1c),
                                       "validation": 1e3 * (t3c - t2c), "total_without_init": 1e3 * (t3c - t0c),
                                       "note": "pose initialisation is not restated on the host: the oracle starts from the GPU's initial poses"}
                g = c0["gpu_ms"]
                c0["gpu_over_cpu_solves_and_validation"] = (1e3 * (t3c - t0c)) / (g["calib_camera_solve"] + g["second_solve"] + g["validation"])
                c0["parity"] = {"iterations_oracle": [r_oa.iterations, r_ob.iterations],
                                "max_rel_dintrinsics": float(np.abs(i_b0[0, :6] / i_ob[0, :6] - 1).max()),
                                "d_avg_99_percent_px": abs(val0[0] - v_o[0]), "d_median_px": abs(val0[1] - v_o[1])}
            except Exception as e:  # noqa: BLE001
                out["extra"]["config0"]["cpu_oracle_ms"] = {"error": repr(e)}
        # ---- parity statement: HIP path against the oracle on a 200-frame sample of the SAME workload, after the
        # timed region - residuals, Jacobians, normal equations, converged intrinsics (GN and LM), and the reference's own quality
        # metric (median / mean of the lowest 99 % of the reprojection errors, src/util.rs:778-795) at each side's optimum
        try:
            ob.set_solve_threads(1)
            sm = sp.shard(0, max(1, args.frames // 200)) if args.frames > 200 else sp
            gp = Problem.from_synth(ctx, sm)
            opp = ob.OracleProblem.from_synth(sm)
            r_g, J_g = gp.eval(sm.intr0, sm.poses0, sm.extr0)
            r_o, J_o = opp.eval(sm.intr0, sm.poses0, sm.extr0)
            S_g, b_g, c_g = gp.build_normal(sm.intr0, sm.poses0, sm.extr0)
            S_o, b_o, c_o = opp.build_normal(sm.intr0, sm.poses0, sm.extr0)
            par = {"sample": f"{sm.n_slots} frames x 144 corners of the benchmark's workload ({args.model.upper()})",
                   "max_abs_dr_px": float(np.abs(r_g - r_o).max()),
                   "max_rel_dJ": float((np.abs(J_g - J_o) / np.maximum(1.0, np.abs(J_o))).max()),
                   "rel_dS": float(np.abs(S_g - S_o).max() / np.abs(S_o).max()), "rel_db": float(np.abs(b_g - b_o).max() / np.abs(b_o).max()),
                   "rel_dcost": float(abs(c_g - c_o) / abs(c_o)),
                   "tolerances": {"dr_px": 1e-10, "rel_dJ": 1e-11, "rel_dS": 1e-9, "intrinsics_rel": 1e-6, "validation_px": 1e-9}}
            P = len(synth.GT_PARAMS[synth.MODEL_NAMES[args.model]])
            for name, method in (("gn", 0), ("lm", 1)):
                ig, pg, eg, rg = gp.solve(sm.intr0, sm.poses0, sm.extr0, opts=default_opts(method))
                io_, po_, eo_, ro_ = opp.solve(sm.intr0, sm.poses0, sm.extr0, opts=default_opts(method))
                ag, mg = gp.validation(0, ig, pg, eg)
                ao, mo = opp.validation(0, io_, po_, eo_)
                par[name] = {"iterations_gpu": rg.iterations, "iterations_oracle": ro_.iterations,
                             "status_gpu": rg.status, "status_oracle": ro_.status,
                             "max_rel_dintrinsics": float(np.abs(ig[0, :P] / io_[0, :P] - 1).max()),
                             "max_abs_dposes": float(np.abs(pg - po_).max()),
                             "rel_dfinal_cost": float(abs(rg.final_cost - ro_.final_cost) / abs(ro_.final_cost)),
                             "max_rel_intrinsics_err_vs_gt": float(np.abs(ig[0, :4] / sm.intr_gt[0, :4] - 1).max()),
                             "validation_gpu": {"avg_99_percent": ag, "median": mg}, "validation_oracle": {"avg_99_percent": ao, "median": mo},
                             "d_avg_99_percent_px": abs(ag - ao), "d_median_px": abs(mg - mo)}
            par["pass"] = bool(par["max_abs_dr_px"] <= 1e-10 and par["max_rel_dJ"] <= 1e-11 and par["rel_dS"] <= 1e-9 and
                               all(par[k]["max_rel_dintrinsics"] <= 1e-6 and par[k]["iterations_gpu"] == par[k]["iterations_oracle"] and
                                   par[k]["d_median_px"] <= 1e-9 and par[k]["d_avg_99_percent_px"] <= 1e-9 for k in ("gn", "lm")))
            par["oracle"] = "oracle/ (CPU restatement of the reference's algorithm; parity unpinned against the absent Rust crates, DESIGN.md 2)"
            out["parity"] = par
            gp.close()
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": repr(e)}

    if rank == 0:
        _emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
