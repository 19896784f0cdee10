#!/usr/bin/env python3
"""Headline benchmark: corner residual+Jacobian evals/s (mode E) on synthetic calib frames.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--model eucm]

One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE).  A "step" is one pass
of the hot path over one batch: every rank evaluates r[2] + J[2 x D] for all corners of its frame
shard (F frames x 144 corners per GPU -- weak scaling; mode E needs no collective).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

The workload defaults to the north-star headline (10 000 frames x 144 corners, EUCM, per GPU);
`--frames 1000` is BASELINE.json configs[1].  PyTorch is used only for the process group, the
barrier and HIP events; the kernels are the hand-written HIP library behind include/ccal.h.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU (144 corners each)")
    ap.add_argument("--model", default="eucm", choices=["ucm", "eucm", "kb4", "opencv5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (mode N / solver) measurements")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # CCAL_BENCH_BACKEND=gloo (developer switch): exercise the multi-rank code path on a box with fewer GPUs than
    # ranks - ranks then share devices; the measured numbers mean nothing there
    backend = os.environ.get("CCAL_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from camera_intrinsic_calibration_rs_amd import synth
    from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts

    # ---- synthetic calib frames for this rank (weak scaling: F frames per GPU) -------------------
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
        # (measured: 20 warm-up launches -> 56.3 us per step, 1000 -> 52.4 us); 0.1 s of untimed launches first, whatever
        # W is, then the W warm-up steps of the contract
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.1:
            for _ in range(50):
                step()
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
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command/workload
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; profiles/<round>/pmc_summary.json); null otherwise
        traffic = None
        try:
            import re
            prof_dirs = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+", d))
            with open(os.path.join(ROOT, "profiles", prof_dirs[-1], "pmc_summary.json")) as f:
                pmc = json.load(f)
            if pmc.get("k_eval_algorithmic_bytes_per_launch") == algo_bytes:
                traffic = pmc["k_eval_hbm_traffic_bytes_per_launch"]
        except Exception:  # noqa: BLE001
            traffic = None
        out = {
            "metric": "corner residual+Jacobian evals/sec; LM iters/sec to converge (EUCM, TUM-VI cam0)",
            "value": total_corners * args.steps / elapsed,
            "unit": "corner residual+Jacobian evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.frames} frames x 144 corners per GPU, {args.model.upper()}, "
                                   f"mode E (r[2] + J[2x{D}] per corner materialised in HBM), 6x6 AprilGrid",
                       "frames_per_gpu": args.frames, "corners_per_frame": 144, "model": args.model,
                       "block_jacobian_cols": D, "sharding": "frames" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_eval", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes},
        }

    # ---- secondary: fused normal equations (mode N) and solver iterations/s ----------------------
    if rank == 0 and not args.no_extra:
        extra = {}
        try:
            with torch.cuda.stream(stream):
                for _ in range(3):
                    prob.build_normal_dev(0.0)
                torch.cuda.synchronize()
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                nb = 50
                a.record(stream)
                for _ in range(nb):
                    prob.build_normal_dev(0.0)
                b.record(stream)
                torch.cuda.synchronize()
            ms = a.elapsed_time(b) / nb
            extra["mode_N_build_ms"] = ms
            extra["mode_N_evals_per_s"] = n_corners / (ms * 1e-3)
            extra["mode_N_note"] = ("ccal_build_normal_dev: reduced normal equations [S | b | cost] from resident parameters "
                                   "(single camera: k_gram1w + k_schur1m + k_reduce1)")
            for name, method in (("gn", 0), ("lm", 1)):
                best = None
                for _ in range(3):                      # wall time of the whole ccal_solve call, best of 3
                    intr, poses, _, rep = prob.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
                    if best is None or rep.solve_ms < best.solve_ms:
                        best = rep
                rep = best
                extra[f"{name}_iterations"] = rep.iterations
                extra[f"{name}_solve_ms"] = rep.solve_ms
                extra[f"{name}_iters_per_s"] = rep.iterations / (rep.solve_ms * 1e-3) if rep.solve_ms > 0 else None
                extra[f"{name}_final_cost"] = rep.final_cost
                extra[f"{name}_status"] = rep.status
                extra[f"{name}_max_rel_intrinsics_err_vs_gt"] = float(
                    np.abs(intr[0, :4] / sp.intr_gt[0, :4] - 1).max())
            # the size of a real single-camera session (BASELINE configs[0]: TUM-VI calib-cam1 has a few hundred
            # frames): a 600-frame and a 1 000-frame (configs[1]) slice of the same synthetic set, whole ccal_solve calls
            for nf in (600, 1000):
                if args.frames < nf:
                    continue
                sub = sp.shard(0, args.frames // nf) if args.frames > nf else sp
                sprob = Problem.from_synth(ctx, sub)
                for name, method in (("gn", 0), ("lm", 1)):
                    best = None
                    for _ in range(3):
                        _, _, _, rep = sprob.solve(sub.intr0, sub.poses0, sub.extr0, opts=default_opts(method))
                        if best is None or rep.solve_ms < best.solve_ms:
                            best = rep
                    extra[f"frames{sub.n_slots}_{name}"] = {"iterations": best.iterations, "solve_ms": best.solve_ms,
                                                           "iters_per_s": best.iterations / (best.solve_ms * 1e-3),
                                                           "status": best.status}
                sprob.close()
        except Exception as e:  # noqa: BLE001
            extra["error"] = repr(e)
        out["extra"] = extra

    # ---- multi-GPU only: frame-sharded Gauss-Newton with the RCCL all-reduce of the reduced system ----
    # every rank solves its shard of a (frames x world)-frame problem; guarded by a watchdog so that a
    # collective that never completes cannot take the headline line down with it
    if world > 1 and not args.no_extra:
        import threading
        from camera_intrinsic_calibration_rs_amd.dist import make_allreduce_hook
        result = {}

        # one shared camera: every rank starts from rank 0's initial intrinsics (each rank's frames are its own)
        intr_shared = torch.from_numpy(np.ascontiguousarray(sp.intr0)).to(dev)
        dist.broadcast(intr_shared, src=0)
        intr_start = intr_shared.cpu().numpy()

        def sharded():
            try:
                prob.set_allreduce(make_allreduce_hook(device=dev))
                with torch.cuda.stream(stream):
                    for name, method in (("gn", 0), ("lm", 1)):
                        best = None
                        for _ in range(3):
                            i2, p2, _, rep = prob.solve(intr_start, sp.poses0, sp.extr0, opts=default_opts(method))
                            if best is None or rep.solve_ms < best.solve_ms:
                                best = rep
                        result[name] = dict(iterations=best.iterations, solve_ms=best.solve_ms, status=best.status,
                                            final_cost=best.final_cost,
                                            iters_per_s=best.iterations / (best.solve_ms * 1e-3))
                result["frames_total"] = args.frames * world
            except Exception as e:  # noqa: BLE001
                result["error"] = repr(e)
            finally:
                prob.set_allreduce(None)

        th = threading.Thread(target=sharded, daemon=True)
        th.start(); th.join(timeout=120.0)
        if th.is_alive():
            result = {"error": "timeout (120 s) in the sharded solve"}
        if rank == 0:
            out.setdefault("extra", {})["sharded_solve"] = result
        if th.is_alive():
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)

    # ---- CPU baseline: the oracle (restatement of the reference's per-corner dual-number path) ----
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import binding as ob
        sample_frames = min(args.frames, 2048)
        sub = sp.shard(0, max(1, args.frames // sample_frames)) if args.frames > sample_frames else sp
        op = ob.OracleProblem.from_synth(sub)
        cores = ob.hardware_threads()
        t_single = op.eval_timed(sub.intr0, sub.poses0, threads=1, reps=1)
        single = op.n_corners / t_single
        # size the all-core run for ~10 s of wall time, every thread repeating its share of the sample
        op.eval_timed(sub.intr0, sub.poses0, threads=cores, reps=8)                # warm-up
        t_cal = op.eval_timed(sub.intr0, sub.poses0, threads=cores, reps=64)       # calibration (~0.5 s)
        reps = int(min(max(8, 10.0 / max(t_cal / 64, 1e-6)), 1e6))
        t_all = op.eval_timed(sub.intr0, sub.poses0, threads=cores, reps=reps)
        out["cpu_baseline"] = {
            "value": op.n_corners * reps / t_all, "unit": "corner residual+Jacobian evals/s", "cores": cores,
            "kind": "port",
            "sample": f"{sub.n_slots} frames x 144 corners of the same workload evaluated {reps} times "
                      f"({t_all:.1f} s wall) by the oracle's per-corner Dual<{D}> path on {cores} threads; "
                      f"single thread {single:.3e}/s",
            "single_thread_value": single,
            "note": "C++ stack-dual restatement of the Rust path (the reference itself cannot be built here); faster "
                    "than tiny-solver's heap-backed duals, so GPU/CPU ratios are conservative",
        }
        out["gpu_over_cpu"] = out["value"] / world / out["cpu_baseline"]["value"]
        if not args.no_extra and "extra" in out:
            # the oracle's Gauss-Newton (reference algorithm, one thread) on a small sample, for the iterations/s line
            try:
                small = sp.shard(0, max(1, args.frames // 200)) if args.frames > 200 else sp
                ops = ob.OracleProblem.from_synth(small)
                _, _, _, orep = ops.solve(small.intr0, small.poses0, small.extr0, opts=default_opts(0))
                out["extra"]["cpu_oracle_gn"] = {"frames": small.n_slots, "iterations": orep.iterations,
                                                 "solve_ms": orep.solve_ms, "threads": 1,
                                                 "iters_per_s": orep.iterations / (orep.solve_ms * 1e-3)}
            except Exception as e:  # noqa: BLE001
                out["extra"]["cpu_oracle_gn"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
