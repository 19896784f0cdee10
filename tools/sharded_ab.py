#!/usr/bin/env python3
"""One process, every GPU: the headline problem (10 000 frames x 144 corners per GPU, EUCM) cut by the library over n devices
(range(n); n = 1: two shards of the one GPU for the in-process transport, one rank for RCCL), Gauss-Newton and LM, with the
transport CCAL_MULTI_TRANSPORT names (inproc | rccl; unset: the library's automatic choice) - one JSON line.  The A/B of
tools/multi_gpu_day.sh: the same solves over both transports give the cost of the step's one collective on this node
(allreduce_us_per_step = (sharded - unsharded) / groups against ONE device holding a shard's share of the frames).
    python tools/sharded_ab.py [n_gpus] [frames_per_gpu]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    from camera_intrinsic_calibration_rs_amd import _ffi, synth
    from camera_intrinsic_calibration_rs_amd.engine import Context, MultiContext, MultiProblem, Problem, default_opts
    nvis = torch.cuda.device_count()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else nvis
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    if n < 1 or n > nvis:
        print(json.dumps({"error": f"{n} GPU(s) asked for, {nvis} visible"})); sys.exit(4)
    env = os.environ.get("CCAL_MULTI_TRANSPORT", "")
    if n > 1:
        devs = list(range(n))
    else:
        devs = [0] if env.startswith("r") else [0, 0]
    total = per * max(n, 1)
    sp = synth.make_problem(total, "eucm", seed=0xC0FFEE)
    out = {"devices": devs, "frames_total": total, "transport_env": env or None}
    try:
        mc = MultiContext(devs)
        mpb = MultiProblem.from_synth(mc, sp)
        out["transport"] = {0: "none", 1: "rccl", 2: "in-process"}[mc.transport]
        out["rccl_ranks"] = mc.rccl_ranks
        out["shards"] = [mpb.slot_range(i) for i in range(mpb.n_shards)]
        # the yardstick: ONE device solving one shard's share on its own (no collective)
        first0, n0 = mpb.slot_range(0)
        one = sp.slot_slice(first0, first0 + n0) if mpb.n_shards > 1 else sp
        solo = Problem.from_synth(Context(devs[0]), one)
        for name, method in (("gn", 0), ("lm", 1)):
            best = None
            for _ in range(5):
                i_m, p_m, _, rep = mpb.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method))
                if best is None or rep.solve_ms < best.solve_ms:
                    best = rep
            bs = None
            for _ in range(5):
                rs = solo.solve(one.intr0, one.poses0, one.extr0, opts=default_opts(method))[3]
                if bs is None or rs.solve_ms < bs.solve_ms:
                    bs = rs
            groups = best.iterations + 1 + (best.lm_spec_misses + best.lm_rejected if method else 0)
            out[name] = {"iterations": best.iterations, "status": best.status, "solve_ms": best.solve_ms,
                         "one_shard_alone_ms": bs.solve_ms, "one_shard_alone_iterations": bs.iterations,
                         "allreduce_us_per_step": 1e3 * (best.solve_ms / max(groups, 1) - bs.solve_ms / max(bs.iterations + 1 + (bs.lm_spec_misses + bs.lm_rejected if method else 0), 1)),
                         "max_rel_intrinsics_err_vs_gt": float(np.abs(i_m[0, :4] / sp.intr_gt[0, :4] - 1).max())}
        solo.close(); mpb.close(); mc.close()
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
