#!/usr/bin/env python3
"""Developer tool: where does mode E lose its rate past ~1.35 GB of output?  Hypothesis: the footprint that is
re-written from launch to launch (address translation reach), not anything inside one launch.
  A  10 000-frame launches (329 MB each) cycling over NB output buffer sets, NB = 1..8
  B  one launch of F frames, F = 40 000 .. 56 000, same buffers every time
  C  the same F frames, but every launch preceded by touching NOTHING else (baseline B) vs a 2 GiB memset in between"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
ctx = Context(0, stream=stream.cuda_stream)

def timeit(fn, reps, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): fn()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

out = {}
sp = synth.make_problem(10000, "eucm")
prob = Problem.from_synth(ctx, sp); prob.upload_params(sp.intr0, sp.poses0, sp.extr0)
bytes_l = prob.n_corners * (36 + 16 * 12) + sp.n_slots * 48
for nb in (1, 2, 3, 4, 5, 6, 8):
    bufs = [(torch.empty(prob.n_corners * 2, dtype=torch.float64, device=dev), torch.empty(prob.j_len, dtype=torch.float64, device=dev)) for _ in range(nb)]
    state = {"i": 0}
    def step():
        r, J = bufs[state["i"] % nb]; state["i"] += 1
        prob.eval_dev(r.data_ptr(), J.data_ptr())
    us = min(timeit(step, 20 * nb) for _ in range(3))
    out[f"A_nb{nb}"] = {"us_per_launch": us, "TBps": bytes_l / us / 1e6, "footprint_GB": nb * bytes_l / 1e9}
    del bufs
print(json.dumps(out), flush=True)
if os.environ.get("CLIFF_B", "1") == "1":
    outB = {}
    for F in (30000, 40000, 46000, 50000, 56000):
        spF = synth.make_problem(F, "eucm")
        pF = Problem.from_synth(ctx, spF); pF.upload_params(spF.intr0, spF.poses0, spF.extr0)
        r = torch.empty(pF.n_corners * 2, dtype=torch.float64, device=dev); J = torch.empty(pF.j_len, dtype=torch.float64, device=dev)
        us = min(timeit(lambda: pF.eval_dev(r.data_ptr(), J.data_ptr()), 20) for _ in range(3))
        b = pF.n_corners * (36 + 16 * 12) + F * 48
        outB[f"B_{F}"] = {"us": us, "TBps": b / us / 1e6, "GB": b / 1e9}
        # the same launch with a COLD start each time: 4 GiB memset between launches (evicts caches / translations)
        pF.close(); del r, J
    print(json.dumps(outB), flush=True)
if os.environ.get("CLIFF_C", "0") == "1":
    # the same 56 000 frames as K launches of 56 000 / K frames each: DISTINCT sub-problems (own inputs, own outputs)
    outC = {}
    for K in (1, 2, 4, 7):
        F = 56000 // K
        probs = []
        for k in range(K):
            spk = synth.make_problem(F, "eucm", seed=1000 + k)
            pk = Problem.from_synth(ctx, spk); pk.upload_params(spk.intr0, spk.poses0, spk.extr0)
            probs.append((pk, torch.empty(pk.n_corners * 2, dtype=torch.float64, device=dev), torch.empty(pk.j_len, dtype=torch.float64, device=dev)))
        def step():
            for pk, r, J in probs: pk.eval_dev(r.data_ptr(), J.data_ptr())
        us = min(timeit(step, 10) for _ in range(3))
        b = sum(pk.n_corners for pk, _, _ in probs) * (36 + 16 * 12) + 56000 * 48
        outC[f"C_K{K}"] = {"us_total": us, "TBps": b / us / 1e6}
        for pk, _, _ in probs: pk.close()
        del probs
    print(json.dumps(outC), flush=True)
if os.environ.get("CLIFF_D", "0") == "1":
    # 40 000 frames (inputs 115 MB: below the cliff) with the inputs EVICTED from the Infinity Cache between launches
    # by reading a 1 GiB buffer: if the rate falls to the above-the-cliff rate, the cliff is "inputs no longer cached"
    outD = {}
    F = 40000
    spF = synth.make_problem(F, "eucm")
    pF = Problem.from_synth(ctx, spF); pF.upload_params(spF.intr0, spF.poses0, spF.extr0)
    r = torch.empty(pF.n_corners * 2, dtype=torch.float64, device=dev); J = torch.empty(pF.j_len, dtype=torch.float64, device=dev)
    big = torch.zeros(1 << 27, dtype=torch.float64, device=dev)          # 1 GiB
    b = pF.n_corners * (36 + 16 * 12) + F * 48
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    def run(evict):
        ts = []
        for _ in range(12):
            if evict:
                with torch.cuda.stream(stream):
                    s_ = big.sum()
            ev[0].record(stream); pF.eval_dev(r.data_ptr(), J.data_ptr()); ev[1].record(stream)
            torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
        return sorted(ts)[len(ts) // 2]
    outD["warm_inputs_us"] = run(False); outD["evicted_inputs_us"] = run(True)
    outD["warm_TBps"] = b / outD["warm_inputs_us"] / 1e6; outD["evicted_TBps"] = b / outD["evicted_inputs_us"] / 1e6
    print(json.dumps(outD), flush=True)
