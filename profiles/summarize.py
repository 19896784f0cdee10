#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/run_profile.sh (gpurun_out/prof_<tag>/) into the small files
committed under profiles/<tag>/:

  kernel_stats_bench_steps50.csv   rocprofv3 --kernel-trace --stats summary of `bench.py --steps 50 --warmup 5`
  kernel_stats_two_camera.csv      the same for tools/time_kernels.py --cams 2 (2 x 10 000 frames, EUCM) + two_camera.json
  kernel_stats_kb4 / _opencv5.csv  the same for --model kb4 / opencv5 (10 000 frames) + model_*.json
  pmc_summary.json                 per-kernel means of the --pmc passes + the HBM bytes per k_eval launch that
                                   bench.py reports as roofline.traffic
  bench_full.json                  the default `python bench.py` line of the same box

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE come from separate passes, are
in KiB, and on gfx950 FETCH_SIZE under-reports streaming reads by 2x.

    python profiles/summarize.py r01
"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main() -> None:
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, "kernel_stats_bench_steps50.csv"))
    hl = os.path.join(src, "stats_headline", "stats_kernel_stats.csv")       # the headline command alone: k_eval's average = roofline.kernel_ms
    if os.path.exists(hl):
        shutil.copyfile(hl, os.path.join(dst, "kernel_stats_headline.csv"))
    line = [l for l in open(os.path.join(src, "bench_full.json")) if l.startswith("{")][-1]
    with open(os.path.join(dst, "bench_full.json"), "w") as f:
        f.write(line)
    bench = json.loads(line)

    counters = defaultdict(lambda: defaultdict(list))
    for extra_dir, name in (("stats2", "kernel_stats_two_camera.csv"), ("stats_kb4", "kernel_stats_kb4.csv"), ("stats_opencv5", "kernel_stats_opencv5.csv")):
        f = os.path.join(src, extra_dir, "stats_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copyfile(f, os.path.join(dst, name))
    for jf in ("two_camera.json", "model_kb4.json", "model_opencv5.json"):
        f = os.path.join(src, jf)
        if os.path.exists(f):
            lines = [l for l in open(f) if l.startswith("{")]
            if lines:
                with open(os.path.join(dst, jf), "w") as o:
                    o.write(lines[-1])
    dropped = defaultdict(int)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mem"):
        path = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        per_dispatch = defaultdict(float)        # a counter is reported once per XCD/instance: sum them per dispatch
        names, dur = {}, {}
        for r in csv.DictReader(open(path)):
            if not r["Kernel_Name"].startswith(("ccal::", "void ccal::")):
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        # The device-resident loops enqueue groups ahead: the kernels of a group that runs after the solve has finished return at
        # their first instruction.  Such dispatches (and re-elimination groups that skip the evaluation) carry the kernel's name
        # and almost none of its work: a dispatch shorter than HALF its kernel's median duration is left out of the means.
        by_kernel = defaultdict(list)
        for disp, k in names.items():
            by_kernel[k].append(dur[disp])
        median = {k: sorted(v)[len(v) // 2] for k, v in by_kernel.items()}
        keep = {disp for disp, k in names.items() if dur[disp] >= 0.5 * median[k]}
        for disp, k in names.items():
            if disp not in keep:
                dropped[k] += 1
        for (disp, cname), v in per_dispatch.items():
            if disp in keep:
                counters[names[disp]][cname].append(v)
        for disp in keep:
            counters[names[disp]]["duration_ns__" + sub].append(dur[disp])

    out_c = {}
    for k, cs in counters.items():
        out_c[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())}
        m = {c: d["mean"] for c, d in out_c[k].items()}
        derived = {}
        # fractions of quantities that one pass measured together (the same dispatches): the two SQ passes of run_profile.sh
        if m.get("SQ_LDS_IDX_ACTIVE"):
            derived["lds_bank_conflict_frac_of_lds_active_cycles"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("SQ_WAVES") and "SQ_INSTS_VALU" in m:
            derived["valu_insts_per_wavefront"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in m and m.get("SQ_ACTIVE_INST_ANY"):
            derived["valu_frac_of_active_inst_cycles"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_ACTIVE_INST_ANY"]
        if "SQ_WAIT_ANY" in m and "SQ_ACTIVE_INST_ANY" in m and "SQ_WAIT_INST_ANY" in m:
            # a wavefront's cycles ~ waiting (s_waitcnt) + issuing + stalled at issue: the pass has no SQ_WAVE_CYCLES, so the
            # fractions are of the sum of the three it measured
            tot = m["SQ_WAIT_ANY"] + m["SQ_ACTIVE_INST_ANY"] + m["SQ_WAIT_INST_ANY"]
            if tot > 0:
                derived["wait_any_frac"] = m["SQ_WAIT_ANY"] / tot
                derived["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] / tot if "SQ_ACTIVE_INST_VALU" in m else None
                derived["issue_stall_frac"] = m["SQ_WAIT_INST_ANY"] / tot
        if m.get("SQ_WAVE_CYCLES") and "SQ_BUSY_CYCLES" in m and m.get("SQ_WAVES"):
            derived["wave_cycles_per_wavefront"] = m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]
        if derived:
            out_c[k]["derived"] = derived
        if dropped.get(k):
            out_c[k]["dispatches_left_out_as_early_exits"] = dropped[k]
    ev = next((k for k in out_c if k.startswith("ccal::k_eval")), None)
    summary = {
        "note": "rocprofv3 --pmc passes of `bench.py --steps 50 --warmup 5 --no-cpu-baseline` "
                f"({bench['config']['workload']}); per-dispatch means over the dispatches that did the kernel's work (a dispatch shorter "
                "than half its kernel's median duration - an early-exit group enqueued ahead of a finished solve - is left out). "
                "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 "
                "FETCH_SIZE under-reports streaming reads by 2x (MI355X_MICROARCH.md, HBM) so read bytes = "
                "2 * FETCH_SIZE * 1024.",
    }
    if ev and "FETCH_SIZE" in out_c[ev] and "WRITE_SIZE" in out_c[ev]:
        rd = 2.0 * out_c[ev]["FETCH_SIZE"]["mean"] * 1024.0
        wr = out_c[ev]["WRITE_SIZE"]["mean"] * 1024.0
        summary.update(k_eval_hbm_traffic_bytes_per_launch=rd + wr, k_eval_read_bytes=rd, k_eval_write_bytes=wr,
                       k_eval_algorithmic_bytes_per_launch=bench["roofline"]["algorithmic_bytes_per_launch"],
                       # what the figure was measured on: bench.py accepts it only while the kernel's sources still hash to this
                       k_eval_source_sha256_16=bench["roofline"].get("traffic_stale_if_kernel_changed"))
    summary["counters"] = out_c
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", dst, "traffic", summary.get("k_eval_hbm_traffic_bytes_per_launch"))


if __name__ == "__main__":
    main()
