#!/usr/bin/env python3
"""Developer tool: whole normal-equation build (ccal_build_normal_dev) against lanes per frame of the Gram kernels,
one subprocess per point (the override is read once per process).  Prints a table and the best LPF per size."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1500,2500,3500,5000,7000,8500,10000,12000,16000,20000,30000,50000").split(",")]
lpfs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,12,16,32").split(",")]
extra = sys.argv[3:]
print("frames " + " ".join(f"lpf{l:>3d}" for l in lpfs) + "   best")
for F in sizes:
    row = []
    for l in lpfs:
        env = dict(os.environ, CCAL_GRAMV_LPF=str(l))
        o = subprocess.run([sys.executable, f"{root}/tools/time_kernels.py", "--what", "normal", "--frames", str(F), "--reps", "100"] + extra,
                           env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        try: row.append(json.loads(o)["normal_us"])
        except Exception: row.append(float("nan"))
    best = lpfs[min(range(len(lpfs)), key=lambda i: row[i] if row[i] == row[i] else 1e9)]
    print(f"{F:6d} " + " ".join(f"{x:6.1f}" for x in row) + f"   {best}", flush=True)
