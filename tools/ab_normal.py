#!/usr/bin/env python3
"""Developer tool: A/B of ccal_build_normal_dev over library variants, one subprocess per (variant, size), round-robin."""
import json, os, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1].split(","); sizes = [int(x) for x in sys.argv[2].split(",")]; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
res = {}
for rd in range(rounds):
    for F in sizes:
        for v in variants:
            env = dict(os.environ)
            if v != "base": env["CCAL_LIB"] = f"{root}/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_{v}.so"
            o = subprocess.run([sys.executable, f"{root}/tools/time_kernels.py", "--what", "normal", "--frames", str(F), "--reps", "200"] + extra,
                               env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
            try: res.setdefault((v, F), []).append(json.loads(o)["normal_us"])
            except Exception: print("ERR", v, F, o[-300:])
for (v, F), xs in sorted(res.items(), key=lambda t: (t[0][1], t[0][0])):
    print(f"{F:6d} {v:12s} median {statistics.median(xs):7.2f} us  all {[round(x, 1) for x in xs]}")
