#!/usr/bin/env python3
"""Developer tool: A/B of ccal_build_normal_dev (and optionally GN solves) over (library variant, environment) pairs, one
subprocess per measurement, round-robin.
    tools/ab_build.py "base,g2:CCAL_GRAM2=1,minw1@minw1" kb4,opencv5 10000 3 [--one-focal]
a spec is  name[@variant][:ENV=VAL[;ENV=VAL]]"""
import json, os, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
specs = sys.argv[1].split(","); models = sys.argv[2].split(","); sizes = [int(x) for x in sys.argv[3].split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
extra = sys.argv[5:]
res = {}
for rd in range(rounds):
    for m in models:
        for F in sizes:
            for sp in specs:
                name, _, envs = sp.partition(":")
                name, _, var = name.partition("@")
                env = dict(os.environ)
                if var: env["CCAL_LIB"] = f"{root}/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_{var}.so"
                for kv in filter(None, envs.split(";")):
                    k, _, v = kv.partition("="); env[k] = v
                o = subprocess.run([sys.executable, f"{root}/tools/time_kernels.py", "--what", "normal,solve", "--model", m, "--frames", str(F), "--reps", "200"] + extra,
                                   env=env, capture_output=True, text=True)
                try:
                    d = json.loads(o.stdout.strip().split("\n")[-1])
                    res.setdefault((m, F, sp), []).append((d["normal_us"], d["gn_ms"], d["gn_iters"], d["lm_ms"]))
                except Exception:
                    print("ERR", m, F, sp, o.stdout[-300:], o.stderr[-300:])
for (m, F, sp), xs in sorted(res.items()):
    print(f"{m:8s} {F:6d} {sp:40s} build median {statistics.median(x[0] for x in xs):7.2f} us  all {[round(x[0], 1) for x in xs]}  GN {min(x[1] for x in xs):.3f} ms ({xs[0][2]} it)  LM {min(x[3] for x in xs):.3f} ms")
