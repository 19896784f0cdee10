#!/usr/bin/env python3
"""Developer tool: the interesting numbers of a bench.py JSON line."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
e = d.get("extra", {})
print("value %.4g evals/s  roofline frac %.4f  kernel_ms %.5f" % (d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"]))
print("mode N build ms", e.get("mode_N_build_ms"), json.dumps(e.get("mode_N_roofline"))[:600])
for k in ("gn", "lm"):
    print(k, {x: e.get(f"{k}_{x}") for x in ("iterations", "solve_ms", "iters_per_s")}, "device-resident", e.get(f"{k}_device_resident"))
for k in e:
    if k.startswith("frames"): print(k, json.dumps(e[k]))
if "sharded_solve" in e: print("sharded", json.dumps(e["sharded_solve"]))
c = d.get("cpu_baseline")
if c: print("cpu", c["value"], "cores", c["cores"], "eff", c.get("scaling_efficiency"), "heap", c["port_heap"]["value"], c["host"])
