#!/bin/bash
# round 5, call c: ccal_solve_batch sizes launches for the share of the GPU; single-launch selection for validation(); calib_cameras(devices)
# radix-select validation statistics instead of the library sort
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05c_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05c_tests.log
tail -5 gpurun_out/r05c_tests.log
python bench.py > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05c_bench.json"))
e = d["extra"]
print("value", d["value"], "frac", d["roofline"]["frac"])
print("build", e.get("mode_N_build_ms"), "gn", e.get("gn_solve_ms"), "lm", e.get("lm_solve_ms"))
print("config0", e["config0"]["gpu_ms"])
print("sps", {k: (v.get("gn", {}).get("solve_ms") if isinstance(v, dict) else v) for k, v in e["single_process_sharded"].items()})
for k in ("1", "2", "4", "8"):
    r = e["concurrent_sessions"]["by_sessions"][k]; print(k, r["ms_per_batch"])
print("parity", d["parity"]["pass"])
PY
