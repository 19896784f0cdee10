#!/bin/bash
# round 5, call a: in-process transport v2 (rank sums in the deciding kernel), bench launcher, multi-device legs
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05a_tests.log
tail -5 gpurun_out/r05a_tests.log
python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
echo "bench rc=$?"
python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r05a_bench2.json 2> gpurun_out/r05a_bench2.err
echo "bench --gpus 2 on one GPU rc=$? (expected non-zero)"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a_bench.json"))
e = d["extra"]
print("value", d["value"], "frac", d["roofline"]["frac"])
print("build", e.get("mode_N_build_ms"), "gn", e.get("gn_solve_ms"), "lm", e.get("lm_solve_ms"))
print("sps", json.dumps(e.get("single_process_sharded"))[:1500])
print("multi_eval", e.get("multi_eval"))
print("conc", json.dumps(e.get("concurrent_sessions"))[:600])
PY
