#!/bin/bash
# round 5, call d: lockstep batch groups (k_gram1v_batch), mode-E kernel with counted stores (A/B: old kernel, new without / with
# prefetch), bench
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05d_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05d_tests.log
tail -5 gpurun_out/r05d_tests.log
{
for m in eucm kb4 opencv5; do
  echo "== mode E $m 10000 frames (GB/s)"; python tools/ab_eval.py old,nopf,pf 10000 3 --model $m
done
echo "== mode E two EUCM cameras x 10000 (GB/s, both blocks)"; python tools/ab_eval.py old,nopf,pf 10000 3 --cams 2
echo "== mode E EUCM 50000 frames"; python tools/ab_eval.py old,nopf,pf 50000 2
echo "== mode E EUCM one-focal / 1000 frames"; python tools/ab_eval.py old,nopf,pf 1000 3
} > gpurun_out/r05d_ab_eval.txt 2>&1
cat gpurun_out/r05d_ab_eval.txt
python bench.py > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d_bench.json"))
e = d["extra"]
print("value", d["value"], "frac", d["roofline"]["frac"])
print("build", e.get("mode_N_build_ms"), "gn", e.get("gn_solve_ms"), "lm", e.get("lm_solve_ms"))
print("config0", e["config0"]["gpu_ms"])
for k in ("1", "2", "4", "8"):
    r = e["concurrent_sessions"]["by_sessions"][k]; print(k, r["ms_per_batch"], r.get("equal_to_sequential_1e-11"))
print("parity", d["parity"]["pass"])
PY
