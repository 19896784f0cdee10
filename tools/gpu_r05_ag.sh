#!/bin/bash
# Round 5, call ag: the driver's round-end sequence on the final tree: smoke(), then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05ag; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05ag/bench_default.json"))
print({k: d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data")})
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
print("modeN", d["extra"]["mode_N_build_ms"], {m: d["extra"]["config2"][m]["mode_N_build_ms"] for m in ("kb4","opencv5")})
PY
