#!/bin/bash
# Round 6, call k: the final library - whole GPU suite, smoke, two randomised sweeps, the round's profile set, the multi-GPU runbook's n = 1 legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06k; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; grep -n "passed\|failed" $O/pytest_full.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.txt
python tools/fuzz_parity.py --seconds 150 --seed 60707 --shards 3 --batch 6 > $O/fuzz_seed60707_shards3_batch6.json 2> $O/fuzz1.err
python tools/fuzz_parity.py --seconds 120 --seed 60808 --shards 3 --batch 6 --big 0.5 > $O/fuzz_seed60808_shards3_batch6_big.json 2> $O/fuzz2.err
python - <<'PY'
import json
for f in ("fuzz_seed60707_shards3_batch6.json", "fuzz_seed60808_shards3_batch6_big.json"):
    d = json.load(open("gpurun_out/r06k/" + f)); print(f, {k: d[k] for k in ("cases", "n_fail", "worst", "sharded_cases", "batched_cases") if k in d})
PY
bash profiles/run_profile.sh r06 > $O/run_profile.log 2>&1; tail -2 $O/run_profile.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06/stats_ragged -o stats -- python3 $R/tools/time_kernels.py --what eval,normal,solve --ragged --reps 50 > $R/gpurun_out/prof_r06/ragged.json 2> $R/gpurun_out/prof_r06/stats_ragged.err
cd $R
CCAL_DAY_OUT=gpurun_out/multi_gpu_r06 bash tools/multi_gpu_day.sh r06 1 > $O/multi_gpu_day.log 2>&1; tail -12 $O/multi_gpu_day.log | cut -c1-300
find gpurun_out/prof_r06 -name "*.csv" -size +20M -delete
