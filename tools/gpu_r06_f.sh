#!/bin/bash
# Round 6, call f: the whole GPU suite, the round's profile set (rocprofv3 kernel stats + PMC passes), kernel stats of the ragged build,
# a randomised sweep weighted to large ragged problems (binned Gram launch), the multi-GPU runbook's n = 1 legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06f; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; tail -4 $O/pytest_full.log
python tools/fuzz_parity.py --seconds 150 --seed 60606 --shards 3 --batch 6 --big 0.5 > $O/fuzz_seed60606_big.json 2> $O/fuzz.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r06f/fuzz_seed60606_big.json")); print("fuzz", {k: d[k] for k in d if k in ("cases", "n_fail", "worst", "sharded_cases", "batched_cases")})
PY
bash profiles/run_profile.sh r06 > $O/run_profile.log 2>&1; tail -3 $O/run_profile.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06/stats_ragged -o stats -- python3 $R/tools/time_kernels.py --what eval,normal,solve --ragged --reps 50 > $R/gpurun_out/prof_r06/ragged.json 2> $R/gpurun_out/prof_r06/stats_ragged.err
cd $R
bash tools/multi_gpu_day.sh r06 1 > $O/multi_gpu_day.log 2>&1; tail -25 $O/multi_gpu_day.log
find gpurun_out/prof_r06 -name "*.csv" -size +20M -delete
