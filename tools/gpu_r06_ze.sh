#!/bin/bash
# Round 6, call ze: the profile set of the final library (fast Huber branch): rocprofv3 kernel stats + PMC passes + the bench line + ragged kernel stats + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06ze; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"
bash profiles/run_profile.sh r06 > $O/run_profile.log 2>&1; tail -2 $O/run_profile.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06/stats_ragged -o stats -- python3 $R/tools/time_kernels.py --what eval,normal,solve --ragged --reps 50 > $R/gpurun_out/prof_r06/ragged.json 2> $R/gpurun_out/prof_r06/stats_ragged.err
cd $R
find gpurun_out/prof_r06 -name "*.csv" -size +20M -delete
bash tools/pmc_normal.sh > $O/pmc_normal.txt 2>&1; tail -12 $O/pmc_normal.txt
tail -c 1500 gpurun_out/prof_r06/bench_full.json
