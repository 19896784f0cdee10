#!/bin/bash
# Round 6, call w: the library with folded plans: whole GPU suite, smoke, randomised parity (big single-camera cases), the profile set again
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06w; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; grep -n "passed\|failed" $O/pytest_full.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"
timeout 900 python tools/fuzz_parity.py --seed 606 --seconds 420 --big 0.25 --huge > $O/fuzz_seed606.json 2> $O/fuzz_seed606.err; echo "fuzz rc $?"; tail -c 400 $O/fuzz_seed606.json
bash profiles/run_profile.sh r06 > $O/run_profile.log 2>&1; tail -2 $O/run_profile.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06/stats_ragged -o stats -- python3 $R/tools/time_kernels.py --what eval,normal,solve --ragged --reps 50 > $R/gpurun_out/prof_r06/ragged.json 2> $R/gpurun_out/prof_r06/stats_ragged.err
cd $R
find gpurun_out/prof_r06 -name "*.csv" -size +20M -delete
tail -c 1500 gpurun_out/prof_r06/bench_full.json
