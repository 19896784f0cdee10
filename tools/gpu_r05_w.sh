#!/bin/bash
# Round 5, call w: the whole GPU suite on the final library, then the profile passes (profiles/run_profile.sh r05)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05w; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
bash profiles/run_profile.sh r05 > $O/run_profile.log 2>&1; tail -5 $O/run_profile.log
