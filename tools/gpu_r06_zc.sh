#!/bin/bash
# Round 6, call zc: the final library once more through the randomised sweep with shards and lockstep batches, and the multi-GPU runbook's one-GPU legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06zc; mkdir -p $O
timeout 900 python tools/fuzz_parity.py --seed 60909 --seconds 420 --shards 3 --batch 6 --big 0.15 --huge > $O/fuzz_seed60909_shards3_batch6_huge.json 2> $O/fuzz.err; echo "fuzz rc $?"; grep -n "\"cases\"\|n_fail\|sharded_cases\|batched_cases" $O/fuzz_seed60909_shards3_batch6_huge.json
CCAL_DAY_OUT=$O/multi_gpu bash tools/multi_gpu_day.sh r06 > $O/multi_gpu_day.log 2>&1; echo "multi_gpu_day rc $?"; tail -25 $O/multi_gpu/summary.txt | cut -c1-300
