#!/bin/bash
# round 5, call f: full GPU suite on the round's library, randomised sweep incl. sharded (3 shards) and batched (6 per call) solves,
# the headline command alone under rocprofv3 --stats (k_eval's average must agree with roofline.kernel_ms)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05f_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05f_tests.log
tail -4 gpurun_out/r05f_tests.log
python tools/fuzz_parity.py --seconds 240 --seed 50505 --shards 3 --batch 6 > gpurun_out/r05f_fuzz.json 2> gpurun_out/r05f_fuzz.err
echo "fuzz rc=$?"; head -c 1500 gpurun_out/r05f_fuzz.json
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r05/stats_headline -o stats -- python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $REPO/gpurun_out/prof_r05/stats_headline_bench.json 2> $REPO/gpurun_out/prof_r05/stats_headline.err
cat $REPO/gpurun_out/prof_r05/stats_headline/stats_kernel_stats.csv | head -5
find $REPO/gpurun_out/prof_r05/stats_headline -name "*.csv" -size +20M -delete
