#!/bin/bash
# Round 6, call b: the binned Gram launch for ragged frames (k_gram2b): parity, A/B against the launch without bins, kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06b; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_batch.py tests/test_gpu_multi.py tests/test_gpu_iter.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
{
echo "== ragged 24..144 corners per frame: bins (base) against the same library without them (nobins)"
python tools/ab_build.py "base,nobins@nobins" eucm,kb4,opencv5 10000 3 --ragged
python tools/ab_build.py "base,nobins@nobins" eucm 5000,20000,50000 3 --ragged
echo "== full frames: must be unchanged"
python tools/ab_build.py "base,nobins@nobins" eucm 10000 3
} > $O/ab_bins.txt 2>&1
cat $O/ab_bins.txt
bash tools/kstats.sh --what normal --ragged --reps 50 > $O/kstats_ragged.txt 2>&1; cat $O/kstats_ragged.txt
