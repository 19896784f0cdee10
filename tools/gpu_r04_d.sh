#!/bin/bash
# round-4 GPU pass D: k_schurq with eight lanes per slot
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_normal.py tests/test_gpu_dist.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
timeout 300 python tools/ab_build.py "s16:CCAL_SCHURQ_SLOTS=16,s8:CCAL_SCHURQ_SLOTS=8" eucm 10000,20000 3 --cams 2 > $O/ab_schurq.txt 2>&1
timeout 200 python tools/ab_build.py "s16:CCAL_SCHURQ_SLOTS=16,s8:CCAL_SCHURQ_SLOTS=8" ucm 10000 2 --cams 2 --one-focal >> $O/ab_schurq.txt 2>&1
timeout 200 python tools/ab_build.py "s16:CCAL_SCHURQ_SLOTS=16;CCAL_SCHURQ=1,s8:CCAL_SCHURQ_SLOTS=8;CCAL_SCHURQ=1,gen:CCAL_SCHURQ=0" eucm 1000,3000,6000 2 --cams 2 >> $O/ab_schurq.txt 2>&1
CCAL_SCHURQ_SLOTS=8 timeout 120 bash tools/kstats.sh --what normal --cams 2 --reps 50 > $O/kstats_s8.txt 2>&1
CCAL_SCHURQ_SLOTS=16 timeout 120 bash tools/kstats.sh --what normal --cams 2 --reps 50 > $O/kstats_s16.txt 2>&1
cat $O/pytest.log $O/ab_schurq.txt; head -4 $O/kstats_s8.txt $O/kstats_s16.txt
