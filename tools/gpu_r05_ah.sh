#!/bin/bash
# Round 5, call ah: k_gram2's prologue with the state-independent requests in front of the state (both parameter sets): parity, A/B, stations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05ah; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (prev = the library before)"; python tools/ab_build.py "prev@prev,early" eucm,kb4,opencv5,ucm 10000 5
echo "== other sizes, ragged"; python tools/ab_build.py "prev@prev,early" eucm 2500,20000 3; python tools/ab_build.py "prev@prev,early" eucm 10000 3 --ragged
} > $O/ab_early.txt 2>&1
cat $O/ab_early.txt
for m in eucm kb4; do CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_stamps1.so timeout 200 python3 tools/stamps_g2.py 10000 $m 2>&1 | grep -v amdgpu.ids; done > $O/stamps1.txt
cat $O/stamps1.txt | cut -c1-300
