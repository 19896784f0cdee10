#!/bin/bash
# Round 5, call aa: merged reduction + record assembly (park area and records side by side in LDS): parity, A/B, stations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05aa; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (swap = the library of commit c15c8ee)"; python tools/ab_build.py "swap@g2swap,new" eucm,kb4,opencv5,ucm 10000 3
echo "== one focal / two cameras / sizes"; python tools/ab_build.py "swap@g2swap,new" eucm,opencv5 10000 2 --one-focal; python tools/ab_build.py "swap@g2swap,new" eucm,kb4 10000 2 --cams 2; python tools/ab_build.py "swap@g2swap,new" eucm 2500,5000,20000,50000 2
} > $O/ab_merge.txt 2>&1
cat $O/ab_merge.txt
for m in eucm kb4 opencv5; do CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_stamps1.so timeout 200 python3 tools/stamps_g2.py 10000 $m 2>&1 | grep -v amdgpu.ids; done > $O/stamps1.txt
cat $O/stamps1.txt
