#!/bin/bash
# Round 5, call r: identity-slot prologue (pose / elimination record requested without waiting for obs_slot) against the previous
# commit's numbers; KB4 on k_gram2 (neighbouring-lane form) against k_gram1v; sessions
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05r; mkdir -p $O
L=$R/camera_intrinsic_calibration_rs_amd/lib/libccal_hip_legacy.so
timeout 900 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (swap = the library of commit c15c8ee)"; python tools/ab_build.py "swap@g2swap,ident" eucm,kb4,opencv5 10000 3
echo "== KB4: k_gram1v (default) against k_gram2 (CCAL_GRAM2=1, second library)"; CCAL_LIB=$L python tools/ab_build.py "g1v,g2:CCAL_GRAM2=1" kb4 10000,2500 3
echo "== KB4 one focal"; CCAL_LIB=$L python tools/ab_build.py "g1v,g2:CCAL_GRAM2=1" kb4 10000 2 --one-focal
echo "== sessions"; python tools/ab_build.py "swap@g2swap,ident" eucm,kb4 625,2500 3
} > $O/ab_ident.txt 2>&1
cat $O/ab_ident.txt
