#!/bin/bash
# Round 6, call h: k_head requests the reduced sums with its other inputs (one round trip instead of two): kernel table + GN, against the library before (prev)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
bash tools/kstats.sh --what solve --reps 50 > $O/kstats_solve.txt 2>&1; cat $O/kstats_solve.txt
CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_nobins.so bash tools/kstats.sh --what solve --reps 50 > $O/kstats_solve_prev.txt 2>&1; cat $O/kstats_solve_prev.txt
python tools/ab_build.py "base,prev@nobins" eucm 10000,3000 5
