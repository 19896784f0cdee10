#!/bin/bash
# Round 6, call x: the finishing single-launch group writes the result with ALL its workgroups (head_finish SPREAD): parity + A/B (legacy library, CCAL_RESULT_SPREAD=0|1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06x; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_iter.py tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
export CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/libccal_hip_legacy.so
for rep in 1 2; do for S in 0 1; do CCAL_RESULT_SPREAD=$S python tools/host_pointer_ab.py "10000,10000:ragged,9000,5000,2500" 10; done; done > $O/ab_result_spread.txt 2>&1
cat $O/ab_result_spread.txt
