#!/bin/bash
# Round 5, call u: stations of k_gram2's epilogue (lib/variants/libccal_stamps1.so: -DCCAL_STAMPS)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05u; mkdir -p $O
for m in eucm kb4 opencv5; do CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_stamps1.so timeout 200 python3 tools/stamps_g2.py 10000 $m 2>&1 | grep -v amdgpu.ids; done > $O/stamps1.txt
cat $O/stamps1.txt
