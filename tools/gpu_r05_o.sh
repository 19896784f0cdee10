#!/bin/bash
# Round 5, call o: where the passes of k_gram2's corner loop go (lib/variants/libccal_stamps2.so: -DCCAL_STAMPS=2, shader cycles per section)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05o; mkdir -p $O
for F in 10000 50000 1000; do CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_stamps2.so timeout 200 python3 tools/stamps_g2.py $F eucm 2>&1 | grep -v amdgpu.ids; done > $O/stamps2_eucm.txt
cat $O/stamps2_eucm.txt
