#!/bin/bash
# round 5, call k: what the 18-column block's time is made of - the kernel with its J stores removed (compute + LDS only) and with
# its projection / chain rule removed (store path only), next to the whole kernel.  "GB/s" = algorithmic bytes / time in all three.
mkdir -p gpurun_out
{
echo "== two EUCM cameras x 10000 (both blocks)"; python tools/ab_eval.py base,nostore,nocompute 10000 3 --cams 2
echo "== EUCM single"; python tools/ab_eval.py base,nostore,nocompute 10000 3
echo "== KB4"; python tools/ab_eval.py base,nostore,nocompute 10000 2 --model kb4
} > gpurun_out/r05k_diag.txt 2>&1
cat gpurun_out/r05k_diag.txt
cd /tmp && export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/camera_intrinsic_calibration_rs_amd/lib/variants
for v in nostore nocompute; do
  CCAL_LIB=$V/libccal_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/diag_$v -o stats -- python3 $GRAFT_REPO_ROOT/tools/time_kernels.py --what eval --cams 2 --reps 30 > /dev/null 2>&1
  echo "== $v"; head -4 $GRAFT_REPO_ROOT/gpurun_out/diag_$v/stats_kernel_stats.csv | cut -c1-140
  find $GRAFT_REPO_ROOT/gpurun_out/diag_$v -name "*trace*.csv" -delete
done
