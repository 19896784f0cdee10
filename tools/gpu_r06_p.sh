#!/bin/bash
# Round 6, call p: where the 0.6 us between this round's and round 5's EUCM build (uniform frames) sit: per-kernel table of both libraries on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06p; mkdir -p $O
export CCAL_LIB_ALLOW_MISSING=1
for rep in 1 2; do
for L in base r05; do
  if [ $L = r05 ]; then export CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_r05.so; else unset CCAL_LIB; fi
  echo "== $L (pass $rep)"; bash tools/kstats.sh --what normal --reps 300 | head -4
done
done > $O/kstats_build_base_vs_r05.txt 2>&1
cat $O/kstats_build_base_vs_r05.txt
