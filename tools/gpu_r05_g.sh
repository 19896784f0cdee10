#!/bin/bash
# round 5, call g: mode E with one wavefront per PASS (k_eval<..., ITEMS>): parity through the variant library, then A/B against the
# per-frame form (base) - 2 wavefronts per workgroup (items), 1 (items1), 4 (itemspf)
mkdir -p gpurun_out
V=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants
CCAL_LIB=$V/libccal_items.so python -m pytest tests/test_gpu_eval.py tests/test_gpu_configs.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -3
{
for m in eucm kb4 opencv5; do
  echo "== mode E $m 10000 frames (GB/s)"; python tools/ab_eval.py base,items,items1,itemspf 10000 3 --model $m
done
echo "== two EUCM cameras x 10000"; python tools/ab_eval.py base,items,items1,itemspf 10000 3 --cams 2
echo "== EUCM 1000 / 2500 / 50000 frames"; python tools/ab_eval.py base,items,items1,itemspf 1000,2500,50000 2
} > gpurun_out/r05g_ab_items.txt 2>&1
cat gpurun_out/r05g_ab_items.txt
# the headline command alone under rocprofv3 --stats (k_eval's average must agree with roofline.kernel_ms)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/prof_r05
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r05/stats_headline -o stats -- python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-traffic > $REPO/gpurun_out/prof_r05/stats_headline_bench.json 2> $REPO/gpurun_out/prof_r05/stats_headline.err
head -5 $REPO/gpurun_out/prof_r05/stats_headline/stats_kernel_stats.csv
find $REPO/gpurun_out/prof_r05/stats_headline -name "*.csv" -size +20M -delete
