#!/bin/bash
# round 5, call i: k_eval_stream - G frames per wavefront as one corner stream (lane utilisation 75 % -> 90 / 100 % on 144-corner frames)
mkdir -p gpurun_out
V=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants
for v in g2 g4; do CCAL_LIB=$V/libccal_$v.so python -m pytest tests/test_gpu_eval.py tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_normal.py -m gpu -x -q 2>&1 | tail -2; done
{
for m in eucm kb4 opencv5; do
  echo "== mode E $m 10000 frames (GB/s)"; python tools/ab_eval.py base,g2,g4,g2w1 10000 3 --model $m
done
echo "== two EUCM cameras x 10000"; python tools/ab_eval.py base,g2,g4,g2w1 10000 3 --cams 2
echo "== EUCM ragged 10000"; python tools/ab_eval.py base,g2,g4 10000 3 --ragged
echo "== EUCM 1000 / 2500 / 50000 frames"; python tools/ab_eval.py base,g2,g4 1000,2500,50000 2
} > gpurun_out/r05i_ab_stream.txt 2>&1
cat gpurun_out/r05i_ab_stream.txt
