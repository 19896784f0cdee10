#!/bin/bash
# round 5, call j: OPENCV5 members in lockstep batches; lockstep on / off side by side (the second library has the switch)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batch.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -3
L=$PWD/camera_intrinsic_calibration_rs_amd/lib/libccal_hip_legacy.so
{
for m in eucm kb4 opencv5; do
  echo "== $m 625 frames, lockstep (product)"; python tools/concurrent_sessions.py 625 $m | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['ms_per_batch'],4), round(v['speedup_vs_1'],2)) for k,v in d['by_sessions'].items()})"
  echo "== $m 625 frames, per-context threads (second library, CCAL_BATCH_LOCKSTEP_OFF=1)"; CCAL_LIB=$L CCAL_BATCH_LOCKSTEP_OFF=1 python tools/concurrent_sessions.py 625 $m | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['ms_per_batch'],4), round(v['speedup_vs_1'],2)) for k,v in d['by_sessions'].items()})"
done
echo "== eucm 300 frames LM lockstep"; python tools/concurrent_sessions.py 300 eucm --lm | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['ms_per_batch'],4), round(v['speedup_vs_1'],2)) for k,v in d['by_sessions'].items()})"
} > gpurun_out/r05j_batch.txt 2>&1
cat gpurun_out/r05j_batch.txt
