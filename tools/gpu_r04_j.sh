#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04j; mkdir -p $O; rm -f $O/*.json
CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_stamps.so timeout 120 python3 tools/stamps_g1v.py 625 eucm 2>&1 | grep -v amdgpu.ids

for rep in 1 2; do
for v in on off; do
  if [ $v = off ]; then export CCAL_ITER_ROWS=0; else unset CCAL_ITER_ROWS; fi
  for f in 100 300 625 1000 2000; do
    timeout 200 python3 tools/time_kernels.py --frames $f --what solve --tag iter_$v > $O/t_${v}_${f}_$rep.json 2>> $O/t.err
  done
  timeout 200 python3 tools/time_kernels.py --frames 625 --model kb4 --what solve --tag iter_$v > $O/t_${v}_kb4_$rep.json 2>> $O/t.err
  timeout 200 python3 tools/time_kernels.py --frames 625 --model opencv5 --what solve --tag iter_$v > $O/t_${v}_ocv5_$rep.json 2>> $O/t.err
done
done
unset CCAL_ITER_ROWS
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04j/t_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms')})
    except Exception as x: print(f,'ERR',x)
PY
timeout 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
