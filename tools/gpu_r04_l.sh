#!/bin/bash
# round-4 GPU pass L: A/B harness for "the decision in front of k_gram2" (CCAL_HEAD_FRONT=0 = off).  The variant was measured with this script and
# NOT kept: the switch does not exist in the committed library, where both legs run the same code
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04l; mkdir -p $O; rm -f $O/*.json
for rep in 1 2; do
for v in on off; do
  if [ $v = off ]; then export CCAL_HEAD_FRONT=0; else unset CCAL_HEAD_FRONT; fi
  for m in eucm ucm opencv5; do
    timeout 200 python3 tools/time_kernels.py --frames 10000 --model $m --what normal,solve --tag hf_$v > $O/t_${v}_${m}_$rep.json 2>> $O/t.err
  done
  timeout 200 python3 tools/time_kernels.py --frames 3000 --model eucm --what normal,solve --tag hf_$v > $O/t_${v}_eucm3k_$rep.json 2>> $O/t.err
done
done
unset CCAL_HEAD_FRONT
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04l/t_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms') or k=='normal_us' or k.endswith('_iters')})
    except Exception as x: print(f,'ERR',x)
PY
timeout 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_configs.py tests/test_gpu_normal.py tests/test_gpu_iter.py -m gpu -x -q 2>&1 | tail -3
