#!/bin/bash
# round-4 GPU pass H: single-launch groups (k_gram1v<.., ITER>): tests, A/B against CCAL_ITER_ROWS=0, fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04h; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest_quick.log 2>&1; echo "pytest quick rc $?" > $O/summary.txt
tail -5 $O/pytest_quick.log
for v in on off; do
  if [ $v = off ]; then export CCAL_ITER_ROWS=0; else unset CCAL_ITER_ROWS; fi
  for f in 300 625 1000 2000; do
    timeout 200 python3 tools/time_kernels.py --frames $f --what solve --tag iter_$v > $O/t_${v}_$f.json 2>> $O/t.err
  done
  timeout 200 python3 tools/time_kernels.py --frames 625 --model kb4 --what solve --tag iter_$v > $O/t_${v}_kb4.json 2>> $O/t.err
  timeout 200 python3 tools/time_kernels.py --frames 625 --model opencv5 --what solve --tag iter_$v > $O/t_${v}_ocv5.json 2>> $O/t.err
  timeout 200 python3 tools/time_kernels.py --frames 10000 --model kb4 --what solve --tag iter_$v > $O/t_${v}_kb4_10k.json 2>> $O/t.err
done
unset CCAL_ITER_ROWS
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04h/t_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith(('gn','lm'))})
    except Exception as x: print(f,'ERR',x)
PY
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/summary.txt
tail -5 $O/pytest.log
timeout 300 python3 tools/fuzz_parity.py --seconds 200 --seed 91 --shards 2 > $O/fuzz.json 2> $O/fuzz.err; echo "fuzz rc $?" >> $O/summary.txt
python3 -c "
import json; d=json.load(open('gpurun_out/r04h/fuzz.json')); print({k:d[k] for k in ('cases','n_fail','fails','worst','worst_sharded_vs_unsharded')})"
cat $O/summary.txt
