#!/bin/bash
# round-4 GPU pass I: single-launch groups A/B (CCAL_ITER_ROWS=0 = off)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04i; mkdir -p $O; rm -f $O/*.json
for v in on off; do
  if [ $v = off ]; then export CCAL_ITER_ROWS=0; else unset CCAL_ITER_ROWS; fi
  for f in 100 300 625 1000 2000; do
    timeout 200 python3 tools/time_kernels.py --frames $f --what solve --tag iter_$v > $O/t_${v}_$f.json 2>> $O/t.err
  done
  timeout 200 python3 tools/time_kernels.py --frames 625 --model kb4 --what solve --tag iter_$v > $O/t_${v}_kb4.json 2>> $O/t.err
  timeout 200 python3 tools/time_kernels.py --frames 625 --model opencv5 --what solve --tag iter_$v > $O/t_${v}_ocv5.json 2>> $O/t.err
done
unset CCAL_ITER_ROWS
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04i/t_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms')})
    except Exception as x: print(f,'ERR',x)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/time_kernels.py --frames 625 --what solve > $O/run.json 2> $O/run.err
python3 - <<PY
import csv,glob
f=glob.glob('$O/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=None
for r in rows[-16:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if t0 is None: t0=s; pe=s
    print(f"{(s-t0)/1e3:9.2f} us  dur {(e-s)/1e3:7.2f}  gap {(s-pe)/1e3:7.2f}  {r['Kernel_Name'][:70]}")
    pe=e
PY
rm -rf $O/trace
