#!/bin/bash
# round-4 GPU pass M: the general (multi-camera) loop at session size: solve times + kernel timeline of one solve
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04m; mkdir -p $O
for c in 2 3; do timeout 200 python3 tools/time_kernels.py --frames 600 --cams $c --what normal,solve | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($c, 'cams x 600 frames: build', round(d['normal_us'],1), 'us  gn', d['gn_ms'], d['gn_iters'], ' lm', d['lm_ms'], d['lm_iters'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/time_kernels.py --frames 600 --cams 2 --what solve > $O/run.json 2> $O/run.err
python3 - <<PY
import csv,glob
f=glob.glob('$O/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
import statistics, collections
by=collections.defaultdict(list)
for r in rows: by[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in by.items():
    big=[x for x in v if x>5.5] or v
    print(f"   median {statistics.median(big):6.2f} us  (n {len(big):3d} of {len(v)})  {k}")
t0=None
for r in rows[-34:-12]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if t0 is None: t0=s; pe=s
    print(f"{(s-t0)/1e3:9.2f} us  dur {(e-s)/1e3:7.2f}  gap {(s-pe)/1e3:7.2f}  {r['Kernel_Name'][:80]}")
    pe=e
PY
rm -rf $O/trace
