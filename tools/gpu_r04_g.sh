#!/bin/bash
# round-4 GPU pass G: timeline of a session-size (625-frame) solve: kernel start / end stamps of every dispatch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04g; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/time_kernels.py --frames 625 --what solve > $O/run.json 2> $O/run.err
echo "rc $?"
python3 - <<PY
import csv,glob
f=glob.glob('$O/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last 3 solves: print the last 60 dispatches
t0=None
out=open('$O/timeline.txt','w')
for r in rows[-70:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if t0 is None: t0=s; pe=s
    print(f"{(s-t0)/1e3:9.2f} us  dur {(e-s)/1e3:7.2f}  gap {(s-pe)/1e3:7.2f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size','')} wg {r.get('Workgroup_Size','')}", file=out)
    pe=e
out.close()
print(open('$O/timeline.txt').read())
PY
rm -rf $O/trace
