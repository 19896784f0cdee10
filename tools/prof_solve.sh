cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${1:-10000}
EXTRA="${2:-}"
NTAIL=${3:-34}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solve -o s -- python3 $R/tools/time_kernels.py --what ${WHAT:-solve} --frames $F $EXTRA > /dev/null 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_solve/s_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'].split('(')[0].replace('void ccal::','').replace('ccal::','') for r in rows]
# last GN solve = find last occurrence of 3+ consecutive groups; print the last 30 kernels with durations
prev=None
for r,n in list(zip(rows,names))[-$NTAIL:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{n:28s} start+{(s-prev)/1e3 if prev else 0:7.1f} us dur {(e-s)/1e3:6.1f} us"); prev=e
PY
