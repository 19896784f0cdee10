#!/bin/bash
# Developer tool: rocprofv3 kernel table (name, calls, average ns) of one run of tools/time_kernels.py with the given arguments.
#   tools/kstats.sh --what normal --cams 2 --reps 50
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_out -o ks -- python3 $REPO/tools/time_kernels.py "$@" > /tmp/kstats_run.json 2>/tmp/kstats_err.txt
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kstats_out/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"]):10.0f}')
PY
tail -1 /tmp/kstats_run.json
