#!/bin/bash
# Developer tool: rocprofv3 --kernel-trace --stats of one time_kernels.py run, printed as a short table.
#   tools/kstats.sh <tag> <time_kernels.py arguments...>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/ks_$tag -o stats -- python3 $R/tools/time_kernels.py "$@" > $R/gpurun_out/r03/ks_$tag.json 2> $R/gpurun_out/r03/ks_$tag.err
python3 - <<PY
import csv
f="$R/gpurun_out/r03/ks_$tag/stats_kernel_stats.csv"
for r in csv.DictReader(open(f)):
    n=r["Name"].replace("void ccal::","")
    print(n[:64].ljust(64), r["Calls"].rjust(6), "%9.1f us avg" % (float(r["AverageNs"])/1e3), "%5.1f %%" % float(r["Percentage"]))
PY
tail -1 $R/gpurun_out/r03/ks_$tag.json
