# Developer tool: per-kernel times of mode E (rocprofv3 --kernel-trace --stats).  EXTRA="--cams 2" etc.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_eval
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eval -o s -- python3 $R/tools/time_kernels.py --what ${WHAT:-eval} --reps 50 ${EXTRA:-} > /dev/null 2>$R/gpurun_out/prof_eval.err
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/prof_eval/s_kernel_stats.csv")):
    if "ccal" in r["Name"]: print("%-90s calls %5s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
