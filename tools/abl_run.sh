cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in abl_W abl_R; do
  if [ -z "$v" ]; then unset CCAL_LIB; else export CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_$v.so; fi
  rm -rf $R/gpurun_out/abl_tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_tmp -o s -- python3 $R/tools/time_kernels.py --what solve > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/abl_tmp/s_kernel_stats.csv")):
    if "k_gram1" in r["Name"]: print("variant ${v:-base}: k_gram1 max %.1f us avg %.1f us" % (float(r["MaxNs"])/1e3, float(r["AverageNs"])/1e3))
PY
done
