cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solve -o s -- python3 $R/tools/time_kernels.py --what solve > /dev/null 2>&1
cat $R/gpurun_out/prof_solve/s_kernel_stats.csv | cut -c1-160
