cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 -L > $R/gpurun_out/counters_all.txt 2>&1
grep -o -E "\b(TCP|TCC|TCA|UTCL|GRBM|TA|TD|SQC|CPC|GL2|ATC|VM)[A-Z0-9_]*" $R/gpurun_out/counters_all.txt | sort -u | tr '\n' ' ' > $R/gpurun_out/counters_mem.txt
