#!/bin/bash
# round-4 GPU pass E: does rocprofv3 still crash on launches from the library's worker threads?  (stand-alone shape first, then
# the library's own batch / sharded paths), every run under its own timeout
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04e; mkdir -p $O
: > $O/summary.txt
for i in 1 2 3; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/ut_stats_$i -o p -- $R/tools/ubench/prof_threads.bin 4 2000 > $O/ut_stats_$i.out 2> $O/ut_stats_$i.err; echo "ubench stats $i rc $? $(tail -1 $O/ut_stats_$i.out)" >> $O/summary.txt
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU -d $O/ut_pmc_$i -o p -- $R/tools/ubench/prof_threads.bin 4 2000 > $O/ut_pmc_$i.out 2> $O/ut_pmc_$i.err; echo "ubench pmc $i rc $? $(tail -1 $O/ut_pmc_$i.out)" >> $O/summary.txt
done
for i in 1 2 3; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/lib_stats_$i -o p -- python3 $R/tools/tsan_drive.py > $O/lib_stats_$i.out 2> $O/lib_stats_$i.err; echo "library stats $i rc $? $(tail -1 $O/lib_stats_$i.out)" >> $O/summary.txt
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU -d $O/lib_pmc_$i -o p -- python3 $R/tools/tsan_drive.py > $O/lib_pmc_$i.out 2> $O/lib_pmc_$i.err; echo "library pmc $i rc $? $(tail -1 $O/lib_pmc_$i.out)" >> $O/summary.txt
done
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt
