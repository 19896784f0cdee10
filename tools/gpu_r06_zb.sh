#!/bin/bash
# Round 6, call zb: the final library (planner fitted to three sweeps, result spread): whole GPU suite, smoke, small-size builds, sanitizers, the profile set
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06zb; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; grep -n "passed\|failed" $O/pytest_full.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"
python tools/ab_build.py "shipped" eucm,ucm 2000,2500,3000,4000,5000,6000,8000,10000,16000 3 --ragged > $O/ab_g2_shipped_final.txt 2>&1; cat $O/ab_g2_shipped_final.txt
TSAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
LD_PRELOAD=$TSAN_RT TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan.supp exitcode=0" \
  CCAL_LIB=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hosttsan.so timeout 900 python tools/tsan_drive.py > $O/tsan.out 2> $O/tsan.err
echo "tsan rc $?" >> $O/tsan.out
grep -c "WARNING: ThreadSanitizer" $O/tsan.err >> $O/tsan.out
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 \
  CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hostasan.so timeout 900 python tools/tsan_drive.py > $O/asan.out 2> $O/asan.err
echo "asan rc $?" >> $O/asan.out
grep -c "ERROR: AddressSanitizer" $O/asan.err >> $O/asan.out
tail -3 $O/tsan.out; tail -3 $O/asan.out; head -c 1500 $O/tsan.err; head -c 1500 $O/asan.err
bash profiles/run_profile.sh r06 > $O/run_profile.log 2>&1; tail -2 $O/run_profile.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06/stats_ragged -o stats -- python3 $R/tools/time_kernels.py --what eval,normal,solve --ragged --reps 50 > $R/gpurun_out/prof_r06/ragged.json 2> $R/gpurun_out/prof_r06/stats_ragged.err
cd $R
find gpurun_out/prof_r06 -name "*.csv" -size +20M -delete
tail -c 1500 gpurun_out/prof_r06/bench_full.json
