#!/bin/bash
# Round 5, call l: host ThreadSanitizer and AddressSanitizer builds (tools/build_hosttsan.sh / build_hostasan.sh) of the final host code
# driven by tools/tsan_drive.py (lockstep batch groups next to rigs on helper threads, three shards of one GPU with the v2 transport,
# the one-shot multi entry points).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05l; mkdir -p $O
TSAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
LD_PRELOAD=$TSAN_RT TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan.supp exitcode=0" \
  CCAL_LIB=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hosttsan.so timeout 900 python tools/tsan_drive.py > $O/tsan.out 2> $O/tsan.err
echo "tsan rc $?" >> $O/tsan.out
grep -c "WARNING: ThreadSanitizer" $O/tsan.err >> $O/tsan.out
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 \
  CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hostasan.so timeout 600 python tools/tsan_drive.py > $O/asan.out 2> $O/asan.err
echo "asan rc $?" >> $O/asan.out
grep -c "ERROR: AddressSanitizer" $O/asan.err >> $O/asan.out
tail -3 $O/tsan.out; tail -3 $O/asan.out; head -c 3000 $O/tsan.err; head -c 1500 $O/asan.err
# the chip's streaming ceilings as a trivial kernel sees them (tools/ubench/hbm_stream.hip)
timeout 300 tools/ubench/hbm_stream.bin > $O/hbm_stream.txt 2>&1; tail -60 $O/hbm_stream.txt
