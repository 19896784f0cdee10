#!/bin/bash
# Round 6, call g: bins for rigs (k_gram2g): parity + A/B; host sanitizer builds over the round's host code (block cache churn, plan, pinned arrays)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06g; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_normal.py tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
{
echo "== two cameras, ragged 24..144 corners: bins (base) against the same library without them (nobins)"
python tools/ab_build.py "base,nobins@nobins" eucm,kb4 10000 3 --ragged --cams 2
python tools/ab_build.py "base,nobins@nobins" eucm 3000,30000 3 --ragged --cams 2
echo "== one camera, KB4 / OPENCV5, larger ragged problems"
python tools/ab_build.py "base,nobins@nobins" kb4,opencv5 20000,50000 3 --ragged
} > $O/ab_bins_rigs.txt 2>&1
cat $O/ab_bins_rigs.txt
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
tail -3 $O/tsan.out; tail -3 $O/asan.out; head -c 2500 $O/tsan.err; head -c 1500 $O/asan.err
