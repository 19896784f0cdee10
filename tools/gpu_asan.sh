#!/bin/bash
# The library's own host threads and sharding code against the host-AddressSanitizer build (tools/build_hostasan.sh; device code
# untouched): ccal_solve_batch over five contexts, ccal_multi_solve over three shards, in ONE process that leaves through os._exit
# (the sanitizer runtime trips over the HIP runtime's teardown order at interpreter exit - and a spawned child then never exits:
# tests/test_gpu_multi.py's children are not run under it).  Then the single-process GPU test files.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0
export CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hostasan.so
timeout 300 python tools/tsan_drive.py > gpurun_out/asan_drive.out 2> gpurun_out/asan_drive.err; echo "drive rc $?" >> gpurun_out/asan_drive.out
grep -c "ERROR: AddressSanitizer" gpurun_out/asan_drive.err >> gpurun_out/asan_drive.out
tail -3 gpurun_out/asan_drive.out
