#!/bin/bash
# round-4 GPU pass A: whole GPU suite, default bench line, host-ThreadSanitizer drive of the library's own threads
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04a
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r04a/pytest.log 2>&1
(time python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err) 2> gpurun_out/r04a/bench.time
TSAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
LD_PRELOAD=$TSAN_RT TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan.supp exitcode=0" \
  CCAL_LIB=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hosttsan.so timeout 600 python tools/tsan_drive.py > gpurun_out/r04a/tsan.out 2> gpurun_out/r04a/tsan.err
echo "tsan rc $?" >> gpurun_out/r04a/tsan.out
grep -c "WARNING: ThreadSanitizer" gpurun_out/r04a/tsan.err >> gpurun_out/r04a/tsan.out
tail -5 gpurun_out/r04a/pytest.log; tail -3 gpurun_out/r04a/tsan.out; head -c 600 gpurun_out/r04a/bench.json
