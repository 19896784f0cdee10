#!/bin/bash
# Developer tool: the WHOLE library with ThreadSanitizer on the host side (-fsanitize=thread -fno-gpu-sanitize, -O1 -g; device
# code untouched, runs on the GPU as usual - no GPU sanitizer is involved) into lib/variants/libccal_hosttsan.so.
# What it watches: the host threads the library owns - ccal_solve_batch's per-context workers, ccal_solve_sharded's one thread
# per shard, the in-process transport's barrier, the dynamic-LDS guard (csrc/ccal_internal.hpp) - while the batch / multi tests run:
#   TSAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
#   LD_PRELOAD=$TSAN_RT TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan.supp" \
#   CCAL_LIB=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hosttsan.so python tools/tsan_drive.py
# (the HIP runtime itself is not instrumented: reports whose stacks lie entirely inside libamdhip64 / libhsa-runtime64 are
# suppressed by tools/tsan.supp; a report with a ccal frame in it is a finding)
set -e
cd "$(dirname "$0")/../camera_intrinsic_calibration_rs_amd/csrc"
mkdir -p build/hosttsan ../lib/variants
for f in ccal_*.hip; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fsanitize=thread -fno-gpu-sanitize -c $f -o build/hosttsan/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -fno-gpu-sanitize -o ../lib/variants/libccal_hosttsan.so build/hosttsan/*.o -ldl -lpthread
echo built ../lib/variants/libccal_hosttsan.so
