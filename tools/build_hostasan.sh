#!/bin/bash
# Developer tool: the WHOLE library with AddressSanitizer on the host side (-fsanitize=address -fno-gpu-sanitize, -O1 -g; the
# device code is untouched by the sanitizer and runs on the GPU as usual - no GPU sanitizer is involved) into
# lib/variants/libccal_hostasan.so.  Run the GPU tests against it with
#   ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
#   LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 \
#   CCAL_LIB=$PWD/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_hostasan.so python -m pytest tests -m gpu -q
# (round 3: 169 tests of the batch / normal / boundary / api / eval / init files, no report - and, being a second compilation of
# every kernel at -O1, it is what exposed the miscompiled pointer select of the elimination kernels)
set -e
cd "$(dirname "$0")/../camera_intrinsic_calibration_rs_amd/csrc"
mkdir -p build/hostasan ../lib/variants
for f in ccal_*.hip; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fsanitize=address -fno-gpu-sanitize -c $f -o build/hostasan/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -o ../lib/variants/libccal_hostasan.so build/hostasan/*.o -ldl -lpthread
echo built ../lib/variants/libccal_hostasan.so
