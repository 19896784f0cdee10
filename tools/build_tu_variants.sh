#!/bin/bash
# Developer tool: A/B variants that differ in ONE translation unit: compile that file with the given macro overrides and
# link it with the base build's other objects into lib/variants/libccal_<name>.so.
#   tools/build_tu_variants.sh ccal_kernels_gram2 "minw1:-DCCAL_GRAM2_MINW=1" "nla8:-DCCAL_G2_NLA(M)=8"
set -e
cd "$(dirname "$0")/../camera_intrinsic_calibration_rs_amd/csrc"
tu=$1; shift
make -s -j8 >/dev/null
mkdir -p ../lib/variants build/var
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast $flags -c $tu.hip -o build/var/${name}_$tu.o
    objs=$(ls build/ccal_*.o | grep -v "/$tu.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libccal_$name.so $objs build/var/${name}_$tu.o -ldl -lpthread
    echo built $name ) &
done
wait
