#!/bin/bash
# Developer tool: build A/B variants of the engine (macro overrides) into lib/variants/.
set -e
cd "$(dirname "$0")/../camera_intrinsic_calibration_rs_amd/csrc"
mkdir -p ../lib/variants build/var
build() { name=$1; shift
  for f in ccal_*.hip; do b=${f%.hip}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast "$@" -c $f -o build/var/${name}_$b.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libccal_$name.so build/var/${name}_ccal_*.o -ldl
}
for spec in "$@"; do name=${spec%%:*}; flags=${spec#*:}; build $name $flags; echo built $name; done
