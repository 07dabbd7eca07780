#!/bin/bash
# A build of one translation unit with extra flags (e.g. the diagnostic stamps): scripts/build_variant.sh NAME "-DCATTUS_STAMPS" [kernels_wino4]
# -> cattus_amd/libcattus_hip_NAME.so (selected at run time with CATTUS_HIP_LIB), the other objects from the regular build.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; tu=${3:-kernels_wino}
python -c "from cattus_amd import build; build.build_hip()"
mkdir -p /tmp/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude \
  $flags -c cattus_amd/csrc/$tu.hip -o /tmp/variants/${tu}_$name.o
objs=""
for s in kernels kernels_t64s kernels_wino kernels_wino4 kernels_wino8 evaluator; do
  if [ $s = $tu ]; then objs="$objs /tmp/variants/${tu}_$name.o"; else objs="$objs cattus_amd/build/hip/$s.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o cattus_amd/libcattus_hip_$name.so $objs -lpthread -ldl
echo built cattus_amd/libcattus_hip_$name.so
