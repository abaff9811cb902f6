#!/bin/bash
# Builds a variant of the library beside the shipped one: tools/build_variant.sh <name> [extra hipcc flags...]
# -> granne_amd/lib/libgranne_hip_<name>.so (use with GRANNE_HIP_LIB=...; kernel experiments and diagnostics)
cd "$(dirname "$0")/.."
NAME=$1; shift
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fPIC -shared -Wall -Wno-unused-function -DGRANNE_HIP_USE_DPP=1 "$@" -I include \
  granne_amd/csrc/granne_hip.hip -o granne_amd/lib/libgranne_hip_$NAME.so
