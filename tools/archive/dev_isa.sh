#!/bin/bash
# tools/dev_isa.sh [extra flags]: ISA of the harness kernels -> /tmp/dev/one.s, with the resource lines
cd "$(dirname "$0")/.."
mkdir -p /tmp/dev
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -DGRANNE_HIP_USE_DPP=1 -I include --cuda-device-only -S -Wno-unused-command-line-argument "$@" \
  tools/dev_one_kernel.hip -o /tmp/dev/one.s || exit 1
grep -E "^\s+\.(vgpr_count|sgpr_spill_count|vgpr_spill_count|name):" /tmp/dev/one.s | paste - - - - | sed 's/ \+/ /g'
