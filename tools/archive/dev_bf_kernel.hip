// Development harness (not part of the library): the scan kernels alone, for their ISA and register counts in seconds.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//     -fno-fast-math -DGRANNE_HIP_USE_DPP=1 -I include --cuda-device-only -S tools/dev_bf_kernel.hip -o /tmp/dev/bf.s
#include <hip/hip_runtime.h>
#include "../granne_amd/csrc/search_kernel.h"
#include "../granne_amd/csrc/brute_force.h"
using namespace granne_hip;
template __global__ void granne_hip::bf_i8_kernel<4, false>(const BruteParams);
template __global__ void granne_hip::bf_i8_kernel<4, true>(const BruteParams);
template __global__ void granne_hip::bf_f32_kernel<52, 4, false>(const BruteParams);
template __global__ void granne_hip::bf_f32_kernel<100, 2, false>(const BruteParams);
template __global__ void granne_hip::bf_f32_kernel<128, 1, false>(const BruteParams);
template __global__ void granne_hip::bf_b16_kernel<7, 4, false>(const BruteParams);
// (bf_i8_ring_kernel is not a template: it is emitted with this file as it is)
