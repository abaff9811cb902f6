// Development harness (not part of the library): instantiates a few walker kernels alone so that their ISA can be read
// and counted in seconds instead of the library's two minutes.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//     -fno-fast-math -DGRANNE_HIP_USE_DPP=1 -I include --cuda-device-only -S tools/dev_one_kernel.hip -o /tmp/dev/one.s
#include <hip/hip_runtime.h>
#include "../granne_amd/csrc/search_kernel.h"
#include "../granne_amd/csrc/walk_fast.h"
using namespace granne_hip;
#ifndef DEV_S
#define DEV_S 1
#endif
template __global__ void granne_hip::fast_kernel<DT_F32, 100, DEV_S, false, 3, false>(const SlowParams);
template __global__ void granne_hip::fast_kernel<DT_I8, 0, DEV_S, false, 3, false>(const SlowParams);
