// mfma_rate.hip -- cycles per matrix instruction on gfx950, one or two waves per SIMD, four independent accumulators
// (development tool: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(uint64_t* out, int* sink, int iters) {
    i32x16 acc[4];
    f32x16 facc[4];
    for (int i = 0; i < 4; ++i)
        for (int v = 0; v < 16; ++v) { acc[i][v] = 0; facc[i][v] = 0.f; }
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    b16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(float)(threadIdx.x + i); hb[i] = (__bf16)1.0f; }
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
                if (KIND == 1) facc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, facc[i], 0, 0, 0);
                if (KIND == 2) { i32x4v c = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]}; c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); acc[i][0] = c[0]; acc[i][1] = c[1]; acc[i][2] = c[2]; acc[i][3] = c[3]; }
            }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][3] + (int)facc[i][5];
    if (s == 0x7fffffff) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    uint64_t* out; int* sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 64);
    const int iters = 4096;
    const char* names[3] = {"v_mfma_i32_32x32x32_i8", "v_mfma_f32_32x32x16_bf16", "v_mfma_i32_16x16x64_i8"};
    for (int kind = 0; kind < 3; ++kind)
        for (int threads : {256, 512}) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            uint64_t c; (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
            const double ops_per = kind == 0 ? 65536.0 : 32768.0, total = ops_per * iters * 16.0 * (threads / 64) * 256;
            printf("%s, %d waves per SIMD: %.1f clocks per instruction and wave, %.1f per SIMD; kernel %.3f ms = %.2f GHz, %.0f T(FL)OP/s on the whole chip (operands: small integers / constants)\n",
                   names[kind], threads / 256, (double)c / (iters * 16.0), (double)c / (iters * 16.0) / (threads / 256), ms, (double)c / (ms * 1e6), total / (ms * 1e9));
        }
    return 0;
}
