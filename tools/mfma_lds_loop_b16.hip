// mfma_lds_loop_b16.hip -- the inner loop of the f32 scan's bf16 path alone (brute_force.h, bf_b16_kernel<7, 4>): 8 waves read the
// same tile of 128 rows x 112 components (two bf16 pieces, rows of 240 bytes) from LDS, three v_mfma_f32_32x32x16_bf16 per
// fragment pair. Clocks per tile for several read-ahead depths and one or two query sets per wave.
// (development tool: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_loop_b16.hip -o tools/mfma_lds_loop_b16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
constexpr int KG = 7, R = 4;
constexpr uint32_t STRIDE_B = 2u * 16u * KG + 16u, ET = 32u * R;
template <int DEPTH, int SETS>
__global__ __launch_bounds__(512) void k(uint64_t* out, float* sink, int tiles) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* tile_hi = smem;
    uint8_t* tile_lo = smem + (size_t)ET * STRIDE_B;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, col = lane & 31u, h = lane >> 5;
    for (uint32_t i = tid; i < 2u * ET * STRIDE_B / 2u; i += 512u) reinterpret_cast<__bf16*>(smem)[i] = (__bf16)(float)((int)((i * 2654435761u) >> 20) - 2048) * (__bf16)0.001f;
    __syncthreads();
    b16x8 qh[SETS][KG], ql[SETS][KG];
    for (int s = 0; s < SETS; ++s)
        for (int g = 0; g < KG; ++g)
            for (int j = 0; j < 8; ++j) { qh[s][g][j] = (__bf16)(0.01f * (float)((tid * 7 + g * 13 + j * 3 + s) % 97)); ql[s][g][j] = (__bf16)(0.0001f * (float)((tid + j) % 31)); }
    float keep = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        f32x16 acc[SETS][R];
        for (int s = 0; s < SETS; ++s)
            for (int r = 0; r < R; ++r)
                for (int v = 0; v < 16; ++v) acc[s][r][v] = 0.f;
        const size_t off0 = (size_t)col * STRIDE_B + (size_t)h * 16u * KG;
        constexpr int N = KG * R;
        b16x8 ah[DEPTH + 1], al[DEPTH + 1];
#pragma unroll
        for (int i = 0; i < DEPTH && i < N; ++i) {
            const int g = i / R, r = i % R;
            ah[i % (DEPTH + 1)] = *reinterpret_cast<const b16x8*>(tile_hi + off0 + (size_t)r * 32u * STRIDE_B + (size_t)g * 16u);
            al[i % (DEPTH + 1)] = *reinterpret_cast<const b16x8*>(tile_lo + off0 + (size_t)r * 32u * STRIDE_B + (size_t)g * 16u);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int g = i / R, r = i % R;
            if (i + DEPTH < N) {
                const int gn = (i + DEPTH) / R, rn = (i + DEPTH) % R;
                ah[(i + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const b16x8*>(tile_hi + off0 + (size_t)rn * 32u * STRIDE_B + (size_t)gn * 16u);
                al[(i + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const b16x8*>(tile_lo + off0 + (size_t)rn * 32u * STRIDE_B + (size_t)gn * 16u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SETS; ++s) {
                acc[s][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i % (DEPTH + 1)], qh[s][g], acc[s][r], 0, 0, 0);
                acc[s][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i % (DEPTH + 1)], ql[s][g], acc[s][r], 0, 0, 0);
                acc[s][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i % (DEPTH + 1)], qh[s][g], acc[s][r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int s = 0; s < SETS; ++s)
            for (int r = 0; r < R; ++r) keep += acc[s][r][0] + acc[s][r][7] + acc[s][r][15];
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (keep == 12345.678f) sink[0] = keep;
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int DEPTH, int SETS>
void run(uint64_t* out, float* sink, const char* what) {
    const int tiles = 1200 / SETS;
    (void)hipFuncSetAttribute((const void*)k<DEPTH, SETS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<DEPTH, SETS>), dim3(256), dim3(512), 65536, 0, out, sink, tiles);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    uint64_t c;
    (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    const double n = 3.0 * KG * R * SETS;
    printf("%s: %.0f clocks per tile and wave (%d matrix instructions: %.1f each), kernel %.3f ms = %.2f GHz, %.0f TFLOP/s\n", what, (double)c / tiles, (int)n,
           (double)c / tiles / n, ms, (double)c / (ms * 1e6), n * 32768.0 * tiles * 8 * 256 / (ms * 1e9));
}
int main() {
    uint64_t* out; float* sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 64);
    run<1, 1>(out, sink, "one set, fragments one step ahead (the kernel's form)");
    run<2, 1>(out, sink, "one set, two steps ahead");
    run<3, 1>(out, sink, "one set, three steps ahead");
    run<1, 2>(out, sink, "two sets, one step ahead");
    run<2, 2>(out, sink, "two sets, two steps ahead");
    return 0;
}
