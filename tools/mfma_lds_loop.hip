// mfma_lds_loop.hip -- the scan's inner loop alone (brute_force.h, bf_i8_ring_kernel): 8 waves of a block read the same
// 16 KB tile from LDS (ds_read_b128, swizzled) and feed v_mfma_i32_32x32x32_i8, two query sets per wave. Clocks per tile
// for several read-ahead depths (development tool: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_loop.hip -o tools/mfma_lds_loop)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(uint32_t lds_dst, const uint8_t* base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
typedef volatile __attribute__((address_space(3))) uint32_t* lds_u32;
template <int DEPTH, int SETS, int MODE>
__global__ __launch_bounds__(512) void k(uint64_t* out, int* sink, int tiles, const uint8_t* rows) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, col = lane & 31u, h = lane >> 5;
    for (uint32_t i = tid; i < 4u * 16384u / 4u; i += 512u) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    i32x4 qr[SETS][4];
    for (int s = 0; s < SETS; ++s)
        for (int g = 0; g < 4; ++g) qr[s][g] = i32x4{(int)(tid * 2654435761u + s), (int)(g * 40503u + tid * 7919u), (int)(tid * 2246822519u), (int)(~tid * 3266489917u)};
    uint32_t aoff[4];
    for (int g = 0; g < 4; ++g) aoff[g] = col * 128u + (((2u * (uint32_t)g + h) ^ ((col >> 1) & 7u)) << 4);
    int keep = 0;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t drow = wave * 16u + (lane >> 3), dslot = lane & 7u;
    const uint32_t v0 = drow * 128u + (dslot ^ ((drow >> 1) & 7u)) * 16u, v1 = (drow + 8u) * 128u + (dslot ^ (((drow + 8u) >> 1) & 7u)) * 16u;
    const uint32_t pairid = (blockIdx.x / 16u) * 8u + (blockIdx.x & 7u); // blocks b and b + 8 (one XCD) stream the same rows
    const uint8_t* mybase = rows + (size_t)pairid * (size_t)tiles * 16384u;
    auto issue = [&](int t) {
        const uint8_t* base = mybase + (size_t)t * 16384u;
        const uint32_t dst = lds0 + (uint32_t)(t & 3) * 16384u + wave * 2048u;
        dma16(dst, base, v0);
        dma16(dst + 1024u, base, v1);
    };
    lds_u32 landed = (lds_u32)(smem + 65536), done = landed + 4;
    if (tid < 8) landed[tid] = 0;
    __syncthreads();
    auto spin = [&](lds_u32 c, uint32_t target) { while (*c < target) __builtin_amdgcn_s_sleep(1); asm volatile("" ::: "memory"); };
    auto signal = [&](lds_u32 c) { if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    uint64_t t0 = __builtin_readcyclecounter();
    if (MODE >= 1) for (int t = 0; t < 3; ++t) issue(t);
    for (int t = 0; t < tiles; ++t) {
        if (MODE == 1) {
            if (tiles - 1 - t >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 3 < tiles) issue(t + 3);
        }
        if (MODE == 2) { // counters, slack: landed signalled at the top of the tile itself (4 slots, 2 ahead)
            if (tiles - 1 - t >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            signal(landed + (t & 3));
            if (t + 3 < tiles) { if (t >= 1) spin(done + ((t + 3) & 3), 8u * (uint32_t)((t + 3) / 4)); issue(t + 3); }
            spin(landed + (t & 3), 8u * (uint32_t)(t / 4 + 1));
        }
        const uint8_t* tile = smem + (uint32_t)(t & 3) * 16384u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            i32x16 acc[SETS][2];
            for (int s = 0; s < SETS; ++s)
                for (int r = 0; r < 2; ++r)
                    for (int v = 0; v < 16; ++v) acc[s][r][v] = 0;
            i32x4 afrag[4][2];
            // DEPTH g-steps of fragments are on their way ahead of the matrix instructions
#pragma unroll
            for (int g = 0; g < DEPTH && g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 2; ++r) afrag[g][r] = *reinterpret_cast<const i32x4*>(tile + aoff[g] + (uint32_t)(p * 2 + r) * 4096u);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + DEPTH < 4) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) afrag[g + DEPTH][r] = *reinterpret_cast<const i32x4*>(tile + aoff[g + DEPTH] + (uint32_t)(p * 2 + r) * 4096u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < SETS; ++s) acc[s][r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[g][r], qr[s][g], acc[s][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int s = 0; s < SETS; ++s)
                for (int r = 0; r < 2; ++r) keep ^= acc[s][r][0] ^ acc[s][r][7] ^ acc[s][r][15];
        }
        if (MODE == 2) { asm volatile("" ::: "memory"); signal(done + (t & 3)); }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (keep == 0x12345678) sink[0] = keep;
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int DEPTH, int SETS, int MODE>
void run(uint64_t* out, int* sink, const char* what, const uint8_t* rows) {
    const int tiles = 600;
    (void)hipFuncSetAttribute((const void*)k<DEPTH, SETS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<DEPTH, SETS, MODE>), dim3(256), dim3(512), 65536 + 64, 0, out, sink, tiles, rows);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    uint64_t c;
    (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("%s: %.0f clocks per tile and wave (%d matrix instructions: %.1f each), kernel %.3f ms = %.2f GHz\n", what, (double)c / tiles, 16 * SETS, (double)c / tiles / (16 * SETS), ms, (double)c / (ms * 1e6));
}
int main() {
    uint64_t* out; int* sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 64);
    uint8_t* rows; (void)hipMalloc(&rows, (size_t)256 * 600 * 16384); {
        std::vector<uint8_t> hr((size_t)128 * 600 * 16384);
        uint32_t x = 12345;
        for (auto& b : hr) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); }
        (void)hipMemcpy(rows, hr.data(), hr.size(), hipMemcpyHostToDevice);
    }
    run<1, 2, 0>(out, sink, "loop only", rows);
    run<1, 2, 1>(out, sink, "+ tiles by LDS-DMA (3 ahead), one barrier per tile", rows);
    run<1, 2, 2>(out, sink, "+ tiles by LDS-DMA, landed / done counters instead of the barrier", rows);
    return 0;
}
