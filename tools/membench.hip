// membench.hip -- what the memory system gives the walker's access pattern on this chip.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench && tools/membench
// 1. burst: a wavefront gathers 32 random rows the way walk_fast.h does (lane = 2R+h, `NL` dwordx4 per lane),
//    waits for all of them, repeats. Reported: cycles per burst (issue -> all data back) and the aggregate
//    rate, for 1 wave, one wave per CU, one per SIMD (= one batch of 1024 queries), three and eight per SIMD.
//    int8 rows: 128-byte stride, 4 loads; f32 100-d rows: 400- and 512-byte stride, 13 loads.
// 2. chain: one lane chases pointers through the same buffer: latency of a dependent load, by footprint
//    (HBM + TLB misses at 5 GB, Infinity Cache at 128 MB, L2 at 2 MB), and of a second load from the same
//    512-byte row straight after the first (the "adjacency in the row's fourth line" case).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <int NL>
__global__ __launch_bounds__(64) void burst_kernel(const uint8_t* base, uint64_t n_rows, uint32_t stride, uint32_t iters,
                                                   uint64_t* out_cycles, uint32_t* out_sink) {
    const uint32_t lane = threadIdx.x, R = lane >> 1, h = lane & 1u;
    uint64_t cycles = 0;
    uint32_t sink = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t row = mix(((uint64_t)blockIdx.x << 32) ^ ((uint64_t)it << 8) ^ R ^ sink) % n_rows;
        const uint8_t* p = base + row * stride + h * 64u;
        uint4 v[NL];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const uint32_t off = (k < 12) ? (uint32_t)(k >> 2) * 128u + (uint32_t)(k & 3) * 16u : 384u - h * 64u;
            v[k] = *reinterpret_cast<const uint4*>(p + off);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        cycles += t1 - t0;
        sink = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            asm volatile("" ::"v"(v[k].x), "v"(v[k].y), "v"(v[k].z), "v"(v[k].w));
            sink |= v[k].x - 0x01010101u; // the buffer is filled with 0x01: a real dependence on the data, value 0
        }
    }
    if (lane == 0) {
        out_cycles[blockIdx.x] = cycles;
        out_sink[blockIdx.x] = sink;
    }
}

// one lane: next = hash(previous data); optionally a second dependent load from the same row (+384 bytes)
__global__ void chain_kernel(const uint8_t* base, uint64_t n_rows, uint32_t stride, uint32_t iters, int second,
                             uint64_t seed, uint64_t* out) {
    uint64_t first = 0, sec = 0;
    uint32_t carry = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t row = mix(seed ^ ((uint64_t)it << 8) ^ carry) % n_rows;
        const uint8_t* p = base + row * stride;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        uint32_t a = *reinterpret_cast<const volatile uint32_t*>(p);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        first += t1 - t0;
        carry = a - 0x01010101u;
        if (second) {
            const uint8_t* q = p + 384u + carry;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const uint64_t t2 = __builtin_amdgcn_s_memtime();
            uint32_t b = *reinterpret_cast<const volatile uint32_t*>(q);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint64_t t3 = __builtin_amdgcn_s_memtime();
            sec += t3 - t2;
            carry |= b - 0x01010101u;
        }
    }
    out[0] = first;
    out[1] = sec;
    out[2] = carry;
}

__global__ void touch_kernel(const uint8_t* base, uint64_t bytes, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t off = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u; off < bytes; off += (uint64_t)gridDim.x * blockDim.x * 16u)
        acc |= reinterpret_cast<const uint4*>(base + off)->x;
    if (acc == 0x12345678u) out[0] = acc;
}

// round 6: a wavefront reads ONE contiguous record of `lines` 128-byte lines at a random place (what an expansion would
// read if a node's record held its neighbors' rows themselves: 25 lines for 30 int8 rows of 100 bytes + the ids, 95 for
// 30 f32 rows of 400 bytes + the ids), 16 bytes per lane and load, all loads in flight, waits, repeats.
template <int NL>
__global__ __launch_bounds__(64) void record_kernel(const uint8_t* base, uint64_t n_records, uint32_t record_bytes, uint32_t lines,
                                                    uint32_t iters, uint64_t* out_cycles, uint32_t* out_sink) {
    const uint32_t lane = threadIdx.x;
    uint64_t cycles = 0;
    uint32_t sink = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t rec = mix(((uint64_t)blockIdx.x << 32) ^ ((uint64_t)it << 8) ^ sink) % n_records;
        const uint8_t* p = base + rec * record_bytes;
        uint4 v[NL];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            uint32_t unit = (uint32_t)k * 64u + lane;
            unit = unit < lines * 8u ? unit : lines * 8u - 1u;
            v[k] = *reinterpret_cast<const uint4*>(p + (size_t)unit * 16u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cycles += __builtin_amdgcn_s_memtime() - t0;
        sink = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            asm volatile("" ::"v"(v[k].x), "v"(v[k].y), "v"(v[k].z), "v"(v[k].w));
            sink |= v[k].x - 0x01010101u;
        }
    }
    if (lane == 0) {
        out_cycles[blockIdx.x] = cycles;
        out_sink[blockIdx.x] = sink;
    }
}

template <int NL>
static void run_record(const char* name, const uint8_t* d, uint64_t n_records, uint32_t record_bytes, uint32_t lines, uint64_t* d_cyc,
                       uint32_t* d_sink) {
    const uint32_t iters = 64;
    const int waves[] = {1024, 3072, 8192};
    for (int w : waves) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(record_kernel<NL>, dim3(w), dim3(64), 0, 0, d, n_records, record_bytes, lines, 4u, d_cyc, d_sink);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(record_kernel<NL>, dim3(w), dim3(64), 0, 0, d, n_records, record_bytes, lines, iters, d_cyc, d_sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s waves %5d: kernel %8.3f ms | %7.1f GB/s of 128-byte lines\n", name, w, ms,
               (double)lines * iters * w * 128.0 / (ms * 1e-3) / 1e9);
    }
}

template <int NL>
static void run_burst(const char* name, const uint8_t* d, uint64_t n_rows, uint32_t stride, uint64_t* d_cyc, uint32_t* d_sink) {
    const uint32_t iters = 64;
    const int waves[] = {1, 256, 1024, 3072, 8192};
    for (int w : waves) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(burst_kernel<NL>, dim3(w), dim3(64), 0, 0, d, n_rows, stride, 4u, d_cyc, d_sink); // warm
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(burst_kernel<NL>, dim3(w), dim3(64), 0, 0, d, n_rows, stride, iters, d_cyc, d_sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> cyc(w);
        CK(hipMemcpy(cyc.data(), d_cyc, (size_t)w * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto c : cyc) sum += (double)c;
        const double lines = (NL == 4 ? 1.0 : NL == 12 ? 3.0 : 4.0) * 32.0 * iters * w; // 128-byte lines touched
        printf("%-22s waves %5d: %8.0f cycles per burst | kernel %8.3f ms | %7.1f GB/s of 128-byte lines\n", name, w,
               sum / w / iters, ms, lines * 128.0 / (ms * 1e-3) / 1e9);
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 6) { // round 6: the ceilings of the layouts (3-line rows; 16 GB footprints; whole records)
        const uint64_t big = 125000000ull; // rows of a C5 shard: 16 GB of 128-byte rows
        uint8_t* d = nullptr;
        CK(hipMalloc((void**)&d, big * 128ull + 65536));
        CK(hipMemset(d, 1, big * 128ull + 65536));
        uint64_t* d_cyc;
        uint32_t* d_sink;
        CK(hipMalloc((void**)&d_cyc, 8192 * 8));
        CK(hipMalloc((void**)&d_sink, 8192 * 4));
        run_burst<4>("i8 rows, stride 128, 1.3 GB", d, 10000000ull, 128, d_cyc, d_sink);
        run_burst<4>("i8 rows, stride 128, 16 GB", d, big, 128, d_cyc, d_sink);
        run_burst<12>("f32 3 of 4 lines, stride 512, 5 GB", d, 10000000ull, 512, d_cyc, d_sink);
        run_burst<13>("f32 rows, stride 400, 4 GB", d, 10000000ull, 400, d_cyc, d_sink);
        run_record<4>("int8 record 3200 B (25 lines), 16 GB", d, big * 128ull / 3200ull, 3200, 25, d_cyc, d_sink);
        run_record<12>("f32 record 12160 B (95 lines), 16 GB", d, big * 128ull / 12160ull, 12160, 95, d_cyc, d_sink);
        run_record<5>("f32 record 5 lines (ids + tails), 6 GB", d, 10000000ull, 640, 5, d_cyc, d_sink);
        return 0;
    }
    const uint64_t n_rows = 10000000ull;
    const size_t bytes = n_rows * 512ull + 4096;
    uint8_t* d = nullptr;
    CK(hipMalloc((void**)&d, bytes));
    CK(hipMemset(d, 1, bytes));
    uint64_t* d_cyc;
    uint32_t* d_sink;
    CK(hipMalloc((void**)&d_cyc, 8192 * 8));
    CK(hipMalloc((void**)&d_sink, 8192 * 4));
    run_burst<4>("i8 rows, stride 128", d, n_rows, 128, d_cyc, d_sink);
    run_burst<13>("f32 rows, stride 400", d, n_rows, 400, d_cyc, d_sink);
    run_burst<13>("f32 rows, stride 512", d, n_rows, 512, d_cyc, d_sink);

    uint64_t* d_out;
    CK(hipMalloc((void**)&d_out, 64));
    struct { const char* name; uint64_t rows; uint32_t stride; int second; } chains[] = {
        {"chain 5.1 GB (512 B rows)", n_rows, 512, 1},
        {"chain 1.3 GB (128 B rows)", n_rows, 128, 0},
        {"chain 128 MB", 262144, 512, 1},
        {"chain 2 MB", 4096, 512, 1},
    };
    for (auto& c : chains) {
        const uint32_t iters = 2000;
        // cache-resident footprints: the second pass repeats the first one's rows; the large ones: fresh rows
        const bool mall = c.rows == 262144; // beyond one L2, inside the Infinity Cache: sweep it in, then fresh rows
        if (mall) hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, 0, d, c.rows * (uint64_t)c.stride, d_sink);
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(1), 0, 0, d, c.rows, c.stride, iters, c.second,
                               (uint64_t)((c.rows > 1000000 || mall) ? 77 * (rep + 1) : 5), d_out);
        CK(hipDeviceSynchronize());
        uint64_t o[3];
        CK(hipMemcpy(o, d_out, 24, hipMemcpyDeviceToHost));
        printf("%-28s first load %7.0f cycles", c.name, (double)o[0] / iters);
        if (c.second) printf(" | second load, same row +384 B: %6.0f cycles", (double)o[1] / iters);
        printf("\n");
    }
    return 0;
}
