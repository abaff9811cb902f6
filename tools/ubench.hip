// ubench.hip -- what single-wavefront code costs on gfx950, in shader clocks per operation (s_memtime around N unrolled
// repetitions, ONE wave of one block: the regime of a walker that has its SIMD to itself). Development tool:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__device__ __forceinline__ uint64_t now() {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// each test: run body REPS times inside a loop of ITER iterations; out[test] = cycles
#define ITER 64
#define TEST_BEGIN(id)                         \
    {                                          \
        const int id_ = id;                    \
        uint64_t t0 = now();                   \
        for (int it_ = 0; it_ < ITER; ++it_) {
#define TEST_END(nops)                                           \
        }                                                        \
        uint64_t t1 = now();                                     \
        if (threadIdx.x == 0) { out[id_ * 2] = t1 - t0; out[id_ * 2 + 1] = (uint64_t)ITER * (nops); } \
    }

__global__ void k(uint64_t* out, uint32_t* sink, const uint32_t* chase_small, const uint32_t* chase_big) {
    uint32_t v = threadIdx.x, w = threadIdx.x * 3 + 1;
    uint64_t a = ((uint64_t)v << 32) | w, b = a * 7 + 3;
    extern __shared__ uint64_t lds[];
    // 0: empty loop (the loop's own back-edge)
    TEST_BEGIN(0) asm volatile("" ::: "memory"); TEST_END(1)
    // 1: 64 dependent v_add_u32
    TEST_BEGIN(1) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w)); TEST_END(64)
    // 2: 64 independent v_add_u32 (4 chains)
    {
        uint32_t x0 = v, x1 = v + 1, x2 = v + 2, x3 = v + 3;
        TEST_BEGIN(2) asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(w)); TEST_END(64)
        v += x0 + x1 + x2 + x3;
    }
    // 3: 64 v_cmp_lt_u64 (independent; result to vcc)
    TEST_BEGIN(3) asm volatile(REP64("v_cmp_lt_u64 vcc, %0, %1\n") ::"v"(a), "v"(b) : "vcc"); TEST_END(64)
    // 4: 64 v_cmp_lt_u32
    TEST_BEGIN(4) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n") ::"v"(v), "v"(w) : "vcc"); TEST_END(64)
    // 5: 16 x (v_cmp_lt_u32 -> s_and_b64 using vcc -> s_bcnt1) VALU->SALU hand-over
    {
        uint32_t s = 0;
        TEST_BEGIN(5) asm volatile(REP16("v_cmp_lt_u32 vcc, %1, %2\n s_bcnt1_i32_b64 s20, vcc\n s_add_u32 %0, %0, s20\n")
                                   : "+s"(s) : "v"(v), "v"(w) : "vcc", "s20", "scc"); TEST_END(16)
        v += s;
    }
    // 6: 16 x (v_cmp -> s_bcnt1 -> v_add with the SGPR -> next v_cmp depends on it) VALU->SALU->VALU round trip
    {
        uint32_t x = v;
        TEST_BEGIN(6) asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n v_add_u32 %0, s20, %0\n")
                                   : "+v"(x) : "v"(w) : "vcc", "s20", "scc"); TEST_END(16)
        v += x;
    }
    // 7: 16 x (v_readlane with SGPR index -> s_add -> ... ) readlane -> SALU chain
    {
        uint32_t s = 3;
        TEST_BEGIN(7) asm volatile(REP16("s_and_b32 s21, %0, 63\n v_readlane_b32 s20, %1, s21\n s_add_u32 %0, %0, s20\n")
                                   : "+s"(s) : "v"(w) : "s20", "s21", "scc"); TEST_END(16)
        v += s;
    }
    // 8: taken branches: 64 x (s_branch to the next instruction's label) -- forward taken branches
    TEST_BEGIN(8) asm volatile(REP16("s_branch 1f\n s_nop 0\n1:\n s_branch 2f\n s_nop 0\n2:\n s_branch 3f\n s_nop 0\n3:\n s_branch 4f\n s_nop 0\n4:\n") ::: "memory"); TEST_END(64)
    // 9: not-taken conditional branches: 64 x (s_cmp_eq 0,1 ; s_cbranch_scc1)
    TEST_BEGIN(9) asm volatile(REP64("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 9f\n") "9:\n" ::: "scc"); TEST_END(64)
    // 10: 16 x (v_cmp -> s_cbranch_vccnz not taken): VALU result consumed by a branch
    TEST_BEGIN(10) asm volatile(REP16("v_cmp_gt_u32 vcc, 0, %0\n s_cbranch_vccnz 9f\n") "9:\n" ::"v"(v) : "vcc"); TEST_END(16)
    // 11: LDS round trip: 16 x (ds_write_b64 ; ds_read_b64 of it ; wait)
    {
        uint32_t addr = threadIdx.x * 8;
        uint64_t x = a;
        TEST_BEGIN(11) asm volatile(REP16("ds_write_b64 %1, %0\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(x) : "v"(addr) : "memory"); TEST_END(16)
        a += x;
    }
    // 12: v_readfirstlane -> v_add using it (VALU -> SGPR -> VALU)
    {
        uint32_t x = v;
        TEST_BEGIN(12) asm volatile(REP16("v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0\n") : "+v"(x)::"s20"); TEST_END(16)
        v += x;
    }
    // 13: DPP wave_shr moves, dependent
    TEST_BEGIN(13) asm volatile(REP64("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v)); TEST_END(64)
    // 14: ds_bpermute dependent
    {
        uint32_t idx = ((threadIdx.x + 1) & 63) * 4, x = v;
        TEST_BEGIN(14) asm volatile(REP16("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(x) : "v"(idx)); TEST_END(16)
        v += x;
    }
    // 15: pointer chase, small buffer (L2 / MALL resident)
    {
        uint32_t p = threadIdx.x;
        TEST_BEGIN(15)
#pragma unroll
        for (int i = 0; i < 16; ++i) p = chase_small[p];
        TEST_END(16)
        v += p;
    }
    // 16: pointer chase over 2 GB (HBM)
    {
        uint32_t p = threadIdx.x * 1024u;
        TEST_BEGIN(16)
#pragma unroll
        for (int i = 0; i < 16; ++i) p = chase_big[p];
        TEST_END(16)
        v += p;
    }
    // 17: 64 dependent v_fma_f32
    {
        float f = (float)v, g = 1.0001f;
        TEST_BEGIN(17) asm volatile(REP64("v_fma_f32 %0, %0, %1, %1\n") : "+v"(f) : "v"(g)); TEST_END(64)
        v += (uint32_t)f;
    }
    // 18: 64 dependent v_pk_fma_f32
    {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 f = {(float)v, 1.0f}, g = {1.0001f, 0.5f};
        TEST_BEGIN(18) asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %1\n") : "+v"(f) : "v"(g)); TEST_END(64)
        v += (uint32_t)f.x;
    }
    // 19: s_nop-free SALU chain: 64 dependent s_add_u32
    {
        uint32_t s = 1;
        TEST_BEGIN(19) asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(s)::"scc"); TEST_END(64)
        v += s;
    }
    // 20: v_cmp -> v_cndmask using vcc (VALU -> VALU through vcc), 32 pairs
    {
        uint32_t x = v;
        TEST_BEGIN(20) asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %1, %0\n s_nop 1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n")
                                    : "+v"(x) : "v"(w) : "vcc"); TEST_END(32)
        v += x;
    }
    // 21: global_load_dwordx4 x13 from one row (issue cost), then wait; rows of 400 bytes, random per pair
    {
        const uint8_t* base = (const uint8_t*)chase_big;
        uint32_t p = (threadIdx.x >> 1) * 7919u;
        uint4 acc = make_uint4(0, 0, 0, 0);
        TEST_BEGIN(21)
        const uint8_t* row = base + (size_t)((p * 2654435761u) >> 10) * 400u + (threadIdx.x & 1) * 64;
        uint4 r[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) r[i] = *(const uint4*)(row + (i / 4) * 128 + (i % 4) * 16);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc.x ^= r[i].x; acc.y += r[i].y; acc.z ^= r[i].z; acc.w += r[i].w; }
        p = acc.x + it_;
        TEST_END(1)
        v += acc.x + acc.y + acc.z + acc.w;
    }

    // ---- crossings ----
    // 22: s_add -> v_add using it (SALU -> VALU), both chains dependent
    {
        uint32_t sc = 1, x = v;
        TEST_BEGIN(22) asm volatile(REP16("s_add_u32 %0, %0, 3\n v_add_u32 %1, %0, %1\n") : "+s"(sc), "+v"(x)::"scc"); TEST_END(16)
        v += x + sc;
    }
    // 23: v_readlane (constant lane) -> v_add using the SGPR (VALU -> SGPR -> VALU)
    {
        uint32_t x = v;
        TEST_BEGIN(23) asm volatile(REP16("v_readlane_b32 s20, %0, 5\n v_add_u32 %0, s20, %0\n") : "+v"(x)::"s20"); TEST_END(16)
        v += x;
    }
    // 24: s_ff1 -> v_readlane by that index -> s_or of the result -> (next s_ff1 depends): SALU -> lane select -> SALU
    {
        uint32_t lo = 0xAAAAAAAAu, hi = 0xAAAAAAAAu;
        TEST_BEGIN(24) asm volatile(REP16("s_ff1_i32_b32 s20, %0\n v_readlane_b32 s21, %1, s20\n s_or_b32 s21, s21, 1\n s_and_b32 s21, s21, 0\n s_or_b32 %0, %0, s21\n")
                                    : "+s"(lo) : "v"(w) : "s20", "s21", "scc"); TEST_END(16)
        v += lo + hi;
    }
    // 25: v_cmp -> s_cmp_eq_u64 vcc -> s_cselect -> v_cmp using it (VALU -> SALU -> VALU -> ...)
    {
        uint32_t sc = 5;
        TEST_BEGIN(25) asm volatile(REP16("v_cmp_eq_u32 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0\n s_cselect_b32 %0, %0, 7\n") : "+s"(sc) : "v"(w) : "vcc", "scc"); TEST_END(16)
        v += sc;
    }
    // 26: the rank loop of walk_fast.h as compiled (28 instructions per candidate), 32 candidates per run
    // 27: without the look-up of the candidate in the list; 28: + v_writelane for the rank; 29: both
#define RANK_PROLOGUE                                                                                     \
    "v_mov_b32 v20, %1\n v_mov_b32 v21, %2\n v_mov_b32 v22, %1\n v_mov_b32 v23, %2\n v_or_b32 v24, 1, v22\n" \
    "v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v29, %3\n"                           \
    "s_mov_b32 s12, 0xAAAAAAAA\n s_mov_b32 s13, 0xAAAAAAAA\n"
    // v20:21 candidate keys (lo, hi), v22:23 list keys, v24 id1, v25 shift, v26 rank, v27 below, v29 lane
    {
        uint32_t lo = w * 2654435761u, hi = v * 40503u;
        TEST_BEGIN(26)
        asm volatile(RANK_PROLOGUE
                     "1:\n"
                     "s_ff1_i32_b64 s42, s[12:13]\n s_add_u32 s10, s12, -1\n s_addc_u32 s11, s13, -1\n v_readlane_b32 s48, v20, s42\n"
                     "s_and_b64 s[12:13], s[10:11], s[12:13]\n s_or_b32 s10, s48, 1\n v_cmp_eq_u32 vcc, s10, v24\n s_cmp_eq_u64 vcc, 0\n"
                     "s_cselect_b64 s[46:47], -1, 0\n v_readlane_b32 s49, v21, s42\n s_and_b64 s[10:11], s[46:47], exec\n"
                     "s_cselect_b32 s49, s49, -1\n s_cselect_b32 s48, s48, -1\n v_cmp_lt_u64 vcc, s[48:49], v[22:23]\n s_bcnt1_i32_b64 s50, vcc\n"
                     "s_and_b64 s[10:11], s[46:47], exec\n v_addc_co_u32 v25, s[10:11], 0, v25, vcc\n s_cselect_b32 s10, 64, 0x41\n"
                     "s_sub_i32 s10, s10, s50\n v_mov_b32 v28, s10\n v_cmp_eq_u32 vcc, s42, v29\n s_cmp_eq_u64 s[12:13], 0\n s_nop 0\n"
                     "v_cndmask_b32 v26, v26, v28, vcc\n v_cmp_lt_u64 vcc, s[48:49], v[20:21]\n s_nop 1\n v_addc_co_u32 v27, vcc, 0, v27, vcc\n"
                     "s_cbranch_scc0 1b\n"
                     "v_add_u32 %0, v25, v26\n v_add_u32 %0, %0, v27\n"
                     : "=v"(lo) : "v"(lo), "v"(hi), "v"((uint32_t)threadIdx.x)
                     : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "s10", "s11", "s12", "s13", "s42", "s46", "s47", "s48", "s49", "s50", "vcc", "scc");
        TEST_END(32)
        v += lo;
    }
    {
        uint32_t lo = w * 2654435761u, hi = v * 40503u;
        TEST_BEGIN(27)
        asm volatile(RANK_PROLOGUE
                     "1:\n"
                     "s_ff1_i32_b64 s42, s[12:13]\n s_add_u32 s10, s12, -1\n s_addc_u32 s11, s13, -1\n v_readlane_b32 s48, v20, s42\n"
                     "s_and_b64 s[12:13], s[10:11], s[12:13]\n v_readlane_b32 s49, v21, s42\n"
                     "v_cmp_lt_u64 vcc, s[48:49], v[22:23]\n s_bcnt1_i32_b64 s50, vcc\n"
                     "v_addc_co_u32 v25, s[10:11], 0, v25, vcc\n"
                     "s_sub_i32 s10, 64, s50\n v_mov_b32 v28, s10\n v_cmp_eq_u32 vcc, s42, v29\n s_cmp_eq_u64 s[12:13], 0\n s_nop 0\n"
                     "v_cndmask_b32 v26, v26, v28, vcc\n v_cmp_lt_u64 vcc, s[48:49], v[20:21]\n s_nop 1\n v_addc_co_u32 v27, vcc, 0, v27, vcc\n"
                     "s_cbranch_scc0 1b\n"
                     "v_add_u32 %0, v25, v26\n v_add_u32 %0, %0, v27\n"
                     : "=v"(lo) : "v"(lo), "v"(hi), "v"((uint32_t)threadIdx.x)
                     : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "s10", "s11", "s12", "s13", "s42", "s46", "s47", "s48", "s49", "s50", "vcc", "scc");
        TEST_END(32)
        v += lo;
    }
    {
        uint32_t lo = w * 2654435761u, hi = v * 40503u;
        TEST_BEGIN(28)
        asm volatile(RANK_PROLOGUE
                     "1:\n"
                     "s_ff1_i32_b64 s42, s[12:13]\n s_add_u32 s10, s12, -1\n s_addc_u32 s11, s13, -1\n v_readlane_b32 s48, v20, s42\n"
                     "s_and_b64 s[12:13], s[10:11], s[12:13]\n v_readlane_b32 s49, v21, s42\n"
                     "v_cmp_lt_u64 vcc, s[48:49], v[22:23]\n s_bcnt1_i32_b64 s50, vcc\n"
                     "v_addc_co_u32 v25, s[10:11], 0, v25, vcc\n"
                     "s_sub_i32 s10, 64, s50\n s_mov_b32 m0, s42\n v_writelane_b32 v26, s10, m0\n s_cmp_eq_u64 s[12:13], 0\n"
                     "v_cmp_lt_u64 vcc, s[48:49], v[20:21]\n s_nop 1\n v_addc_co_u32 v27, vcc, 0, v27, vcc\n"
                     "s_cbranch_scc0 1b\n"
                     "v_add_u32 %0, v25, v26\n v_add_u32 %0, %0, v27\n"
                     : "=v"(lo) : "v"(lo), "v"(hi), "v"((uint32_t)threadIdx.x)
                     : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "s10", "s11", "s12", "s13", "s42", "s46", "s47", "s48", "s49", "s50", "vcc", "scc", "m0");
        TEST_END(32)
        v += lo;
    }
    // 29: two candidates per trip, interleaved by hand (the second's scalar chain under the first's vector ops), no look-up, v_writelane
    {
        uint32_t lo = w * 2654435761u, hi = v * 40503u;
        TEST_BEGIN(29)
        asm volatile(RANK_PROLOGUE
                     "1:\n"
                     "s_ff1_i32_b64 s42, s[12:13]\n s_add_u32 s10, s12, -1\n s_addc_u32 s11, s13, -1\n s_and_b64 s[12:13], s[10:11], s[12:13]\n"
                     "s_ff1_i32_b64 s43, s[12:13]\n s_add_u32 s10, s12, -1\n s_addc_u32 s11, s13, -1\n s_and_b64 s[12:13], s[10:11], s[12:13]\n"
                     "v_readlane_b32 s48, v20, s42\n v_readlane_b32 s49, v21, s42\n v_readlane_b32 s52, v20, s43\n v_readlane_b32 s53, v21, s43\n"
                     "v_cmp_lt_u64 vcc, s[48:49], v[22:23]\n v_cmp_lt_u64 s[46:47], s[52:53], v[22:23]\n"
                     "v_addc_co_u32 v25, s[10:11], 0, v25, vcc\n v_addc_co_u32 v25, s[10:11], 0, v25, s[46:47]\n"
                     "v_cmp_lt_u64 s[54:55], s[48:49], v[20:21]\n v_cmp_lt_u64 s[56:57], s[52:53], v[20:21]\n"
                     "s_bcnt1_i32_b64 s50, vcc\n s_bcnt1_i32_b64 s51, s[46:47]\n"
                     "v_addc_co_u32 v27, s[10:11], 0, v27, s[54:55]\n v_addc_co_u32 v27, s[10:11], 0, v27, s[56:57]\n"
                     "s_sub_i32 s50, 64, s50\n s_sub_i32 s51, 64, s51\n s_mov_b32 m0, s42\n v_writelane_b32 v26, s50, m0\n s_mov_b32 m0, s43\n v_writelane_b32 v26, s51, m0\n"
                     "s_cmp_eq_u64 s[12:13], 0\n s_cbranch_scc0 1b\n"
                     "v_add_u32 %0, v25, v26\n v_add_u32 %0, %0, v27\n"
                     : "=v"(lo) : "v"(lo), "v"(hi), "v"((uint32_t)threadIdx.x)
                     : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "s10", "s11", "s12", "s13", "s42", "s43", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "vcc", "scc", "m0");
        TEST_END(32)
        v += lo;
    }
    sink[threadIdx.x] = v + (uint32_t)a;
}

int main() {
    const int NT = 30;
    uint64_t* d_out; uint32_t *d_sink, *d_small, *d_big;
    hipMalloc(&d_out, NT * 16); hipMalloc(&d_sink, 256 * 4);
    const size_t nsmall = 1 << 16, nbig = (size_t)1 << 29; // 256 KB, 2 GB
    hipMalloc(&d_small, nsmall * 4); hipMalloc(&d_big, nbig * 4);
    std::vector<uint32_t> hs(nsmall);
    for (size_t i = 0; i < nsmall; ++i) hs[i] = (uint32_t)((i * 40503u + 12345u) & (nsmall - 1));
    hipMemcpy(d_small, hs.data(), nsmall * 4, hipMemcpyHostToDevice);
    {   // big: p -> (p * A + C) mod nbig, filled on the host in pieces
        std::vector<uint32_t> hb((size_t)1 << 24);
        for (size_t c = 0; c < nbig; c += hb.size()) {
            for (size_t i = 0; i < hb.size(); ++i) hb[i] = (uint32_t)((((c + i) * 2654435761ull) + 1013904223ull) & (nbig - 1));
            hipMemcpy(d_big + c, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        }
    }
    hipMemset(d_out, 0, NT * 16);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d_out, d_sink, d_small, d_big); hipDeviceSynchronize(); }
    uint64_t h[NT * 2];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[NT] = {"empty loop iteration (back-edge)", "v_add_u32 dependent", "v_add_u32 independent x4", "v_cmp_lt_u64", "v_cmp_lt_u32",
                             "v_cmp -> s_bcnt1 -> s_add (VALU->SALU)", "v_cmp -> s_bcnt1 -> v_add (VALU->SALU->VALU)", "s_and; v_readlane(sgpr idx); s_add",
                             "s_branch taken (forward, next line)", "s_cmp + s_cbranch not taken", "v_cmp -> s_cbranch_vccnz not taken", "ds_write_b64 + ds_read_b64 + wait",
                             "v_readfirstlane -> v_add", "v_mov_dpp wave_shr dependent", "ds_bpermute + wait", "pointer chase 256 KB", "pointer chase 2 GB (HBM)",
                             "v_fma_f32 dependent", "v_pk_fma_f32 dependent", "s_add_u32 dependent", "v_cmp->cndmask / v_cmp->addc pair (with s_nop 1)", "13 x dwordx4 row gather + wait", "s_add -> v_add (SALU->VALU)", "v_readlane const lane -> v_add", "s_ff1 -> v_readlane -> s_or x3 (5 instr)",
                             "v_cmp -> s_cmp_eq_u64 -> s_cselect (VALU->SALU->SALU)", "rank loop as compiled (per candidate)", "rank loop, no look-up", "rank loop, no look-up, v_writelane", "rank loop x2 interleaved, no look-up, v_writelane"};
    const double base = (double)h[0] / (double)h[1];
    printf("loop overhead per iteration: %.1f clocks\n", base);
    for (int i = 1; i < NT; ++i) {
        const double per = ((double)h[i * 2] - base * ITER) / (double)h[i * 2 + 1];
        printf("%-55s %8.1f clocks per op (%llu clocks / %llu ops)\n", names[i], per, (unsigned long long)h[i * 2], (unsigned long long)h[i * 2 + 1]);
    }
    return 0;
}
