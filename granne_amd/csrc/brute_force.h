// brute_force.h -- exact k nearest elements of every query by scanning ALL elements: the one piece of granne's
// element side that is a real contraction (ElementContainer::dists for every index at once,
// /root/reference/src/elements/mod.rs:35-39 over src/elements/dense_vector.rs:157-163), so the one piece that
// belongs on the matrix cores. It is the recall ground truth of bench.py and an operator of its own
// (granne_hip_brute_force_device).
//
//   scores   v_mfma_f32_32x32x2_f32 (f32 rows) / v_mfma_i32_32x32x32_i8 (int8 rows: exact integer dots). The A operand
//            is a tile of ELEMENTS staged in LDS, the B operand the wave's 32 QUERIES held in registers for the whole
//            scan, so that in the 32x32 result a lane owns ONE query (column) and sees 16 elements per block: the
//            running top-k of that query lives in the lane's registers and is touched only when a score beats its
//            k-th (k ln(n/k) times per scan). The two K halves of the MFMA (lanes 0-31 / 32-63) take the two halves
//            of the vector; which physical component meets which K step is immaterial to a dot product.
//   layout   grid = (query tiles of 256, element ranges): the tiles of one range run side by side and stream the same
//            rows; block = 8 waves x 32 queries; the lane pair (l, l+32) of a query joins its two lists at the end of
//            the range; the lists of all ranges are merged by merge_topk_kernel.
//   exact    the MFMA sums in another order than the reference's 32 accumulators (src/math.rs:17-42): the scan SELECTS
//            k + BF_EXTRA candidates per query by that score, then their distances are recomputed by dists_kernel --
//            the reference's arithmetic, bit for bit -- and the k best by (distance, id) are returned. Distances are
//            therefore the reference's; the id SET can differ from a scalar scan only where two elements' distances to
//            the query differ by less than the MFMA's rounding (~1e-6) at the k + BF_EXTRA boundary.
// Roofline: all three forms are bound by the matrix cores -- the f32 form by the f32 matrix rate (2 * nq * n * dim flops at
// 157 TFLOP/s dense f32: 13 ms for 1024 x 10M x 100; 17 ms measured), the bf16 form (round 6, bf_b16_kernel: 6.6 ms) and the
// int8 form (bf_i8_ring_kernel: 1.75-2.0 ms; the rows once from HBM are 1.28 GB, 0.2 ms) by what the chip SUSTAINS on them:
// under dense matrix load it halves its shader clock (tools/mfma_rate.hip), and the scans' inner loops alone take 3.75-5.1 ms
// and 0.95 ms (tools/mfma_lds_loop*.hip, profiles/r6_scan_matrix_rates.log). bench.py reports the achieved rate against the
// guide's dense peaks and says so.
#pragma once

#include "util_kernels.h"

namespace granne_hip {

typedef float bf_f32x16 __attribute__((ext_vector_type(16)));
typedef int bf_i32x16 __attribute__((ext_vector_type(16)));
typedef int bf_i32x4 __attribute__((ext_vector_type(4)));

#ifndef GRANNE_BF_EXP
#define GRANNE_BF_EXP 0
#endif
#ifndef GRANNE_BF_SUB
#define GRANNE_BF_SUB 2
#endif
constexpr uint32_t BF_I8_SUB = GRANNE_BF_SUB; // int8 scan: tiles per staging round (bf_i8_kernel)
constexpr uint32_t BF_QT = 256;    // queries per block (8 waves x 32: two per SIMD, one scores while the other is checked)
constexpr uint32_t BF_THREADS = 512;
constexpr uint32_t BF_EXTRA = 6;   // candidates selected beyond k, re-ranked by the exact distance
constexpr uint32_t BF_KMAX = 16;   // longest per-lane list (k + BF_EXTRA <= BF_KMAX)

struct BruteParams {
    const uint8_t* elements; // device rows: row_bytes of (zero padded) data every row_stride bytes
    uint64_t n;
    uint32_t row_bytes, row_stride, dim;
    const uint8_t* queries;  // dense [nq][dim]
    uint32_t nq, kk;         // kk = k + BF_EXTRA entries per list
    uint64_t per_range;      // elements per range (a multiple of the tile)
    uint64_t* part_ids;      // [ranges][nq][kk]
    float* part_d;           // [ranges][nq][kk]
    uint32_t* part_c;        // [ranges][nq]
    const float* inv_norm;   // int8: [n] 1 / |x| of every row (made once per index: inv_norm_rows_kernel)
    const float* inv_gmax;   // int8: [ceil(n / 32)][2] the largest 1 / |x| among the 16 rows a lane half sees of a 32-row block (inv_gmax_kernel)
    const float* tau_in;     // [nq] or null: a score that at least kk elements of the set reach (the priming pass's kk-th best)
    uint32_t* share_hist;    // [nq][BF_SHARE_BUCKETS] or null: elements seen so far per score bucket, by ALL ranges (BfShare)
    const uint8_t* qpad;     // int8 rows of more than 128 bytes: the queries zero padded to row_bytes each, [nq][row_bytes] (bf_i8_chunked_kernel)
};

// Which (query tile, element range) a block takes. Workgroups go to the 8 XCDs round robin by their linear id, each XCD
// with an L2 of its own: with the plain (blockIdx.x, blockIdx.y) reading, the 4 query tiles that stream the SAME range sat
// on 4 different XCDs and every row came from HBM four times (5.1 GB per 1024 x 10M int8 scan, with one 16 KB tile in
// flight per block: the scan was bound by that latency, not by anything it computes). Here the tiles of a range share
// an XCD: its rows come from HBM once and from that L2 three times.
struct BfBlock { uint32_t qt, range; };
__device__ __forceinline__ BfBlock bf_block() {
    const uint32_t nqt = gridDim.x, G = gridDim.y;
    const uint32_t L = blockIdx.y * nqt + blockIdx.x; // dispatch order: x fastest
    BfBlock b;
    if ((G & 7u) == 0u) {
        const uint32_t xcd = L & 7u, slot = L >> 3; // slot: 0 .. nqt * G / 8 - 1 within the XCD
        b.range = xcd * (G >> 3) + slot / nqt;
        b.qt = slot % nqt;
    } else {
        b.qt = blockIdx.x;
        b.range = blockIdx.y;
    }
    return b;
}

// per-lane top list: KK scores descending (a larger dot is a smaller distance), always full length -- the lists are
// cut to k + BF_EXTRA when they are written
template <int KK>
struct BfList {
    float s[KK];
    uint32_t id[KK];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < KK; ++i) { s[i] = -3.0e38f; id[i] = 0xFFFFFFFFu; }
    }
    __device__ __forceinline__ float worst() const { return s[KK - 1]; }
    __device__ __forceinline__ void insert(float sc, uint32_t e) {
#pragma unroll
        for (int i = 0; i < KK; ++i) {
            const bool take = sc > s[i];
            const float ts = s[i];
            const uint32_t ti = id[i];
            s[i] = take ? sc : ts;
            id[i] = take ? e : ti;
            sc = take ? ts : sc;
            e = take ? ti : e;
        }
    }
};

// end of a range: lane l takes the list of lane l + 32 (the other K half's rows of the same query) and writes the joint one
// `scale`: what a list's scores are multiplied by on the way out (int8 lists hold dot / |x|; the query's 1 / |q| comes here)
template <int KK>
__device__ __forceinline__ void bf_write_list(const BruteParams& P, BfList<KK>& L, uint32_t range, uint32_t q, bool qlive, uint32_t h, float scale = 1.0f) {
#pragma unroll
    for (int i = 0; i < KK; ++i) {
        const float sc = __shfl_xor(L.s[i], 32, 64);
        const uint32_t e = (uint32_t)__shfl_xor((int)L.id[i], 32, 64);
        if (h == 0u && e != 0xFFFFFFFFu) L.insert(sc, e);
    }
    if (qlive && h == 0u) {
        const size_t list = (size_t)range * P.nq + q;
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < KK; ++i) {
            if ((uint32_t)i < P.kk) {
                const bool ok = L.id[i] != 0xFFFFFFFFu;
                P.part_ids[list * P.kk + i] = ok ? (uint64_t)L.id[i] : ~0ull;
                const float sc = L.s[i] * scale;
                const float d = 1.0f - sc;
                P.part_d[list * P.kk + i] = ok ? (d > 0.0f ? d : 0.0f) : __builtin_inff();
                cnt += ok ? 1u : 0u;
            }
        }
        P.part_c[list] = cnt;
    }
}

// The threshold a scan starts from. Every element range used to warm its lists up from nothing -- k ln(n / k) inserts of a
// 16-deep register list per query and range, under exec masks: 14 of the 18 vector instructions per 64 scores of the
// round-4 int8 scan (profiles/r4_bruteforce_i8_sq_pmc.csv). A priming pass over the first 1/64 of the rows leaves its
// kk-th best score per query; no element below it can be among the kk best of the whole set (the sample alone holds kk
// that reach it), so the scan proper starts there (one ulp below: the sample's own kk-th must pass `>`).
// the priming pass: the best score of (range, query), the two lanes of a query joined
__device__ __forceinline__ void bf_write_max(const BruteParams& P, float best, uint32_t range, uint32_t q, bool qlive, uint32_t h) {
    best = __builtin_fmaxf(best, __shfl_xor(best, 32, 64));
    if (qlive && h == 0u) P.part_d[(size_t)range * P.nq + q] = best;
}

__device__ __forceinline__ float bf_next_below(float t) {
    if (!(t > -1.0e38f)) return t;
    if (t > 0.0f) return __uint_as_float(__float_as_uint(t) - 1u);
    if (t < 0.0f) return __uint_as_float(__float_as_uint(t) + 1u);
    return -1.0e-30f;
}
__device__ __forceinline__ float bf_start_tau(const BruteParams& P, uint32_t q, bool qlive) {
    return (P.tau_in && qlive) ? bf_next_below(P.tau_in[q]) : -3.0e38f;
}

// One threshold per QUERY instead of one per (range, lane). The 64 ranges x 2 lane halves of a query each keep a list of
// their own, and a list only knows its own rows: over 1/128 of 10M rows its kk-th best sits at ~3.6 sigma of the score
// distribution where the set's kk-th best sits at ~4.6 -- a quarter of all 32x32 blocks held a score that beat SOME lane's
// threshold and went through the exec-masked insert chain (2/3 of the int8 scan's time). Every insert is an element
// nobody else sees, so the inserts of all lists are counted per query in a histogram over score buckets [edge(j),
// edge(j+1)), edge(j) = base * (1 + j/64), base = the primed threshold: once the buckets from j up hold kk elements,
// nothing below edge(j) is among the kk best of the set. A wave looks its 32 queries' histograms up after 1, 3, 7, 15, ...
// tiles (the kk-th best of the rows seen so far moves with the LOGARITHM of their number): the two lanes of a query read
// half the buckets each and join. Counts that arrive late only leave a threshold lower than
// it could be; the lists' union still holds the kk best, so the merged result does not depend on timing.
// (First form: every insert walked the histogram and published the threshold through an atomic max -- 32 dependent
// agent-scope loads under an exec mask per insert: 3.5 -> 13.9 ms.)
constexpr int BF_SHARE_BUCKETS = 32;
struct BfShare {
    float step;                // base / 64: edge(j) = step * (64 + j); 0 = this lane does not share
    uint32_t tiles, next_poll; // wave-uniform
    __device__ __forceinline__ void init(const BruteParams& P, bool qlive, float start) {
        const bool on = P.share_hist != nullptr && qlive && start > 1.0e-30f && start < 1.0e30f;
        step = on ? start * (1.0f / 64.0f) : 0.0f;
        tiles = 0;
        next_poll = 2; // (counted at the top of a tile: after the first tile has been scored)
    }
    __device__ __forceinline__ bool on() const { return step > 0.0f; }
    __device__ __forceinline__ float edge(int j) const { return step * (float)(64 + j); }
    // an element with score sc (> base) has just been inserted
    __device__ __forceinline__ void count(const BruteParams& P, uint32_t q, float sc) {
        // (clamped as a float first: sc may be many orders above a base of 1e-30, and a float beyond int's range converts
        //  to nothing C++ defines)
        const float jf = __builtin_fminf(sc * __builtin_amdgcn_rcpf(step), 64.0f + (float)BF_SHARE_BUCKETS);
        int j = (int)jf - 64;
        j = j < 0 ? 0 : (j > BF_SHARE_BUCKETS - 1 ? BF_SHARE_BUCKETS - 1 : j);
        while (j > 0 && sc < edge(j)) --j; // the count of bucket j vouches for sc >= edge(j): never round up into one
        __hip_atomic_fetch_add(P.share_hist + (size_t)q * BF_SHARE_BUCKETS + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // once per tile, the whole wave: the threshold the counts support (or nothing new). Two rounds of 8 loads per lane,
    // the upper 16 buckets first: lane h = 1 takes the upper 8 of a round, h = 0 the lower 8, the pair joins.
    __device__ __forceinline__ float poll(const BruteParams& P, uint32_t q, uint32_t h) {
        tiles += 1;
        if (tiles != next_poll) return -3.0e38f;
        next_poll = tiles * 2u; // the kk-th best of the rows seen so far moves with the logarithm of their number
        const uint32_t* hist = P.share_hist + (size_t)(on() ? q : 0u) * BF_SHARE_BUCKETS;
        uint32_t carry = 0;
        int top = -1;
#pragma unroll 1
        for (int round = 1; round >= 0 && __ballot(on() && top < 0); --round) {
            const int b0 = round * 16 + (int)h * 8;
            uint32_t c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                c[i] = on() ? __hip_atomic_load(hist + b0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) mine += c[i];
            const uint32_t other = (uint32_t)__shfl_xor((int)mine, 32, 64);
            uint32_t sum = carry + (h == 0u ? other : 0u); // what the buckets above this lane's eight hold
            int found = -1;
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                sum += c[i];
                if (found < 0 && sum >= P.kk) found = b0 + i;
            }
            const int pair = max(found, __shfl_xor(found, 32, 64));
            if (top < 0) top = pair;
            carry += mine + other;
        }
        return (on() && top > 0) ? bf_next_below(edge(top)) : -3.0e38f; // scores must beat it strictly: edge(top) itself passes
    }
};

// f32: KH = K entries per half (vector components h*KH .. h*KH+KH-1, zero padded), R = 32-element blocks per tile
// (the shared threshold is compiled out of the f32 scan: it is MFMA-bound -- 0.76 of the f32 matrix peak -- and at 245-252
// registers the histogram's few more cost it spills: 17.1 -> 18.0 ms with it)
template <int KH, int R, bool PRIME = false, bool SHARE = false>
__global__ __launch_bounds__(BF_THREADS) void bf_f32_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t ET = 32u * R;        // elements per tile
    constexpr uint32_t STRIDE = 2u * KH + 4u; // floats per LDS row: 4 x odd -> conflict-free ds_read_b128 down a column
    float* tile = reinterpret_cast<float*>(smem_bf);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const BfBlock blk = bf_block();
    const uint32_t q = blk.qt * BF_QT + wave * 32u + col;
    const bool qlive = q < P.nq;

    // the lane's half of its query, in registers for the whole scan
    float qr[KH];
    {
        const float* qp = reinterpret_cast<const float*>(P.queries) + (size_t)(qlive ? q : 0u) * P.dim;
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            const uint32_t c = h * KH + t;
            qr[t] = (qlive && c < P.dim) ? qp[c] : 0.0f;
        }
    }
    BfList<PRIME ? 1 : BF_KMAX> L;
    L.init();
    float tau = bf_start_tau(P, q, qlive); // what a score must beat: the list's kk-th, never below the priming pass's
    [[maybe_unused]] float best = -3.0e38f; // PRIME: the largest score of the range, nothing else
    [[maybe_unused]] BfShare share;
    if constexpr (!PRIME && SHARE) share.init(P, qlive, tau);

    const uint64_t r0 = (uint64_t)blk.range * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    const uint32_t row_f4 = P.row_bytes / 16u; // float4 units per device row (rows are zero padded to 16 bytes)
    // The tile of the NEXT step travels from HBM to registers while the matrix cores work on this one
    // ([row][component], components beyond dim zero); it is written to LDS at the top of its step.
    constexpr uint32_t UNITS = STRIDE / 4u;                  // float4 per LDS row
    constexpr uint32_t NPF = (ET * UNITS + BF_THREADS - 1u) / BF_THREADS; // float4 per thread and tile
    float4 pf[NPF];
    auto fetch = [&](uint64_t e0) {
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t u = tid + BF_THREADS * j;
            const uint32_t row = u / UNITS, c4 = u - row * UNITS;
            pf[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (u < ET * UNITS && e0 + row < r1 && c4 < row_f4)
                pf[j] = *reinterpret_cast<const float4*>(P.elements + (e0 + row) * P.row_stride + c4 * 16u);
        }
    };
    fetch(r0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ET) {
        __syncthreads(); // the previous tile has been consumed
        if constexpr (!PRIME && SHARE) {
            if (P.share_hist) tau = __builtin_fmaxf(tau, share.poll(P, q, h));
        }
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t u = tid + BF_THREADS * j;
            const uint32_t row = u / UNITS, c4 = u - row * UNITS;
            if (u < ET * UNITS) *reinterpret_cast<float4*>(tile + (size_t)row * STRIDE + c4 * 4u) = pf[j];
        }
        __syncthreads();
        if (e0 + ET < r1) fetch(e0 + ET);
        bf_f32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][v] = 0.0f;
#pragma unroll
        for (int t4 = 0; t4 < KH / 4; ++t4) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 a = *reinterpret_cast<const float4*>(tile + (size_t)(r * 32 + col) * STRIDE + h * KH + t4 * 4);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qr[t4 * 4 + 0], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qr[t4 * 4 + 1], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qr[t4 * 4 + 2], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qr[t4 * 4 + 3], acc[r], 0, 0, 0);
            }
        }
        // result block r: acc[r][v] = dot(element e0 + r*32 + 8*(v/4) + 4*h + v%4, query `col` of this wave).
        // Almost no block holds a score that beats a lane's kk-th: one max over the block decides for the wave.
        const bool whole = e0 + ET <= r1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (PRIME) { // (rows past the range's end are zero rows: score 0, below every real maximum of this data or equal to it)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    best = (whole || e < r1) ? __builtin_fmaxf(best, acc[r][v]) : best;
                }
                continue;
            }
            float mx = acc[r][0];
#pragma unroll
            for (int v = 1; v < 16; ++v) mx = __builtin_fmaxf(mx, acc[r][v]);
            if (__ballot(mx > tau)) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float sc = acc[r][v];
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    if (sc > tau && (whole || e < r1)) {
                        L.insert(sc, (uint32_t)e);
                        tau = __builtin_fmaxf(tau, L.worst());
                        if constexpr (SHARE) {
                            if (share.on()) share.count(P, q, sc);
                        }
                    }
                }
            }
        }
    }
    if constexpr (PRIME) bf_write_max(P, best, blk.range, q, qlive, h);
    else bf_write_list(P, L, blk.range, q, qlive, h);
}

// f32 rows scored on the bf16 matrix path (round 6): the f32 scan above is MFMA-bound at 0.76 of the f32 matrix peak
// (157 TFLOP/s: 17.2 ms per 1024 x 10M x 100) while the bf16 path is 16 x as wide. A scan only SELECTS candidates (their
// distances are recomputed exactly afterwards), so its score needs to be accurate to a small fraction of the gap between
// the k-th and the (k + BF_EXTRA)-th best, not to the last bit: every component is split into two bf16 pieces,
// x = xh + xl (xh = bf16(x), xl = bf16(x - xh): 16 bits of mantissa between them), and a product is three matrix
// instructions, xh qh + xh ql + xl qh, accumulated in f32 -- what is dropped (xl ql and the pieces' own rounding) is below
// 2^-15 of |x||q| per product: ~3e-6 of a unit-vector dot in the typical case, 3e-5 at worst, against gaps of 1e-3 and
// more between neighbours of rank 10 and 16 in the benchmark's sets. Same tile / range / priming / list structure as the
// f32 scan; per 16 components three v_mfma_f32_32x32x16_bf16 (96 cycles) where that one issues eight 32x32x2 (512).
// KG = groups of 8 components per lane half (the two halves of the wave take the two halves of the vector: 16 KG >= dim).
typedef __bf16 bf_b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf_b16x4 __attribute__((ext_vector_type(4)));
// (Blocks of 256 threads -- two per CU with barriers of their own, so that one block's conversion runs under the other's
// matrix instructions -- were tried: 11.2 ms where the block of eight waves takes 7.3; every block converts and stores the
// whole tile, and eight blocks instead of four stream every range. Two tiles in LDS with the next tile's conversion
// interleaved into the matrix loop, one barrier per tile: 7.2 ms where this form takes 6.6 -- the conversion then waits
// for the tile's loads inside the loop, and a second set of load registers does not fit. Two query sets per wave with the
// lists as sets in LDS, bf_i8_ring_kernel's shape without the LDS-DMA: 6.6-7.0 ms at 245 registers, no gain, dropped.)
#ifndef GRANNE_BF_B16_AHEAD
#define GRANNE_BF_B16_AHEAD 1 // fragment pairs read ahead of the matrix instructions (2: +8 registers, the same 6.2-6.8 ms)
#endif
constexpr uint32_t BF_B16_THREADS = 512, BF_B16_QT = 256;
template <int KG, int R, bool PRIME = false>
__global__ __launch_bounds__(BF_B16_THREADS) void bf_b16_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t ET = 32u * R;                // elements per tile
    constexpr uint32_t COMPS = 16u * KG;            // components per (zero padded) row
    constexpr uint32_t STRIDE_B = 2u * COMPS + 16u; // bytes per LDS row of one piece: an odd number of 16-byte units
    uint8_t* tile_hi = smem_bf;
    uint8_t* tile_lo = smem_bf + (size_t)ET * STRIDE_B;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const BfBlock blk = bf_block();
    const uint32_t q = blk.qt * BF_B16_QT + wave * 32u + col;
    const bool qlive = q < P.nq;

    // the lane's half of its query as bf16 pieces, in registers for the whole scan
    bf_b16x8 qh[KG], ql[KG];
    {
        const float* qp = reinterpret_cast<const float*>(P.queries) + (size_t)(qlive ? q : 0u) * P.dim;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t c = h * 8u * KG + (uint32_t)g * 8u + (uint32_t)j;
                const float v = (qlive && c < P.dim) ? qp[c] : 0.0f;
                const __bf16 hi = (__bf16)v;
                qh[g][j] = hi;
                ql[g][j] = (__bf16)(v - (float)hi);
            }
        }
    }
    BfList<PRIME ? 1 : BF_KMAX> L;
    L.init();
    float tau = bf_start_tau(P, q, qlive);
    [[maybe_unused]] float best = -3.0e38f;

    const uint64_t r0 = (uint64_t)blk.range * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    const uint32_t row_f4 = P.row_bytes / 16u;      // float4 units of a device row that hold data
    constexpr uint32_t UNITS = COMPS / 4u;          // float4 units per (padded) row
    constexpr uint32_t NPF = (ET * UNITS + BF_B16_THREADS - 1u) / BF_B16_THREADS;
    float4 pf[NPF];
    auto fetch = [&](uint64_t e0) {
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t u = tid + BF_B16_THREADS * j;
            const uint32_t row = u / UNITS, c4 = u - row * UNITS;
            pf[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (u < ET * UNITS && e0 + row < r1 && c4 < row_f4)
                pf[j] = *reinterpret_cast<const float4*>(P.elements + (e0 + row) * P.row_stride + c4 * 16u);
        }
    };
    fetch(r0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ET) {
        __syncthreads(); // the previous tile has been consumed
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) { // split into the two pieces on the way to LDS
            const uint32_t u = tid + BF_B16_THREADS * j;
            const uint32_t row = u / UNITS, c4 = u - row * UNITS;
            if (u < ET * UNITS) {
                bf_b16x4 hi, lo;
                const float v[4] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    hi[t] = (__bf16)v[t];
                    lo[t] = (__bf16)(v[t] - (float)hi[t]);
                }
                *reinterpret_cast<bf_b16x4*>(tile_hi + (size_t)row * STRIDE_B + c4 * 8u) = hi;
                *reinterpret_cast<bf_b16x4*>(tile_lo + (size_t)row * STRIDE_B + c4 * 8u) = lo;
            }
        }
        __syncthreads();
        if (e0 + ET < r1) fetch(e0 + ET);
        bf_f32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][v] = 0.0f;
        // the fragments of the next (step, block) are read while the matrix cores work on this one (left alone the scheduler
        // sinks every read to just before its instruction, and each waits for LDS; a whole step ahead -- 2 x R fragments of
        // each piece -- spilled: 10.1 ms where this takes less)
        const size_t off0 = (size_t)col * STRIDE_B + (size_t)h * 16u * KG;
        constexpr int AHEAD = KG == 13 ? 1 : GRANNE_BF_B16_AHEAD, NF = KG * R; // (KG = 13: a second step ahead spills)
        bf_b16x8 ah[AHEAD + 1], al[AHEAD + 1];
#pragma unroll
        for (int i = 0; i < AHEAD && i < NF; ++i) {
            ah[i] = *reinterpret_cast<const bf_b16x8*>(tile_hi + off0 + (size_t)(i % R) * 32u * STRIDE_B + (size_t)(i / R) * 16u);
            al[i] = *reinterpret_cast<const bf_b16x8*>(tile_lo + off0 + (size_t)(i % R) * 32u * STRIDE_B + (size_t)(i / R) * 16u);
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int g = i / R, r = i % R;
            if (i + AHEAD < NF) {
                const int gn = (i + AHEAD) / R, rn = (i + AHEAD) % R;
                ah[(i + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf_b16x8*>(tile_hi + off0 + (size_t)rn * 32u * STRIDE_B + (size_t)gn * 16u);
                al[(i + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf_b16x8*>(tile_lo + off0 + (size_t)rn * 32u * STRIDE_B + (size_t)gn * 16u);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i % (AHEAD + 1)], qh[g], acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i % (AHEAD + 1)], ql[g], acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i % (AHEAD + 1)], qh[g], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // result block r: acc[r][v] ~ dot(element e0 + r*32 + 8*(v/4) + 4*h + v%4, query `col` of this wave)
        const bool whole = e0 + ET <= r1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (PRIME) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    best = (whole || e < r1) ? __builtin_fmaxf(best, acc[r][v]) : best;
                }
                continue;
            }
            float mx = acc[r][0];
#pragma unroll
            for (int v = 1; v < 16; ++v) mx = __builtin_fmaxf(mx, acc[r][v]);
            if (__ballot(mx > tau)) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float sc = acc[r][v];
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    if (sc > tau && (whole || e < r1)) {
                        L.insert(sc, (uint32_t)e);
                        tau = __builtin_fmaxf(tau, L.worst());
                    }
                }
            }
        }
    }
    if constexpr (PRIME) bf_write_max(P, best, blk.range, q, qlive, h);
    else bf_write_list(P, L, blk.range, q, qlive, h);
}

// f32 rows of ANY length on the bf16 path (round 6: bf_b16_kernel keeps a lane's half of its query in registers for the
// whole scan, which ends at 256 dims). Here the vector is walked in chunks of 128 components: the accumulators of a
// tile of 64 rows stay in registers across the chunks, a chunk of the tile goes HBM -> registers -> LDS as in
// bf_b16_kernel, and a wave reads its 32 queries' pieces of the chunk again for every tile (from L2: 16 KB per wave and
// chunk against 12 Mflop of matrix work). Slower per flop than the resident form -- two barriers and a query read per
// 128 components -- and there for the dims it cannot take (768-d: bench.py's recall ground truth on embeddings).
template <int R, bool PRIME = false>
__global__ __launch_bounds__(BF_B16_THREADS) void bf_b16_chunked_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr int KG = 8;                           // groups of 8 components per lane half and chunk
    constexpr uint32_t ET = 32u * R;                // elements per tile
    constexpr uint32_t COMPS = 16u * KG;            // components per chunk
    constexpr uint32_t STRIDE_B = 2u * COMPS + 16u; // bytes per LDS row of one piece: an odd number of 16-byte units
    uint8_t* tile_hi = smem_bf;
    uint8_t* tile_lo = smem_bf + (size_t)ET * STRIDE_B;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const BfBlock blk = bf_block();
    const uint32_t q = blk.qt * BF_B16_QT + wave * 32u + col;
    const bool qlive = q < P.nq;
    const float* qp = reinterpret_cast<const float*>(P.queries) + (size_t)(qlive ? q : 0u) * P.dim;
    const uint32_t nchunks = (P.dim + COMPS - 1u) / COMPS;
    const bool qvec_ok = (P.dim & 3u) == 0u && (reinterpret_cast<uintptr_t>(P.queries) & 15u) == 0u; // every query starts on 16 bytes
    BfList<PRIME ? 1 : BF_KMAX> L;
    L.init();
    float tau = bf_start_tau(P, q, qlive);
    [[maybe_unused]] float best = -3.0e38f;

    const uint64_t r0 = (uint64_t)blk.range * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    const uint32_t row_f4 = P.row_bytes / 16u;      // float4 units of a device row that hold data
    constexpr uint32_t UNITS = COMPS / 4u;          // float4 units per row and chunk
    constexpr uint32_t NPF = (ET * UNITS + BF_B16_THREADS - 1u) / BF_B16_THREADS;
    float4 pf[NPF];
    auto fetch = [&](uint64_t e0, uint32_t c) { // chunk c of the tile at e0
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t u = tid + BF_B16_THREADS * j;
            const uint32_t row = u / UNITS, c4 = c * UNITS + (u - row * UNITS);
            pf[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (u < ET * UNITS && e0 + row < r1 && c4 < row_f4)
                pf[j] = *reinterpret_cast<const float4*>(P.elements + (e0 + row) * P.row_stride + (size_t)c4 * 16u);
        }
    };
    if (r0 < r1) fetch(r0, 0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ET) {
        bf_f32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][v] = 0.0f;
#pragma unroll 1
        for (uint32_t c = 0; c < nchunks; ++c) {
            __syncthreads(); // the previous chunk has been consumed
#pragma unroll
            for (uint32_t j = 0; j < NPF; ++j) { // split into the two pieces on the way to LDS
                const uint32_t u = tid + BF_B16_THREADS * j;
                const uint32_t row = u / UNITS, c4 = u - row * UNITS;
                if (u < ET * UNITS) {
                    bf_b16x4 hi, lo;
                    const float v[4] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        hi[t] = (__bf16)v[t];
                        lo[t] = (__bf16)(v[t] - (float)hi[t]);
                    }
                    *reinterpret_cast<bf_b16x4*>(tile_hi + (size_t)row * STRIDE_B + c4 * 8u) = hi;
                    *reinterpret_cast<bf_b16x4*>(tile_lo + (size_t)row * STRIDE_B + c4 * 8u) = lo;
                }
            }
            __syncthreads();
            if (c + 1u < nchunks) fetch(e0, c + 1u);
            else if (e0 + ET < r1) fetch(e0 + ET, 0);
            // the lane's half of its query's chunk as bf16 pieces
            bf_b16x8 qh[KG], ql[KG];
            const bool q_vec = qvec_ok && (c + 1u) * COMPS <= P.dim; // whole chunk inside the vector, 16-byte aligned: 16 loads instead of 64
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const uint32_t c0 = c * COMPS + h * 8u * KG + (uint32_t)g * 8u;
                float v8[8];
                if (q_vec) {
                    const float4 a = *reinterpret_cast<const float4*>(qp + c0), b = *reinterpret_cast<const float4*>(qp + c0 + 4u);
                    v8[0] = a.x, v8[1] = a.y, v8[2] = a.z, v8[3] = a.w, v8[4] = b.x, v8[5] = b.y, v8[6] = b.z, v8[7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = c0 + (uint32_t)j < P.dim ? qp[c0 + (uint32_t)j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = qlive ? v8[j] : 0.0f;
                    const __bf16 hi = (__bf16)v;
                    qh[g][j] = hi;
                    ql[g][j] = (__bf16)(v - (float)hi);
                }
            }
            const size_t off0 = (size_t)col * STRIDE_B + (size_t)h * 16u * KG;
            constexpr int NF = KG * R;
            bf_b16x8 ah[2], al[2];
            ah[0] = *reinterpret_cast<const bf_b16x8*>(tile_hi + off0);
            al[0] = *reinterpret_cast<const bf_b16x8*>(tile_lo + off0);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int g = i / R, r = i % R;
                if (i + 1 < NF) {
                    const int gn = (i + 1) / R, rn = (i + 1) % R;
                    ah[(i + 1) & 1] = *reinterpret_cast<const bf_b16x8*>(tile_hi + off0 + (size_t)rn * 32u * STRIDE_B + (size_t)gn * 16u);
                    al[(i + 1) & 1] = *reinterpret_cast<const bf_b16x8*>(tile_lo + off0 + (size_t)rn * 32u * STRIDE_B + (size_t)gn * 16u);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], qh[g], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], ql[g], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], qh[g], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // result block r: acc[r][v] ~ dot(element e0 + r*32 + 8*(v/4) + 4*h + v%4, query `col` of this wave)
        const bool whole = e0 + ET <= r1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (PRIME) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    best = (whole || e < r1) ? __builtin_fmaxf(best, acc[r][v]) : best;
                }
                continue;
            }
            float mx = acc[r][0];
#pragma unroll
            for (int v = 1; v < 16; ++v) mx = __builtin_fmaxf(mx, acc[r][v]);
            if (__ballot(mx > tau)) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float sc = acc[r][v];
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    if (sc > tau && (whole || e < r1)) {
                        L.insert(sc, (uint32_t)e);
                        tau = __builtin_fmaxf(tau, L.worst());
                    }
                }
            }
        }
    }
    if constexpr (PRIME) bf_write_max(P, best, blk.range, q, qlive, h);
    else bf_write_list(P, L, blk.range, q, qlive, h);
}

// int8: device rows of up to 128 bytes (dims up to 128; the scan refuses longer rows). K = 32 per MFMA
// (v_mfma_i32_32x32x32_i8, gfx950): lanes 0-31 carry bytes 0-15 of a 32-byte group, lanes 32-63 bytes 16-31.
// Score = dot / (|x| |q|) with the exact integer dot.
template <int R, bool PRIME = false>
__global__ __launch_bounds__(BF_THREADS) void bf_i8_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t ET = 32u * R;        // rows per tile: what one pass of the matrix cores scores
    constexpr uint32_t SUB = BF_I8_SUB;     // tiles per stage: what travels HBM -> registers -> LDS between two barriers
    constexpr uint32_t ST = ET * SUB;
    constexpr uint32_t STRIDE = 128u + 16u; // bytes per LDS row: an odd number of 16-byte units
    uint8_t* tile = smem_bf;
    float* inv = reinterpret_cast<float*>(smem_bf + (size_t)ST * STRIDE); // [ST] 1 / |x|
    float* gm = inv + ST;                                                   // [SUB][2][R] inv_gmax of the tiles' blocks, by lane half
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const BfBlock blk = bf_block();
    const uint32_t q = blk.qt * BF_QT + wave * 32u + col;
    const bool qlive = q < P.nq;
    bf_i32x4 qr[4]; // bytes 32*g + 16*h .. +15 of the query, g = 0..3 (v_mfma_i32_32x32x32_i8: lanes 0-31 carry K 0-15 of a step, lanes 32-63 K 16-31)
    float qinv = 0.0f;
    {
        // (every byte read unconditionally at a clamped index, then selected: 64 loads in flight instead of 64 round trips;
        // |q|^2 from the packed words -- the byte-by-byte forms cost each block ~60 us before its first tile)
        const uint8_t* qp = P.queries + (size_t)(qlive ? q : 0u) * P.dim;
        const uint32_t last = P.dim - 1u;
        int dy = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t c = (uint32_t)g * 32u + h * 16u + (uint32_t)w * 4u + (uint32_t)b;
                    const uint32_t raw = qp[c < last ? c : last];
                    const uint32_t keep = (qlive && c < P.dim) ? 0xFFu : 0u; // (a mask, not a select: a select pulls the load under a branch)
                    v |= (raw & keep) << (8 * b);
                }
                qr[g][w] = (int)v;
                dy = dot4_i8(v, v, dy);
            }
        }
        dy += __shfl_xor(dy, 32, 64); // the other half of the vector
        qinv = dy > 0 ? 1.0f / __builtin_sqrtf((float)dy) : 0.0f;
    }
    BfList<PRIME ? 1 : BF_KMAX> L; // scores WITHOUT the query's 1 / |q| (one factor per lane: it does not change the order)
    L.init();
    [[maybe_unused]] float best = -3.0e38f; // PRIME: the largest score of the range, nothing else
    float tau = bf_start_tau(P, q, qlive);
    tau = (tau > -1.0e38f && qinv > 0.0f) ? bf_next_below(tau / qinv) : -3.0e38f; // the primed threshold in the lists' unit
    [[maybe_unused]] BfShare share; // (every lane of a query derives the same `tau` here: one histogram scale per query)
    if constexpr (!PRIME) share.init(P, qlive, tau);
    const uint64_t r0 = (uint64_t)blk.range * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    // 8 threads per row (16 bytes each; device rows shorter than 128 bytes are zero extended); the tile of the NEXT
    // step travels to registers while this one is scored. (Round 4 took the rows' norms here, from the same bytes: an
    // integer sum, three cross-lane adds, a correctly rounded square root and division per row and per QUERY TILE -- 1.5 of
    // the scan's 8.5 ms for a quantity that belongs to the index.)
    constexpr uint32_t NPF = (ST * 8u + BF_THREADS - 1u) / BF_THREADS;
    const uint32_t row_u4 = P.row_bytes / 16u;
    uint4 pf[NPF];
    float pfn[NPF]; // the row's 1 / |x| travels with its first 16 bytes
    [[maybe_unused]] float pfg = 0.0f; // threads 0 .. 2 R SUB - 1: inv_gmax of the stage's block tid / 2, half tid & 1
    // (a stage wholly inside the range -- all but the last -- loads without per-row bounds: one uniform base, 32-bit
    // offsets; the guarded form's 64-bit address products and zero selects were a third of the instructions of a stage)
    static_assert((ST * 8u) % BF_THREADS == 0u, "every thread carries NPF units of a stage");
    const uint32_t my_row = tid >> 3, my_c = tid & 7u; // unit j: row my_row + 64 j, 16-byte column my_c
    const bool col_live = my_c < row_u4;
    auto fetch = [&](uint64_t e0) {
        if constexpr (!PRIME) {
            if (tid < 2u * R * SUB) pfg = e0 + 32u * (tid >> 1) < r1 ? P.inv_gmax[((e0 >> 5) + (tid >> 1)) * 2u + (tid & 1u)] : 0.0f;
        }
        const uint8_t* base = P.elements + e0 * P.row_bytes;
        const float* nbase = P.inv_norm + e0;
        if (e0 + ST <= r1) {
#pragma unroll
            for (uint32_t j = 0; j < NPF; ++j) {
                const uint32_t row = my_row + (BF_THREADS / 8u) * j;
                pf[j] = make_uint4(0, 0, 0, 0);
                pfn[j] = 0.0f;
                if (col_live) pf[j] = *reinterpret_cast<const uint4*>(base + row * P.row_bytes + my_c * 16u);
                if (my_c == 0u) pfn[j] = nbase[row];
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < NPF; ++j) {
                const uint32_t row = my_row + (BF_THREADS / 8u) * j;
                pf[j] = make_uint4(0, 0, 0, 0);
                pfn[j] = 0.0f;
                if (e0 + row < r1 && col_live) pf[j] = *reinterpret_cast<const uint4*>(base + row * P.row_bytes + my_c * 16u);
                if (e0 + row < r1 && my_c == 0u) pfn[j] = nbase[row];
            }
        }
    };
    fetch(r0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ST) {
        __syncthreads();
        if constexpr (!PRIME) {
            if (P.share_hist) tau = __builtin_fmaxf(tau, share.poll(P, q, h));
        }
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t row = my_row + (BF_THREADS / 8u) * j;
            *reinterpret_cast<uint4*>(tile + (size_t)row * STRIDE + my_c * 16u) = pf[j];
            if (my_c == 0u) inv[row] = pfn[j];
        }
        if constexpr (!PRIME) {
            if (tid < 2u * R * SUB) gm[((tid >> 1) / R) * 2u * R + (tid & 1u) * R + (tid >> 1) % R] = pfg;
        }
        __syncthreads();
#if GRANNE_BF_EXP != 1 // (diagnostic builds, tools/build_variant.sh: 1 = no row traffic after the first tile, 2 = no matrix work, 3 = no fragment reads)
        if (e0 + ST < r1) fetch(e0 + ST);
#endif
        // (one stage of SUB tiles per pair of barriers: with one tile the block's load round trip -- ~0.9 us of every 2 us
        // tile, nothing of it hidden by the matrix work between the same two barriers -- was paid per 128 rows)
#pragma unroll 1
        for (uint32_t sub = 0; sub < SUB; ++sub) {
        const uint64_t es = e0 + (uint64_t)sub * ET;
        if (es >= r1) break;
        const uint8_t* tile_s = tile + (size_t)sub * ET * STRIDE;
        const float* inv_s = inv + sub * ET;
        const float* gm_s = gm + sub * 2u * R;
        bf_i32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][v] = 0;
        // 16 bytes per lane and step: one conflict-free ds_read_b128 down a column of rows. The R fragments of step g + 1 are
        // on their way while the matrix cores take the R independent products of step g. The scheduler is held to that
        // order: left alone it sank every read to just before its MFMA and paired the two MFMAs of one accumulator back to
        // back -- 16 LDS latencies and 8 dependent-issue stalls per tile and wave, 4,900 clocks per tile where the matrix
        // cores need ~1,000 (rocprofv3: 2.5 ms of the 2.9 ms scan).
        bf_i32x4 afrag[2][R];
#if GRANNE_BF_EXP == 2
        if (tid == 0xFFFFu)
#endif
        {
#pragma unroll
        for (int r = 0; r < R; ++r) afrag[0][r] = *reinterpret_cast<const bf_i32x4*>(tile_s + (size_t)(r * 32 + col) * STRIDE + h * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4 && (GRANNE_BF_EXP != 3)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    afrag[(g + 1) & 1][r] = *reinterpret_cast<const bf_i32x4*>(tile_s + (size_t)(r * 32 + col) * STRIDE + (g + 1) * 32 + h * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[GRANNE_BF_EXP == 3 ? 0 : (g & 1)][r], qr[g], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // Almost no block holds a score that beats a lane's threshold, and with the queries' shared threshold (BfShare) that
        // threshold is sharp: the largest integer dot of the lane's 16 rows times the largest 1 / |x| among them (per index:
        // inv_gmax) bounds every score of the block from above -- 8 v_max3_i32, one conversion, one product per block
        // instead of 16 + 16 + 8 and four LDS reads. The float scores are made only where the bound passes. (The norms of
        // quantized rows spread by +-10 %, so the bound sits ~0.35 sigma of the score distribution below the scores: with
        // thresholds per (range, lane), at 3.3-3.7 sigma, nearly every block passed it -- round 4's finding; at 4.6 one in
        // a hundred does.) Negative dots: their scores are <= 0 <= the bound (max with 0).
        const bool whole = es + ET <= r1;
        [[maybe_unused]] float gmv[R];
        if constexpr (!PRIME) {
#pragma unroll
            for (int r = 0; r < R; ++r) gmv[r] = gm_s[h * R + r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (!PRIME) {
                int im = 0;
#pragma unroll
                for (int v = 0; v < 16; ++v) im = max(im, acc[r][v]);
                if (!__ballot((float)im * gmv[r] > tau)) continue;
            }
            float sc[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { // elements r*32 + 8*g4 + 4*h + 0..3: their 1 / |x| in one read
                const float4 iv = *reinterpret_cast<const float4*>(inv_s + r * 32 + 8 * g4 + 4 * h);
                sc[g4 * 4 + 0] = (float)acc[r][g4 * 4 + 0] * iv.x;
                sc[g4 * 4 + 1] = (float)acc[r][g4 * 4 + 1] * iv.y;
                sc[g4 * 4 + 2] = (float)acc[r][g4 * 4 + 2] * iv.z;
                sc[g4 * 4 + 3] = (float)acc[r][g4 * 4 + 3] * iv.w;
            }
            if constexpr (PRIME) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint64_t e = es + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    best = (whole || e < r1) ? __builtin_fmaxf(best, sc[v]) : best;
                }
                continue;
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const uint64_t e = es + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                if (sc[v] > tau && (whole || e < r1)) {
                    L.insert(sc[v], (uint32_t)e);
                    tau = __builtin_fmaxf(tau, L.worst());
                    if (share.on()) share.count(P, q, sc[v]);
                }
            }
        }
        } // sub
    }
    if constexpr (PRIME) bf_write_max(P, best * qinv, blk.range, q, qlive, h);
    else bf_write_list(P, L, blk.range, q, qlive, h, qinv);
}

// int8 rows of MORE than 128 bytes (round 6; bf_i8_kernel and the ring keep a lane's part of its query in registers for the
// whole scan: 128 bytes). The row is walked in chunks of 128 bytes: the integer accumulators of a tile of 32 R rows stay in
// registers across the chunks, a chunk of the tile goes HBM -> registers -> LDS as in bf_i8_kernel, and a wave reads its 32
// queries' 64 bytes per lane of the chunk again for every tile -- from a zero-padded, 16-byte aligned copy of the queries
// (BruteParams::qpad: [nq][row_bytes], made per call by bf_pad_queries_kernel). Norms, block bounds, shared threshold and
// lists as in bf_i8_kernel.
__global__ void bf_pad_queries_kernel(const uint8_t* __restrict__ queries, uint32_t nq, uint32_t dim, uint32_t row_bytes, uint8_t* __restrict__ out) {
    const uint64_t total = (uint64_t)nq * row_bytes;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t qi = (uint32_t)(t / row_bytes), c = (uint32_t)(t - (uint64_t)qi * row_bytes);
        out[t] = c < dim ? queries[(size_t)qi * dim + c] : (uint8_t)0;
    }
}

template <int R, bool PRIME = false>
__global__ __launch_bounds__(BF_THREADS) void bf_i8_chunked_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t ET = 32u * R;
    constexpr uint32_t STRIDE = 128u + 16u; // bytes per LDS row: an odd number of 16-byte units
    uint8_t* tile = smem_bf;
    float* inv = reinterpret_cast<float*>(smem_bf + (size_t)ET * STRIDE); // [ET] 1 / |x|
    float* gm = inv + ET;                                                   // [2][R] inv_gmax of the tile's blocks, by lane half
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const BfBlock blk = bf_block();
    const uint32_t q = blk.qt * BF_QT + wave * 32u + col;
    const bool qlive = q < P.nq;
    const uint32_t nchunks = P.row_bytes / 128u; // (device rows of more than 128 bytes are whole 128-byte blocks, zero padded)
    const uint8_t* qrow = P.qpad + (size_t)(qlive ? q : 0u) * P.row_bytes;
    float qinv = 0.0f;
    {
        int dy = 0;
        for (uint32_t u = h; u < P.row_bytes / 16u; u += 2u) {
            const uint4 v = *reinterpret_cast<const uint4*>(qrow + (size_t)u * 16u);
            dy = dot4_i8(v.x, v.x, dy);
            dy = dot4_i8(v.y, v.y, dy);
            dy = dot4_i8(v.z, v.z, dy);
            dy = dot4_i8(v.w, v.w, dy);
        }
        dy += __shfl_xor(dy, 32, 64);
        qinv = (qlive && dy > 0) ? 1.0f / __builtin_sqrtf((float)dy) : 0.0f;
    }
    BfList<PRIME ? 1 : BF_KMAX> L; // scores WITHOUT the query's 1 / |q|
    L.init();
    [[maybe_unused]] float best = -3.0e38f;
    float tau = bf_start_tau(P, q, qlive);
    tau = (tau > -1.0e38f && qinv > 0.0f) ? bf_next_below(tau / qinv) : -3.0e38f;
    [[maybe_unused]] BfShare share;
    if constexpr (!PRIME) share.init(P, qlive, tau);
    const uint64_t r0 = (uint64_t)blk.range * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    constexpr uint32_t NPF = (ET * 8u + BF_THREADS - 1u) / BF_THREADS;
    static_assert((ET * 8u) % BF_THREADS == 0u, "every thread carries NPF units of a chunk");
    uint4 pf[NPF];
    float pfn[NPF];
    [[maybe_unused]] float pfg = 0.0f;
    const uint32_t my_row = tid >> 3, my_c = tid & 7u;
    auto fetch = [&](uint64_t e0, uint32_t c) {
        if (c == 0u) {
            if constexpr (!PRIME) {
                if (tid < 2u * R) pfg = e0 + 32u * (tid >> 1) < r1 ? P.inv_gmax[((e0 >> 5) + (tid >> 1)) * 2u + (tid & 1u)] : 0.0f;
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t row = my_row + (BF_THREADS / 8u) * j;
            pf[j] = make_uint4(0, 0, 0, 0);
            if (e0 + row < r1) pf[j] = *reinterpret_cast<const uint4*>(P.elements + (e0 + row) * P.row_stride + (size_t)c * 128u + my_c * 16u);
            if (c == 0u) pfn[j] = (e0 + row < r1 && my_c == 0u) ? P.inv_norm[e0 + row] : 0.0f;
        }
    };
    if (r0 < r1) fetch(r0, 0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ET) {
        bf_i32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][v] = 0;
#pragma unroll 1
        for (uint32_t c = 0; c < nchunks; ++c) {
            __syncthreads(); // the previous chunk (and the previous tile's norms) have been consumed
            if constexpr (!PRIME) {
                if (c == 0u && P.share_hist) tau = __builtin_fmaxf(tau, share.poll(P, q, h));
            }
#pragma unroll
            for (uint32_t j = 0; j < NPF; ++j) {
                const uint32_t row = my_row + (BF_THREADS / 8u) * j;
                *reinterpret_cast<uint4*>(tile + (size_t)row * STRIDE + my_c * 16u) = pf[j];
                if (c == 0u && my_c == 0u) inv[row] = pfn[j];
            }
            if constexpr (!PRIME) {
                if (c == 0u && tid < 2u * R) gm[(tid & 1u) * R + (tid >> 1)] = pfg;
            }
            __syncthreads();
            if (c + 1u < nchunks) fetch(e0, c + 1u);
            else if (e0 + ET < r1) fetch(e0 + ET, 0);
            // the lane's 64 bytes of its query's chunk: bytes 32 g + 16 h .. + 15, g = 0..3
            bf_i32x4 qr[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf_i32x4 v = *reinterpret_cast<const bf_i32x4*>(qrow + (size_t)c * 128u + (uint32_t)g * 32u + h * 16u);
                qr[g] = qlive ? v : bf_i32x4{0, 0, 0, 0};
            }
            bf_i32x4 afrag[2][R];
#pragma unroll
            for (int r = 0; r < R; ++r) afrag[0][r] = *reinterpret_cast<const bf_i32x4*>(tile + (size_t)(r * 32 + col) * STRIDE + h * 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        afrag[(g + 1) & 1][r] = *reinterpret_cast<const bf_i32x4*>(tile + (size_t)(r * 32 + col) * STRIDE + (g + 1) * 32 + h * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[g & 1][r], qr[g], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const bool whole = e0 + ET <= r1;
        [[maybe_unused]] float gmv[R];
        if constexpr (!PRIME) {
#pragma unroll
            for (int r = 0; r < R; ++r) gmv[r] = gm[h * R + r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (!PRIME) {
                int im = 0;
#pragma unroll
                for (int v = 0; v < 16; ++v) im = max(im, acc[r][v]);
                if (!__ballot((float)im * gmv[r] > tau)) continue;
            }
            float sc[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 iv = *reinterpret_cast<const float4*>(inv + r * 32 + 8 * g4 + 4 * h);
                sc[g4 * 4 + 0] = (float)acc[r][g4 * 4 + 0] * iv.x;
                sc[g4 * 4 + 1] = (float)acc[r][g4 * 4 + 1] * iv.y;
                sc[g4 * 4 + 2] = (float)acc[r][g4 * 4 + 2] * iv.z;
                sc[g4 * 4 + 3] = (float)acc[r][g4 * 4 + 3] * iv.w;
            }
            if constexpr (PRIME) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    best = (whole || e < r1) ? __builtin_fmaxf(best, sc[v]) : best;
                }
                continue;
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                if (sc[v] > tau && (whole || e < r1)) {
                    L.insert(sc[v], (uint32_t)e);
                    tau = __builtin_fmaxf(tau, L.worst());
                    if (share.on()) share.count(P, q, sc[v]);
                }
            }
        }
    }
    if constexpr (PRIME) bf_write_max(P, best * qinv, blk.range, q, qlive, h);
    else bf_write_list(P, L, blk.range, q, qlive, h, qinv);
}

// ---- int8 rows of 128 bytes: the tiles travel HBM -> LDS by LDS-DMA into a ring, two query sets per wave (round 6) ----
// What bounded bf_i8_kernel (profiles/r5_bruteforce_i8_*): a stage's rows went HBM -> registers -> LDS between two
// barriers, with ONE stage in flight per block -- the staging alone took 1.1 ms of the 2.6 ms scan, the matrix work 0.6, and
// the two did not overlap (waves parked 56 %). Here
//   * a tile (128 rows = 16 KB) is written to LDS by the memory system itself (global_load_lds_dwordx4: 1 KB per wave
//     instruction, no registers, no ds_write), into a ring of BF_RING_STAGES tiles: three tiles (48 KB per CU) are on
//     their way while one is scored, ONE barrier per tile, and a wave waits for its own part of the next tile only
//     (s_waitcnt vmcnt(N) counted by hand: the compiler does not see the transfers -- they are issued from asm -- and so
//     does not drain them before every LDS read, which is what it does for the builtin);
//   * LDS-DMA writes a wave's 64 x 16 bytes in lane order, so a row cannot be padded: the 16-byte chunk c of tile row r
//     stands in slot c ^ ((r >> 1) & 7) of the row's 128 bytes (the swizzle is applied to the SOURCE address of each
//     lane and to the fragment reads): the 16 lanes that ds_read_b128 serves per cycle hit 16 different slots of the
//     256-byte bank row;
//   * a wave holds TWO sets of 32 queries (64 per wave, 512 per block): every fragment read from LDS feeds two matrix
//     instructions, and half as many blocks stream every range (L2 -> LDS traffic halves with it);
//   * the rows' 1 / |x| and the block bounds (inv_gmax) ride in a ring of their own (one 4-byte LDS-DMA per wave and
//     tile, 16 + 2 lanes): one block in fifty passes its bound and reads its norms, and an ordinary vector load there
//     made the compiler wait for everything the ring has in flight (5.3 ms);
//   * the running top-16 of a query is a SET in LDS, one per query for both lane halves (64 KB; 64 registers back, and
//     a query's threshold is that of both halves' rows): an element takes the place of the worst one and the new worst
//     is looked up (16 independent reads); the set is put in order once, when it is written out. (Through a generic
//     `volatile` pointer these were FLAT instructions, counted by vmcnt like the transfers: the pointers carry their
//     address space.) The path that puts candidates in is written for latency -- a block's eight waves meet at the
//     barrier of every tile, ~3 visits per tile among them: the float scores of a lane's 16 rows at once, one bit
//     each, then one candidate per lane and round (a loop over the 16 scores with a norm read and a branch each:
//     2.8 ms; unrolled with the insert at each of the 16: 200 KB of code, 2.6 ms).
// 1024 x 10M x 100-d: 2.6 -> 1.75-2.0 ms for the whole operator (the kernel 1.55 ms). What is left, measured
// (tools/mfma_lds_loop.hip, tools/mfma_rate.hip): this loop alone, fragments from LDS, no transfers, takes 1,400 clocks
// per tile and wave (44 per matrix instruction; two waves of a SIMD interleave theirs -- 17 clocks per instruction and
// SIMD where one wave alone issues one per 32) -- and the chip answers that rate on random int8 data with a shader clock
// of 0.9-1.1 GHz: 0.95 ms for this scan's 2.6e15 operations, 1.1 ms with the transfers, whatever the structure. Counters
// per ring slot instead of the barrier (a wave could fall a tile behind) changed nothing; neither did two tiles in
// flight instead of three, nor a fifth slot.
// The priming pass stays bf_i8_kernel<4, true>. Rows shorter than 128 bytes keep bf_i8_kernel.
constexpr uint32_t BF_RING_THREADS = 512, BF_RING_QT = 512, BF_RING_STAGES = 4, BF_RING_ROWS = 128;
constexpr uint32_t BF_RING_LDS = BF_RING_STAGES * BF_RING_ROWS * 128u + (BF_KMAX * 512u * 8u + 512u * 8u) + BF_RING_STAGES * 8u * 80u; // tiles + sets (BF_SET_BYTES) + norms
constexpr uint32_t BF_RING_TILE_BYTES = BF_RING_ROWS * 128u;
typedef const __attribute__((address_space(4))) float* bf_cptr_f32;
// (volatile accesses through a generic pointer stay FLAT instructions -- counted by vmcnt, so each made the compiler wait for
// the whole ring: the lists' pointers carry their address space)
typedef volatile __attribute__((address_space(3))) float* bf_lds_f32;
typedef volatile __attribute__((address_space(3))) uint32_t* bf_lds_u32;

// 64 lanes x 16 bytes from base + voff (per lane) to LDS at lds_dst + 16 * lane (lds_dst, base: wave-uniform)
__device__ __forceinline__ void bf_dma16(uint32_t lds_dst, const uint8_t* base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
}

// the active lanes' 4 bytes from base + voff (per lane) to LDS at lds_dst + 4 * lane
__device__ __forceinline__ void bf_dma4(uint32_t lds_dst, const float* base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
}

// the lane's half of query q as the B operand of four K = 32 steps, and 1 / |q|
__device__ __forceinline__ void bf_i8_load_query(const BruteParams& P, uint32_t q, bool qlive, uint32_t h, bf_i32x4 (&qr)[4], float& qinv) {
    const uint8_t* qp = P.queries + (size_t)(qlive ? q : 0u) * P.dim;
    const uint32_t last = P.dim - 1u;
    int dy = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t c = (uint32_t)g * 32u + h * 16u + (uint32_t)w * 4u + (uint32_t)b;
                const uint32_t raw = qp[c < last ? c : last];
                const uint32_t keep = (qlive && c < P.dim) ? 0xFFu : 0u;
                v |= (raw & keep) << (8 * b);
            }
            qr[g][w] = (int)v;
            dy = dot4_i8(v, v, dy);
        }
    }
    dy += __shfl_xor(dy, 32, 64);
    qinv = dy > 0 ? 1.0f / __builtin_sqrtf((float)dy) : 0.0f;
}

// The running top-16 of a block's 512 queries as SETS in LDS (bf_i8_ring_kernel): entry i of query ql
// at [i][ql], ONE set per query for both lane halves (64 KB), beside it the set's worst score and its place (4 KB). An
// element takes the place of the worst one and the new worst is looked up -- 16 independent reads, no chain of shifts; the
// set is put in order once, when it is written out.
constexpr uint32_t BF_SET_QT = 512;                                          // queries per block
static_assert(BF_SET_QT == BF_RING_QT, "the ring kernel's blocks hold BF_SET_QT queries");
constexpr uint32_t BF_SET_BYTES = BF_KMAX * BF_SET_QT * 8u + BF_SET_QT * 8u; // scores + ids + (worst, place)
struct BfSets {
    bf_lds_f32 ls; // [BF_KMAX][BF_SET_QT] scores (no particular order)
    bf_lds_u32 li; // [BF_KMAX][BF_SET_QT] element ids
    bf_lds_f32 lw; // [BF_SET_QT] the set's smallest score ...
    bf_lds_u32 lp; // ... and which entry holds it
    __device__ __forceinline__ void at(uint8_t* lds) {
        ls = (bf_lds_f32)lds;
        li = (bf_lds_u32)(lds + BF_KMAX * BF_SET_QT * 4u);
        lw = (bf_lds_f32)(lds + BF_KMAX * BF_SET_QT * 8u);
        lp = (bf_lds_u32)(lds + BF_KMAX * BF_SET_QT * 8u + BF_SET_QT * 4u);
    }
    __device__ __forceinline__ void clear(uint32_t ql) {
#pragma unroll
        for (int i = 0; i < (int)BF_KMAX; ++i) {
            ls[(uint32_t)i * BF_SET_QT + ql] = -3.0e38f;
            li[(uint32_t)i * BF_SET_QT + ql] = 0xFFFFFFFFu;
        }
        lw[ql] = -3.0e38f;
        lp[ql] = 0u;
    }
    // the smallest score of query ql's 16 entries and where it stands
    __device__ __forceinline__ float worst(uint32_t ql, uint32_t& pos) const {
        float sv[BF_KMAX];
#pragma unroll
        for (int i = 0; i < (int)BF_KMAX; ++i) sv[i] = ls[(uint32_t)i * BF_SET_QT + ql];
        float m = sv[0];
        pos = 0;
#pragma unroll
        for (int i = 1; i < (int)BF_KMAX; ++i) {
            const bool lower = sv[i] < m;
            m = lower ? sv[i] : m;
            pos = lower ? (uint32_t)i : pos;
        }
        return m;
    }
    // The scores sc16[v] of the lane's 16 rows of a 32-row block (row 8 (v / 4) + 4 h + v % 4; id0 = the id of the block's
    // row 0; a score that must not enter is NaN or -inf) against query ql's threshold `tau`: what beats it enters the set.
    // Written for LATENCY -- a block's eight waves meet at a barrier per tile and a few of them come here per tile, so the
    // tile takes as long as its slowest visit: one bit per score, then one candidate per lane and round (1.5 per visit;
    // the index differs between lanes: a ladder of selects); the insert exists once per place. The two lane halves of a
    // query share its set and take their turns (a wave's LDS operations complete in order). `share`: the int8 scan's
    // histogram of inserts, or null.
    template <typename Scores> // float[16] or a 16-float vector
    __device__ __forceinline__ void candidates(const BruteParams& P, const Scores& sc16, float& tau, uint32_t ql, uint32_t q, uint32_t h,
                                               uint32_t id0, BfShare* share) const {
        uint32_t cm = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) cm |= (sc16[v] > tau ? 1u : 0u) << v;
#pragma unroll 1
        while (__ballot(cm != 0u)) {
            const bool want = cm != 0u;
            const uint32_t v = want ? (uint32_t)__builtin_ctz(cm) : 0u;
            cm &= cm - 1u;
            float sc = sc16[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) sc = v == (uint32_t)j ? sc16[j] : sc;
#pragma unroll 1
            for (uint32_t hh = 0; hh < 2u; ++hh) {
                if (h == hh && want) {
                    tau = __builtin_fmaxf(tau, lw[ql]); // (the other half may have put something in)
                    if (sc > tau) {
                        uint32_t wpos = lp[ql];
                        ls[wpos * BF_SET_QT + ql] = sc;
                        li[wpos * BF_SET_QT + ql] = id0 + 8u * (v >> 2) + 4u * h + (v & 3u);
                        const float w = worst(ql, wpos);
                        lw[ql] = w;
                        lp[ql] = wpos;
                        tau = __builtin_fmaxf(tau, w);
                        if (share && share->on()) share->count(P, q, sc);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        tau = __builtin_fmaxf(tau, lw[ql]);
    }
    // the set of query ql in order -- an entry's rank among the 16 is its place -- as list (range, q); scores x scale
    __device__ __forceinline__ void write(const BruteParams& P, uint32_t ql, uint32_t range, uint32_t q, float scale) const {
        const size_t list = (size_t)range * P.nq + q;
        float sv[BF_KMAX];
        uint32_t iv[BF_KMAX];
#pragma unroll
        for (int i = 0; i < (int)BF_KMAX; ++i) {
            sv[i] = ls[(uint32_t)i * BF_SET_QT + ql];
            iv[i] = li[(uint32_t)i * BF_SET_QT + ql];
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < (int)BF_KMAX; ++i) {
            const bool ok = iv[i] != 0xFFFFFFFFu;
            uint32_t rank = 0; // entries that stand before this one: a larger score, or the same score and a smaller id
#pragma unroll
            for (int j = 0; j < (int)BF_KMAX; ++j)
                rank += (j != i && iv[j] != 0xFFFFFFFFu && (sv[j] > sv[i] || (sv[j] == sv[i] && iv[j] < iv[i]))) ? 1u : 0u;
            if (ok && rank < P.kk) {
                P.part_ids[list * P.kk + rank] = (uint64_t)iv[i];
                const float d = 1.0f - sv[i] * scale;
                P.part_d[list * P.kk + rank] = d > 0.0f ? d : 0.0f;
            }
            cnt += ok ? 1u : 0u;
        }
        cnt = cnt < P.kk ? cnt : P.kk;
        for (uint32_t i = cnt; i < P.kk; ++i) {
            P.part_ids[list * P.kk + i] = ~0ull;
            P.part_d[list * P.kk + i] = __builtin_inff();
        }
        P.part_c[list] = cnt;
    }
};

__global__ __launch_bounds__(BF_RING_THREADS) void bf_i8_ring_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t NS = BF_RING_STAGES, TR = BF_RING_ROWS, TB = BF_RING_TILE_BYTES;
    constexpr uint32_t A = NS - 1u; // tiles on their way while one is scored
    static_assert(NS == 4, "the waits below are written for three tiles in flight");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t col = lane & 31u, h = lane >> 5;
    const BfBlock blk = bf_block();
    uint32_t q[2];
    bool qlive[2];
    bf_i32x4 qr[2][4];
    float qinv[2], tau[2];
    BfShare share[2];
    BfSets sets; // the running top-16 of the block's 512 queries
    sets.at(smem_bf + NS * TB);
    // the tiles' norms, a ring like the rows': per tile 8 x [16 x 1 / |x| of rows 16 w ..][inv_gmax of block w, both lane halves][2 unused]
    constexpr uint32_t NB = 8u * 80u;
    const uint32_t norms0 = NS * TB + BF_SET_BYTES;
    uint32_t ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ql[s] = wave * 64u + (uint32_t)s * 32u + col;
        q[s] = blk.qt * BF_RING_QT + ql[s];
        qlive[s] = q[s] < P.nq;
        bf_i8_load_query(P, q[s], qlive[s], h, qr[s], qinv[s]);
        if (h == 0u) sets.clear(ql[s]);
        tau[s] = bf_start_tau(P, q[s], qlive[s]);
        tau[s] = (tau[s] > -1.0e38f && qinv[s] > 0.0f) ? bf_next_below(tau[s] / qinv[s]) : -3.0e38f;
        share[s].init(P, qlive[s], tau[s]);
    }
    const uint64_t r0 = (uint64_t)blk.range * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    const uint32_t T = r0 < r1 ? (uint32_t)((r1 - r0 + TR - 1u) / TR) : 0u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_bf;
    // LDS-DMA, three instructions per wave and tile. Rows: tile rows 16 wave + 8 j + (lane >> 3), slot lane & 7. Norms: lanes
    // 0-15 the rows' 1 / |x|, lanes 16-17 of waves 0-3 the block bound of row block `wave` (lane half lane - 16).
    const uint32_t n_pad = (uint32_t)((P.n + 31u) & ~31ull); // (the host takes this path for sets below 2^29 rows: 32-bit offsets)
    const bool dnorm = lane < (wave < 4u ? 18u : 16u);
    const uint32_t drow = wave * 16u + (lane >> 3), dslot = lane & 7u;
    const uint32_t dc0 = dslot ^ ((drow >> 1) & 7u), dc1 = dslot ^ (((drow + 8u) >> 1) & 7u);
    auto issue = [&](uint32_t t, uint32_t slot) {
        const uint64_t e0 = r0 + (uint64_t)t * TR;
        const uint8_t* base = P.elements + e0 * 128u;
        const uint32_t dst = lds0 + slot * TB + wave * 2048u;
        uint32_t v0 = drow * 128u + dc0 * 16u, v1 = (drow + 8u) * 128u + dc1 * 16u;
        if (e0 + TR > P.n) { // the set's last rows: a row past the end reads the last row instead (its scores are never taken)
            const uint32_t lastr = (uint32_t)(P.n - 1u - e0);
            v0 = (drow < lastr ? drow : lastr) * 128u + dc0 * 16u;
            v1 = (drow + 8u < lastr ? drow + 8u : lastr) * 128u + dc1 * 16u;
        }
        bf_dma16(dst, base, v0);
        bf_dma16(dst + 1024u, base, v1);
        if (dnorm) { // (inv_gmax stands behind inv_norm's n_pad entries in one allocation)
            const uint32_t e32 = (uint32_t)e0, row = e32 + wave * 16u + lane;
            const uint32_t idx = lane < 16u ? (row < n_pad ? row : n_pad - 1u) : n_pad + (e32 >> 5) * 2u + wave * 2u + (lane - 16u);
            bf_dma4(lds0 + norms0 + slot * NB + wave * 80u, P.inv_norm, idx * 4u);
        }
    };
    // fragment reads: row rb * 32 + col, chunk 2 g + h -> slot (2 g + h) ^ ((col >> 1) & 7)
    uint32_t aoff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) aoff[g] = col * 128u + (((2u * (uint32_t)g + h) ^ ((col >> 1) & 7u)) << 4);

    __syncthreads(); // (the lists are set)
    for (uint32_t t = 0; t < A && t < T; ++t) issue(t, t);
    uint32_t slot = 0;          // tile t stands in slot t % NS,
    uint32_t fslot = A % NS;    // tile t + A goes to slot (t + A) % NS: where tile t - 1 stood (A = NS - 1)
    for (uint32_t t = 0; t < T; ++t) {
        // this wave's part of tile t has landed (three transfers per tile): at most its parts of the A - 1 tiles after it are
        // still on their way
        const uint32_t rem = T - 1u - t;
        if (rem >= 2u) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (rem == 1u) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // every wave's part has; and every wave is done with tile t - 1, whose slot the next transfer takes
        if (t + A < T) issue(t + A, fslot);
        if (P.share_hist) {
            tau[0] = __builtin_fmaxf(tau[0], share[0].poll(P, q[0], h));
            tau[1] = __builtin_fmaxf(tau[1], share[1].poll(P, q[1], h));
        }
        const uint64_t es = r0 + (uint64_t)t * TR;
        const uint8_t* tile = smem_bf + slot * TB;
        const uint32_t lim = r1 - es < TR ? (uint32_t)(r1 - es) : TR; // rows of this tile inside the range (the set's last tile: < 128)
        const float* norms = reinterpret_cast<const float*>(smem_bf + norms0 + slot * NB);
        float gmv4[4]; // the bound of the lane's 16 rows of each row block
#pragma unroll
        for (int i = 0; i < 4; ++i) gmv4[i] = norms[i * 20 + 16 + (int)h];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bf_i32x16 acc[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[s][r][v] = 0;
            bf_i32x4 afrag[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) afrag[0][r] = *reinterpret_cast<const bf_i32x4*>(tile + aoff[0] + (uint32_t)(p * 2 + r) * 4096u);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        afrag[(g + 1) & 1][r] = *reinterpret_cast<const bf_i32x4*>(tile + aoff[g + 1] + (uint32_t)(p * 2 + r) * 4096u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        acc[s][r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[g & 1][r], qr[s][g], acc[s][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t rb = (uint32_t)(p * 2 + r);
                if (rb * 32u >= lim) continue; // a block wholly past the set's end (rows past it inside a block: their norm is NaN)
                const float gmv = gmv4[rb];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    int im = 0;
#pragma unroll
                    for (int v = 0; v < 16; ++v) im = max(im, acc[s][r][v]);
                    if (!__ballot((float)im * gmv > tau[s])) continue;
                    // One block in fifty comes here: the float scores of the lane's 16 rows at once (four reads of the norms in
                    // flight together; a row past the set's end has a NaN there and compares false), then BfSets::candidates.
                    // (A loop over the 16 scores with a norm read, a compare and a branch each: 3,000+ clocks per visit, the
                    // scan 2.8 ms.)
                    float sc16[16];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint32_t rw = rb * 32u + 8u * (uint32_t)g4 + 4u * h;
                        const float4 iv = *reinterpret_cast<const float4*>(norms + (rw >> 4) * 20u + (rw & 15u));
                        sc16[g4 * 4 + 0] = (float)acc[s][r][g4 * 4 + 0] * iv.x;
                        sc16[g4 * 4 + 1] = (float)acc[s][r][g4 * 4 + 1] * iv.y;
                        sc16[g4 * 4 + 2] = (float)acc[s][r][g4 * 4 + 2] * iv.z;
                        sc16[g4 * 4 + 3] = (float)acc[s][r][g4 * 4 + 3] * iv.w;
                    }
                    sets.candidates(P, sc16, tau[s], ql[s], q[s], h, (uint32_t)es + rb * 32u, &share[s]);
                }
            }
        }
        if (++slot == NS) slot = 0;
        if (++fslot == NS) fslot = 0;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (qlive[s] && h == 0u) sets.write(P, ql[s], blk.range, q[s], qinv[s]);
}

// 1 / |x| of every int8 row (0 for a zero row), once per index: eight lanes per 128-byte row
__global__ void inv_norm_rows_kernel(const uint8_t* __restrict__ elements, uint64_t n, uint32_t row_bytes, float* __restrict__ out) {
    const uint32_t c = threadIdx.x & 7u, units = row_bytes / 16u;
    for (uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; row < ((n + 31u) & ~31ull); row += ((uint64_t)gridDim.x * blockDim.x) >> 3) {
        int dx = 0;
        for (uint32_t u = c; u < units; u += 8u) { // (rows of more than 128 bytes: every lane takes every eighth unit)
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < n) v = *reinterpret_cast<const uint4*>(elements + row * row_bytes + (size_t)u * 16u);
            dx = dot4_i8(v.x, v.x, dx);
            dx = dot4_i8(v.y, v.y, dx);
            dx = dot4_i8(v.z, v.z, dx);
            dx = dot4_i8(v.w, v.w, dx);
        }
        dx += __shfl_xor(dx, 1, 64);
        dx += __shfl_xor(dx, 2, 64);
        dx += __shfl_xor(dx, 4, 64);
        // (rows n .. n_pad - 1: NaN -- bf_i8_ring_kernel scores a whole 32-row block and drops what does not compare)
        if (c == 0u) out[row] = row < n ? (dx > 0 ? 1.0f / __builtin_sqrtf((float)dx) : 0.0f) : __builtin_nanf("");
    }
}

// the largest 1 / |x| among the 16 rows that lane half h of a wave sees of the 32-row block b (rows 32b + 8g + 4h + 0..3,
// g = 0..3: the 32x32 MFMA result's row order); rows past the end count 0
__global__ void inv_gmax_kernel(const float* __restrict__ inv_norm, uint64_t n, float* __restrict__ out) {
    const uint64_t total = ((n + 31u) >> 5) * 2u; // (block, half) pairs
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = t >> 1;
        const uint32_t h = (uint32_t)(t & 1u);
        float m = 0.0f;
        for (uint32_t g = 0; g < 4; ++g)
            for (uint32_t i = 0; i < 4; ++i) {
                const uint64_t row = b * 32u + 8u * g + 4u * h + i;
                if (row < n) m = __builtin_fmaxf(m, inv_norm[row]);
            }
        out[t] = m;
    }
}

// The priming pass leaves the best score of each of its `ranges` sub-ranges per query; the kk-th largest of those is a
// score that kk DIFFERENT elements reach (the maxima of kk sub-ranges), so nothing below it is among the kk best of the
// whole set: the threshold the scan proper starts from (a few ulps of 1 lower: the exact re-ranking has the last word).
// Fewer than kk sub-ranges: no threshold.
// One wave per query, lane r holds sub-range r's maximum (at most 64 sub-ranges), a value's rank = the lanes that hold a
// larger one. (One thread per query ranked them with a 16-deep insertion per value: 26 us of a 1.7 ms scan.)
__global__ __launch_bounds__(64) void bf_tau_kernel(const float* __restrict__ maxima, uint32_t ranges, uint32_t nq, uint32_t kk, float* __restrict__ tau) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) return;
    if (ranges < kk || ranges > 64u) { // fewer sub-ranges than entries: no threshold
        if (lane == 0u) tau[q] = -3.0e38f;
        return;
    }
    const float v = lane < ranges ? maxima[(size_t)lane * nq + q] : -3.0e38f;
    uint32_t rank = 0; // values that stand before this one: larger, or equal in a lower lane
    for (uint32_t j = 0; j < ranges; ++j) {
        const float o = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)j));
        rank += (o > v || (o == v && j < lane)) ? 1u : 0u;
    }
    if (lane < ranges && rank + 1u == kk) tau[q] = v > -1.0e38f ? v - 2.0e-6f : -3.0e38f;
}

// merged candidates [nq][kk] u64 -> u32 ids for dists_kernel (entries beyond the count: UNUSED -> +inf)
__global__ void bf_narrow_ids_kernel(const uint64_t* __restrict__ ids, const uint32_t* __restrict__ counts, uint32_t nq,
                                     uint32_t kk, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * kk) return;
    const uint32_t qi = t / kk, j = t - qi * kk;
    out[t] = j < counts[qi] ? (uint32_t)ids[t] : 0xFFFFFFFFu;
}

// per query: the k best of its kk candidates by (exact distance, id), ascending -- Granne::search's order.
// One thread per (query, candidate): the candidate's rank among the query's candidates is its place in the output.
__global__ void bf_final_kernel(const uint32_t* __restrict__ cand, const float* __restrict__ exact, uint32_t nq, uint32_t kk,
                                uint32_t k, uint64_t* __restrict__ out_ids, float* __restrict__ out_d,
                                uint32_t* __restrict__ out_c) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * kk) return;
    const uint32_t qi = t / kk, j = t - qi * kk;
    const uint32_t* c = cand + (size_t)qi * kk;
    const float* d = exact + (size_t)qi * kk;
    const uint32_t my_id = c[j];
    const uint64_t mine = ((uint64_t)__float_as_uint(d[j]) << 32) | my_id;
    uint32_t rank = 0, valid = 0;
    for (uint32_t i = 0; i < kk; ++i) {
        const uint32_t id = c[i];
        const bool ok = id != 0xFFFFFFFFu;
        valid += ok ? 1u : 0u;
        const uint64_t other = ((uint64_t)__float_as_uint(d[i]) << 32) | id;
        rank += (ok && (other < mine || (other == mine && i < j))) ? 1u : 0u;
    }
    const uint32_t cnt = valid < k ? valid : k;
    if (my_id != 0xFFFFFFFFu && rank < k) {
        out_ids[(size_t)qi * k + rank] = my_id;
        out_d[(size_t)qi * k + rank] = d[j];
    }
    if (j == 0) {
        out_c[qi] = cnt;
        for (uint32_t r = cnt; r < k; ++r) {
            out_ids[(size_t)qi * k + r] = ~0ull;
            out_d[(size_t)qi * k + r] = __builtin_inff();
        }
    }
}

} // namespace granne_hip
