// brute_force.h -- exact k nearest elements of every query by scanning ALL elements: the one piece of granne's
// element side that is a real contraction (ElementContainer::dists for every index at once,
// /root/reference/src/elements/mod.rs:35-39 over src/elements/dense_vector.rs:157-163), so the one piece that
// belongs on the matrix cores. It is the recall ground truth of bench.py and an operator of its own
// (granne_hip_brute_force_device).
//
//   scores   v_mfma_f32_32x32x2_f32 (f32 rows) / v_mfma_i32_32x32x16_i8 (int8 rows: exact integer dots). The A operand
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
// Roofline: f32 is MFMA-bound (2 * nq * n * dim flops at 157 TFLOP/s dense f32: 13 ms for 1024 x 10M x 100), int8 is
// HBM-bound. bench.py reports the achieved rate; profiles/ holds the MFMA-busy counter.
#pragma once

#include "util_kernels.h"

namespace granne_hip {

typedef float bf_f32x16 __attribute__((ext_vector_type(16)));
typedef int bf_i32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t BF_QT = 256;    // queries per block (8 waves x 32: two per SIMD, one scores while the other is checked)
constexpr uint32_t BF_THREADS = 512;
constexpr uint32_t BF_EXTRA = 6;   // candidates selected beyond k, re-ranked by the exact distance
constexpr uint32_t BF_KMAX = 16;   // longest per-lane list (k + BF_EXTRA <= BF_KMAX)

struct BruteParams {
    const uint8_t* elements; // device rows: [n][row_bytes]
    uint64_t n;
    uint32_t row_bytes, dim;
    const uint8_t* queries;  // dense [nq][dim]
    uint32_t nq, kk;         // kk = k + BF_EXTRA entries per list
    uint64_t per_range;      // elements per range (a multiple of the tile)
    uint64_t* part_ids;      // [ranges][nq][kk]
    float* part_d;           // [ranges][nq][kk]
    uint32_t* part_c;        // [ranges][nq]
};

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
template <int KK>
__device__ __forceinline__ void bf_write_list(const BruteParams& P, BfList<KK>& L, uint32_t q, bool qlive, uint32_t h) {
#pragma unroll
    for (int i = 0; i < KK; ++i) {
        const float sc = __shfl_xor(L.s[i], 32, 64);
        const uint32_t e = (uint32_t)__shfl_xor((int)L.id[i], 32, 64);
        if (h == 0u && e != 0xFFFFFFFFu) L.insert(sc, e);
    }
    if (qlive && h == 0u) {
        const size_t list = (size_t)blockIdx.y * P.nq + q;
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < KK; ++i) {
            if ((uint32_t)i < P.kk) {
                const bool ok = L.id[i] != 0xFFFFFFFFu;
                P.part_ids[list * P.kk + i] = ok ? (uint64_t)L.id[i] : ~0ull;
                const float d = 1.0f - L.s[i];
                P.part_d[list * P.kk + i] = ok ? (d > 0.0f ? d : 0.0f) : __builtin_inff();
                cnt += ok ? 1u : 0u;
            }
        }
        P.part_c[list] = cnt;
    }
}

// f32: KH = K entries per half (vector components h*KH .. h*KH+KH-1, zero padded), R = 32-element blocks per tile
template <int KH, int R>
__global__ __launch_bounds__(BF_THREADS) void bf_f32_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t ET = 32u * R;        // elements per tile
    constexpr uint32_t STRIDE = 2u * KH + 4u; // floats per LDS row: 4 x odd -> conflict-free ds_read_b128 down a column
    float* tile = reinterpret_cast<float*>(smem_bf);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const uint32_t q = blockIdx.x * BF_QT + wave * 32u + col;
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
    BfList<BF_KMAX> L;
    L.init();
    float tau = -3.0e38f; // the list's kk-th score

    const uint64_t r0 = (uint64_t)blockIdx.y * P.per_range;
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
                pf[j] = *reinterpret_cast<const float4*>(P.elements + (e0 + row) * P.row_bytes + c4 * 16u);
        }
    };
    fetch(r0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ET) {
        __syncthreads(); // the previous tile has been consumed
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
                        tau = L.worst();
                    }
                }
            }
        }
    }
    bf_write_list(P, L, q, qlive, h);
}

// int8: device rows of up to 128 bytes (dims up to 128; the scan refuses longer rows). K = 16 per MFMA: lanes 0-31 carry
// bytes 0-7 of a 16-byte group, lanes 32-63 bytes 8-15. Score = dot / (|x| |q|) with the exact integer dot.
template <int R>
__global__ __launch_bounds__(BF_THREADS) void bf_i8_kernel(const BruteParams P) {
    extern __shared__ __align__(16) uint8_t smem_bf[];
    constexpr uint32_t ET = 32u * R;
    constexpr uint32_t STRIDE = 128u + 16u; // bytes per LDS row: an odd number of 16-byte units
    uint8_t* tile = smem_bf;
    float* inv = reinterpret_cast<float*>(smem_bf + (size_t)ET * STRIDE); // [ET] 1 / |x|
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, h = lane >> 5;
    const uint32_t q = blockIdx.x * BF_QT + wave * 32u + col;
    const bool qlive = q < P.nq;
    long qr[8]; // bytes 16*g + 8*h .. +7 of the query, g = 0..7
    float qinv = 0.0f;
    {
        const int8_t* qp = reinterpret_cast<const int8_t*>(P.queries) + (size_t)(qlive ? q : 0u) * P.dim;
        int dy = 0;
        for (uint32_t c = 0; c < P.dim; ++c) dy += (int)qp[c] * (int)qp[c];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            unsigned long v = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t c = (uint32_t)g * 16u + h * 8u + (uint32_t)b;
                const unsigned long byte = (qlive && c < P.dim) ? (unsigned long)(uint8_t)qp[c] : 0ul;
                v |= byte << (8 * b);
            }
            qr[g] = (long)v;
        }
        qinv = dy > 0 ? 1.0f / __builtin_sqrtf((float)dy) : 0.0f;
    }
    BfList<BF_KMAX> L;
    L.init();
    float tau = -3.0e38f;
    const uint64_t r0 = (uint64_t)blockIdx.y * P.per_range;
    const uint64_t r1 = r0 + P.per_range < P.n ? r0 + P.per_range : P.n;
    // 8 threads per row (16 bytes each; device rows shorter than 128 bytes are zero extended); the tile of the NEXT
    // step travels to registers while this one is scored; the row's squared norm falls out of the same bytes
    constexpr uint32_t NPF = (ET * 8u + BF_THREADS - 1u) / BF_THREADS;
    const uint32_t row_u4 = P.row_bytes / 16u;
    uint4 pf[NPF];
    auto fetch = [&](uint64_t e0) {
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t u = tid + BF_THREADS * j;
            const uint32_t row = u >> 3, c = u & 7u;
            pf[j] = make_uint4(0, 0, 0, 0);
            if (u < ET * 8u && e0 + row < r1 && c < row_u4) pf[j] = *reinterpret_cast<const uint4*>(P.elements + (e0 + row) * P.row_bytes + c * 16u);
        }
    };
    fetch(r0);
    for (uint64_t e0 = r0; e0 < r1; e0 += ET) {
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < NPF; ++j) {
            const uint32_t u = tid + BF_THREADS * j;
            const uint32_t row = u >> 3, c = u & 7u;
            const uint4 v = pf[j];
            if (u < ET * 8u) *reinterpret_cast<uint4*>(tile + (size_t)row * STRIDE + c * 16u) = v;
            int dx = dot4_i8(v.x, v.x, 0);
            dx = dot4_i8(v.y, v.y, dx);
            dx = dot4_i8(v.z, v.z, dx);
            dx = dot4_i8(v.w, v.w, dx);
            dx += __shfl_xor(dx, 1, 64);
            dx += __shfl_xor(dx, 2, 64);
            dx += __shfl_xor(dx, 4, 64);
            if (c == 0 && u < ET * 8u) inv[row] = dx > 0 ? 1.0f / __builtin_sqrtf((float)dx) : 0.0f;
        }
        __syncthreads();
        if (e0 + ET < r1) fetch(e0 + ET);
        bf_i32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][v] = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long a = *reinterpret_cast<const long*>(tile + (size_t)(r * 32 + col) * STRIDE + g * 16 + h * 8);
                acc[r] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, qr[g], acc[r], 0, 0, 0);
            }
        }
        const bool whole = e0 + ET <= r1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float sc[16];
            float mx = -3.0e38f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { // elements r*32 + 8*g4 + 4*h + 0..3: their 1/|x| in one read
                const float4 iv = *reinterpret_cast<const float4*>(inv + r * 32 + 8 * g4 + 4 * h);
                sc[g4 * 4 + 0] = (float)acc[r][g4 * 4 + 0] * iv.x * qinv;
                sc[g4 * 4 + 1] = (float)acc[r][g4 * 4 + 1] * iv.y * qinv;
                sc[g4 * 4 + 2] = (float)acc[r][g4 * 4 + 2] * iv.z * qinv;
                sc[g4 * 4 + 3] = (float)acc[r][g4 * 4 + 3] * iv.w * qinv;
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) mx = __builtin_fmaxf(mx, sc[v]);
            if (__ballot(mx > tau)) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const uint64_t e = e0 + (uint32_t)(r * 32 + 8 * (v / 4) + v % 4) + 4u * h;
                    if (sc[v] > tau && (whole || e < r1)) {
                        L.insert(sc[v], (uint32_t)e);
                        tau = L.worst();
                    }
                }
            }
        }
    }
    bf_write_list(P, L, q, qlive, h);
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
