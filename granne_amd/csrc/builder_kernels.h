// builder_kernels.h -- device side of the batched HNSW builder (GranneBuilder on the GPU).
//
// Restates /root/reference/src/index/mod.rs:805-960 (index_element, select_neighbors,
// initialize_node, connect_nodes, add_and_limit_neighbors) under the BATCHED schedule that
// oracle/granne_oracle.h documents at gro_build_config.batch_max:
//   phase A  every member of a batch searches the frozen graph (search_kernel.h, the same
//            kernel Granne::search uses) and runs select_neighbors       -> select_kernel
//   phase B  the link updates of the whole batch are turned into (target row, order) keyed ops,
//            radix-sorted, and each target row replays its ops in the order the sequential
//            phase B would have issued them                               -> apply_kernel
//   finally  every row is re-limited to the layer's num_neighbors          -> final_prune_kernel
// One wavefront owns one node / target row. All distances use the exact routines of dist.h, so
// a build is a deterministic function of (elements, config) and equals the oracle's batched
// build bit for bit (tests/test_gpu_builder.py).
#pragma once

#include "dist.h"
#include "search_kernel.h"
#include "wave_prims.h"

namespace granne_hip {

constexpr uint64_t OP_INVALID = ~0ull;
// op key = target row : 32 | position in batch : 20 | phase (0 own forward, 1 reverse) : 1 | k : 8
__host__ __device__ inline uint64_t op_key(uint32_t target, uint32_t p, uint32_t phase, uint32_t k) {
    return ((uint64_t)target << 29) | ((uint64_t)p << 9) | ((uint64_t)phase << 8) | (uint64_t)k;
}
__host__ __device__ inline uint32_t op_target(uint64_t key) { return (uint32_t)(key >> 29); }
__host__ __device__ inline uint32_t op_pos(uint64_t key) { return (uint32_t)(key >> 9) & 0xFFFFFu; }
__host__ __device__ inline uint32_t op_phase(uint64_t key) { return (uint32_t)(key >> 8) & 1u; }
constexpr int OP_KEY_BITS = 61;
constexpr uint32_t BUILD_MAX_NEIGHBORS = 63; // one lane per neighbor, +1 for the extra candidate
constexpr uint32_t BUILD_MAX_CAND = 1024;    // search candidates per element: max_search up to the register walker's longest list
constexpr uint32_t BUILD_MIN_CAND_CAP = 256; // the candidate arrays' LDS size follows the build's max_search from here up
constexpr uint32_t BUILD_CHUNK = 32;         // candidate rows staged per gather round (BuildParams.chunk: fewer when rows are long)
constexpr float EPS100 = 100.0f * 1.1920929e-07f; // 100.0 * f32::EPSILON (mod.rs:813, 829)

struct BuildParams {
    const uint8_t* elements;
    uint32_t row_bytes, row_stride, dim, lrow; // row_stride: bytes between rows in HBM; lrow: LDS row stride (odd multiple of 16 bytes)
    uint32_t* adj;                 // the layer being built: [len][W]
    uint32_t W;                    // device row width
    uint32_t cap;                  // logical row capacity = BuildConfig.num_neighbors (node.len())
    uint32_t m_layer;              // num_neighbors used for this layer (halved above the bottom)
    uint64_t layer_len;
    // phase A
    int64_t first_idx, idx_step;   // batch member t is element first_idx + t * idx_step
    uint32_t batch;
    uint32_t efc;                  // stride of the search outputs
    const uint64_t* s_ids;
    const float* s_dists;
    const uint32_t* s_counts;
    uint64_t* op_keys;             // [batch * cap * 2]
    uint64_t* op_vals;
    // phase B
    const uint64_t* sorted_keys;
    const uint64_t* sorted_vals;
    uint32_t n_ops;
    uint32_t* seg_start;
    uint32_t* n_seg;
    uint8_t* selected;             // [layer_len]: 1 = the row is the untouched output of select_neighbors (see apply_kernel)
    uint32_t cand_cap;             // entries of the candidate arrays in LDS (>= max_search, >= cap + 1; a multiple of 64)
    uint32_t chunk;                // candidate rows staged per gather round: BUILD_CHUNK, or 16 / 8 / 4 when rows are long
                                   // (build_chunk_for); what is computed does not depend on it, only how many rounds it takes
    uint32_t sel_lds;              // how many of the rows select_neighbors has selected stay in LDS: cap (all of them), or -- rows too
                                   // long for that, beyond ~1150-d f32 at 30 neighbors -- as many as fit; the others are read from
                                   // the elements where they lie (L2 / HBM)
};

// LDS carve-up shared by the three kernels
struct BuildLds {
    uint8_t* qrow;    // [lrow]
    uint8_t* chunk;   // [P.chunk][lrow]
    uint8_t* selrows; // [cap][lrow]
    uint32_t* cid;    // [cand_cap]
    float* cd;        // [cand_cap]
    uint32_t* sid;    // [64]
    float* sd;        // [64]
    uint32_t* cur;    // [64]
    uint32_t* slot;   // [64]
    float* pair;      // [BUILD_CHUNK][BUILD_CHUNK], aliases selrows when that is large enough
};
constexpr uint32_t PAIR_BYTES = BUILD_CHUNK * BUILD_CHUNK * 4; // pairwise distances of one chunk of candidates
__host__ __device__ inline uint32_t build_cand_cap(uint32_t max_search) {
    const uint32_t c = (max_search + 63u) & ~63u;
    return c < BUILD_MIN_CAND_CAP ? BUILD_MIN_CAND_CAP : c;
}
__host__ __device__ inline uint32_t build_lds_bytes(uint32_t lrow, uint32_t cap, uint32_t cand_cap, uint32_t chunk = BUILD_CHUNK,
                                                    uint32_t sel_lds = 0xFFFFFFFFu) {
    if (sel_lds < cap) return lrow * (1 + chunk + sel_lds) + PAIR_BYTES + cand_cap * 8 + 64 * 4 * 4; // a partial selected-rows stage
    // the pairwise matrix shares the selected-rows stage when that is large enough
    return lrow * (1 + chunk + cap) + (lrow * cap < PAIR_BYTES ? PAIR_BYTES : 0u) + cand_cap * 8 + 64 * 4 * 4;
}
// the largest chunk stage with which select_neighbors' LDS fits a CU (0: not even 4 rows do). 100-d f32 rows take
// the full 32; 768-d f32 rows (3 KB) take 16; the selected-rows stage (cap rows) holds up to about 1150-d f32 / 4600-d int8
// at 30 neighbors -- longer rows stage as many selected rows as fit beside the node's row and 4 candidates' (sel_lds < cap:
// the others are read where they lie), down to none: ~8000-d f32
__host__ __device__ inline uint32_t build_chunk_for(uint32_t lrow, uint32_t cap, uint32_t cand_cap, uint32_t lds_max) {
    for (uint32_t c = BUILD_CHUNK; c >= 4u; c >>= 1)
        if (build_lds_bytes(lrow, cap, cand_cap, c) <= lds_max) return c;
    return 0u;
}
// rows too long for the whole stage (build_chunk_for == 0): how many selected rows fit with 4 candidate rows per round
// (cap: not even none does)
__host__ __device__ inline uint32_t build_sel_lds_for(uint32_t lrow, uint32_t cap, uint32_t cand_cap, uint32_t lds_max) {
    if (build_lds_bytes(lrow, cap, cand_cap, 4u, 0u) > lds_max) return cap;
    uint32_t s = 0;
    while (s + 1u < cap && build_lds_bytes(lrow, cap, cand_cap, 4u, s + 1u) <= lds_max) ++s;
    return s;
}
// apply_kernel / final_prune_kernel only ever limit a row of at most cap + 1 candidates: while that is one chunk
// (select_neighbors_pairs) the selected-rows stage is never touched and is left out -- 21 KB instead of 29 KB per
// wave at 100-d f32, seven waves per CU instead of five
__host__ __device__ inline bool build_lds_compact(uint32_t cap, uint32_t chunk = BUILD_CHUNK) { return cap + 1u <= chunk; }
__host__ __device__ inline uint32_t build_lds_bytes_rows(uint32_t lrow, uint32_t cap, uint32_t cand_cap, uint32_t chunk = BUILD_CHUNK,
                                                         uint32_t sel_lds = 0xFFFFFFFFu) {
    if (sel_lds < cap || !build_lds_compact(cap, chunk)) return build_lds_bytes(lrow, cap, cand_cap, chunk, sel_lds);
    return lrow * (1 + chunk) + PAIR_BYTES + cand_cap * 8 + 64 * 4 * 4;
}

template <int DT, int DIM>
struct RowWork {
    const BuildParams& P;
    BuildLds L;
    uint32_t lane;

    // rows_only: the layout of build_lds_bytes_rows (no selected-rows stage)
    __device__ __forceinline__ RowWork(const BuildParams& p, uint8_t* smem, bool rows_only = false) : P(p) {
        lane = threadIdx.x;
        L.qrow = smem;
        L.chunk = smem + p.lrow;
        L.selrows = L.chunk + (size_t)p.chunk * p.lrow;
        uint8_t* a = L.selrows + (size_t)p.cap * p.lrow;
        L.pair = reinterpret_cast<float*>(L.selrows);
        if (p.sel_lds < p.cap) { // a partial selected-rows stage, the pairwise matrix behind it
            a = L.selrows + (size_t)p.sel_lds * p.lrow;
            L.pair = reinterpret_cast<float*>(a);
            a += PAIR_BYTES;
        } else if (rows_only && build_lds_compact(p.cap, p.chunk)) {
            a = L.selrows + PAIR_BYTES;
        } else if (p.lrow * p.cap < PAIR_BYTES) {
            L.pair = reinterpret_cast<float*>(a);
            a += PAIR_BYTES;
        }
        L.cid = reinterpret_cast<uint32_t*>(a);
        L.cd = reinterpret_cast<float*>(a + p.cand_cap * 4);
        a += p.cand_cap * 8;
        L.sid = reinterpret_cast<uint32_t*>(a);
        L.sd = reinterpret_cast<float*>(a + 256);
        L.cur = reinterpret_cast<uint32_t*>(a + 512);
        L.slot = reinterpret_cast<uint32_t*>(a + 768);
    }

    // copy one element row into LDS (zero padded device row)
    __device__ __forceinline__ void load_row(uint8_t* dst, uint32_t id) {
        const uint32_t row16 = P.row_bytes >> 4;
        const uint8_t* src = P.elements + (size_t)id * P.row_stride;
        for (uint32_t u = lane; u < row16; u += 64)
            *reinterpret_cast<uint4*>(dst + (size_t)u * 16) = *reinterpret_cast<const uint4*>(src + (size_t)u * 16);
    }
    // copy a row LDS -> LDS
    __device__ __forceinline__ void copy_row(uint8_t* dst, const uint8_t* src) {
        const uint32_t row16 = P.row_bytes >> 4;
        for (uint32_t u = lane; u < row16; u += 64)
            *reinterpret_cast<uint4*>(dst + (size_t)u * 16) = *reinterpret_cast<const uint4*>(src + (size_t)u * 16);
    }
    // gather rows ids[0..n) (n <= P.chunk, ids in LDS) into the chunk stage
    __device__ __forceinline__ void gather_chunk(const uint32_t* ids, uint32_t n) {
        const uint32_t row16 = P.row_bytes >> 4;
        const uint32_t lrow16 = P.lrow >> 4;
        const uint32_t total = n * row16;
        for (uint32_t f0 = 0; f0 < total; f0 += 64 * 4) {
            uint4 v0, v1, v2, v3;
            uint32_t d0, d1, d2, d3;
#define GRANNE_BGATHER(U, V, D)                                                                           \
    {                                                                                                     \
        uint32_t f = f0 + (U) * 64u + lane;                                                               \
        uint32_t fc = f < total ? f : total - 1u;                                                         \
        uint32_t row;                                                                                     \
        if constexpr (DIM > 0 && DT == DT_F32) row = fc / (uint32_t)(DIM / 4);                            \
        else row = fc / row16;                                                                            \
        uint32_t part = fc - row * row16;                                                                 \
        V = *reinterpret_cast<const uint4*>(P.elements + (size_t)ids[row] * P.row_stride + (size_t)part * 16); \
        D = f < total ? (row * lrow16 + part) * 16u : 0xFFFFFFFFu;                                        \
    }
            GRANNE_BGATHER(0, v0, d0)
            GRANNE_BGATHER(1, v1, d1)
            GRANNE_BGATHER(2, v2, d2)
            GRANNE_BGATHER(3, v3, d3)
#undef GRANNE_BGATHER
            if (d0 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(L.chunk + d0) = v0;
            if (d1 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(L.chunk + d1) = v1;
            if (d2 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(L.chunk + d2) = v2;
            if (d3 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(L.chunk + d3) = v3;
        }
    }

    // exact distance between two LDS rows (one lane)
    __device__ __forceinline__ float dist_lds(const uint8_t* x, const uint8_t* y) {
        if constexpr (DT == DT_F32) {
            float r;
            if constexpr (DIM > 0) r = dot_f32_exact<DIM>(reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(y));
            else r = dot_f32_exact_rt(reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(y), P.dim);
            return angular_from_dot(r);
        } else {
            const uint32_t row16 = P.row_bytes >> 4;
            int r = 0, dx = 0, dy = 0;
            for (uint32_t u = 0; u < row16; ++u) {
                uint4 a = *reinterpret_cast<const uint4*>(x + (size_t)u * 16);
                uint4 b = *reinterpret_cast<const uint4*>(y + (size_t)u * 16);
                r = dot4_i8(a.x, b.x, r); r = dot4_i8(a.y, b.y, r); r = dot4_i8(a.z, b.z, r); r = dot4_i8(a.w, b.w, r);
                dx = dot4_i8(a.x, a.x, dx); dx = dot4_i8(a.y, a.y, dx); dx = dot4_i8(a.z, a.z, dx); dx = dot4_i8(a.w, a.w, dx);
                dy = dot4_i8(b.x, b.x, dy); dy = dot4_i8(b.y, b.y, dy); dy = dot4_i8(b.z, b.z, dy); dy = dot4_i8(b.w, b.w, dy);
            }
            return angular_int_from_sums(r, dx, dy);
        }
    }

    // select_neighbors (mod.rs:849-883). Candidates cid/cd[0..n) sorted ascending, rows gathered a
    // chunk at a time. Result in sid/sd.
    __device__ __forceinline__ uint32_t select_neighbors(uint32_t n, uint32_t max_neighbors) {
        if (n <= max_neighbors) { // :854-856
            if (lane < n) {
                L.sid[lane] = L.cid[lane];
                L.sd[lane] = L.cd[lane];
            }
            __syncthreads();
            return n;
        }
        uint32_t nsel = 0;
        for (uint32_t c0 = 0; c0 < n && nsel < max_neighbors; c0 += P.chunk) {
            const uint32_t cn = min(P.chunk, n - c0);
            __syncthreads();
            gather_chunk(L.cid + c0, cn);
            __syncthreads();
            for (uint32_t j = c0; j < c0 + cn && nsel < max_neighbors; ++j) { // :866-881
                const float dj = L.cd[j];
                const uint8_t* rj = L.chunk + (size_t)(j - c0) * P.lrow;
                bool bad = false;
                if (lane < nsel) { // dist_to_element(n, &element_j): the selected row from its stage, or from the elements
                    float dd;
                    if (lane < P.sel_lds) dd = dist_lds(L.selrows + (size_t)lane * P.lrow, rj);
                    else dd = dist_lds(P.elements + (size_t)L.sid[lane] * P.row_stride, rj);
                    bad = !(dj <= dd);
                }
                if (wave_ballot(bad) == 0) {
                    if (nsel < P.sel_lds) copy_row(L.selrows + (size_t)nsel * P.lrow, rj);
                    if (lane == 0) {
                        L.sid[nsel] = L.cid[j];
                        L.sd[nsel] = dj;
                    }
                    nsel += 1;
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        return nsel;
    }

    // select_neighbors (mod.rs:849-883) for n <= P.chunk candidates whose rows already sit in the
    // chunk stage (candidate j in chunk[slot[j]]): the distances the reference evaluates one
    // candidate at a time -- dist(selected, candidate j) -- are a pure function of the pair, so all
    // n(n-1)/2 of them are computed first, 64 pairs at a time, and the selection loop only compares.
    // Pair rows a and n-2-a are folded into one row of n entries so that every lane has work.
    __device__ __forceinline__ uint32_t select_neighbors_pairs(uint32_t n, uint32_t max_neighbors) {
        if (n <= max_neighbors) { // :854-856
            if (lane < n) {
                L.sid[lane] = L.cid[lane];
                L.sd[lane] = L.cd[lane];
            }
            __syncthreads();
            return n;
        }
        const uint32_t folded = (n >> 1) * n;
        for (uint32_t p0 = 0; p0 < folded; p0 += 64) {
            const uint32_t p = p0 + lane;
            const uint32_t r = p / n, c = p - r * n;
            const uint32_t len1 = n - 1u - r;
            uint32_t a, b;
            bool valid = p < folded;
            if (c < len1) {
                a = r;
                b = r + 1u + c;
            } else {
                a = n - 2u - r;
                b = a + 1u + (c - len1);
                valid = valid && a != r; // the middle row of an odd triangle has no partner
            }
            if (valid) {
                const float dd = dist_lds(L.chunk + (size_t)L.slot[a] * P.lrow, L.chunk + (size_t)L.slot[b] * P.lrow);
                L.pair[a * BUILD_CHUNK + b] = dd;
                L.pair[b * BUILD_CHUNK + a] = dd;
            }
        }
        __syncthreads();
        uint32_t nsel = 0;
        uint32_t mine = 0; // lane s < nsel: sorted index of the s-th selected candidate
        for (uint32_t j = 0; j < n && nsel < max_neighbors; ++j) { // :866-881
            const float dj = L.cd[j];
            const bool bad = lane < nsel && !(dj <= L.pair[mine * BUILD_CHUNK + j]); // dist_to_element(n, &element_j)
            if (wave_ballot(bad) == 0) {
                if (lane == nsel) {
                    mine = j;
                    L.sid[nsel] = L.cid[j];
                    L.sd[nsel] = dj;
                }
                nsel += 1;
            }
        }
        __syncthreads();
        return nsel;
    }

    // add_and_limit_neighbors (mod.rs:923-959) on the row held in L.cur[0..c): optional extra
    // candidate (ex_id, ex_d). qrow must hold the node's own element. Returns the new count.
    __device__ __forceinline__ uint32_t add_and_limit(uint32_t c, bool has_extra, uint32_t ex_id, float ex_d,
                                                      uint32_t num_neighbors) {
        const uint32_t n = c + (has_extra ? 1u : 0u);
        // elements.dists(node_id, &neighbors), :938 -- candidate rows go through the chunk stage
        float d = 0.0f;
        uint32_t id = ID_EMPTY;
        if (lane < c) id = L.cur[lane];
        if (has_extra && lane == c) { id = ex_id; d = ex_d; }
        __syncthreads();
        if (lane < n) L.cid[lane] = id; // unsorted for now: gather source
        __syncthreads();
        const bool one_chunk = n <= P.chunk;
        for (uint32_t c0 = 0; c0 < n; c0 += P.chunk) {
            const uint32_t cn = min(P.chunk, n - c0);
            gather_chunk(L.cid + c0, cn);
            __syncthreads();
            if (lane >= c0 && lane < c0 + cn && lane < c)
                d = dist_lds(L.qrow, L.chunk + (size_t)(lane - c0) * P.lrow); // element(node).dist(element(j))
            __syncthreads();
        }
        // sort by (d, id), :943 (ties by id: the reference's unstable sort leaves them unspecified)
        const uint64_t key = (lane < n) ? make_key(d, id) : KEY_INF;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += (readlane64(key, j) < key) ? 1u : 0u;
        if (lane < n) {
            L.cid[rank] = id;
            L.cd[rank] = d;
            L.slot[rank] = lane; // where this candidate's row sits when everything fits one chunk
        }
        __syncthreads();
        const uint32_t ns = one_chunk ? select_neighbors_pairs(n, num_neighbors) : select_neighbors(n, num_neighbors); // :947
        if (lane < 64) L.cur[lane] = (lane < ns) ? L.sid[lane] : ID_EMPTY;  // :950-958
        __syncthreads();
        return ns;
    }

    // add_and_limit_neighbors (mod.rs:923-959) with one extra candidate x for a FULL row (c == num_neighbors)
    // that is itself the untouched output of select_neighbors for this node. Such a row is sorted by (distance
    // to the node, id) and every entry is at most as far from the node as from each entry before it, so
    // re-running select_neighbors on row + x decides nothing new about the old entries among themselves:
    //   * entries that sort before x are selected again (same predecessors as when they were selected);
    //   * x is selected iff it is at most as far from the node as from every entry before it (and fewer than
    //     num_neighbors entries sort before it) -- otherwise the result is the old row;
    //   * with x selected, an entry after x survives iff it is at most as far from the node as from x; the list
    //     is cut at num_neighbors (mod.rs:866: the loop breaks before it evaluates anything further).
    // Only the distances node<->entries (the reference's elements.dists, :938) and x<->entries are evaluated:
    // 2c row products instead of c + (c+1)c/2, the same values the full pass would compare. Returns the new
    // count, or 0xFFFFFFFF when the row is not sorted after all (the caller takes the full pass).
    // Invariant this rests on: a distance is a pure function of its two rows whichever kernel evaluates it -- the
    // walker that proposed an earlier extra (walk_fast.h / search_kernel.h) and dist_lds here produce the same bits
    // (dist.h: one operation order for all of them). Held by tests/test_gpu_builder.py: every GPU build must equal,
    // row for row, the oracle's build, which runs the reference's FULL add_and_limit_neighbors every time -- for the
    // unrolled f32 dims, the streamed run-time dims, tiny dims and int8.
    __device__ __forceinline__ uint32_t add_one_to_selected(uint32_t c, uint32_t ex_id, float ex_d, uint32_t num_neighbors) {
        uint32_t id = ID_EMPTY;
        if (lane < c) id = L.cur[lane];
        if (lane == c) id = ex_id;
        __syncthreads();
        if (lane <= c) L.cid[lane] = id;
        __syncthreads();
        gather_chunk(L.cid, c + 1u);
        __syncthreads();
        float d = 0.0f, e = 0.0f;
        if (lane < c) {
            const uint8_t* rj = L.chunk + (size_t)lane * P.lrow;
            d = dist_lds(L.qrow, rj);                              // element(node).dist(element(j)), :938
            e = dist_lds(rj, L.chunk + (size_t)c * P.lrow);        // dist_to_element(j, &element_x) = dist(x, j)
        }
        const uint64_t key = (lane < c) ? make_key(d, id) : KEY_INF;
        const uint64_t next = shift_down1(key); // lane i: key of lane i+1
        if (wave_ballot(lane + 1u < c && key > next)) return 0xFFFFFFFFu;
        const uint64_t kx = make_key(ex_d, ex_id);
        const uint32_t p = (uint32_t)__popcll(wave_ballot(lane < c && key < kx));
        if (p >= num_neighbors) return c;                              // :866: the loop breaks before it reaches x
        if (wave_ballot(lane < p && !(ex_d <= e))) return c;           // x is closer to an earlier neighbor: not selected
        const bool keep = lane < c && (lane < p || d <= e);           // :873-876 with x among the selected
        const uint64_t km = wave_ballot(keep);
        const uint64_t after = km & ~((1ull << p) - 1ull) & ((1ull << lane) - 1ull); // survivors in [p, lane)
        const uint32_t pos = lane < p ? lane : p + 1u + (uint32_t)__popcll(after);
        uint32_t total = (uint32_t)__popcll(km) + 1u;
        if (total > num_neighbors) total = num_neighbors;
        __syncthreads();
        L.cur[lane] = ID_EMPTY;
        __syncthreads();
        if (keep && pos < total) L.cur[pos] = id;
        if (lane == 0) L.cur[p] = ex_id;
        __syncthreads();
        return total;
    }
};

// ---- phase A tail: filter, select_neighbors, dead-node rule, emit ops (mod.rs:813-845) ------------
template <int DT, int DIM>
__global__ __launch_bounds__(64) void select_kernel(const BuildParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    RowWork<DT, DIM> w(P, smem);
    const uint32_t lane = threadIdx.x;
    const uint32_t t = blockIdx.x;
    if (t >= P.batch) return;
    const uint32_t idx = (uint32_t)(P.first_idx + (int64_t)t * P.idx_step);

    w.load_row(w.L.qrow, idx);
    __syncthreads();
    // do not index elements that are zero, :813-815
    float dself = 0.0f;
    if (lane == 0) dself = w.dist_lds(w.L.qrow, w.L.qrow);
    const bool skip = __uint_as_float(readlane32(__float_as_uint(dself), 0)) > EPS100;

    uint32_t nsel = 0;
    if (!skip) {
        // candidates.filter(id != idx), :822
        const uint32_t cnt = min(P.s_counts[t], P.cand_cap);
        uint32_t m = 0;
        for (uint32_t base = 0; base < cnt; base += 64) {
            uint32_t i = base + lane;
            uint32_t id = ID_EMPTY;
            float d = 0.0f;
            if (i < cnt) {
                id = (uint32_t)P.s_ids[(size_t)t * P.efc + i];
                d = P.s_dists[(size_t)t * P.efc + i];
            }
            bool keep = (i < cnt) && id != idx;
            uint64_t km = wave_ballot(keep);
            uint32_t pos = m + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
            if (keep) {
                w.L.cid[pos] = id;
                w.L.cd[pos] = d;
            }
            m += (uint32_t)__popcll(km);
        }
        __syncthreads();
        nsel = w.select_neighbors(m, P.m_layer); // :824
        // duplicates rule, :828-832
        const uint32_t half = P.m_layer / 2;
        if (half < nsel && w.L.sd[half] < EPS100) nsel = 0;
    }
    // link updates of this element as ops: own row first (:834-841), then the reverse links (:843-845)
    for (uint32_t k = lane; k < P.cap; k += 64) {
        size_t o = ((size_t)t * P.cap + k) * 2;
        if (k < nsel) {
            uint32_t nb = w.L.sid[k];
            uint64_t dbits = (uint64_t)__float_as_uint(w.L.sd[k]);
            P.op_keys[o] = op_key(idx, t, 0, k);
            P.op_vals[o] = ((uint64_t)nb << 32) | dbits;
            P.op_keys[o + 1] = op_key(nb, t, 1, k);
            P.op_vals[o + 1] = ((uint64_t)idx << 32) | dbits;
        } else {
            P.op_keys[o] = OP_INVALID;
            P.op_vals[o] = 0;
            P.op_keys[o + 1] = OP_INVALID;
            P.op_vals[o + 1] = 0;
        }
    }
}

// segment heads of the sorted op list (one segment per target row)
__global__ void mark_heads_kernel(const uint64_t* __restrict__ keys, uint32_t n_ops, uint32_t* __restrict__ seg_start,
                                  uint32_t* __restrict__ n_seg) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += gridDim.x * blockDim.x) {
        uint64_t k = keys[i];
        if (k == OP_INVALID) continue;
        if (i == 0 || op_target(keys[i - 1]) != op_target(k)) seg_start[atomicAdd(n_seg, 1u)] = i;
    }
}

// ---- phase B: replay one target row's ops in order (initialize_node / connect_nodes) --------------
template <int DT, int DIM>
__global__ __launch_bounds__(64) void apply_kernel(const BuildParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    RowWork<DT, DIM> w(P, smem, true);
    const uint32_t lane = threadIdx.x;
    const uint32_t n_seg = *P.n_seg;
    for (uint32_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        uint32_t i = P.seg_start[seg];
        const uint32_t t = op_target(P.sorted_keys[i]);
        uint32_t* row = P.adj + (size_t)t * P.W;
        __syncthreads();
        uint32_t nb = (lane < P.W) ? row[lane] : ID_EMPTY;
        uint64_t un = wave_ballot(nb == ID_EMPTY);
        uint32_t c = un ? (uint32_t)__builtin_ctzll(un) : 64u;
        w.L.cur[lane] = (lane < c) ? nb : ID_EMPTY;
        bool qloaded = false;
        // the row is the untouched output of select_neighbors (set by add_and_limit, cleared by any plain write)
        bool selected = P.selected[t] != 0;
        __syncthreads();

        while (i < P.n_ops) {
            const uint64_t key = P.sorted_keys[i];
            if (key == OP_INVALID || op_target(key) != t) break;
            const uint64_t val = P.sorted_vals[i];
            const uint32_t other = (uint32_t)(val >> 32);
            const float d = __uint_as_float((uint32_t)val);
            if (op_phase(key) == 0 && (key & 0xFF) == 0 && c == 0) {
                // initialize_node, :886-896: the whole own-forward group of this source at once
                const uint32_t p0 = op_pos(key);
                while (i < P.n_ops) {
                    const uint64_t k2 = P.sorted_keys[i];
                    if (k2 == OP_INVALID || op_target(k2) != t || op_phase(k2) != 0 || op_pos(k2) != p0) break;
                    if (c < P.cap) {
                        if (lane == 0) w.L.cur[c] = (uint32_t)(P.sorted_vals[i] >> 32);
                        c += 1;
                    }
                    i += 1;
                }
                selected = false;
                __syncthreads();
                continue;
            }
            // connect_nodes(t <- other, d), :898-921
            if (other != t) {
                uint32_t v = (lane < c) ? w.L.cur[lane] : ID_EMPTY;
                uint64_t found = wave_ballot(lane < c && v == other);
                uint32_t pos = found ? (uint32_t)__builtin_ctzll(found) : c; // first UNUSED or j_id, :912
                if (pos < P.cap) {
                    if (pos == c) {
                        if (lane == 0) w.L.cur[c] = other;
                        c += 1;
                        selected = false;
                        __syncthreads();
                    }
                } else {
                    if (!qloaded) {
                        w.load_row(w.L.qrow, t);
                        qloaded = true;
                        __syncthreads();
                    }
                    // num_neighbors = node.len(), :916-917. A full row that select_neighbors produced takes the
                    // one-candidate form of the same computation
                    uint32_t nc = 0xFFFFFFFFu;
                    if (selected && c == P.cap && c + 1u <= P.chunk) nc = w.add_one_to_selected(c, other, d, P.cap);
                    if (nc == 0xFFFFFFFFu) nc = w.add_and_limit(c, true, other, d, P.cap);
                    c = nc;
                    selected = true;
                }
            }
            i += 1;
        }
        __syncthreads();
        if (lane < P.W) row[lane] = (lane < c) ? w.L.cur[lane] : ID_EMPTY;
        if (lane == 0) P.selected[t] = selected ? 1 : 0;
    }
}

// ---- the final pass of index_elements (mod.rs:795-797): limit every row to num_neighbors ----------
template <int DT, int DIM>
__global__ __launch_bounds__(64) void final_prune_kernel(const BuildParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    RowWork<DT, DIM> w(P, smem, true);
    const uint32_t lane = threadIdx.x;
    for (uint64_t t = blockIdx.x; t < P.layer_len; t += gridDim.x) {
        uint32_t* row = P.adj + (size_t)t * P.W;
        __syncthreads();
        uint32_t nb = (lane < P.W) ? row[lane] : ID_EMPTY;
        uint64_t un = wave_ballot(nb == ID_EMPTY);
        uint32_t c = un ? (uint32_t)__builtin_ctzll(un) : 64u;
        if (c == 0) continue; // nothing to sort or limit; row stays all-UNUSED
        w.L.cur[lane] = (lane < c) ? nb : ID_EMPTY;
        w.load_row(w.L.qrow, (uint32_t)t);
        __syncthreads();
        c = w.add_and_limit(c, false, 0, 0.0f, P.m_layer);
        if (lane < P.W) row[lane] = (lane < c) ? w.L.cur[lane] : ID_EMPTY;
    }
}

} // namespace granne_hip
