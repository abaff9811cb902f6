// search_kernel.h -- Granne::search as one gfx950 kernel, the GENERAL shapes: one wavefront walks one
// query through every layer (find_entrypoint + search_for_neighbors,
// /root/reference/src/index/mod.rs:963-1037). The common shapes (every layer 32 ids wide; f32 rows
// of a compile-time dim or int8 rows of 128 bytes) are served by walk_fast.h; this kernel takes the
// rest: f32 with a run-time dim, other int8 row sizes, layers wider than 32 ids, ids beyond 31 bits.
//
// Per expansion (one iteration of the reference's `while let Some(..) = pq.pop()` loop):
//   1. the expanded node's adjacency row is already in registers when the node was the queue head
//      one expansion earlier (prefetch), else it is read with one coalesced load, one id per lane;
//   2. the exact visited set (LDS front table + global overflow, wave_prims.h) filters the ids; the
//      fresh ones are compacted through LDS;
//   3. f32: their rows are staged in LDS, one lane per candidate evaluates the reference's 32
//      fused accumulators (dist.h explains why that association must be kept); i8: v_dot4_i32_i8
//      partial sums over a power-of-two group of lanes per row, xor-shuffle reduce, float tail;
//   4. candidate keys (dist,id) that pass the reference's filter enter the register-resident
//      sorted queue (ballot rank + DPP shift).
// The walk is a strict restatement of the reference's control flow, so results are identical;
// only memory (never logic) is parallel. One bounded structure can change results: the candidate
// queue (64*S entries; dropping its largest entry is provably safe unless that entry ties with
// the max_search-th smallest distance). That event -- or running out of visited-set overflow --
// hands the query, untouched, to the unbounded global-memory walker (slow_kernel.h), which is the
// same algorithm with literal heaps -- still on the GPU, never on the CPU.
#pragma once

#include "dist.h"
#include "wave_prims.h"

namespace granne_hip {

constexpr int DT_F32 = 0;
constexpr int DT_I8 = 1;

struct LayerDev {
    const uint32_t* adj; // [len][width] u32, UNUSED-padded, valid ids first
    uint64_t len;
    uint32_t width;      // device row width (multiple of 32 -> rows are 128-byte aligned)
    uint32_t flags;      // LAYER_*
    // The register walker's own copy of the layer (walk_fast.h, f32 rows with a tail: 100-d, 200-d): node i's 32 ids
    // FOLLOWED BY the tails (the dim % 32 last floats) of those 32 neighbors' element rows, adjx_stride bytes per node
    // (128 + 32 * tail bytes: whole 128-byte lines). An expansion then reads whole lines only: the neighbors' 32-float
    // chunks from their rows (rows start on a line), the tails from here, next to the ids. Null: no such copy (a
    // builder's layers in the making, every other shape) -- the tails are read from the rows themselves.
    const uint8_t* adjx;
    uint32_t adjx_stride;
    uint32_t reserved;
};
constexpr uint32_t LAYER_TWIN_ROWS = 1u; // some row names a neighbor twice (no builder of the reference makes such rows; a foreign file may hold them)

// one batch of a launch that serves several (SearchParams::batch)
constexpr uint32_t MAX_LAUNCH_BATCHES = 32;
struct BatchIO {
    const uint8_t* queries;
    uint64_t* out_ids;
    float* out_dists;
    uint32_t* out_counts;
    uint64_t* out_stats; // or null
};

struct SearchParams {
    const uint8_t* elements; // device layout: [n][row_bytes], zero padded
    uint64_t n_elements;
    uint32_t dim;
    uint32_t row_bytes;      // bytes of a row's data, zero padded: multiple of 16
    uint32_t row_stride;     // bytes from one row to the next (>= row_bytes; f32 rows of 256 bytes and more start on a 128-byte line)
    const LayerDev* layers;
    uint32_t n_layers;
    const uint8_t* queries;  // query i starts at queries + i * q_stride (dim scalars each)
    int64_t q_stride;        // bytes; dense batches: dim * sizeof(scalar); may be negative
    uint32_t nq;
    uint32_t ef;             // max_search
    uint32_t k;              // num_neighbors
    uint64_t* out_ids;       // [nq][k]
    float* out_dists;        // [nq][k]
    uint32_t* out_counts;    // [nq]
    uint64_t* out_stats;     // [nq][3] or null
    uint32_t visited_slots;  // bottom layer, power of two
    uint32_t upper_slots;    // upper layers, power of two <= visited_slots
    uint32_t front_eighths;  // bottom layer: the front table freezes at this many eighths of its slots (7 = default)
    uint32_t maxc;           // f32: rows the LDS stage holds (<= 64)
    uint32_t lrow_bytes;     // f32: LDS stage row stride (odd multiple of 16)
    uint32_t stage_bytes;    // LDS bytes of the stage
    uint32_t adjspec_bytes;  // LDS bytes of the speculative adjacency table (0: kept in registers)
    uint32_t* slow_count;    // queries handed to the slow path
    uint32_t* slow_list;     // [nq]
    uint32_t force_slow;
    uint32_t spec;           // 1: speculative adjacency prefetch (narrow layers)
    OverflowPool ovf;        // global overflow tables of the visited sets (wave_prims.h)
    // trail mode (Granne::reorder, src/index/reorder.rs:180-208): instead of a search, walk layers
    // 0..trail_layers-1 with max_search 1, each from node 0, and record the ids found
    uint32_t* trail_out;     // [nq][8] or null
    uint32_t trail_layers;
    // Several batches in ONE launch (granne_hip_search_batches_device): walker w serves query w % batch_nq of batch
    // w / batch_nq, whose buffers are batch[w / batch_nq]; `nq` is then the number of walkers of the launch and the
    // plain pointers above are unused. n_batches == 0: one batch, the plain pointers.
    uint32_t n_batches;
    uint32_t batch_nq;
    BatchIO batch[MAX_LAUNCH_BATCHES];
};
constexpr uint32_t TRAIL_WIDTH = 8; // NUM_LAYERS, reorder.rs:177

// where walker qi of a launch reads its query and writes its result
struct QueryIO {
    const uint8_t* q;
    uint64_t* ids;   // [k]
    float* dists;    // [k]
    uint32_t* count;
    uint64_t* stats; // [3] or null
};
__device__ __forceinline__ QueryIO query_io(const SearchParams& p, uint32_t qi) {
    const uint8_t* q = p.queries;
    uint64_t* ids = p.out_ids;
    float* dists = p.out_dists;
    uint32_t* counts = p.out_counts;
    uint64_t* stats = p.out_stats;
    uint32_t i = qi;
    if (p.n_batches) {
        const uint32_t b = qi / p.batch_nq;
        i = qi - b * p.batch_nq;
        q = p.batch[b].queries;
        ids = p.batch[b].out_ids;
        dists = p.batch[b].out_dists;
        counts = p.batch[b].out_counts;
        stats = p.batch[b].out_stats;
    }
    QueryIO io;
    io.q = q + (int64_t)i * p.q_stride;
    io.ids = ids + (size_t)i * p.k;
    io.dists = dists + (size_t)i * p.k;
    io.count = counts + i;
    io.stats = stats ? stats + (size_t)i * 3 : nullptr;
    return io;
}

struct WalkStats {
    uint64_t n_dist, n_expand, n_adj;
};

// LDS carve-up (dynamic shared memory), all offsets multiples of 16:
//   [query: row_bytes][cand: 64 u32][dout: 64 f32][adjspec: adjspec_bytes][stage: stage_bytes][visited]
__host__ __device__ inline uint32_t lds_query_bytes(uint32_t row_bytes) { return (row_bytes + 15u) & ~15u; }
constexpr uint32_t LDS_FIXED_BYTES = 512;                // cand + dout
constexpr uint32_t LDS_ADJSPEC_BYTES = 4096 + 16;        // 32x32 u32 + a 16-byte dump slot (paths without register spec)
constexpr uint32_t COLSTAGE_ROW_BYTES = 144;             // column-streamed stage: 32 floats + pad = 9 x 16 B (odd)
constexpr uint32_t COLSTAGE_BYTES = 32 * COLSTAGE_ROW_BYTES;

// Hand query qi, untouched, to the exact global-memory walker. The list is read by other CUs (possibly of
// another XCD) inside the same launch: the entry is written with a device-scope atomic and fenced before the
// block reports itself done (slow_kernel.h).
__device__ __forceinline__ void hand_over(const SearchParams& p, uint32_t qi) {
    if (threadIdx.x == 0) {
        const uint32_t at = atomicAdd(p.slow_count, 1u);
        atomicExch(p.slow_list + at, qi);
        __threadfence();
    }
}

template <int DT, int DIM, int S>
struct Walker {
    // ---- immutable per-launch state
    const SearchParams& p;
    uint32_t lane;
    uint8_t* lds_q;
    uint32_t* cand;
    float* dout;
    uint32_t* adjspec; // speculatively fetched adjacency rows of the last expansion's candidates
    uint8_t* stage;
    uint32_t* vis_tab;
    int dy; // i8: sum of squares of the query
    // paths whose speculative adjacency rows stay in registers (everything but runtime-dim f32)
    static constexpr bool REGSPEC = (DT == DT_I8) || (DIM > 0);
    uint4 sa0, sa1, sa2, sa3; // REGSPEC: speculative adjacency rows stay in registers
    // ---- per-walk state
    VisitedSet vis;
    SortedList<S> res; // `res`, capped at ef entries
    SortedList<S> pq;  // `pq`, bounded at 64*S entries
    WalkStats st;
    bool bail; // visited table full or unsafe queue drop: hand over to the slow path

    __device__ __forceinline__ Walker(const SearchParams& p_, uint8_t* smem) : p(p_) {
        lane = threadIdx.x;
        uint32_t qb = lds_query_bytes(p.row_bytes);
        lds_q = smem;
        cand = reinterpret_cast<uint32_t*>(smem + qb);
        dout = reinterpret_cast<float*>(smem + qb + 256);
        adjspec = reinterpret_cast<uint32_t*>(smem + qb + 512);
        stage = smem + qb + LDS_FIXED_BYTES + p.adjspec_bytes;
        vis_tab = reinterpret_cast<uint32_t*>(stage + p.stage_bytes);
        sa0 = sa1 = sa2 = sa3 = make_uint4(ID_EMPTY, ID_EMPTY, ID_EMPTY, ID_EMPTY);
        dy = 0;
        st.n_dist = st.n_expand = st.n_adj = 0;
        bail = false;
        vis.init_walker();
    }

    // stage the query in LDS, zero padded to row_bytes
    __device__ __forceinline__ void load_query(uint32_t qi) {
        if (DT == DT_F32) {
            const float* q = reinterpret_cast<const float*>(query_io(p, qi).q);
            float* l = reinterpret_cast<float*>(lds_q);
            for (uint32_t i = lane; i < p.row_bytes / 4; i += 64) l[i] = (i < p.dim) ? q[i] : 0.0f;
        } else {
            const int8_t* q = reinterpret_cast<const int8_t*>(query_io(p, qi).q);
            int8_t* l = reinterpret_cast<int8_t*>(lds_q);
            int part = 0;
            for (uint32_t i = lane; i < p.row_bytes; i += 64) {
                int v = (i < p.dim) ? (int)q[i] : 0;
                l[i] = (int8_t)v;
                part += v * v;
            }
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            dy = part; // exact i32, identical in every lane
        }
        __syncthreads();
    }

    // distances of candidates cand[0..m) (m <= 64) to the query; lane c < m returns d(c)
    __device__ __forceinline__ float distances(uint32_t m) {
        float d = 0.0f;
        if constexpr (DT == DT_F32) {
            const uint32_t row16 = p.row_bytes >> 4;
            const uint32_t lrow16 = p.lrow_bytes >> 4;
            for (uint32_t g0 = 0; g0 < m; g0 += p.maxc) {
                uint32_t gm = min(p.maxc, m - g0);
                uint32_t total = gm * row16;
                // gather: 64 lanes x 16 bytes per step, 8 steps in flight before the first LDS write
                for (uint32_t f0 = 0; f0 < total; f0 += 64 * 8) {
                    uint4 v0, v1, v2, v3, v4, v5, v6, v7;
                    uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
#define GRANNE_GATHER(U, V, D)                                                                       \
    {                                                                                                \
        uint32_t f = f0 + (U) * 64u + lane;                                                          \
        uint32_t fc = f < total ? f : total - 1u; /* clamp: the load itself is unconditional */      \
        uint32_t row;                                                                                \
        if constexpr (DIM > 0) row = fc / (uint32_t)(DIM / 4);                                       \
        else row = fc / row16;                                                                       \
        uint32_t part = fc - row * row16;                                                            \
        uint32_t id = cand[g0 + row];                                                                \
        V = *reinterpret_cast<const uint4*>(p.elements + (size_t)id * p.row_stride + (size_t)part * 16); \
        D = f < total ? (row * lrow16 + part) * 16u : 0xFFFFFFFFu;                                   \
    }
                    GRANNE_GATHER(0, v0, d0)
                    GRANNE_GATHER(1, v1, d1)
                    GRANNE_GATHER(2, v2, d2)
                    GRANNE_GATHER(3, v3, d3)
                    GRANNE_GATHER(4, v4, d4)
                    GRANNE_GATHER(5, v5, d5)
                    GRANNE_GATHER(6, v6, d6)
                    GRANNE_GATHER(7, v7, d7)
#undef GRANNE_GATHER
                    if (d0 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d0) = v0;
                    if (d1 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d1) = v1;
                    if (d2 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d2) = v2;
                    if (d3 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d3) = v3;
                    if (d4 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d4) = v4;
                    if (d5 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d5) = v5;
                    if (d6 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d6) = v6;
                    if (d7 != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(stage + d7) = v7;
                }
                __syncthreads();
                if (lane >= g0 && lane < g0 + gm) {
                    const float* x = reinterpret_cast<const float*>(stage + (size_t)(lane - g0) * p.lrow_bytes);
                    const float* q = reinterpret_cast<const float*>(lds_q);
                    float r;
                    if constexpr (DIM > 0) r = dot_f32_exact<DIM>(x, q);
                    else r = dot_f32_exact_rt(x, q, p.dim);
                    d = angular_from_dot(r);
                }
                __syncthreads();
            }
        } else {
            // lanes-per-row: 16 bytes per lane, up to 64 lanes (rows of up to 1024 bytes per step)
            const uint32_t row16 = p.row_bytes >> 4;
            const uint32_t lpr = min(64u, row16); // power of two by construction of row_bytes
            const uint32_t rpp = 64u / lpr;       // rows per pass
            const uint32_t sub = lane & (lpr - 1);
            const uint32_t rip = lane / lpr;
            for (uint32_t c0 = 0; c0 < m; c0 += rpp) {
                uint32_t ci = c0 + rip;
                int r = 0, dx = 0;
                if (ci < m) {
                    const uint8_t* row = p.elements + (size_t)cand[ci] * p.row_stride;
                    for (uint32_t u = sub; u < row16; u += lpr) {
                        uint4 x = *reinterpret_cast<const uint4*>(row + (size_t)u * 16);
                        uint4 y = *reinterpret_cast<const uint4*>(lds_q + (size_t)u * 16);
                        r = dot4_i8(x.x, y.x, r); r = dot4_i8(x.y, y.y, r);
                        r = dot4_i8(x.z, y.z, r); r = dot4_i8(x.w, y.w, r);
                        dx = dot4_i8(x.x, x.x, dx); dx = dot4_i8(x.y, x.y, dx);
                        dx = dot4_i8(x.z, x.z, dx); dx = dot4_i8(x.w, x.w, dx);
                    }
                }
                for (uint32_t o = lpr >> 1; o > 0; o >>= 1) {
                    r += __shfl_xor(r, (int)o, 64);
                    dx += __shfl_xor(dx, (int)o, 64);
                }
                if (ci < m && sub == 0) dout[ci] = angular_int_from_sums(r, dx, dy);
            }
            __syncthreads();
            if (lane < m) d = dout[lane];
            __syncthreads();
        }
        return d;
    }

    // pq.push with the bounded queue; flags `bail` when dropping an entry is not provably safe
    __device__ __forceinline__ void pq_push(uint64_t ck, uint32_t ef) {
        constexpr uint32_t CAP = 64u * S;
        uint32_t r = pq.rank(ck);
        uint64_t dropped;
        if (r >= CAP) {
            dropped = ck;
        } else {
            dropped = pq.get(CAP - 1);
            pq.insert_at(r, ck, lane);
        }
        if (dropped != KEY_INF) {
            // every entry left in pq sorts before `dropped`; it could only ever be popped after
            // >= ef smaller ones, i.e. when res.max.dist <= pq[ef-1].dist. Safe unless tied.
            if (key_dist(dropped) == key_dist(pq.get(ef - 1))) bail = true;
        }
    }

    // ---- narrow layers (device row width 32) on the paths without a fast expansion -----------------
    // (f32 with a runtime dim, i8 rows other than 128 bytes): candidates are compacted through LDS
    // and `distances` gathers their rows; the candidates' adjacency rows are still fetched
    // speculatively in the same round trip (the next node to expand is either the queue's current
    // head -- prefetched into registers at the start of the expansion -- or one of these
    // candidates). Speculation only moves loads of immutable data earlier.
    __device__ __forceinline__ float distances_narrow(uint32_t m, gptr_u32 adj) {
        float d = 0.0f;
        // adjacency rows of the candidates: 8 lanes x 16 bytes per row, 8 rows per step
        const uint32_t au = m * 8u;
        uint4 a0, a1, a2, a3;
#define GRANNE_ASPEC(U, A)                                                                            \
    {                                                                                                 \
        uint32_t f = (U) * 64u + lane;                                                                \
        uint32_t fc = f < au ? f : au - 1u;                                                           \
        A = load_global_u4(adj + (size_t)cand[fc >> 3] * 32u + (fc & 7u) * 4u);                        \
    }
        GRANNE_ASPEC(0, a0) GRANNE_ASPEC(1, a1) GRANNE_ASPEC(2, a2) GRANNE_ASPEC(3, a3)
#undef GRANNE_ASPEC
        asm volatile("" ::: "memory"); // the loads above may not sink below this line
        if constexpr (REGSPEC) {
            sa0 = a0; sa1 = a1; sa2 = a2; sa3 = a3;
            d = distances(m); // its own row loads overlap with the ones in flight
        } else {
            d = distances(m);
            park_spec(au, a0, a1, a2, a3);
            __syncthreads();
        }
        return d;
    }

    // park the speculative adjacency rows in LDS: row c at adjspec[c*32 ..] = linear in the unit
    // index f; every lane stores (out-of-range lanes into the dump slot) so that the stores -- and
    // with them the loads -- stay in straight-line code
    __device__ __forceinline__ void park_spec(uint32_t au, uint4 a0, uint4 a1, uint4 a2, uint4 a3) {
        uint4* as4 = reinterpret_cast<uint4*>(adjspec);
        const uint32_t dump = 256u; // the 16 bytes after the 32x32 table
        as4[lane < au ? lane : dump] = a0;
        as4[64u + lane < au ? 64u + lane : dump] = a1;
        as4[128u + lane < au ? 128u + lane : dump] = a2;
        as4[192u + lane < au ? 192u + lane : dump] = a3;
    }

    // mod.rs:1029-1031 + pq.push for the candidates of one expansion (lane c < m holds d, cid)
    __device__ __forceinline__ void offer_candidates(uint32_t m, float d, uint32_t cid, bool full, float worst,
                                                     uint32_t ef) {
        uint64_t ck = make_key(d, cid);
        // `res` is frozen during an expansion, so the filter is too
        bool pass = (lane < m) && (!full || d < worst);
        // entries that cannot enter a full queue are dropped right here
        uint64_t last = pq.get(64u * S - 1);
        if (last != KEY_INF) {
            bool dropnow = pass && ck > last;
            if (wave_ballot(dropnow && d == key_dist(pq.get(ef - 1)))) bail = true;
            pass = pass && !dropnow;
        }
        uint64_t pm = wave_ballot(pass);
        while (pm) {
            uint32_t src = (uint32_t)__builtin_ctzll(pm);
            pm &= pm - 1;
            pq_push(readlane64(ck, src), ef);
        }
    }

    // search_for_neighbors (mod.rs:999-1037) on one layer. Result: `res` (ascending).
    __device__ __forceinline__ void search_layer(const LayerDev& L, uint32_t entrypoint, uint32_t ef,
                                                 uint32_t slots) {
        vis.reset(vis_tab, slots, lane);
        res.init();
        pq.init();
        __syncthreads();
        uint32_t n_popped = 0;

        const bool narrow = p.spec && L.width == 32 && (DT == DT_I8 || DIM > 0 || p.maxc >= 32);
        const gptr_u32 adjg = (gptr_u32)L.adj;
        // speculative adjacency state (narrow layers)
        uint32_t specA_id = ID_EMPTY, specA_nb = ID_EMPTY; // row of the queue head, one id per lane
        uint32_t specB_m = 0, specB_cid = ID_EMPTY;        // last expansion's candidates (rows in sa* / adjspec)

        // distance to the entry point (mod.rs:1012-1016)
        {
            if (lane == 0) cand[0] = entrypoint;
            vis.insert(entrypoint, lane == 0, p.ovf);
            vis.count = 1;
            __syncthreads();
            float d0 = distances(1);
            st.n_dist += 1;
            uint64_t k0 = readlane64(make_key(d0, entrypoint), 0);
            pq.insert_at(0, k0, lane);
        }

        for (;;) {
            uint64_t x = pq.get(0); // pq.pop(), mod.rs:1018
            if (x == KEY_INF) break;
            bool full = n_popped >= ef;
            if (full && key_dist(x) > key_dist(res.get(ef - 1))) break; // mod.rs:1019-1021
            pq.pop_front(lane);

            // res.push((d, idx)), max_size_heap.rs:18-32
            {
                uint32_t r = res.rank(x);
                if (r < ef) {
                    res.insert_at(r, x, lane);
                    if (ef < 64u * S) res.clear_at(ef, lane);
                }
            }
            n_popped += 1;
            full = n_popped >= ef;
            const float worst = full ? key_dist(res.get(ef - 1)) : 0.0f;

            // layer.get_neighbors(idx), mod.rs:1025 / 540-552: row prefix until UNUSED
            const uint32_t xid = key_id(x);
            const gptr_u32 row = adjg + (size_t)xid * L.width;
            st.n_expand += 1;

            if (narrow) {
                uint32_t nb;
                const uint64_t hitB = wave_ballot(lane < specB_m && specB_cid == xid);
                if (specA_id == xid) {
                    nb = specA_nb;
                } else if (hitB) {
                    const uint32_t c = (uint32_t)__builtin_ctzll(hitB);
                    if constexpr (REGSPEC) {
                        // row c sits in register sa[c/8], lanes 8*(c%8)..+7, four ids per lane
                        const uint4 sel = (c < 8) ? sa0 : (c < 16) ? sa1 : (c < 24) ? sa2 : sa3;
                        const int src = (int)(((c & 7u) << 3) + ((lane & 31u) >> 2));
                        const uint32_t w0 = (uint32_t)__shfl((int)sel.x, src, 64);
                        const uint32_t w1 = (uint32_t)__shfl((int)sel.y, src, 64);
                        const uint32_t w2 = (uint32_t)__shfl((int)sel.z, src, 64);
                        const uint32_t w3 = (uint32_t)__shfl((int)sel.w, src, 64);
                        const uint32_t comp = lane & 3u;
                        const uint32_t w = comp == 0 ? w0 : comp == 1 ? w1 : comp == 2 ? w2 : w3;
                        nb = (lane < 32) ? w : ID_EMPTY;
                    } else {
                        nb = (lane < 32) ? adjspec[c * 32u + lane] : ID_EMPTY;
                    }
                } else {
                    nb = (lane < 32) ? row[lane] : ID_EMPTY;
                }
                // prefetch the adjacency of the queue's new head for the next iteration
                const uint64_t head = pq.get(0);
                specA_id = (head != KEY_INF) ? key_id(head) : ID_EMPTY;
                if (specA_id != ID_EMPTY) specA_nb = (lane < 32) ? adjg[(size_t)specA_id * 32u + lane] : ID_EMPTY;

                uint64_t unused = wave_ballot(nb == ID_EMPTY);
                uint32_t nvalid = unused ? (uint32_t)__builtin_ctzll(unused) : 64u;
                st.n_adj += nvalid;
                {
                    bool fresh = vis.insert(nb, lane < nvalid, p.ovf); // visited.insert(neighbor_idx), mod.rs:1026
                    uint64_t fm = wave_ballot(fresh);
                    uint32_t m = (uint32_t)__popcll(fm);
                    if (m) {
                        vis.added(m);
                        uint32_t pos = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
                        __syncthreads(); // adjspec / cand of the previous expansion are dead from here on
                        if (fresh) cand[pos] = nb;
                        __syncthreads();
                        float d = distances_narrow(m, adjg); // mod.rs:1027
                        st.n_dist += m;
                        uint32_t cid = (lane < m) ? cand[lane] : ID_EMPTY;
                        specB_m = m;
                        specB_cid = cid;
                        offer_candidates(m, d, cid, full, worst, ef);
                    }
                }
                if (!vis.make_room(p.ovf, lane)) bail = true;
            } else {
                for (uint32_t base = 0; base < L.width; base += 64) {
                    uint32_t nb = (base + lane < L.width) ? row[base + lane] : ID_EMPTY;
                    uint64_t unused = wave_ballot(nb == ID_EMPTY);
                    uint32_t nvalid = unused ? (uint32_t)__builtin_ctzll(unused) : 64u;
                    st.n_adj += nvalid;
                    bool fresh = vis.insert(nb, lane < nvalid, p.ovf); // visited.insert(neighbor_idx), mod.rs:1026
                    uint64_t fm = wave_ballot(fresh);
                    uint32_t m = (uint32_t)__popcll(fm);
                    if (m) {
                        vis.added(m);
                        uint32_t pos = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
                        if (fresh) cand[pos] = nb;
                        __syncthreads();
                        float d = distances(m); // mod.rs:1027
                        st.n_dist += m;
                        uint32_t cid = (lane < m) ? cand[lane] : 0u;
                        offer_candidates(m, d, cid, full, worst, ef);
                        __syncthreads();
                    }
                    if (!vis.make_room(p.ovf, lane)) bail = true;
                    if (nvalid < 64u) break;
                }
            }
            if (bail) return;
        }
    }
};

template <int DT, int DIM, int S, bool TRAIL>
__device__ __forceinline__ void walk_one(const SearchParams& p, const uint32_t qi, uint8_t* smem) {
    const uint32_t lane = threadIdx.x;
    if (p.force_slow) {
        hand_over(p, qi);
        return;
    }

    Walker<DT, DIM, S> w(p, smem);
    w.load_query(qi);

    if constexpr (TRAIL) { // find_entrypoint_trail: `ep` reads the still-zero eps[i], every walk starts at node 0
        const uint32_t take = min(min(p.trail_layers, TRAIL_WIDTH), p.n_layers);
        uint32_t mine = 0;
        for (uint32_t l = 0; l < take; ++l) {
            w.search_layer(p.layers[l], 0u, 1u, p.upper_slots);
            if (w.bail) break;
            const uint32_t found = key_id(w.res.get(0));
            if (lane == l) mine = found;
        }
        w.vis.release(p.ovf, lane);
        if (w.bail) {
            hand_over(p, qi);
        } else if (lane < TRAIL_WIDTH) {
            p.trail_out[(size_t)qi * TRAIL_WIDTH + lane] = mine;
        }
        return;
    } else {

    uint32_t entrypoint = 0; // mod.rs:989
    for (uint32_t l = 0; l < p.n_layers; ++l) {
        const LayerDev L = p.layers[l];
        const bool bottom = (l + 1 == p.n_layers);
        w.search_layer(L, entrypoint, bottom ? p.ef : 1u, bottom ? p.visited_slots : p.upper_slots);
        if (w.bail) break;
        if (!bottom) entrypoint = key_id(w.res.get(0)); // res[0].0, mod.rs:993
    }

    w.vis.release(p.ovf, lane);
    if (w.bail) { // hand the untouched query to the exact global-memory walker
        hand_over(p, qi);
        return;
    }

    // .take(num_neighbors), mod.rs:974-977
    uint32_t count = 0;
    if (p.n_layers > 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) count += (uint32_t)__popcll(wave_ballot(w.res.key[s] != KEY_INF));
        count = min(count, p.k);
    }
    const QueryIO io = query_io(p, qi);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint32_t e = (uint32_t)s * 64u + lane;
        if (e < p.k) {
            bool ok = e < count;
            io.ids[e] = ok ? (uint64_t)key_id(w.res.key[s]) : ~0ull;
            io.dists[e] = ok ? key_dist(w.res.key[s]) : __builtin_inff();
        }
    }
    for (uint32_t e = 64u * S + lane; e < p.k; e += 64) { // k beyond the list capacity: padding
        io.ids[e] = ~0ull;
        io.dists[e] = __builtin_inff();
    }
    if (lane == 0) {
        *io.count = count;
        if (io.stats) {
            io.stats[0] = w.st.n_dist;
            io.stats[1] = w.st.n_expand;
            io.stats[2] = w.st.n_adj;
        }
    }
    } // !TRAIL
}


// The __global__ entry point (search_kernel) lives in slow_kernel.h, beside the tail blocks that serve the
// hand-over list inside the same launch.

} // namespace granne_hip
