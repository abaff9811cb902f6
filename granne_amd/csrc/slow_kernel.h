// slow_kernel.h -- the unbounded exact walker. Same algorithm as search_kernel.h
// (/root/reference/src/index/mod.rs:963-1037), but `res`, `pq` and `visited` are the reference's
// literal containers (binary heaps and an open-addressing set) in GLOBAL memory, sized by the
// host, so no walk can outgrow them short of GRANNE_HIP_ERR_OVERFLOW. It serves
//   - queries the register/LDS walkers hand over (a queue drop that ties with the max_search-th
//     distance -- search_kernel.h explains why only those are unsafe -- or no visited-set
//     overflow table left),
//   - max_search beyond the register walkers' lists, and GRANNE_HIP_OPT_FORCE_SLOW (tests).
// It is slow on purpose of simplicity: lane 0 runs the heaps; the 64 lanes split the neighbor
// row (visited insert + one exact distance each, rows read straight from HBM).
//
// One launch per search. The hand-over list is served INSIDE the walker's launch: the grid is nq walker
// blocks plus a few tail blocks; a tail block sleeps until the launch's `done` counter says every walker
// block has finished, reads the list (device-scope atomics on both sides: the list crosses XCDs) and walks
// its share; the last tail block out re-arms the control words, so the scratch block needs no memset
// between launches and a search is ONE kernel on the stream (round 2: memset + walker + slow_kernel +
// stream-ordered malloc/free per call, 20-37 us per step). Tail blocks never hold a walker up: walkers do
// not wait for anything, so the order in which the dispatcher places blocks does not matter. Searches that
// are known to go to the exact walker wholesale launch `slow_kernel` alone.
#pragma once

#include "search_kernel.h"

namespace granne_hip {

// control words at the start of a search's scratch block (all zero between launches)
constexpr uint32_t CTL_SLOW_COUNT = 0; // hand-over list length (SearchParams::slow_count)
constexpr uint32_t CTL_DONE = 1;       // walker blocks finished
constexpr uint32_t CTL_TAIL_DONE = 2;  // tail blocks finished
constexpr uint32_t CTL_EXHAUSTED = 3;  // a walk outgrew the exact walker's containers in THIS launch
constexpr uint32_t CTL_LAST_SLOW = 4;  // previous launch: queries the exact walker served
constexpr uint32_t CTL_LAST_EXHAUSTED = 5; // previous launch: CTL_EXHAUSTED
constexpr uint32_t CTL_SPILLED = 6;    // walks that borrowed an overflow table (statistics, monotonic)
constexpr uint32_t CTL_WORDS = 16;

struct SlowParams {
    SearchParams sp;
    uint32_t* ctl;   // the control words above
    uint32_t* vis;   // [blocks][slots]
    uint64_t* pq;    // [blocks][slots]   min-heap of keys
    uint64_t* res;   // [blocks][ef]      max-heap of keys
    uint32_t slots;  // power of two
    uint32_t all;    // 1: no list, every query of the launch is walked here (slow_kernel launched alone)
    uint32_t* status2; // caller's u32[2] (optional): [0] = 1 when a walk exhausts `slots`, [1] += queries that took this path
    uint32_t* host_status; // optional u32[2] in host-mapped memory, written with PLAIN stores (no PCIe atomics):
                           // [0] = 1 when a walk exhausts `slots`, [1] = queries that took this path
};

// binary heaps over u64 keys, run by one lane
__device__ inline void gheap_push_min(uint64_t* h, uint32_t& n, uint64_t k) {
    uint32_t i = n++;
    while (i > 0) {
        uint32_t pa = (i - 1) >> 1;
        if (h[pa] <= k) break;
        h[i] = h[pa];
        i = pa;
    }
    h[i] = k;
}
__device__ inline uint64_t gheap_pop_min(uint64_t* h, uint32_t& n) {
    uint64_t top = h[0];
    uint64_t k = h[--n];
    uint32_t i = 0;
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && h[c + 1] < h[c]) ++c;
        if (h[c] >= k) break;
        h[i] = h[c];
        i = c;
    }
    if (n > 0) h[i] = k;
    return top;
}
__device__ inline void gheap_push_max(uint64_t* h, uint32_t& n, uint64_t k) {
    uint32_t i = n++;
    while (i > 0) {
        uint32_t pa = (i - 1) >> 1;
        if (h[pa] >= k) break;
        h[i] = h[pa];
        i = pa;
    }
    h[i] = k;
}
__device__ inline void gheap_replace_max(uint64_t* h, uint32_t n, uint64_t k) { // pop max, push k
    uint32_t i = 0;
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && h[c + 1] > h[c]) ++c;
        if (h[c] <= k) break;
        h[i] = h[c];
        i = c;
    }
    h[i] = k;
}

template <int DT>
__device__ inline float slow_dist(const SearchParams& p, const uint8_t* lds_q, uint32_t id, int dy) {
    const uint8_t* row = p.elements + (size_t)id * p.row_stride;
    if constexpr (DT == DT_F32) {
        float r = dot_f32_exact_rt(reinterpret_cast<const float*>(row), reinterpret_cast<const float*>(lds_q), p.dim);
        return angular_from_dot(r);
    } else {
        const int8_t* x = reinterpret_cast<const int8_t*>(row);
        const int8_t* q = reinterpret_cast<const int8_t*>(lds_q);
        int r = 0, dx = 0;
        for (uint32_t i = 0; i < p.dim; ++i) {
            int xi = x[i], qi = q[i];
            r += xi * qi;
            dx += xi * xi;
        }
        return angular_int_from_sums(r, dx, dy);
    }
}

// Block `me` of `n_blocks` walks entries me, me + n_blocks, ... of the hand-over list (or of 0..nq-1 when P.all).
template <int DT>
__device__ inline void slow_walk_list(const SlowParams& P, const uint32_t me, const uint32_t n_blocks,
                                      const uint32_t n_slow, uint8_t* smem) {
    const SearchParams& p = P.sp;
    const uint32_t lane = threadIdx.x;
    uint8_t* lds_q = smem;
    uint64_t* ckey = reinterpret_cast<uint64_t*>(smem + lds_query_bytes(p.row_bytes)); // [64]

    uint32_t* vis = P.vis + (size_t)me * P.slots;
    uint64_t* pq = P.pq + (size_t)me * P.slots;
    uint64_t* res = P.res + (size_t)me * p.ef;
    if (me == 0 && threadIdx.x == 0 && P.status2 && n_slow) atomicAdd(P.status2 + 1, n_slow);
    if (me == 0 && threadIdx.x == 0 && P.host_status) P.host_status[1] = n_slow;

    for (uint32_t si = me; si < n_slow; si += n_blocks) {
        const uint32_t qi = P.all ? si : __hip_atomic_load(p.slow_list + si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // query to LDS
        int dy = 0;
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
            dy = part;
        }
        __syncthreads();

        uint64_t n_dist = 0, n_expand = 0, n_adj = 0;
        uint32_t entrypoint = 0;
        uint32_t n_res = 0;
        bool overflow = false;

        const bool trail = p.trail_out != nullptr; // reorder.rs:180-208, see search_kernel.h
        const uint32_t walk_layers = trail ? min(min(p.trail_layers, TRAIL_WIDTH), p.n_layers) : p.n_layers;
        uint32_t trail_mine = 0;
        for (uint32_t l = 0; l < walk_layers && !overflow; ++l) {
            const LayerDev L = p.layers[l];
            const bool bottom = !trail && (l + 1 == p.n_layers);
            if (trail) entrypoint = 0;
            const uint32_t ef = bottom ? p.ef : 1u;
            const uint32_t slots = bottom ? P.slots : min(P.slots, 65536u);
            const uint32_t mask = slots - 1;
            const uint32_t limit = slots / 2;
            for (uint32_t i = lane; i < slots; i += 64) vis[i] = ID_EMPTY;
            __threadfence_block();
            __syncthreads();

            uint32_t n_pq = 0, n_vis = 1;
            n_res = 0;
            if (lane == 0) {
                vis[VisitedSet::hash(entrypoint) & mask] = entrypoint;
                float d0 = slow_dist<DT>(p, lds_q, entrypoint, dy);
                gheap_push_min(pq, n_pq, make_key(d0, entrypoint));
            }
            n_dist += 1;
            n_pq = 1;
            __syncthreads();

            for (;;) {
                // lane 0: pop, break test, res.push
                uint64_t x = KEY_INF;
                uint32_t worst_bits = 0;
                if (lane == 0) {
                    if (n_pq > 0) {
                        uint64_t top = pq[0];
                        bool was_full = n_res >= ef;
                        if (!(was_full && key_dist(top) > key_dist(res[0]))) {
                            x = gheap_pop_min(pq, n_pq);
                            if (!was_full) {
                                gheap_push_max(res, n_res, x);
                            } else if (x < res[0]) {
                                gheap_replace_max(res, n_res, x);
                            }
                            if (n_res >= ef) worst_bits = (uint32_t)(res[0] >> 32);
                        }
                    }
                }
                x = readlane64(x, 0);
                n_pq = readlane32(n_pq, 0);
                n_res = readlane32(n_res, 0);
                worst_bits = readlane32(worst_bits, 0);
                if (x == KEY_INF) break;
                const bool full = n_res >= ef;
                const float worst = __uint_as_float(worst_bits);

                const gptr_u32 row = (gptr_u32)L.adj + (size_t)key_id(x) * L.width;
                n_expand += 1;
                for (uint32_t base = 0; base < L.width; base += 64) {
                    uint32_t nb = (base + lane < L.width) ? row[base + lane] : ID_EMPTY;
                    uint64_t unused = wave_ballot(nb == ID_EMPTY);
                    uint32_t nvalid = unused ? (uint32_t)__builtin_ctzll(unused) : 64u;
                    n_adj += nvalid;
                    bool fresh = false;
                    if (lane < nvalid) {
                        uint32_t slot = VisitedSet::hash(nb) & mask;
                        for (;;) {
                            uint32_t old = atomicCAS(&vis[slot], ID_EMPTY, nb);
                            if (old == ID_EMPTY) { fresh = true; break; }
                            if (old == nb) break;
                            slot = (slot + 1) & mask;
                        }
                    }
                    uint64_t fm = wave_ballot(fresh);
                    uint32_t m = (uint32_t)__popcll(fm);
                    n_vis += m;
                    n_dist += m;
                    float d = 0.0f;
                    if (fresh) d = slow_dist<DT>(p, lds_q, nb, dy);
                    bool pass = fresh && (!full || d < worst);
                    uint64_t pm = wave_ballot(pass);
                    uint32_t np = (uint32_t)__popcll(pm);
                    if (n_pq + np > slots || n_vis > limit) { overflow = true; break; }
                    uint32_t pos = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                    __syncthreads();
                    if (pass) ckey[pos] = make_key(d, nb);
                    __syncthreads();
                    if (lane == 0)
                        for (uint32_t i = 0; i < np; ++i) gheap_push_min(pq, n_pq, ckey[i]);
                    n_pq = readlane32(n_pq, 0);
                    __threadfence_block();
                    if (nvalid < 64u) break;
                }
                if (overflow) break;
            }
            if (overflow) break;
            // smallest entry of res = next entry point (upper layers hold one entry)
            if (!bottom) {
                uint32_t ep = 0;
                if (lane == 0) ep = key_id(res[0]);
                entrypoint = readlane32(ep, 0);
                if (trail && lane == l) trail_mine = entrypoint;
            }
        }
        if (trail) {
            if (overflow) {
                if (lane == 0) {
                    atomicExch(P.ctl + CTL_EXHAUSTED, 1u);
                    if (P.status2) atomicExch(P.status2, 1u);
                    if (P.host_status) P.host_status[0] = 1u;
                }
            } else if (lane < TRAIL_WIDTH) {
                p.trail_out[(size_t)qi * TRAIL_WIDTH + lane] = trail_mine;
            }
            __syncthreads();
            continue;
        }

        // output: ascending (dist, id) = repeatedly take the max of `res` from the back
        uint32_t count = 0;
        if (overflow) {
            if (lane == 0) {
                atomicExch(P.ctl + CTL_EXHAUSTED, 1u);
                if (P.status2) atomicExch(P.status2, 1u);
                if (P.host_status) P.host_status[0] = 1u;
            }
        } else if (p.n_layers > 0) {
            if (lane == 0) {
                // heap-sort in place: res[0..n_res) ascending
                uint32_t n = n_res;
                while (n > 1) {
                    uint64_t mx = res[0];
                    uint64_t k = res[--n];
                    uint32_t i = 0;
                    for (;;) {
                        uint32_t c = 2 * i + 1;
                        if (c >= n) break;
                        if (c + 1 < n && res[c + 1] > res[c]) ++c;
                        if (res[c] <= k) break;
                        res[i] = res[c];
                        i = c;
                    }
                    res[i] = k;
                    res[n] = mx;
                }
            }
            __threadfence_block();
            __syncthreads();
            count = min(n_res, p.k);
        }
        if (lane == 0) {
            const QueryIO io = query_io(p, qi);
            for (uint32_t e = 0; e < p.k; ++e) {
                bool ok = e < count;
                uint64_t key = ok ? res[e] : KEY_INF;
                io.ids[e] = ok ? (uint64_t)key_id(key) : ~0ull;
                io.dists[e] = ok ? key_dist(key) : __builtin_inff();
            }
            *io.count = count;
            if (io.stats) {
                io.stats[0] = n_dist;
                io.stats[1] = n_expand;
                io.stats[2] = n_adj;
            }
        }
        __syncthreads();
    }
}

// The last block out publishes the launch's status words and re-arms the control words.
__device__ inline void slow_epilogue(const SlowParams& P, const uint32_t n_blocks, const uint32_t n_slow) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(P.ctl + CTL_TAIL_DONE, 1u) == n_blocks - 1u) {
            const uint32_t ex = atomicExch(P.ctl + CTL_EXHAUSTED, 0u);
            atomicExch(P.ctl + CTL_LAST_SLOW, n_slow);
            atomicExch(P.ctl + CTL_LAST_EXHAUSTED, ex);
            atomicExch(P.ctl + CTL_SLOW_COUNT, 0u);
            atomicExch(P.ctl + CTL_DONE, 0u);
            atomicExch(P.ctl + CTL_TAIL_DONE, 0u);
        }
    }
}

// a walker block has finished (its hand-over, if any, is already on the list: hand_over() fences)
__device__ __forceinline__ void walker_done(const SlowParams& P) {
    if (threadIdx.x == 0) atomicAdd(P.ctl + CTL_DONE, 1u);
}

// blocks nq.. of a walker launch
template <int DT>
__device__ inline void tail_block(const SlowParams& P, uint8_t* smem) {
    const uint32_t nq = P.sp.nq;
    const uint32_t me = blockIdx.x - nq, n_blocks = gridDim.x - nq;
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(P.ctl + CTL_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nq)
            __builtin_amdgcn_s_sleep(24);
    }
    __syncthreads();
    const uint32_t n_slow = __hip_atomic_load(P.ctl + CTL_SLOW_COUNT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n_slow) slow_walk_list<DT>(P, me, n_blocks, n_slow, smem);
    else if (me == 0 && threadIdx.x == 0 && P.host_status) P.host_status[1] = 0u;
    slow_epilogue(P, n_blocks, n_slow);
}

// every query of the launch on the exact walker (max_search beyond the register walkers, GRANNE_HIP_OPT_FORCE_SLOW)
template <int DT>
__global__ __launch_bounds__(64) void slow_kernel(const SlowParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    slow_walk_list<DT>(P, blockIdx.x, gridDim.x, P.sp.nq, smem);
    slow_epilogue(P, gridDim.x, P.sp.nq);
}

// The general walker's launch (search_kernel.h): block b < nq walks query b; the blocks after them are the tail.
// TRAIL = true is the variant Granne::reorder launches (SearchParams::trail_out): a kernel of its own,
// so that the search kernel carries one copy of the walker and nothing else.
template <int DT, int DIM, int S, bool TRAIL = false>
__global__ __launch_bounds__(64) void search_kernel(const SlowParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    if (blockIdx.x < P.sp.nq) {
        walk_one<DT, DIM, S, TRAIL>(P.sp, blockIdx.x, smem);
        walker_done(P);
    } else {
        tail_block<DT>(P, smem);
    }
}

} // namespace granne_hip
