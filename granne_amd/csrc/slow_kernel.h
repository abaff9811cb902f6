// slow_kernel.h -- the unbounded exact walker. Same algorithm as search_kernel.h
// (/root/reference/src/index/mod.rs:963-1037), but `res`, `pq` and `visited` are the reference's
// literal containers (binary heaps and an open-addressing set) in GLOBAL memory, sized by the
// host, so no walk can outgrow them short of GRANNE_HIP_ERR_OVERFLOW. It serves
//   - queries the register/LDS walker hands over (a queue drop that ties with the max_search-th
//     distance -- search_kernel.h explains why only those are unsafe -- or no visited-set
//     overflow table left),
//   - max_search > 256, and GRANNE_HIP_OPT_FORCE_SLOW (tests).
// It is slow on purpose of simplicity: lane 0 runs the heaps; the 64 lanes split the neighbor
// row (visited insert + one exact distance each, rows read straight from HBM).
#pragma once

#include "search_kernel.h"

namespace granne_hip {

struct SlowParams {
    SearchParams sp;
    uint32_t* vis;   // [blocks][slots]
    uint64_t* pq;    // [blocks][slots]   min-heap of keys
    uint64_t* res;   // [blocks][ef]      max-heap of keys
    uint32_t slots;  // power of two
    uint32_t* status;// set to 1 when a walk exhausts `slots`
    uint32_t* status2; // caller's u32[2] (optional): [0] same flag, [1] += queries that took this path
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
    const uint8_t* row = p.elements + (size_t)id * p.row_bytes;
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

template <int DT>
__global__ __launch_bounds__(64) void slow_kernel(const SlowParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    const SearchParams& p = P.sp;
    const uint32_t lane = threadIdx.x;
    uint8_t* lds_q = smem;
    uint64_t* ckey = reinterpret_cast<uint64_t*>(smem + lds_query_bytes(p.row_bytes)); // [64]

    uint32_t* vis = P.vis + (size_t)blockIdx.x * P.slots;
    uint64_t* pq = P.pq + (size_t)blockIdx.x * P.slots;
    uint64_t* res = P.res + (size_t)blockIdx.x * p.ef;
    const uint32_t n_slow = *p.slow_count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.status2 && n_slow) atomicAdd(P.status2 + 1, n_slow);
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.host_status) P.host_status[1] = n_slow;

    for (uint32_t si = blockIdx.x; si < n_slow; si += gridDim.x) {
        const uint32_t qi = p.slow_list[si];
        __syncthreads();
        // query to LDS
        int dy = 0;
        if (DT == DT_F32) {
            const float* q = reinterpret_cast<const float*>(p.queries + (int64_t)qi * p.q_stride);
            float* l = reinterpret_cast<float*>(lds_q);
            for (uint32_t i = lane; i < p.row_bytes / 4; i += 64) l[i] = (i < p.dim) ? q[i] : 0.0f;
        } else {
            const int8_t* q = reinterpret_cast<const int8_t*>(p.queries + (int64_t)qi * p.q_stride);
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
                    atomicExch(P.status, 1u);
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
                atomicExch(P.status, 1u);
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
            for (uint32_t e = 0; e < p.k; ++e) {
                bool ok = e < count;
                uint64_t key = ok ? res[e] : KEY_INF;
                p.out_ids[(size_t)qi * p.k + e] = ok ? (uint64_t)key_id(key) : ~0ull;
                p.out_dists[(size_t)qi * p.k + e] = ok ? key_dist(key) : __builtin_inff();
            }
        }
        if (lane == 0) {
            p.out_counts[qi] = count;
            if (p.out_stats) {
                p.out_stats[(size_t)qi * 3 + 0] = n_dist;
                p.out_stats[(size_t)qi * 3 + 1] = n_expand;
                p.out_stats[(size_t)qi * 3 + 2] = n_adj;
            }
        }
        __syncthreads();
    }
}

} // namespace granne_hip
