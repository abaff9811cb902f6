// util_kernels.h -- layout conversion, element preparation (Vector::from) and the plain
// Dist operator, all on device. Reference citations relative to /root/reference.
#pragma once

#include "dist.h"

namespace granne_hip {

// dense [n][src_bytes] -> padded [n][dst_bytes] (zero fill); one 16-byte unit per thread
__global__ void relayout_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n,
                                     uint32_t src_bytes, uint32_t dst_bytes) {
    const uint32_t units = dst_bytes >> 4;
    uint64_t total = n * units;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t row = t / units;
        uint32_t u = (uint32_t)(t - row * units);
        const uint8_t* s = src + row * src_bytes + (uint64_t)u * 16;
        uint32_t avail = (u * 16 < src_bytes) ? min(16u, src_bytes - u * 16) : 0u;
        uint32_t w[4] = {0, 0, 0, 0};
        uint8_t* wb = reinterpret_cast<uint8_t*>(w);
        if (avail == 16 && ((reinterpret_cast<uintptr_t>(s) & 3u) == 0)) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
            w[0] = s4[0]; w[1] = s4[1]; w[2] = s4[2]; w[3] = s4[3];
        } else {
            for (uint32_t b = 0; b < avail; ++b) wb[b] = s[b];
        }
        *reinterpret_cast<uint4*>(dst + row * dst_bytes + (uint64_t)u * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// LayerDev::adjx of one layer (search_kernel.h): node i's 32 ids, then the tails of those neighbors' element rows
// (`tu` 16-byte units each, found `body_bytes` into a row; zeros where the row has no neighbor). One 16-byte unit of
// the copy per thread: units 0..7 of a node are its ids, the next 32 * tu its tails.
__global__ void inline_tails_kernel(const uint32_t* __restrict__ adj, uint64_t len, const uint8_t* __restrict__ elements,
                                    uint32_t row_stride, uint32_t body_bytes, uint32_t tu, uint8_t* __restrict__ adjx,
                                    uint32_t adjx_stride) {
    const uint32_t units = adjx_stride >> 4;
    const uint64_t total = len * units;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t node = t / units;
        const uint32_t u = (uint32_t)(t - node * units);
        uint4 v;
        if (u < 8u) {
            v = *reinterpret_cast<const uint4*>(adj + node * 32u + u * 4u);
        } else {
            const uint32_t slot = (u - 8u) / tu, part = (u - 8u) - slot * tu;
            const uint32_t id = adj[node * 32u + slot];
            v = make_uint4(0, 0, 0, 0);
            if (id != 0xFFFFFFFFu) v = *reinterpret_cast<const uint4*>(elements + (size_t)id * row_stride + body_bytes + part * 16u);
        }
        *reinterpret_cast<uint4*>(adjx + node * adjx_stride + (size_t)u * 16u) = v;
    }
}

// fixed-width adjacency [len][w] -> [len][W] (W >= w), UNUSED padded
__global__ void relayout_adj_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t len,
                                    uint32_t w, uint32_t W) {
    uint64_t total = len * W;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t row = t / W;
        uint32_t c = (uint32_t)(t - row * W);
        dst[t] = (c < w) ? src[row * w + c] : 0xFFFFFFFFu;
    }
}

// neighbor ids of a layer must be nodes of that layer (the walk indexes the layer's rows with them):
// counts the entries that are neither UNUSED nor < limit
__global__ void check_adj_kernel(const uint32_t* __restrict__ rows, uint64_t total, uint64_t limit,
                                 uint32_t* __restrict__ bad) {
    uint32_t mine = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = rows[t];
        mine += (v != 0xFFFFFFFFu && (uint64_t)v >= limit) ? 1u : 0u;
    }
    if (mine) atomicAdd(bad, mine);
}

// Does some row of a layer name a neighbor twice? (No builder of the reference writes such a row, mod.rs:913-917; a foreign
// file may hold one.) The register walker ranks an expansion's candidates against each other and needs to know whether two
// lanes can hold one node (walk_fast.h, LAYER_TWIN_ROWS). One wavefront per row of W <= 64 ids (wider layers are not the
// register walker's): lane i holds id i and meets the ids at distance 1..W/2 by rotation.
__global__ void twin_rows_kernel(const uint32_t* __restrict__ adj, uint64_t len, uint32_t W, uint32_t* __restrict__ found) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    bool hit = false;
    for (uint64_t row = wave; row < len; row += waves) {
        const uint32_t mine = lane < W ? adj[row * W + lane] : 0xFFFFFFFFu;
        for (uint32_t o = 1; o <= W / 2u; ++o) {
            const uint32_t other = (uint32_t)__shfl((int)mine, (int)((lane + o) % W), 64);
            hit = hit || (lane < W && mine != 0xFFFFFFFFu && mine == other);
        }
    }
    if (__ballot(hit) && lane == 0) atomicOr(found, 1u);
}

// CSR adjacency -> [len][W], UNUSED padded
__global__ void csr_to_adj_kernel(const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ ids,
                                  uint32_t* __restrict__ dst, uint64_t len, uint32_t W) {
    uint64_t total = len * W;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t row = t / W;
        uint32_t c = (uint32_t)(t - row * W);
        uint64_t b = offsets[row], e = offsets[row + 1];
        dst[t] = (b + c < e) ? ids[b + c] : 0xFFFFFFFFu;
    }
}

// synthetic rows: must equal oracle gro_synth_component bit for bit
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void synth_rows_kernel(float* __restrict__ out, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim) {
    uint64_t total = n * dim;
    uint64_t s = splitmix64(seed);
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = splitmix64(s ^ (row0 * dim + t));
        out[t] = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
}

// angular::Vector::from(Vec<f32>) = math::normalize_f32 (src/math.rs:123-150), in place.
// One block stages ROWS rows in LDS (coalesced), one lane per row runs the exact dot product
// and the division, rows are written back coalesced. LDS row stride is dim+pad floats.
__global__ void normalize_rows_kernel(float* __restrict__ rows, uint64_t n, uint32_t dim, uint32_t rows_per_block,
                                      uint32_t lstride) {
    extern __shared__ __align__(16) uint8_t smem_u[];
    float* lds = reinterpret_cast<float*>(smem_u);
    for (uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block; r0 < n; r0 += (uint64_t)gridDim.x * rows_per_block) {
        uint32_t nr = (uint32_t)min((uint64_t)rows_per_block, n - r0);
        uint32_t total = nr * dim;
        for (uint32_t t = threadIdx.x; t < total; t += blockDim.x) {
            uint32_t r = t / dim, c = t - r * dim;
            lds[r * lstride + c] = rows[r0 * dim + t];
        }
        __syncthreads();
        if (threadIdx.x < nr) {
            float* x = lds + threadIdx.x * lstride;
            float norm = __builtin_sqrtf(dot_f32_exact_rt(x, x, dim)); // :132
            if (norm > 0.0f)
                for (uint32_t c = 0; c < dim; ++c) x[c] = x[c] / norm; // :134-138
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < total; t += blockDim.x) {
            uint32_t r = t / dim, c = t - r * dim;
            rows[r0 * dim + t] = lds[r * lstride + c];
        }
        __syncthreads();
    }
}

// Rust `f32 as i8`: truncate toward zero, saturate, NaN -> 0
__device__ __forceinline__ int8_t f32_as_i8(float v) {
    if (v != v) return 0;
    if (v <= -128.0f) return -128;
    if (v >= 127.0f) return 127;
    return (int8_t)(int)v;
}

// angular_int::Vector::quantize (src/elements/angular_int.rs:27-45); one wave per row
__global__ void quantize_rows_kernel(const float* __restrict__ rows, int8_t* __restrict__ out, uint64_t n,
                                     uint32_t dim) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += n_waves) {
        const float* x = rows + r * dim;
        float mx = 0.0f; // |x| >= 0; max over NotNan (angular_int.rs:28-32)
        for (uint32_t c = lane; c < dim; c += 64) mx = fmaxf(mx, fabsf(x[c]));
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (dim == 0) mx = 127.0f;
        for (uint32_t c = lane; c < dim; c += 64) {
            float vi = x[c] * 127.0f / mx; // left to right (angular_int.rs:38)
            out[r * dim + c] = f32_as_i8(vi);
        }
    }
}

// ElementContainer::dists / dist_to_element (src/elements/mod.rs:35-39, dense_vector.rs:149-163) as a
// stand-alone operator: pair t = (query qidx[t] or t / m, element ids[t]). Eight lanes share one
// element row, lane `sub` reading the 16-byte piece `sub` of every 128-byte block, so a wave
// fetches eight rows with fully used 128-byte lines -- the same lane layout as the walk's
// fast_rows. f32 keeps the reference's association: lane `sub` owns accumulators 4*sub..4*sub+3
// of the 32, the ordered sum runs down the eight lanes, the tail is folded by sequential fmas.
__device__ __forceinline__ float lane_shr1(float v) { // value of lane-1 (within a 16-lane row)
#if GRANNE_HIP_USE_DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
#else
    return __shfl_up(v, 1, 64);
#endif
}

template <int DT>
__global__ __launch_bounds__(256) void dists_kernel(const uint8_t* __restrict__ elements, uint64_t n_elements,
                                                    uint32_t row_bytes, uint32_t row_stride, uint32_t dim,
                                                    const uint8_t* __restrict__ queries,
                                                    const uint32_t* __restrict__ qidx, uint32_t m,
                                                    const uint32_t* __restrict__ ids, uint64_t n_pairs,
                                                    float* __restrict__ out, uint32_t* __restrict__ status) {
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const uint64_t n_groups = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    // wave-uniform trip count: every lane of a wave runs the same number of rounds
    const uint64_t wave_first = group - (lane >> 3);
    for (uint64_t t0 = wave_first; t0 < n_pairs; t0 += n_groups) {
        const uint64_t t = t0 + (lane >> 3);
        const bool live = t < n_pairs;
        const uint64_t tc = live ? t : n_pairs - 1;
        const uint32_t id = ids[tc];
        const bool valid = id < n_elements;
        const uint64_t qi = qidx ? qidx[tc] : tc / m;
        const uint8_t* row = elements + (uint64_t)(valid ? id : 0u) * row_stride;
        float d;
        if constexpr (DT == 0) {
            const float* q = reinterpret_cast<const float*>(queries) + qi * dim;
            const uint32_t nfull = dim >> 5, tail = dim & 31u;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll 4
            for (uint32_t c = 0; c < nfull; ++c) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + (size_t)c * 128u + sub * 16u);
                const float* qc = q + c * 32u + sub * 4u;
                a0 = __builtin_fmaf(__uint_as_float(v.x), qc[0], a0);
                a1 = __builtin_fmaf(__uint_as_float(v.y), qc[1], a1);
                a2 = __builtin_fmaf(__uint_as_float(v.z), qc[2], a2);
                a3 = __builtin_fmaf(__uint_as_float(v.w), qc[3], a3);
            }
            uint4 vt = make_uint4(0, 0, 0, 0); // the (zero padded) tail block
            if (nfull * 128u + sub * 16u + 16u <= row_bytes)
                vt = *reinterpret_cast<const uint4*>(row + (size_t)nfull * 128u + sub * 16u);
            float r = 0.0f; // ordered sum acc[0] .. acc[31]: lane s is right after step s
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                float u = (ps == 0) ? 0.0f : lane_shr1(r);
                u = u + a0; u = u + a1; u = u + a2; u = u + a3;
                r = u;
            }
            r = __shfl(r, (int)(lane | 7u), 64);
            for (uint32_t k = 0; k < tail; ++k) { // src/math.rs:47-49
                const uint32_t w = (k & 3u) == 0 ? vt.x : (k & 3u) == 1 ? vt.y : (k & 3u) == 2 ? vt.z : vt.w;
                const float xv = __uint_as_float((uint32_t)__shfl((int)w, (int)((lane & ~7u) + (k >> 2)), 64));
                r = __builtin_fmaf(xv, q[nfull * 32u + k], r);
            }
            d = angular_from_dot(r);
        } else {
            const int8_t* q = reinterpret_cast<const int8_t*>(queries) + qi * dim;
            int r = 0, dx = 0, dy = 0;
            for (uint32_t b = sub * 16u; b < row_bytes; b += 128u) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + b);
                uint32_t qw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t o = b + 4u * j;
                    if ((dim & 3u) == 0) {
                        qw[j] = (o < dim) ? *reinterpret_cast<const uint32_t*>(q + o) : 0u;
                    } else {
                        uint32_t w = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (o + e < dim) w |= (uint32_t)(uint8_t)q[o + e] << (8 * e);
                        qw[j] = w;
                    }
                }
                r = dot4_i8(v.x, qw[0], r); r = dot4_i8(v.y, qw[1], r);
                r = dot4_i8(v.z, qw[2], r); r = dot4_i8(v.w, qw[3], r);
                dx = dot4_i8(v.x, v.x, dx); dx = dot4_i8(v.y, v.y, dx);
                dx = dot4_i8(v.z, v.z, dx); dx = dot4_i8(v.w, v.w, dx);
                dy = dot4_i8(qw[0], qw[0], dy); dy = dot4_i8(qw[1], qw[1], dy);
                dy = dot4_i8(qw[2], qw[2], dy); dy = dot4_i8(qw[3], qw[3], dy);
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                r += __shfl_xor(r, o, 64);
                dx += __shfl_xor(dx, o, 64);
                dy += __shfl_xor(dy, o, 64);
            }
            d = angular_int_from_sums(r, dx, dy);
        }
        if (live && sub == 0) {
            out[t] = valid ? d : __builtin_inff();
            if (!valid && status) atomicAdd(status, 1u);
        }
    }
}

struct MergeParams {
    // shard s: ids at ids + s * ids_stride bytes ([nq][k] u64), dists and counts likewise. The plain
    // form passes three [n_shards][...] arrays, the packed form three offsets into records of one
    // stride (granne_hip_packed_topk_bytes).
    const uint8_t* ids;
    const uint8_t* dists;
    const uint8_t* counts;
    uint64_t ids_stride, dists_stride, counts_stride;
    uint64_t offsets[64];
    uint32_t n_shards, nq, k;
    uint64_t* out_ids;
    float* out_dists;
    uint32_t* out_counts;
};

// smallest value of the wavefront, in every lane (DPP row shifts + the two row broadcasts of gfx9, no LDS)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t wave_min_step(uint32_t v) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xf, false);
    return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = wave_min_step<0x111, 0xf>(v); // row_shr:1
    v = wave_min_step<0x112, 0xf>(v); // row_shr:2
    v = wave_min_step<0x114, 0xf>(v); // row_shr:4
    v = wave_min_step<0x118, 0xf>(v); // row_shr:8   -> lane 15 of every row holds its row's minimum
    v = wave_min_step<0x142, 0xa>(v); // row_bcast:15 -> lanes 31 and 63 hold the minimum of rows 0-1 / 2-3
    v = wave_min_step<0x143, 0xc>(v); // row_bcast:31 -> lane 63 holds the wave's
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// one wave per query: the k best of the shards' candidates by (dist, global id). Every shard's list is ascending (or
// is made so first, below), so the answer is a k-way merge: lane s stands at the head of shard s's list, the wave's smallest head is the next result
// and that lane moves on -- k rounds of one wave-wide minimum. (Ranking all C = n_shards * k candidates against each
// other was 1.4 ms of the 8.5 ms of a 1024-query scan whose 64 element ranges bring 16 candidates each.)
__global__ __launch_bounds__(64) void merge_topk_kernel(const MergeParams P) {
    extern __shared__ __align__(16) uint8_t smem_m[];
    uint32_t* kd = reinterpret_cast<uint32_t*>(smem_m); // [C] distance bits
    uint64_t* ki = reinterpret_cast<uint64_t*>(smem_m + (size_t)((P.n_shards * P.k * 4 + 7) & ~7u)); // [C] global ids
    const uint32_t lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= P.nq) return;
    const uint32_t C = P.n_shards * P.k;
    for (uint32_t c = lane; c < C; c += 64) {
        uint32_t s = c / P.k, j = c - s * P.k;
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(P.counts + (size_t)s * P.counts_stride);
        const uint64_t* sid = reinterpret_cast<const uint64_t*>(P.ids + (size_t)s * P.ids_stride);
        const float* sd = reinterpret_cast<const float*>(P.dists + (size_t)s * P.dists_stride);
        bool ok = j < cnt[q];
        size_t src = (size_t)q * P.k + j;
        kd[c] = ok ? __float_as_uint(sd[src]) : 0xFFFFFFFFu;
        ki[c] = ok ? sid[src] + P.offsets[s] : ~0ull;
    }
    __syncthreads();
    // The merge below needs every list ascending by (dist, id), valid entries first. The searches' lists are; a caller's
    // own may not be (the entry points promise the k best by (dist, global id) of WHATEVER they are given: round 4 ranked
    // all candidates against each other): lane s checks list s on its way, and a list out of order is put in order
    // first -- an insertion sort by its own lane, in LDS, the price of k reads when nothing is to be done.
    if (lane < P.n_shards && P.k > 1) {
        uint32_t* d = kd + lane * P.k;
        uint64_t* i = ki + lane * P.k;
        bool sorted = true;
        for (uint32_t j = 1; j < P.k; ++j) sorted = sorted && (d[j - 1] < d[j] || (d[j - 1] == d[j] && i[j - 1] <= i[j]));
        if (!sorted) {
            for (uint32_t j = 1; j < P.k; ++j) {
                const uint32_t dj = d[j];
                const uint64_t ij = i[j];
                uint32_t t = j;
                while (t > 0 && (d[t - 1] > dj || (d[t - 1] == dj && i[t - 1] > ij))) {
                    d[t] = d[t - 1];
                    i[t] = i[t - 1];
                    --t;
                }
                d[t] = dj;
                i[t] = ij;
            }
        }
    }
    __syncthreads();
    uint32_t pos = 0; // this lane's place in its shard's list
    uint32_t count = 0;
    for (uint32_t r = 0; r < P.k; ++r) {
        const bool live = lane < P.n_shards && pos < P.k;
        const uint32_t d = live ? kd[lane * P.k + pos] : 0xFFFFFFFFu; // (an exhausted list reads its 0xFFFFFFFF padding)
        const uint32_t dmin = wave_min_u32(d);
        if (dmin == 0xFFFFFFFFu) break; // every list is exhausted
        uint64_t tied = __ballot(d == dmin);
        uint32_t win = (uint32_t)__builtin_ctzll(tied);
        const uint64_t id = live ? ki[lane * P.k + pos] : ~0ull;
        uint64_t best = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(id >> 32), (int)win) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)id, (int)win);
        for (tied &= tied - 1; tied; tied &= tied - 1) { // equal distances: the smaller global id first
            const uint32_t l = (uint32_t)__builtin_ctzll(tied);
            const uint64_t oi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(id >> 32), (int)l) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)id, (int)l);
            if (oi < best) {
                best = oi;
                win = l;
            }
        }
        if (lane == win) {
            P.out_ids[(size_t)q * P.k + r] = best;
            P.out_dists[(size_t)q * P.k + r] = __uint_as_float(dmin);
            pos += 1;
        }
        count = r + 1;
    }
    for (uint32_t e = count + lane; e < P.k; e += 64) {
        P.out_ids[(size_t)q * P.k + e] = ~0ull;
        P.out_dists[(size_t)q * P.k + e] = __builtin_inff();
    }
    if (lane == 0) P.out_counts[q] = count;
}

// The status words of the G shards of a partitioned search (four u32 after each shard's packed top-k, stride bytes
// apart) folded into the caller's four: [0] |= a shard's exact-search scratch ran out, [1] += hand-overs, [2] += walks
// that borrowed an overflow table. One wave (G <= 64).
__global__ __launch_bounds__(64) void fold_status_kernel(const uint8_t* gathered, uint64_t stride, uint64_t status_off,
                                                         uint32_t n_shards, uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t a = 0, b = 0, c = 0;
    if (lane < n_shards) {
        const uint32_t* st = reinterpret_cast<const uint32_t*>(gathered + (size_t)lane * stride + status_off);
        a = st[0];
        b = st[1];
        c = st[2];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a |= __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
        c += __shfl_xor(c, o, 64);
    }
    if (lane == 0) {
        if (a) atomicOr(out + 0, 1u);
        if (b) atomicAdd(out + 1, b);
        if (c) atomicAdd(out + 2, c);
    }
}

} // namespace granne_hip
