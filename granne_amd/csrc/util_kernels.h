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

// ElementContainer::dist_to_element for explicit pairs; one lane per pair, rows straight from HBM
template <int DT>
__global__ void dist_pairs_kernel(const uint8_t* __restrict__ elements, uint32_t row_bytes, uint32_t dim,
                                  const uint8_t* __restrict__ queries, const uint32_t* __restrict__ qidx,
                                  const uint32_t* __restrict__ ids, uint64_t n_pairs, float* __restrict__ out) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_pairs;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t* row = elements + (uint64_t)ids[t] * row_bytes;
        if constexpr (DT == 0) {
            const float* q = reinterpret_cast<const float*>(queries) + (uint64_t)qidx[t] * dim;
            out[t] = angular_from_dot(dot_f32_exact_rt(reinterpret_cast<const float*>(row), q, dim));
        } else {
            const int8_t* q = reinterpret_cast<const int8_t*>(queries) + (uint64_t)qidx[t] * dim;
            const int8_t* x = reinterpret_cast<const int8_t*>(row);
            int r = 0, dx = 0, dy = 0;
            for (uint32_t i = 0; i < dim; ++i) {
                int xi = x[i], qi = q[i];
                r += xi * qi;
                dx += xi * xi;
                dy += qi * qi;
            }
            out[t] = angular_int_from_sums(r, dx, dy);
        }
    }
}

struct MergeParams {
    const uint64_t* ids;
    const float* dists;
    const uint32_t* counts;
    uint64_t offsets[64];
    uint32_t n_shards, nq, k;
    uint64_t* out_ids;
    float* out_dists;
    uint32_t* out_counts;
};

// one wave per query: rank every candidate among all candidates of the query by (dist, global id)
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
        bool ok = j < P.counts[(size_t)s * P.nq + q];
        size_t src = ((size_t)s * P.nq + q) * P.k + j;
        kd[c] = ok ? __float_as_uint(P.dists[src]) : 0xFFFFFFFFu;
        ki[c] = ok ? P.ids[src] + P.offsets[s] : ~0ull;
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t c = lane; c < C; c += 64) total += (kd[c] != 0xFFFFFFFFu) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    const uint32_t count = total < P.k ? total : P.k;
    for (uint32_t c = lane; c < C; c += 64) {
        const uint32_t d = kd[c];
        const uint64_t id = ki[c];
        if (d == 0xFFFFFFFFu) continue;
        uint32_t rank = 0;
        for (uint32_t o = 0; o < C; ++o) {
            uint32_t od = kd[o];
            uint64_t oi = ki[o];
            rank += (od < d || (od == d && oi < id)) ? 1u : 0u;
        }
        if (rank < P.k) {
            P.out_ids[(size_t)q * P.k + rank] = id;
            P.out_dists[(size_t)q * P.k + rank] = __uint_as_float(d);
        }
    }
    for (uint32_t e = count + lane; e < P.k; e += 64) {
        P.out_ids[(size_t)q * P.k + e] = ~0ull;
        P.out_dists[(size_t)q * P.k + e] = __builtin_inff();
    }
    if (lane == 0) P.out_counts[q] = count;
}

} // namespace granne_hip
