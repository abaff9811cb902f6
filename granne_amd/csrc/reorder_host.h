// reorder_host.h -- Granne::reorder / reorder_by_keys (/root/reference/src/index/reorder.rs) on the
// device. Included by granne_hip.hip.
//
//   compute_order (:135-175)        trails = the search kernel in trail mode (one wave per element,
//                                   max_search 1 from node 0 in every upper layer, :180-208); keys are
//                                   mapped through order_inv and the (eps, idx) tuples sorted by an LSD
//                                   radix sort, two trail columns per 64-bit pass, stable in idx.
//   reorder_layers (:210-281)       one wave per new row: gather the old row of order[i], map ids
//                                   through the reverse mapping, sort ascending (MultiSetVector::push
//                                   sorts, src/slice_vector/set_vector.rs:41-47).
//   elements.permute (mod.rs:437)   row gather, 16 bytes per lane.
#pragma once

#include <hipcub/hipcub.hpp>

namespace granne_hip {

__global__ void iota_u32_kernel(uint32_t* out, uint64_t n, uint32_t first) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = first + (uint32_t)t;
}

// key[j] = (order_inv[trail[perm[j]-lo][c]] << 32) | order_inv[trail[perm[j]-lo][c+1]]   (reorder.rs:159)
__global__ void trail_keys_kernel(const uint32_t* __restrict__ trail, const uint32_t* __restrict__ perm,
                                  const uint32_t* __restrict__ order_inv, uint64_t n, uint32_t lo, uint32_t c,
                                  uint64_t* __restrict__ keys) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t* row = trail + (size_t)(perm[t] - lo) * TRAIL_WIDTH;
        keys[t] = ((uint64_t)order_inv[row[c]] << 32) | (uint64_t)order_inv[row[c + 1]];
    }
}

// order_inv[order[i]] = i for i in [lo, lo+n)   (reorder.rs:167-171); order points at position lo
__global__ void scatter_inverse_kernel(const uint32_t* __restrict__ order, uint64_t n, uint32_t lo,
                                       uint32_t* __restrict__ inv) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        inv[order[t]] = lo + (uint32_t)t;
}

__global__ void gather_u64_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                                  uint64_t* __restrict__ out) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = src[idx[t]];
}

__global__ void widen_u32_kernel(const uint32_t* __restrict__ src, uint64_t n, uint64_t* __restrict__ out) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = src[t];
}

// new_rows[i] = rows[order[i]], 16 bytes per lane
__global__ void permute_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                    const uint32_t* __restrict__ order, uint64_t n, uint32_t row_bytes /* the stride: padding travels too */) {
    const uint32_t units = row_bytes >> 4;
    const uint64_t total = n * units;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = t / units;
        const uint32_t u = (uint32_t)(t - row * units);
        *reinterpret_cast<uint4*>(dst + row * row_bytes + (size_t)u * 16) =
            *reinterpret_cast<const uint4*>(src + (size_t)order[row] * row_bytes + (size_t)u * 16);
    }
}

// reorder_layer (:230-281): one wave per new row
__global__ __launch_bounds__(64) void remap_rows_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                        const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ rev, uint64_t len, uint32_t W) {
    extern __shared__ __align__(16) uint8_t smem_r[];
    uint32_t* v = reinterpret_cast<uint32_t*>(smem_r); // [W]
    __shared__ uint32_t first_unused;
    const uint32_t lane = threadIdx.x;
    for (uint64_t i = blockIdx.x; i < len; i += gridDim.x) {
        __syncthreads();
        if (lane == 0) first_unused = W;
        __syncthreads();
        const uint32_t* row = src + (size_t)order[i] * W;
        for (uint32_t c = lane; c < W; c += 64) {
            const uint32_t x = row[c];
            v[c] = x;
            if (x == 0xFFFFFFFFu) atomicMin(&first_unused, c);
        }
        __syncthreads();
        const uint32_t nvalid = first_unused; // get_neighbors takes the prefix before the first UNUSED (mod.rs:540-552)
        __syncthreads();
        for (uint32_t c = lane; c < W; c += 64) v[c] = (c < nvalid) ? rev[v[c]] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t c = lane; c < W; c += 64) {
            const uint32_t x = v[c];
            uint32_t rank = 0;
            for (uint32_t o = 0; o < W; ++o) {
                const uint32_t y = v[o];
                rank += (y < x || (y == x && o < c)) ? 1u : 0u;
            }
            dst[(size_t)i * W + rank] = x;
        }
    }
}

} // namespace granne_hip

namespace {

struct ReorderScratch {
    std::vector<void*> ptrs;
    ~ReorderScratch() {
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, count ? count * sizeof(T) : 16);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T*)p;
        return e;
    }
};

// stable sort of (keys, vals) by the full 64-bit key
static int sort_pairs_u64(ReorderScratch& S, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in,
                          uint32_t* vals_out, uint64_t n, void*& tmp, size_t& tmp_bytes, hipStream_t s) {
    size_t need = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 64, s));
    if (need > tmp_bytes) {
        HIP_TRY(S.alloc((uint8_t**)&tmp, need));
        tmp_bytes = need;
    }
    size_t use = tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, use, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 64, s));
    return GRANNE_HIP_OK;
}

// reorder_layers + elements.permute with a device order (u32, all of [0, n))
static int apply_order(granne_hip_index* ix, const uint32_t* d_order, ReorderScratch& S, hipStream_t s) {
    using namespace granne_hip;
    const uint64_t n = ix->n_elements;
    uint32_t* d_rev = nullptr;
    HIP_TRY(S.alloc(&d_rev, n));
    hipLaunchKernelGGL(scatter_inverse_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_order, n, 0u, d_rev);
    HIP_TRY(hipGetLastError());
    // new buffers first, swap only once everything has been produced
    std::vector<uint32_t*> new_adj(ix->layers.size(), nullptr);
    uint8_t* new_el = nullptr;
    auto drop = [&]() {
        for (auto p : new_adj)
            if (p) (void)hipFree(p);
        if (new_el) (void)hipFree(new_el);
    };
    auto body = [&]() -> int {
        for (size_t l = 0; l < ix->layers.size(); ++l) {
            const LayerHost& L = ix->layers[l];
            size_t bytes = (size_t)L.len * L.dev_width * 4;
            HIP_TRY(hipMalloc((void**)&new_adj[l], bytes ? bytes : 16));
            if (L.len == 0) continue;
            uint32_t grid = L.len < 65536 ? (uint32_t)L.len : 65536u;
            hipLaunchKernelGGL(remap_rows_kernel, dim3(grid), dim3(64), L.dev_width * 4, s, L.d_adj, new_adj[l], d_order,
                               d_rev, L.len, L.dev_width);
            HIP_TRY(hipGetLastError());
        }
        size_t el_bytes = (size_t)n * ix->row_stride;
        HIP_TRY(hipMalloc((void**)&new_el, el_bytes ? el_bytes : 16));
        if (n) {
            hipLaunchKernelGGL(permute_rows_kernel, dim3(grid_for(n * (ix->row_stride >> 4), 256)), dim3(256), 0, s,
                               ix->d_elements, new_el, d_order, n, ix->row_stride);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(s));
        return GRANNE_HIP_OK;
    };
    int rc = body();
    if (rc) {
        drop();
        return rc;
    }
    for (size_t l = 0; l < ix->layers.size(); ++l) {
        (void)hipFree(ix->layers[l].d_adj);
        ix->layers[l].d_adj = new_adj[l];
    }
    (void)hipFree(ix->d_elements);
    ix->d_elements = new_el;
    {   // the rows have moved: the scan's per-row norms are taken again when it next runs
        std::lock_guard<std::mutex> lk(ix->norm_mu);
        if (ix->d_inv_norm) {
            (void)hipFree(ix->d_inv_norm);
            ix->hbm_bytes -= inv_norm_bytes(ix->n_elements);
            ix->d_inv_norm = nullptr;
        }
    }
    return finish_layers(ix, s); // (the walkers' copies of the layers, LayerDev::adjx, are made again there)
}

static int reorder_precheck(granne_hip_index* ix) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (ix->layers.empty()) return fail(GRANNE_HIP_ERR_INVALID, "reorder of an index without layers (the reference panics)");
    if (ix->layers.back().len != ix->n_elements)
        return fail(GRANNE_HIP_ERR_INVALID,
                    "reorder needs len() == number of elements (the reference asserts, src/slice_vector/mod.rs:438)");
    return GRANNE_HIP_OK;
}

static int order_to_host(const uint32_t* d_order, uint64_t n, uint64_t* out_order, ReorderScratch& S, hipStream_t s) {
    if (!out_order || n == 0) return GRANNE_HIP_OK;
    uint64_t* d64 = nullptr;
    HIP_TRY(S.alloc(&d64, n));
    hipLaunchKernelGGL(granne_hip::widen_u32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_order, n, d64);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_order, d64, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GRANNE_HIP_OK;
}

} // namespace

extern "C" int granne_hip_index_reorder(granne_hip_index* ix, uint64_t* out_order) {
    using namespace granne_hip;
    int rc = reorder_precheck(ix);
    if (rc) return rc;
    const uint32_t n_layers = (uint32_t)ix->layers.size();
    if (n_layers < 2)
        return fail(GRANNE_HIP_ERR_INVALID, "reorder needs at least two layers (the reference panics, src/index/reorder.rs:137)");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    ReorderScratch S;
    auto body = [&]() -> int {
        const uint64_t n = ix->n_elements;
        const uint64_t inv_len = ix->layers[n_layers - 2].len;
        uint64_t widest = 0;
        for (uint32_t l = 1; l < n_layers; ++l) widest = std::max(widest, ix->layers[l].len - ix->layers[l - 1].len);
        uint32_t *d_order = nullptr, *d_inv = nullptr, *d_trail = nullptr, *d_perm_b = nullptr, *d_overflow = nullptr;
        uint64_t *d_keys_a = nullptr, *d_keys_b = nullptr;
        HIP_TRY(S.alloc(&d_order, n));
        HIP_TRY(S.alloc(&d_inv, inv_len));
        HIP_TRY(S.alloc(&d_trail, widest * TRAIL_WIDTH));
        HIP_TRY(S.alloc(&d_perm_b, widest));
        HIP_TRY(S.alloc(&d_keys_a, widest));
        HIP_TRY(S.alloc(&d_keys_b, widest));
        HIP_TRY(S.alloc(&d_overflow, 4));
        HIP_TRY(hipMemsetAsync(d_inv, 0, inv_len * 4, s));     // vec![0; layer_len(num_layers - 2)], :137
        HIP_TRY(hipMemsetAsync(d_overflow, 0, 16, s));
        hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_order, n, 0u); // :136
        HIP_TRY(hipGetLastError());
        void* tmp = nullptr;
        size_t tmp_bytes = 0;
        SearchTarget T = target_of(ix);
        for (uint32_t layer = 1; layer < n_layers; ++layer) { // :149
            const uint64_t lo = ix->layers[layer - 1].len, hi = ix->layers[layer].len;
            const uint64_t m = hi - lo;
            if (m == 0) continue;
            // find_entrypoint_trail for every idx in [lo, hi); the element rows are the queries
            int r = search_launch(&T, ix->d_elements + (size_t)lo * ix->row_stride, (int64_t)ix->row_stride, (uint32_t)m, 1, 1,
                                  nullptr, nullptr, nullptr, nullptr, d_overflow, s, nullptr, d_trail, layer);
            if (r) return r;
            // (eps, idx) ascending: LSD over the column pairs, starting from idx order
            uint32_t* perm = d_order + lo; // holds lo..hi-1 from the iota
            uint32_t* other = d_perm_b;
            const uint32_t cols = layer < TRAIL_WIDTH ? layer : TRAIL_WIDTH; // columns beyond map to order_inv[0] for all
            for (int c = (int)((cols - 1) & ~1u); c >= 0; c -= 2) {
                hipLaunchKernelGGL(trail_keys_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, d_trail, perm, d_inv, m,
                                   (uint32_t)lo, (uint32_t)c, d_keys_a);
                HIP_TRY(hipGetLastError());
                r = sort_pairs_u64(S, d_keys_a, d_keys_b, perm, other, m, tmp, tmp_bytes, s);
                if (r) return r;
                std::swap(perm, other);
            }
            if (perm != d_order + lo) HIP_TRY(hipMemcpyAsync(d_order + lo, perm, m * 4, hipMemcpyDeviceToDevice, s));
            if (layer < n_layers - 1) { // :167-171
                hipLaunchKernelGGL(scatter_inverse_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, d_order + lo, m,
                                   (uint32_t)lo, d_inv);
                HIP_TRY(hipGetLastError());
            }
        }
        uint32_t h_over[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(h_over, d_overflow, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (h_over[0]) return fail(GRANNE_HIP_ERR_OVERFLOW, "a trail walk outgrew the slow-path containers (raise GRANNE_HIP_OPT_SLOW_SLOTS)");
        int r = order_to_host(d_order, n, out_order, S, s);
        if (r) return r;
        return apply_order(ix, d_order, S, s);
    };
    rc = body();
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return rc;
}

extern "C" int granne_hip_index_reorder_by_keys(granne_hip_index* ix, const uint64_t* keys, uint64_t* out_order) {
    using namespace granne_hip;
    int rc = reorder_precheck(ix);
    if (rc) return rc;
    if (!keys && ix->n_elements) return fail(GRANNE_HIP_ERR_INVALID, "keys is null");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    ReorderScratch S;
    auto body = [&]() -> int {
        const uint64_t n = ix->n_elements;
        uint32_t *d_iota = nullptr, *d_order = nullptr;
        uint64_t *d_keys = nullptr, *d_keys_out = nullptr;
        HIP_TRY(S.alloc(&d_iota, n));
        HIP_TRY(S.alloc(&d_order, n));
        HIP_TRY(S.alloc(&d_keys, n));
        HIP_TRY(S.alloc(&d_keys_out, n));
        HIP_TRY(hipMemcpyAsync(d_keys, keys, n * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_iota, n, 0u);
        HIP_TRY(hipGetLastError());
        void* tmp = nullptr;
        size_t tmp_bytes = 0;
        for (size_t layer = 0; layer < ix->layers.size(); ++layer) { // reorder.rs:96-104: (key, idx) within each layer
            const uint64_t lo = layer ? ix->layers[layer - 1].len : 0, hi = ix->layers[layer].len;
            if (hi == lo) continue;
            int r = sort_pairs_u64(S, d_keys + lo, d_keys_out + lo, d_iota + lo, d_order + lo, hi - lo, tmp, tmp_bytes, s);
            if (r) return r;
        }
        int r = order_to_host(d_order, n, out_order, S, s);
        if (r) return r;
        return apply_order(ix, d_order, S, s);
    };
    rc = body();
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return rc;
}
