// builder_host.h -- host side of the GPU GranneBuilder (included by granne_hip.hip).
// Restates the control flow of /root/reference/src/index/mod.rs:364-402 (build_partial),
// :634-643 (compute_num_elements_in_layer), :646-713 (index_elements_in_last_layer) and :715-802
// (index_elements) around the batched device kernels of builder_kernels.h.
#pragma once

#include <hipcub/hipcub.hpp>

#include <cmath>

#include "builder_kernels.h"

struct BuilderLayer {
    uint64_t len = 0;      // rows in use
    uint64_t cap_rows = 0; // rows allocated
    uint32_t* d_adj = nullptr;
};

struct granne_hip_builder {
    int device = 0;
    granne_hip_build_config cfg;
    uint32_t dim = 0;
    int dtype = 0;
    uint64_t n_elements = 0;
    uint32_t row_bytes = 0, row_stride = 0; // data bytes of a device row; bytes from one row to the next
    uint8_t* d_elements = nullptr;
    uint32_t W = 32; // device row width
    std::vector<BuilderLayer> layers;
    uint64_t hbm_bytes = 0;
    ScratchCache scratch; // the build's searches (search_launch)
};

extern "C" void granne_hip_build_config_default(granne_hip_build_config* c) {
    if (!c) return;
    c->layer_multiplier = 15.0f;
    c->expected_num_elements = 0;
    c->num_neighbors = 30;
    c->max_search = 200;
    c->reinsert_elements = 1;
    c->show_progress = 0;
    c->batch_max = 0;
    c->batch_div = 0;
}

// compute_num_elements_in_layer, src/index/mod.rs:634-643
static uint64_t num_elements_in_layer(uint64_t total, float layer_multiplier, uint64_t layer_idx) {
    double m = (double)layer_multiplier;
    double t = (double)total;
    double e = std::floor(std::log(t) / std::log(m)) - (double)layer_idx;
    double v = std::ceil(t / std::pow(m, e));
    uint64_t r;
    if (!(v >= 0.0)) r = 0;
    else if (v >= 18446744073709551615.0) r = UINT64_MAX;
    else r = (uint64_t)v;
    return r < total ? r : total;
}

static void destroy_builder(granne_hip_builder* b) {
    if (!b) return;
    DeviceGuard g(b->device);
    if (b->d_elements) (void)hipFree(b->d_elements);
    for (auto& L : b->layers)
        if (L.d_adj) (void)hipFree(L.d_adj);
    b->scratch.free_all();
    delete b;
}

static int builder_validate(granne_hip_builder** out, const granne_hip_build_config* cfg, uint64_t n, uint32_t dim,
                            int dtype) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!cfg) return fail(GRANNE_HIP_ERR_INVALID, "config is null");
    if (dtype != GRANNE_HIP_F32 && dtype != GRANNE_HIP_I8) return fail(GRANNE_HIP_ERR_INVALID, "unknown dtype %d", dtype);
    if (dim == 0) return fail(GRANNE_HIP_ERR_INVALID, "dim must be > 0");
    if (n >= 0xFFFFFFFFull) return fail(GRANNE_HIP_ERR_INVALID, "too many elements (src/index/mod.rs:420)");
    if (cfg->num_neighbors < 1 || cfg->num_neighbors > BUILD_MAX_NEIGHBORS)
        return fail(GRANNE_HIP_ERR_INVALID, "num_neighbors must be in [1, %u] on the GPU builder", BUILD_MAX_NEIGHBORS);
    if (cfg->max_search < 1 || cfg->max_search > BUILD_MAX_CAND)
        return fail(GRANNE_HIP_ERR_INVALID, "max_search must be in [1, %u] on the GPU builder", BUILD_MAX_CAND);
    if (!(cfg->layer_multiplier > 1.0f)) return fail(GRANNE_HIP_ERR_INVALID, "layer_multiplier must be > 1");
    if (cfg->batch_max > (1u << 20)) return fail(GRANNE_HIP_ERR_INVALID, "batch_max must be <= 2^20");
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_builder_create_device(granne_hip_builder** out, const granne_hip_build_config* cfg,
                                                const void* d_elements, uint64_t n_elements, uint32_t dim, int dtype,
                                                int device_id, void* stream) {
    int rc = builder_validate(out, cfg, n_elements, dim, dtype);
    if (rc) return rc;
    if (n_elements && !d_elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    granne_hip_builder* b = new granne_hip_builder();
    b->device = device_id;
    b->cfg = *cfg;
    if (b->cfg.batch_max == 0) b->cfg.batch_max = 65536;
    if (b->cfg.batch_div == 0) b->cfg.batch_div = 8;
    b->dim = dim;
    b->dtype = dtype;
    b->n_elements = n_elements;
    b->row_bytes = device_row_bytes(dim, dtype);
    b->row_stride = device_row_stride(dim, dtype);
    b->W = (cfg->num_neighbors + 31u) & ~31u;
    // reuse the index's element upload (re-layout to the padded device rows)
    granne_hip_index tmp;
    tmp.device = device_id;
    tmp.dim = dim;
    tmp.dtype = dtype;
    tmp.n_elements = n_elements;
    tmp.row_bytes = b->row_bytes;
    tmp.row_stride = b->row_stride;
    rc = upload_elements_from_device(&tmp, d_elements, (hipStream_t)stream);
    if (rc == 0 && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = fail(GRANNE_HIP_ERR_HIP, "sync failed");
    b->d_elements = tmp.d_elements;
    b->hbm_bytes = tmp.hbm_bytes;
    tmp.d_elements = nullptr;
    if (rc) {
        destroy_builder(b);
        return rc;
    }
    *out = b;
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_builder_create(granne_hip_builder** out, const granne_hip_build_config* cfg,
                                         const void* elements, uint64_t n_elements, uint32_t dim, int dtype,
                                         int device_id) {
    int rc = builder_validate(out, cfg, n_elements, dim, dtype);
    if (rc) return rc;
    if (n_elements && !elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    void* d = nullptr;
    size_t bytes = (size_t)n_elements * dim * elem_size(dtype);
    HIP_TRY(hipMalloc(&d, bytes ? bytes : 16));
    hipError_t e = bytes ? hipMemcpy(d, elements, bytes, hipMemcpyHostToDevice) : hipSuccess;
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(GRANNE_HIP_ERR_HIP, "element upload failed: %s", hipGetErrorString(e));
    }
    rc = granne_hip_builder_create_device(out, cfg, d, n_elements, dim, dtype, device_id, nullptr);
    (void)hipFree(d);
    return rc;
}

// Builder::push for a batch of rows (src/index/mod.rs:303-315, dense_vector.rs push): the element
// container grows, nothing is indexed until the next build.
extern "C" int granne_hip_builder_append(granne_hip_builder* b, const void* elements, uint64_t n_new) {
    if (!b) return fail(GRANNE_HIP_ERR_INVALID, "builder is null");
    if (n_new == 0) return GRANNE_HIP_OK;
    if (!elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    if (b->n_elements + n_new >= 0xFFFFFFFFull) return fail(GRANNE_HIP_ERR_INVALID, "too many elements (src/index/mod.rs:420)");
    DeviceGuard g(b->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", b->device);
    const size_t dense = (size_t)b->dim * elem_size(b->dtype);
    const size_t old_bytes = (size_t)b->n_elements * b->row_stride, add_bytes = (size_t)n_new * b->row_stride;
    uint8_t* grown = nullptr;
    void* staged = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&grown, old_bytes + add_bytes));
        HIP_TRY(hipMalloc(&staged, n_new * dense));
        HIP_TRY(hipMemcpy(staged, elements, n_new * dense, hipMemcpyHostToDevice));
        if (old_bytes) HIP_TRY(hipMemcpyAsync(grown, b->d_elements, old_bytes, hipMemcpyDeviceToDevice, nullptr));
        if (dense == b->row_stride) {
            HIP_TRY(hipMemcpyAsync(grown + old_bytes, staged, add_bytes, hipMemcpyDeviceToDevice, nullptr));
        } else {
            const uint64_t units = n_new * (b->row_stride / 16);
            hipLaunchKernelGGL(relayout_rows_kernel, dim3(grid_for(units, 256)), dim3(256), 0, nullptr,
                               (const uint8_t*)staged, grown + old_bytes, n_new, (uint32_t)dense, b->row_stride);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipDeviceSynchronize());
        return GRANNE_HIP_OK;
    };
    int rc = body();
    if (staged) (void)hipFree(staged);
    if (rc) {
        if (grown) (void)hipFree(grown);
        return rc;
    }
    if (b->d_elements) (void)hipFree(b->d_elements);
    b->d_elements = grown;
    b->n_elements += n_new;
    b->hbm_bytes += add_bytes;
    return GRANNE_HIP_OK;
}

extern "C" void granne_hip_builder_destroy(granne_hip_builder* b) { destroy_builder(b); }
extern "C" uint64_t granne_hip_builder_len(const granne_hip_builder* b) {
    return (b && !b->layers.empty()) ? b->layers.back().len : 0;
}
extern "C" uint64_t granne_hip_builder_num_elements(const granne_hip_builder* b) { return b ? b->n_elements : 0; }
extern "C" uint32_t granne_hip_builder_num_layers(const granne_hip_builder* b) { return b ? (uint32_t)b->layers.size() : 0; }
extern "C" uint64_t granne_hip_builder_layer_len(const granne_hip_builder* b, uint32_t l) {
    return (b && l < b->layers.size()) ? b->layers[l].len : 0;
}

// ------------------------------------------------------------------------------------------------
// one index_elements pass (first insertion or reinsertion) over the last layer
// ------------------------------------------------------------------------------------------------
typedef void (*build_fn)(const BuildParams);
struct BuildKernels {
    build_fn select, apply, final_prune;
};
template <int DT, int DIM>
static BuildKernels build_kernels_of() {
    return {select_kernel<DT, DIM>, apply_kernel<DT, DIM>, final_prune_kernel<DT, DIM>};
}
static BuildKernels pick_build_kernels(int dtype, uint32_t dim) {
    if (dtype == GRANNE_HIP_I8) return build_kernels_of<DT_I8, 0>();
    switch (dim) {
    case 100: return build_kernels_of<DT_F32, 100>();
    case 200: return build_kernels_of<DT_F32, 200>();
    default: return build_kernels_of<DT_F32, 0>();
    }
}

struct BuildScratch {
    uint64_t *s_ids = nullptr, *op_keys = nullptr, *op_vals = nullptr, *sorted_keys = nullptr, *sorted_vals = nullptr;
    float* s_dists = nullptr;
    uint32_t *s_counts = nullptr, *seg_start = nullptr, *counters = nullptr; // counters: [0]=n_seg [1]=status [2]=slow count
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    LayerDev* d_layers = nullptr;
    uint8_t* selected = nullptr; // per row of the layer being built: BuildParams::selected
    void free_all() {
        void* ps[] = {s_ids, op_keys, op_vals, sorted_keys, sorted_vals, s_dists, s_counts, seg_start, counters, sort_tmp, d_layers, selected};
        for (void* p : ps)
            if (p) (void)hipFree(p);
    }
};

static int index_elements_pass(granne_hip_builder* b, uint32_t m_layer, uint32_t max_search, bool reinsert,
                               uint64_t already, BuildScratch& S, hipStream_t s) {
    const uint32_t last = (uint32_t)b->layers.size() - 1;
    BuilderLayer& L = b->layers[last];
    const uint64_t layer_len = L.len;
    const uint32_t cap = b->cfg.num_neighbors;
    const uint64_t total = reinsert ? layer_len : layer_len - already;
    const uint32_t bmax = b->cfg.batch_max;
    const uint32_t lrow = ((b->row_bytes / 16) | 1u) * 16u;
    const BuildKernels K = pick_build_kernels(b->dtype, b->dim);
    // (apply / final_prune only ever see cap + 1 candidates: their arrays keep the minimum size)
    const uint32_t cand_cap = build_cand_cap(max_search);
    // long rows stage fewer candidates per gather round (same results, more rounds): 32 up to 640-d f32, 16 at 768-d
    uint32_t sel_lds = cap; // the rows select_neighbors has selected stay in LDS -- all of them, unless they are too long for that
    uint32_t chunk = build_chunk_for(lrow, cap, cand_cap, 160u * 1024u);
    if (chunk == 0) {
        sel_lds = build_sel_lds_for(lrow, cap, cand_cap, 160u * 1024u);
        if (sel_lds >= cap)
            return fail(GRANNE_HIP_ERR_INVALID, "dimension too large for the GPU builder: select_neighbors stages the node's row and at "
                        "least 4 candidate rows of %u bytes in LDS (%u bytes, a CU has 163840)", lrow,
                        build_lds_bytes(lrow, cap, cand_cap, 4, 0u));
        chunk = 4;
    }
    const uint32_t lds = build_lds_bytes(lrow, cap, cand_cap, chunk, sel_lds);
    const uint32_t lds_rows = build_lds_bytes_rows(lrow, cap, cand_cap, chunk, sel_lds); // apply / final_prune (<= lds)
    if (lds > 64u * 1024u) {
        HIP_TRY(hipFuncSetAttribute((const void*)K.select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)K.apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)K.final_prune, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }

    SearchTarget T;
    T.device = b->device;
    T.d_elements = b->d_elements;
    T.n_elements = b->n_elements;
    T.dim = b->dim;
    T.dtype = b->dtype;
    T.row_bytes = b->row_bytes;
    T.row_stride = b->row_stride;
    T.d_layers = S.d_layers;
    T.n_layers = last + 1;
    T.max_dev_width = b->W;
    T.opt_visited_slots = 0;
    T.opt_force_slow = 0;
    T.opt_slow_slots = 1u << 18;
    T.opt_slow_blocks = 16;
    T.opt_overflow_slots = 0;
    T.scratch = &b->scratch;

    BuildParams P;
    P.elements = b->d_elements;
    P.row_bytes = b->row_bytes;
    P.row_stride = b->row_stride;
    P.dim = b->dim;
    P.lrow = lrow;
    P.adj = L.d_adj;
    P.W = b->W;
    P.cap = cap;
    P.m_layer = m_layer;
    P.layer_len = layer_len;
    P.efc = max_search;
    P.s_ids = S.s_ids;
    P.s_dists = S.s_dists;
    P.s_counts = S.s_counts;
    P.op_keys = S.op_keys;
    P.op_vals = S.op_vals;
    P.sorted_keys = S.sorted_keys;
    P.sorted_vals = S.sorted_vals;
    P.seg_start = S.seg_start;
    P.n_seg = S.counters;
    P.selected = S.selected;
    P.cand_cap = cand_cap;
    P.chunk = chunk;
    P.sel_lds = sel_lds;
    HIP_TRY(hipMemsetAsync(S.selected, 0, layer_len ? layer_len : 1, s)); // nothing is known about the rows of a pass

    const bool debug = getenv("GRANNE_HIP_DEBUG") != nullptr;
    auto dbg = [&](const char* what, uint64_t pos_, uint64_t B_) {
        if (!debug) return;
        hipError_t e = hipStreamSynchronize(s);
        fprintf(stderr, "[granne_hip build] layer %u %s pos %llu batch %llu: %s\n", last, what,
                (unsigned long long)pos_, (unsigned long long)B_, hipGetErrorString(e));
        fflush(stderr);
    };
    uint64_t pos = 0;
    while (pos < total) {
        const uint64_t n_in_graph = reinsert ? layer_len : already + pos;
        uint64_t B = n_in_graph / b->cfg.batch_div;
        if (B < 1) B = 1;
        if (B > bmax) B = bmax;
        if (B > total - pos) B = total - pos;
        const int64_t first = reinsert ? (int64_t)(layer_len - 1 - pos) : (int64_t)(already + pos);
        const int64_t step = reinsert ? -1 : 1;

        // phase A: entry search through the previous layers + search_for_neighbors on this layer
        int rc = search_launch(&T, b->d_elements + (size_t)first * b->row_stride, step * (int64_t)b->row_stride, (uint32_t)B,
                               max_search, max_search, S.s_ids, S.s_dists, S.s_counts, nullptr, S.counters + 1, s,
                               nullptr);
        if (rc) return rc;
        dbg("search", pos, B);
        P.first_idx = first;
        P.idx_step = step;
        P.batch = (uint32_t)B;
        P.n_ops = (uint32_t)(B * cap * 2);
        hipLaunchKernelGGL(K.select, dim3((uint32_t)B), dim3(64), lds, s, P);
        HIP_TRY(hipGetLastError());
        dbg("select", pos, B);

        // phase B: sort the ops by (target row, order), one wave per target row replays them
        size_t tmp_bytes = S.sort_tmp_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(S.sort_tmp, tmp_bytes, S.op_keys, S.sorted_keys, S.op_vals,
                                                   S.sorted_vals, (int)P.n_ops, 0, OP_KEY_BITS, s));
        dbg("sort", pos, B);
        HIP_TRY(hipMemsetAsync(S.counters, 0, 4, s));
        hipLaunchKernelGGL(mark_heads_kernel, dim3(grid_for(P.n_ops, 256)), dim3(256), 0, s, S.sorted_keys, P.n_ops,
                           S.seg_start, S.counters);
        HIP_TRY(hipGetLastError());
        uint32_t grid = P.n_ops < 4096 ? P.n_ops : 4096;
        hipLaunchKernelGGL(K.apply, dim3(grid), dim3(64), lds_rows, s, P);
        HIP_TRY(hipGetLastError());
        dbg("apply", pos, B);
        pos += B;
    }
    // limit number of neighbors, src/index/mod.rs:795-797
    uint32_t grid = layer_len < 8192 ? (uint32_t)layer_len : 8192u;
    hipLaunchKernelGGL(K.final_prune, dim3(grid ? grid : 1), dim3(64), lds_rows, s, P);
    HIP_TRY(hipGetLastError());
    dbg("final_prune", total, 0);
    return GRANNE_HIP_OK;
}

// index_elements_in_last_layer, src/index/mod.rs:646-713
static int index_elements_in_last_layer(granne_hip_builder* b, uint64_t max_num_elements, hipStream_t s) {
    const uint64_t total = b->cfg.expected_num_elements ? b->cfg.expected_num_elements : b->n_elements;
    const uint64_t t2 = total > b->n_elements ? total : b->n_elements;
    const uint32_t last = (uint32_t)b->layers.size() - 1;
    BuilderLayer& L = b->layers[last];
    const uint64_t ideal = num_elements_in_layer(t2, b->cfg.layer_multiplier, last);
    if (ideal <= L.len) return GRANNE_HIP_OK; // nothing to index in this layer, :654-657
    const uint64_t num_in_layer = max_num_elements < ideal ? max_num_elements : ideal;
    uint32_t m_layer = b->cfg.num_neighbors;
    if (ideal < total) m_layer = m_layer / 2 > 1 ? m_layer / 2 : 1; // half num_neighbors on upper layers, :665-668
    uint32_t max_search = b->cfg.max_search;
    if (b->cfg.show_progress) {
        printf("Building layer %u with %llu elements...\n", last, (unsigned long long)num_in_layer);
        fflush(stdout);
    }

    // layer.resize(num_elements, UNUSED), :730 (capacity: `ideal` rows, :670)
    const uint64_t already = L.len;
    if (num_in_layer > L.cap_rows) {
        uint32_t* nrows = nullptr;
        size_t bytes = (size_t)ideal * b->W * 4;
        HIP_TRY(hipMalloc((void**)&nrows, bytes));
        b->hbm_bytes += bytes;
        HIP_TRY(hipMemsetAsync(nrows, 0xFF, bytes, s));
        if (L.len) HIP_TRY(hipMemcpyAsync(nrows, L.d_adj, (size_t)L.len * b->W * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (L.d_adj) {
            b->hbm_bytes -= (size_t)L.cap_rows * b->W * 4;
            (void)hipFree(L.d_adj);
        }
        L.d_adj = nrows;
        L.cap_rows = ideal;
    }
    L.len = num_in_layer;

    // scratch sized for the largest batch of this pass
    BuildScratch S;
    const uint64_t bmaxu = std::min<uint64_t>(b->cfg.batch_max, num_in_layer);
    const uint64_t n_ops_max = bmaxu * b->cfg.num_neighbors * 2;
    int rc = GRANNE_HIP_OK;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&S.s_ids, bmaxu * max_search * 8));
        HIP_TRY(hipMalloc((void**)&S.s_dists, bmaxu * max_search * 4));
        HIP_TRY(hipMalloc((void**)&S.s_counts, bmaxu * 4));
        HIP_TRY(hipMalloc((void**)&S.op_keys, n_ops_max * 8));
        HIP_TRY(hipMalloc((void**)&S.op_vals, n_ops_max * 8));
        HIP_TRY(hipMalloc((void**)&S.sorted_keys, n_ops_max * 8));
        HIP_TRY(hipMalloc((void**)&S.sorted_vals, n_ops_max * 8));
        HIP_TRY(hipMalloc((void**)&S.seg_start, n_ops_max * 4));
        HIP_TRY(hipMalloc((void**)&S.counters, 32));
        HIP_TRY(hipMalloc((void**)&S.selected, num_in_layer ? num_in_layer : 1));
        HIP_TRY(hipMemsetAsync(S.counters, 0, 32, s));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, S.sort_tmp_bytes, S.op_keys, S.sorted_keys, S.op_vals,
                                                   S.sorted_vals, (int)n_ops_max, 0, OP_KEY_BITS, s));
        HIP_TRY(hipMalloc(&S.sort_tmp, S.sort_tmp_bytes ? S.sort_tmp_bytes : 16));
        // layer table: finished layers + the one being built
        std::vector<LayerDev> h(last + 1);
        for (uint32_t l = 0; l <= last; ++l) {
            h[l].adj = b->layers[l].d_adj;
            h[l].len = b->layers[l].len;
            h[l].width = b->W;
            h[l].flags = 0; // connect_nodes never lists a neighbor twice (mod.rs:913-917)
        }
        HIP_TRY(hipMalloc((void**)&S.d_layers, sizeof(LayerDev) * h.size()));
        HIP_TRY(hipMemcpyAsync(S.d_layers, h.data(), sizeof(LayerDev) * h.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));

        int r = index_elements_pass(b, m_layer, max_search, false, already, S, s);
        if (r) return r;
        if (b->cfg.reinsert_elements) { // :692-710
            if (b->cfg.show_progress) {
                printf("Reinserting elements...\n");
                fflush(stdout);
            }
            uint32_t ms2 = max_search / 2 > 1 ? max_search / 2 : 1;
            r = index_elements_pass(b, m_layer, ms2, true, 0, S, s);
            if (r) return r;
        }
        uint32_t hc[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(hc, S.counters, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (hc[1]) return fail(GRANNE_HIP_ERR_OVERFLOW, "exact-search scratch exhausted during build");
        return GRANNE_HIP_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(s);
    S.free_all();
    return rc;
}

// build_partial, src/index/mod.rs:374-402
extern "C" int granne_hip_builder_build(granne_hip_builder* b, uint64_t num_elements) {
    if (!b) return fail(GRANNE_HIP_ERR_INVALID, "builder is null");
    if (num_elements == GRANNE_HIP_BUILD_ALL) num_elements = b->n_elements; // Builder::build(), mod.rs:370-372
    if (num_elements == 0) return GRANNE_HIP_OK; // build_partial(0) is a no-op, mod.rs:375
    if (num_elements > b->n_elements) return fail(GRANNE_HIP_ERR_INVALID, "Cannot index more elements than exist.");
    if (!b->layers.empty() && num_elements < b->layers.back().len)
        return fail(GRANNE_HIP_ERR_INVALID, "Cannot index fewer elements than already in index.");
    DeviceGuard g(b->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", b->device);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int rc = GRANNE_HIP_OK;
    if (!b->layers.empty()) rc = index_elements_in_last_layer(b, num_elements, s);
    while (rc == 0 && granne_hip_builder_len(b) < num_elements) {
        BuilderLayer nl; // with_width(num_neighbors) or prev_layer.clone(), :393-398
        if (!b->layers.empty()) {
            const BuilderLayer& pl = b->layers.back();
            size_t bytes = (size_t)pl.len * b->W * 4;
            hipError_t e = hipMalloc((void**)&nl.d_adj, bytes ? bytes : 16);
            // on the builder's own stream: a device-to-device hipMemcpy on the null stream is
            // asynchronous and a non-blocking stream would not wait for it
            if (e == hipSuccess && bytes) e = hipMemcpyAsync(nl.d_adj, pl.d_adj, bytes, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                rc = fail(GRANNE_HIP_ERR_HIP, "layer clone failed: %s", hipGetErrorString(e));
                break;
            }
            nl.len = pl.len;
            nl.cap_rows = pl.len;
            b->hbm_bytes += bytes;
        }
        if (b->layers.size() >= 64) {
            rc = fail(GRANNE_HIP_ERR_INVALID, "too many layers");
            break;
        }
        b->layers.push_back(nl);
        rc = index_elements_in_last_layer(b, num_elements, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return rc;
}

extern "C" int granne_hip_builder_get_layer(const granne_hip_builder* b, uint32_t layer, uint32_t* out_rows) {
    if (!b || !out_rows) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (layer >= b->layers.size()) return fail(GRANNE_HIP_ERR_INVALID, "layer out of range");
    DeviceGuard g(b->device);
    const BuilderLayer& L = b->layers[layer];
    const uint32_t nn = b->cfg.num_neighbors;
    HIP_TRY(hipMemcpy2D(out_rows, (size_t)nn * 4, L.d_adj, (size_t)b->W * 4, (size_t)nn * 4, L.len, hipMemcpyDeviceToHost));
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_builder_get_index(const granne_hip_builder* b, granne_hip_index** out) {
    if (!b || !out) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    *out = nullptr;
    DeviceGuard g(b->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", b->device);
    granne_hip_index* ix = new granne_hip_index();
    ix->device = b->device;
    ix->dim = b->dim;
    ix->dtype = b->dtype;
    ix->n_elements = b->n_elements;
    ix->row_bytes = b->row_bytes;
    ix->row_stride = b->row_stride;
    auto body = [&]() -> int {
        size_t eb = (size_t)b->n_elements * b->row_stride;
        HIP_TRY(hipMalloc((void**)&ix->d_elements, eb ? eb : 16));
        ix->hbm_bytes += eb;
        if (eb) HIP_TRY(hipMemcpy(ix->d_elements, b->d_elements, eb, hipMemcpyDeviceToDevice));
        for (const auto& BL : b->layers) {
            LayerHost L;
            L.len = BL.len;
            L.width = b->cfg.num_neighbors;
            L.dev_width = b->W;
            size_t bytes = (size_t)BL.len * b->W * 4;
            HIP_TRY(hipMalloc((void**)&L.d_adj, bytes ? bytes : 16));
            ix->hbm_bytes += bytes;
            ix->layers.push_back(L);
            if (bytes) HIP_TRY(hipMemcpy(L.d_adj, BL.d_adj, bytes, hipMemcpyDeviceToDevice));
        }
        return finish_layers(ix, nullptr);
    };
    int rc = body();
    if (rc) {
        destroy_index(ix);
        return rc;
    }
    *out = ix;
    return GRANNE_HIP_OK;
}
