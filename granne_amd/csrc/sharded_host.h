// sharded_host.h -- a partitioned index driven by ONE host process (included by granne_hip.hip).
//
// The reference partitions by splitting the element set into independent indexes
// (/root/reference/src/elements/embeddings/parsing.rs:63-100: shards of consecutive elements); a search
// asks every shard and keeps the best num_neighbors by (dist, global id), global id = shard offset +
// the shard's local id. Here shard s is a granne_hip_index of its own, on whatever device it was created
// on: every shard searches the same batch on its own stream (concurrently across devices), its packed
// top-k (granne_hip_packed_topk_bytes) is copied to the merge device (peer copy over xGMI when the
// shard lives elsewhere), and merge_topk_kernel ranks the n_shards*k candidates of each query.
// One process per GPU with a collective instead of peer copies is granne_amd/sharded.py; both use the
// same search and merge entry points and return the same bits.
#pragma once

struct granne_hip_sharded {
    struct Shard {
        granne_hip_index* ix = nullptr;
        uint64_t offset = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        uint8_t* d_queries = nullptr; // on the shard's device
        uint8_t* d_packed = nullptr;  // on the shard's device
        uint32_t* d_status = nullptr; // u32[4] on the shard's device
        size_t q_cap = 0, p_cap = 0;
    };
    std::vector<Shard> shards;
    int merge_device = 0;
    hipStream_t merge_stream = nullptr;
    uint8_t* d_gather = nullptr; // [n_shards][packed] on the merge device
    uint8_t* d_out = nullptr;    // merged ids | dists | counts
    size_t g_cap = 0, o_cap = 0;
    uint32_t dim = 0;
    int dtype = 0;
    std::mutex mu; // one search at a time per handle (the buffers above are the handle's)
};

static void sharded_free(granne_hip_sharded* sh) {
    if (!sh) return;
    for (auto& S : sh->shards) {
        if (!S.ix) continue;
        DeviceGuard g(S.ix->device);
        if (S.stream) (void)hipStreamDestroy(S.stream);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.d_queries) (void)hipFree(S.d_queries);
        if (S.d_packed) (void)hipFree(S.d_packed);
        if (S.d_status) (void)hipFree(S.d_status);
    }
    {
        DeviceGuard g(sh->merge_device);
        if (sh->merge_stream) (void)hipStreamDestroy(sh->merge_stream);
        if (sh->d_gather) (void)hipFree(sh->d_gather);
        if (sh->d_out) (void)hipFree(sh->d_out);
    }
    delete sh;
}

extern "C" int granne_hip_sharded_create(granne_hip_sharded** out, granne_hip_index* const* shards,
                                         const uint64_t* id_offsets, uint32_t n_shards) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!shards || !id_offsets) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (n_shards == 0 || n_shards > 64) return fail(GRANNE_HIP_ERR_INVALID, "n_shards must be in [1, 64]");
    for (uint32_t s = 0; s < n_shards; ++s) {
        if (!shards[s]) return fail(GRANNE_HIP_ERR_INVALID, "shard %u is null", s);
        if (shards[s]->dim != shards[0]->dim || shards[s]->dtype != shards[0]->dtype)
            return fail(GRANNE_HIP_ERR_INVALID, "shard %u has another element type than shard 0", s);
    }
    granne_hip_sharded* sh = new granne_hip_sharded();
    sh->dim = shards[0]->dim;
    sh->dtype = shards[0]->dtype;
    sh->merge_device = shards[0]->device;
    sh->shards.resize(n_shards);
    auto body = [&]() -> int {
        for (uint32_t s = 0; s < n_shards; ++s) {
            auto& S = sh->shards[s];
            S.ix = shards[s];
            S.offset = id_offsets[s];
            DeviceGuard g(S.ix->device);
            if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", S.ix->device);
            HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
            HIP_TRY(hipMalloc((void**)&S.d_status, 16));
            if (S.ix->device != sh->merge_device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, S.ix->device, sh->merge_device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(sh->merge_device, 0); // already enabled is fine
            }
        }
        DeviceGuard g(sh->merge_device);
        HIP_TRY(hipStreamCreateWithFlags(&sh->merge_stream, hipStreamNonBlocking));
        return GRANNE_HIP_OK;
    };
    int rc = body();
    if (rc) {
        sharded_free(sh);
        return rc;
    }
    *out = sh;
    return GRANNE_HIP_OK;
}

extern "C" void granne_hip_sharded_destroy(granne_hip_sharded* sh) { sharded_free(sh); }
extern "C" uint32_t granne_hip_sharded_num_shards(const granne_hip_sharded* sh) { return sh ? (uint32_t)sh->shards.size() : 0; }
extern "C" uint64_t granne_hip_sharded_len(const granne_hip_sharded* sh) {
    uint64_t n = 0;
    if (sh)
        for (auto& S : sh->shards) n += granne_hip_index_len(S.ix);
    return n;
}

static int grow(uint8_t** p, size_t* cap, size_t want) {
    if (*cap >= want) return GRANNE_HIP_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)p, want));
    *cap = want;
    return GRANNE_HIP_OK;
}

// queries: host, dense [nq][dim], prepared like the elements. Outputs: host, global ids.
extern "C" int granne_hip_sharded_search_batch(granne_hip_sharded* sh, const void* queries, uint32_t nq, uint32_t max_search,
                                               uint32_t num_neighbors, uint64_t* out_ids, float* out_dists,
                                               uint32_t* out_counts) {
    if (!sh) return fail(GRANNE_HIP_ERR_INVALID, "sharded index is null");
    if (max_search == 0) return fail(GRANNE_HIP_ERR_INVALID, "max_search must be > 0 (the reference panics, src/index/mod.rs:1019)");
    if (nq == 0) return GRANNE_HIP_OK;
    if (num_neighbors == 0) {
        if (!out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
        memset(out_counts, 0, (size_t)nq * 4);
        return GRANNE_HIP_OK;
    }
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    const uint32_t G = (uint32_t)sh->shards.size();
    if ((uint64_t)G * num_neighbors > 4096) return fail(GRANNE_HIP_ERR_INVALID, "n_shards * num_neighbors must be <= 4096");
    std::lock_guard<std::mutex> lk(sh->mu);
    // Whatever way this call ends, nothing it enqueued may still be running when it returns: the shard streams read the
    // caller's `queries` and write buffers the next call may regrow (grow() frees), the merge stream writes the caller's
    // outputs. On success everything has been waited for already and this costs nothing.
    struct Quiesce {
        granne_hip_sharded* sh;
        ~Quiesce() {
            for (auto& S : sh->shards) {
                DeviceGuard g(S.ix->device);
                (void)hipStreamSynchronize(S.stream);
            }
            DeviceGuard g(sh->merge_device);
            (void)hipStreamSynchronize(sh->merge_stream);
        }
    } quiesce{sh};
    const size_t k = num_neighbors;
    const size_t qb = (size_t)nq * sh->dim * elem_size(sh->dtype);
    const size_t pb = (size_t)granne_hip_packed_topk_bytes(nq, num_neighbors);
    const size_t ob = (size_t)nq * k * 12 + (size_t)nq * 4;
    {
        DeviceGuard g(sh->merge_device);
        int rc = grow(&sh->d_gather, &sh->g_cap, pb * G);
        if (rc == 0) rc = grow(&sh->d_out, &sh->o_cap, ob);
        if (rc) return rc;
    }
    // every shard: upload the batch, search, send its packed top-k to the merge device
    for (uint32_t s = 0; s < G; ++s) {
        auto& S = sh->shards[s];
        DeviceGuard g(S.ix->device);
        if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", S.ix->device);
        int rc = grow(&S.d_queries, &S.q_cap, qb);
        if (rc == 0) rc = grow(&S.d_packed, &S.p_cap, pb);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(S.d_status, 0, 16, S.stream));
        HIP_TRY(hipMemcpyAsync(S.d_queries, queries, qb, hipMemcpyHostToDevice, S.stream));
        rc = granne_hip_search_batch_packed_device(S.ix, S.d_queries, nq, max_search, num_neighbors, S.d_packed, S.d_status,
                                                   S.stream);
        if (rc) return rc;
        if (S.ix->device == sh->merge_device)
            HIP_TRY(hipMemcpyAsync(sh->d_gather + pb * s, S.d_packed, pb, hipMemcpyDeviceToDevice, S.stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(sh->d_gather + pb * s, sh->merge_device, S.d_packed, S.ix->device, pb, S.stream));
        HIP_TRY(hipEventRecord(S.done, S.stream));
    }
    DeviceGuard g(sh->merge_device);
    for (uint32_t s = 0; s < G; ++s) HIP_TRY(hipStreamWaitEvent(sh->merge_stream, sh->shards[s].done, 0));
    uint64_t offs[64];
    for (uint32_t s = 0; s < G; ++s) offs[s] = sh->shards[s].offset;
    uint64_t* d_ids = (uint64_t*)sh->d_out;
    float* d_d = (float*)(sh->d_out + (size_t)nq * k * 8);
    uint32_t* d_c = (uint32_t*)(sh->d_out + (size_t)nq * k * 12);
    int rc = granne_hip_merge_topk_packed_device(sh->d_gather, offs, G, nq, num_neighbors, d_ids, d_d, d_c, sh->merge_device,
                                                 sh->merge_stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_ids, d_ids, (size_t)nq * k * 8, hipMemcpyDeviceToHost, sh->merge_stream));
    HIP_TRY(hipMemcpyAsync(out_dists, d_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost, sh->merge_stream));
    HIP_TRY(hipMemcpyAsync(out_counts, d_c, (size_t)nq * 4, hipMemcpyDeviceToHost, sh->merge_stream));
    HIP_TRY(hipStreamSynchronize(sh->merge_stream));
    // a shard whose exact-search scratch ran out wrote empty results for those queries: report, never merge silently
    for (uint32_t s = 0; s < G; ++s) {
        auto& S = sh->shards[s];
        DeviceGuard gs(S.ix->device);
        uint32_t st[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(st, S.d_status, 16, hipMemcpyDeviceToHost));
        if (st[0]) return fail(GRANNE_HIP_ERR_OVERFLOW, "shard %u: exact-search scratch exhausted (raise GRANNE_HIP_OPT_SLOW_SLOTS)", s);
    }
    return GRANNE_HIP_OK;
}

// Granne::search on a partitioned index: one query
extern "C" int granne_hip_sharded_search(granne_hip_sharded* sh, const void* query, uint32_t max_search,
                                         uint32_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_count) {
    if (!out_count) return fail(GRANNE_HIP_ERR_INVALID, "out_count is null");
    return granne_hip_sharded_search_batch(sh, query, 1, max_search, num_neighbors, out_ids, out_dists, out_count);
}
