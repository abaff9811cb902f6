// sharded_host.h -- a partitioned index driven by ONE host process (included by granne_hip.hip).
//
// The reference partitions by splitting the element set into independent indexes
// (/root/reference/src/elements/embeddings/parsing.rs:63-100: shards of consecutive elements); a search
// asks every shard and keeps the best num_neighbors by (dist, global id), global id = shard offset +
// the shard's local id. Here shard s is a granne_hip_index of its own, on whatever device it was created
// on. One batch:
//   * every shard searches the batch on a stream of its own (concurrently across and within devices) and
//     writes its packed top-k + four status words (granne_hip_packed_topk_bytes + 16) straight into its
//     place of the merge device's gather buffer when it lives there -- no copy at all -- else into a buffer
//     on its own device;
//   * the exchange step: peer copies over xGMI into the gather buffer (default), or ONE grouped
//     ncclAllGather over a communicator of the shard devices (RCCL, loaded with dlopen when the option asks
//     for it: the collective BASELINE.json's north star names; this library does not link librccl);
//   * merge_topk_kernel ranks the n_shards*k candidates of each query, fold_status_kernel folds the
//     shards' status words into the caller's -- nothing is read back shard by shard.
// Everything is stream-ordered: `begin` orders the batch after what the caller's stream holds and returns
// at once, `end` makes the caller's stream wait for the merged result; between the two the caller may
// begin further batches (up to the handle's depth, two by default), so batch b+1 is searched while batch
// b is exchanged and merged. The host-pointer calls stage through pinned memory and pipeline the same way.
// One process per GPU with torch.distributed's collective is granne_amd/sharded.py; both use the same
// search and merge entry points and return the same bits.
#pragma once

#include <dlfcn.h>

#include <algorithm>
#include <thread>

// ---- RCCL, bound at run time ------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why; // when it could not be loaded
};
constexpr int RCCL_UINT8 = 1; // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

static RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process has ONE HIP runtime; an RCCL built against another copy of it would see no devices. So: a librccl
        // the process has loaded already (PyTorch-ROCm brings its own) is taken first, then GRANNE_HIP_RCCL_LIB, then
        // the system's.
        const char* env = getenv("GRANNE_HIP_RCCL_LIB");
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (!api.handle && env && *env) api.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) {
            const char* e = dlerror();
            api.why = e ? e : "librccl.so not found";
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(api.handle, name);
            if (!p && api.why.empty()) api.why = std::string("librccl lacks ") + name;
            return p;
        };
        api.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
        api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return &api;
}

#define RCCL_TRY(api, expr)                                                                                         \
    do {                                                                                                            \
        int r_ = (expr);                                                                                            \
        if (r_ != 0)                                                                                                \
            return fail(GRANNE_HIP_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(r_) : "?"); \
    } while (0)

// ---- the handle ---------------------------------------------------------------------------------------------
constexpr uint32_t SHARDED_MAX_DEPTH = 8;
constexpr size_t SHARD_STATUS_BYTES = 16;

struct granne_hip_sharded {
    struct Shard {
        granne_hip_index* ix = nullptr;
        uint64_t offset = 0;
        hipStream_t stream = nullptr;
        uint32_t dev = 0;   // index into `devices`
        uint32_t local = 0; // index among the shards of its device
    };
    struct Device {
        int id = 0;
        uint32_t n_local = 0;
        hipStream_t xstream = nullptr; // the exchange (RCCL) on this device; the merge device's is merge_stream
        void* comm = nullptr;          // ncclComm_t
    };
    // One batch in flight: every buffer a batch touches belongs to its slot.
    struct Slot {
        uint32_t nq = 0, k = 0;
        size_t stride = 0;
        bool busy = false;    // begun, not ended
        bool used = false;    // `merged` (and `xdone`) have been recorded at least once
        uint64_t seq = 0;     // the ticket's upper half: an `end` must name the begin it belongs to
        std::vector<uint8_t*> d_recv;   // per device: [G][stride]; the merge device's is the gather buffer
        std::vector<size_t> recv_cap;
        std::vector<uint8_t*> d_q;      // per device other than the merge device: the batch's queries
        std::vector<size_t> q_cap;
        hipEvent_t ready = nullptr;     // caller's stream: queries (and the previous use of the outputs) are in order
        hipEvent_t merged = nullptr;    // merge stream: the merged result is written
        std::vector<hipEvent_t> searched; // per shard
        std::vector<hipEvent_t> q_there;  // per device: the queries have arrived
        std::vector<hipEvent_t> xdone;    // per device: its part of the exchange is over (buffers reusable)
    };
    // host-pointer calls: [queries | ids | dists | counts | status] of one batch in pinned memory and on the merge device
    // (the calls' own buffers, `depth` of them in rotation: independent of which slot a batch's begin happens to take)
    struct HostIO {
        uint8_t* h_pin = nullptr;
        uint8_t* d_io = nullptr;
        size_t h_cap = 0, io_cap = 0;
        hipEvent_t h_done = nullptr;
    };
    HostIO host_io[SHARDED_MAX_DEPTH];
    std::vector<Shard> shards;
    std::vector<Device> devices;
    std::vector<Slot> slots;
    uint32_t depth = 2;
    uint32_t next_slot = 0;
    uint64_t next_seq = 1;
    int exchange = GRANNE_HIP_SHARDED_EXCHANGE_PEER;
    bool uniform = false; // shards in device-major order, the same number on every device (what one all-gather needs)
    int merge_device = 0;
    hipStream_t merge_stream = nullptr;
    hipStream_t io_stream = nullptr; // the host-pointer calls' copies
    uint32_t dim = 0;
    int dtype = 0;
    std::mutex mu;      // slot bookkeeping + the enqueue of one begin / end (host work only; nothing waits for a GPU under it)
    std::mutex host_mu; // the host-pointer calls of a handle run one at a time (they are synchronous anyway)
    bool owns_shards = false; // granne_hip_sharded_build: the shard indexes are the handle's and go with it
};

static void sharded_free_slot(granne_hip_sharded* sh, granne_hip_sharded::Slot& L) {
    for (size_t d = 0; d < sh->devices.size(); ++d) {
        DeviceGuard g(sh->devices[d].id);
        if (d < L.d_recv.size() && L.d_recv[d]) (void)hipFree(L.d_recv[d]);
        if (d < L.d_q.size() && L.d_q[d]) (void)hipFree(L.d_q[d]);
        if (d < L.q_there.size() && L.q_there[d]) (void)hipEventDestroy(L.q_there[d]);
        if (d < L.xdone.size() && L.xdone[d]) (void)hipEventDestroy(L.xdone[d]);
    }
    for (size_t s = 0; s < L.searched.size(); ++s) {
        DeviceGuard g(sh->shards[s].ix->device);
        if (L.searched[s]) (void)hipEventDestroy(L.searched[s]);
    }
    DeviceGuard g(sh->merge_device);
    if (L.ready) (void)hipEventDestroy(L.ready);
    if (L.merged) (void)hipEventDestroy(L.merged);
    L = granne_hip_sharded::Slot();
}

static void sharded_quiesce(granne_hip_sharded* sh) {
    for (auto& S : sh->shards) {
        if (!S.ix || !S.stream) continue;
        DeviceGuard g(S.ix->device);
        (void)hipStreamSynchronize(S.stream);
    }
    for (auto& D : sh->devices) {
        if (!D.xstream) continue;
        DeviceGuard g(D.id);
        (void)hipStreamSynchronize(D.xstream);
    }
    DeviceGuard g(sh->merge_device);
    if (sh->merge_stream) (void)hipStreamSynchronize(sh->merge_stream);
    if (sh->io_stream) (void)hipStreamSynchronize(sh->io_stream);
}

static void sharded_free(granne_hip_sharded* sh) {
    if (!sh) return;
    sharded_quiesce(sh);
    for (auto& L : sh->slots) sharded_free_slot(sh, L);
    RcclApi* api = rccl_api();
    for (auto& D : sh->devices) {
        DeviceGuard g(D.id);
        if (D.comm && api->CommDestroy) (void)api->CommDestroy(D.comm);
        if (D.xstream && D.xstream != sh->merge_stream) (void)hipStreamDestroy(D.xstream);
    }
    for (auto& S : sh->shards) {
        if (!S.ix) continue;
        DeviceGuard g(S.ix->device);
        if (S.stream) (void)hipStreamDestroy(S.stream);
    }
    if (sh->owns_shards)
        for (auto& S : sh->shards)
            if (S.ix) granne_hip_index_destroy(S.ix);
    {
        DeviceGuard g(sh->merge_device);
        for (auto& H : sh->host_io) {
            if (H.h_done) (void)hipEventDestroy(H.h_done);
            if (H.h_pin) (void)hipHostFree(H.h_pin);
            if (H.d_io) (void)hipFree(H.d_io);
        }
        if (sh->merge_stream) (void)hipStreamDestroy(sh->merge_stream);
        if (sh->io_stream) (void)hipStreamDestroy(sh->io_stream);
    }
    delete sh;
}

static int sharded_init_slot(granne_hip_sharded* sh, granne_hip_sharded::Slot& L) {
    const size_t nd = sh->devices.size(), G = sh->shards.size();
    L.d_recv.assign(nd, nullptr);
    L.recv_cap.assign(nd, 0);
    L.d_q.assign(nd, nullptr);
    L.q_cap.assign(nd, 0);
    L.q_there.assign(nd, nullptr);
    L.xdone.assign(nd, nullptr);
    L.searched.assign(G, nullptr);
    for (size_t d = 0; d < nd; ++d) {
        DeviceGuard g(sh->devices[d].id);
        if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", sh->devices[d].id);
        HIP_TRY(hipEventCreateWithFlags(&L.q_there[d], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.xdone[d], hipEventDisableTiming));
    }
    for (size_t s = 0; s < G; ++s) {
        DeviceGuard g(sh->shards[s].ix->device);
        HIP_TRY(hipEventCreateWithFlags(&L.searched[s], hipEventDisableTiming));
    }
    DeviceGuard g(sh->merge_device);
    HIP_TRY(hipEventCreateWithFlags(&L.ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&L.merged, hipEventDisableTiming));
    return GRANNE_HIP_OK;
}

// `groups` (or null): the exchange group of every shard. A group is what the exchange step treats as one device: its
// first shard fetches the batch's queries for all of them, its results travel to the merge device together (peer copies,
// or its part of the all-gather), its buffers are released by one event. Null: one group per HIP device, which is what a
// deployment wants. Several groups on ONE device walk every branch of the multi-device exchange on a single GPU (the
// tests do; the copies are then device-local); the all-gather needs one group per device (RCCL refuses duplicates).
extern "C" int granne_hip_sharded_create_grouped(granne_hip_sharded** out, granne_hip_index* const* shards,
                                                 const uint64_t* id_offsets, uint32_t n_shards, const uint32_t* groups) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!shards || !id_offsets) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (n_shards == 0 || n_shards > 64) return fail(GRANNE_HIP_ERR_INVALID, "n_shards must be in [1, 64]");
    for (uint32_t s = 0; s < n_shards; ++s) {
        if (!shards[s]) return fail(GRANNE_HIP_ERR_INVALID, "shard %u is null", s);
        if (shards[s]->dim != shards[0]->dim || shards[s]->dtype != shards[0]->dtype)
            return fail(GRANNE_HIP_ERR_INVALID, "shard %u has another element type than shard 0", s);
        if (groups)
            for (uint32_t t = 0; t < s; ++t)
                if (groups[t] == groups[s] && shards[t]->device != shards[s]->device)
                    return fail(GRANNE_HIP_ERR_INVALID, "shards %u and %u share exchange group %u but live on devices %d and %d", t, s,
                                groups[s], shards[t]->device, shards[s]->device);
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
            uint32_t d = 0;
            if (groups) { // the group of an earlier shard with the same label, else a new one
                uint32_t t = 0;
                while (t < s && groups[t] != groups[s]) ++t;
                d = t < s ? sh->shards[t].dev : (uint32_t)sh->devices.size();
            } else {
                while (d < sh->devices.size() && sh->devices[d].id != S.ix->device) ++d;
            }
            if (d == sh->devices.size()) {
                sh->devices.emplace_back();
                sh->devices.back().id = S.ix->device;
            }
            S.dev = d;
            S.local = sh->devices[d].n_local++;
            DeviceGuard g(S.ix->device);
            if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", S.ix->device);
            HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
            if (S.ix->device != sh->merge_device) { // peer copies in both directions (queries out, results back)
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, S.ix->device, sh->merge_device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(sh->merge_device, 0); // already enabled is fine
                DeviceGuard gm(sh->merge_device);
                can = 0;
                if (gm.ok && hipDeviceCanAccessPeer(&can, sh->merge_device, S.ix->device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(S.ix->device, 0);
            }
        }
        // one all-gather needs equal contributions in rank order: shard s on device s / n_local
        sh->uniform = true;
        for (uint32_t s = 0; s < n_shards; ++s) {
            const auto& S = sh->shards[s];
            const uint32_t per = sh->devices[0].n_local;
            if (sh->devices[S.dev].n_local != per || S.dev != s / per || S.local != s % per) sh->uniform = false;
        }
        (void)hipGetLastError(); // (a "peer access already enabled" above is not an error of ours)
        DeviceGuard g(sh->merge_device);
        HIP_TRY(hipStreamCreateWithFlags(&sh->merge_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&sh->io_stream, hipStreamNonBlocking));
        sh->devices[0].xstream = sh->merge_stream; // (device 0 of the list is shard 0's = the merge device)
        sh->slots.resize(sh->depth);
        for (auto& L : sh->slots) {
            int rc = sharded_init_slot(sh, L);
            if (rc) return rc;
        }
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

extern "C" int granne_hip_sharded_create(granne_hip_sharded** out, granne_hip_index* const* shards,
                                         const uint64_t* id_offsets, uint32_t n_shards) {
    return granne_hip_sharded_create_grouped(out, shards, id_offsets, n_shards, nullptr);
}

// SURVEY.md 8b's `index_create(..., device_ids, n_devices, partitioned)`: the whole element set in, a searchable
// partitioned index out. Shard s takes the elements [s * ceil(n / n_shards), ...) -- the split of
// src/elements/embeddings/parsing.rs:72-98 -- and is built with the GPU builder (GranneBuilder::new(config, shard).build())
// on device_ids[s / ceil(n_shards / n_devices)]: device-major, so that the all-gather exchange is available when the
// shards divide evenly. The handle owns the shard indexes.
extern "C" int granne_hip_sharded_build(granne_hip_sharded** out, const granne_hip_build_config* config, const void* elements,
                                        uint64_t n_elements, uint32_t dim, int dtype, uint32_t n_shards,
                                        const int* device_ids, uint32_t n_devices) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!config || !device_ids) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (n_shards == 0 || n_shards > 64) return fail(GRANNE_HIP_ERR_INVALID, "n_shards must be in [1, 64]");
    if (n_devices == 0 || n_devices > n_shards) return fail(GRANNE_HIP_ERR_INVALID, "n_devices must be in [1, n_shards]");
    if (dtype != GRANNE_HIP_F32 && dtype != GRANNE_HIP_I8) return fail(GRANNE_HIP_ERR_INVALID, "unknown dtype %d", dtype);
    if (dim == 0) return fail(GRANNE_HIP_ERR_INVALID, "dim must be > 0");
    if (n_elements && !elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    const uint64_t per = (n_elements + n_shards - 1) / n_shards;
    const uint32_t per_dev = (n_shards + n_devices - 1) / n_devices;
    const size_t row = (size_t)dim * elem_size(dtype);
    // one host thread per entry of device_ids builds that entry's shards one after another: the devices build at the same
    // time (eight GPUs build the eight shards of configs[4] in the time of one). Entries that name the same device share
    // it -- their builds interleave on the device, which is what the tests of this path on one GPU do.
    std::vector<granne_hip_index*> ixs(n_shards, nullptr);
    std::vector<uint64_t> offs(n_shards, 0);
    std::vector<uint32_t> groups(n_shards, 0);
    std::vector<int> rcs(n_devices, GRANNE_HIP_OK);
    std::vector<std::string> whys(n_devices);
    auto build_group = [&](uint32_t d) {
        for (uint32_t s = d * per_dev; s < std::min(n_shards, (d + 1) * per_dev) && rcs[d] == GRANNE_HIP_OK; ++s) {
            const uint64_t lo = std::min(n_elements, (uint64_t)s * per), hi = std::min(n_elements, (uint64_t)(s + 1) * per);
            granne_hip_builder* b = nullptr;
            int rc = granne_hip_builder_create(&b, config, (const uint8_t*)elements + lo * row, hi - lo, dim, dtype, device_ids[d]);
            if (rc == GRANNE_HIP_OK) rc = granne_hip_builder_build(b, GRANNE_HIP_BUILD_ALL);
            if (rc == GRANNE_HIP_OK) rc = granne_hip_builder_get_index(b, &ixs[s]);
            if (rc != GRANNE_HIP_OK) whys[d] = g_last_error; // (the message is the building thread's own)
            if (b) granne_hip_builder_destroy(b);
            offs[s] = lo;
            groups[s] = d;
            rcs[d] = rc;
        }
    };
    if (n_devices == 1) {
        build_group(0);
    } else {
        std::vector<std::thread> workers;
        for (uint32_t d = 0; d < n_devices; ++d) workers.emplace_back(build_group, d);
        for (auto& w : workers) w.join();
    }
    int rc = GRANNE_HIP_OK;
    for (uint32_t d = 0; d < n_devices && rc == GRANNE_HIP_OK; ++d)
        if (rcs[d] != GRANNE_HIP_OK) {
            rc = rcs[d];
            g_last_error = whys[d];
        }
    granne_hip_sharded* sh = nullptr;
    if (rc == GRANNE_HIP_OK) rc = granne_hip_sharded_create_grouped(&sh, ixs.data(), offs.data(), n_shards, groups.data());
    if (rc != GRANNE_HIP_OK) {
        const std::string why = g_last_error; // (destroying the shards must not lose the message)
        for (auto* ix : ixs)
            if (ix) granne_hip_index_destroy(ix);
        g_last_error = why;
        return rc;
    }
    sh->owns_shards = true;
    *out = sh;
    return GRANNE_HIP_OK;
}

// the shard indexes of a handle (borrowed: they live as long as the handle when it owns them, else as long as their owner)
extern "C" granne_hip_index* granne_hip_sharded_shard(const granne_hip_sharded* sh, uint32_t shard) {
    return (sh && shard < sh->shards.size()) ? sh->shards[shard].ix : nullptr;
}
extern "C" uint64_t granne_hip_sharded_shard_offset(const granne_hip_sharded* sh, uint32_t shard) {
    return (sh && shard < sh->shards.size()) ? sh->shards[shard].offset : 0;
}

extern "C" void granne_hip_sharded_destroy(granne_hip_sharded* sh) { sharded_free(sh); }
extern "C" uint32_t granne_hip_sharded_num_shards(const granne_hip_sharded* sh) { return sh ? (uint32_t)sh->shards.size() : 0; }
extern "C" uint64_t granne_hip_sharded_len(const granne_hip_sharded* sh) {
    uint64_t n = 0;
    if (sh)
        for (auto& S : sh->shards) n += granne_hip_index_len(S.ix);
    return n;
}
extern "C" int granne_hip_sharded_device(const granne_hip_sharded* sh) { return sh ? sh->merge_device : -1; }

extern "C" int granne_hip_sharded_set_option(granne_hip_sharded* sh, int option, uint64_t value) {
    if (!sh) return fail(GRANNE_HIP_ERR_INVALID, "sharded index is null");
    std::lock_guard<std::mutex> hk(sh->host_mu);
    std::lock_guard<std::mutex> lk(sh->mu);
    for (auto& L : sh->slots)
        if (L.busy) return fail(GRANNE_HIP_ERR_INVALID, "options cannot change while a batch is in flight");
    switch (option) {
    case GRANNE_HIP_SHARDED_OPT_DEPTH: {
        if (value < 1 || value > SHARDED_MAX_DEPTH) return fail(GRANNE_HIP_ERR_INVALID, "depth must be in [1, %u]", SHARDED_MAX_DEPTH);
        sharded_quiesce(sh);
        for (auto& L : sh->slots) sharded_free_slot(sh, L);
        sh->depth = (uint32_t)value;
        sh->slots.assign(sh->depth, granne_hip_sharded::Slot());
        sh->next_slot = 0;
        for (auto& L : sh->slots) {
            int rc = sharded_init_slot(sh, L);
            if (rc) { // a half-built slot never stays behind: back to one slot, or to none (then every begin fails cleanly)
                for (auto& X : sh->slots) sharded_free_slot(sh, X);
                sh->depth = 1;
                sh->slots.assign(1, granne_hip_sharded::Slot());
                if (sharded_init_slot(sh, sh->slots[0])) {
                    sharded_free_slot(sh, sh->slots[0]);
                    sh->slots.clear();
                    sh->depth = 0;
                }
                return rc;
            }
        }
        return GRANNE_HIP_OK;
    }
    case GRANNE_HIP_SHARDED_OPT_EXCHANGE: {
        if (value != GRANNE_HIP_SHARDED_EXCHANGE_PEER && value != GRANNE_HIP_SHARDED_EXCHANGE_RCCL)
            return fail(GRANNE_HIP_ERR_INVALID, "exchange must be 0 (peer copies) or 1 (RCCL all-gather)");
        if (value == GRANNE_HIP_SHARDED_EXCHANGE_RCCL) {
            if (!sh->uniform)
                return fail(GRANNE_HIP_ERR_INVALID, "the all-gather needs the shards in device order, the same number on every device");
            for (size_t a = 0; a < sh->devices.size(); ++a)
                for (size_t b = a + 1; b < sh->devices.size(); ++b)
                    if (sh->devices[a].id == sh->devices[b].id)
                        return fail(GRANNE_HIP_ERR_INVALID, "the all-gather needs one exchange group per device (groups %zu and %zu share device %d)",
                                    a, b, sh->devices[a].id);
            RcclApi* api = rccl_api();
            if (!api->handle || !api->why.empty())
                return fail(GRANNE_HIP_ERR_HIP, "RCCL is not available: %s", api->why.c_str());
            if (!sh->devices[0].comm) {
                const int nd = (int)sh->devices.size();
                std::vector<int> devs(nd);
                std::vector<void*> comms(nd, nullptr);
                for (int d = 0; d < nd; ++d) devs[d] = sh->devices[d].id;
                RCCL_TRY(api, api->CommInitAll(comms.data(), nd, devs.data()));
                for (int d = 0; d < nd; ++d) {
                    sh->devices[d].comm = comms[d];
                    if (!sh->devices[d].xstream) {
                        DeviceGuard g(sh->devices[d].id);
                        HIP_TRY(hipStreamCreateWithFlags(&sh->devices[d].xstream, hipStreamNonBlocking));
                    }
                }
            }
        }
        sharded_quiesce(sh);
        sh->exchange = (int)value;
        return GRANNE_HIP_OK;
    }
    default:
        return fail(GRANNE_HIP_ERR_INVALID, "unknown option %d", option);
    }
}

extern "C" int granne_hip_sharded_get_option(const granne_hip_sharded* sh, int option, uint64_t* value) {
    if (!sh || !value) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    switch (option) {
    case GRANNE_HIP_SHARDED_OPT_DEPTH: *value = sh->depth; return GRANNE_HIP_OK;
    case GRANNE_HIP_SHARDED_OPT_EXCHANGE: *value = (uint64_t)sh->exchange; return GRANNE_HIP_OK;
    default: return fail(GRANNE_HIP_ERR_INVALID, "unknown option %d", option);
    }
}

// a buffer of a slot grows: nothing of the slot's previous batch may still be running
static int slot_grow(uint8_t** p, size_t* cap, size_t want) {
    if (*cap >= want) return GRANNE_HIP_OK;
    if (*p) (void)hipFree(*p); // (hipFree waits for the device)
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)p, want + (want >> 2)));
    *cap = want + (want >> 2);
    return GRANNE_HIP_OK;
}

// ---- one batch: begin (enqueue everything) / end (order the caller's stream after the merge) -------------------
// The caller holds sh->mu.
static int sharded_begin_locked(granne_hip_sharded* sh, const void* d_queries, uint32_t nq, uint32_t max_search,
                                uint32_t num_neighbors, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                                uint32_t* d_status, hipStream_t stream, uint64_t* out_ticket) {
    using Slot = granne_hip_sharded::Slot;
    const uint32_t G = (uint32_t)sh->shards.size();
    const uint32_t nd = (uint32_t)sh->devices.size();
    // any slot that is free, looked for from the one after the last begun (tickets may be ended in any order)
    uint32_t si = sh->depth;
    for (uint32_t t = 0; t < sh->depth && t < sh->slots.size(); ++t) {
        const uint32_t c = (sh->next_slot + t) % sh->depth;
        if (!sh->slots[c].busy) {
            si = c;
            break;
        }
    }
    if (si >= sh->slots.size())
        return fail(GRANNE_HIP_ERR_INVALID, "%u batches are in flight already (GRANNE_HIP_SHARDED_OPT_DEPTH): end one first", sh->depth);
    Slot& L = sh->slots[si];
    const size_t pb = (size_t)granne_hip_packed_topk_bytes(nq, num_neighbors);
    const size_t stride = pb + SHARD_STATUS_BYTES;
    const size_t qb = (size_t)nq * sh->dim * elem_size(sh->dtype);
    const bool rccl = sh->exchange == GRANNE_HIP_SHARDED_EXCHANGE_RCCL;

    // buffers (sized once per (nq, k); a change waits for the slot's previous batch: hipFree synchronises)
    for (uint32_t d = 0; d < nd; ++d) {
        DeviceGuard g(sh->devices[d].id);
        if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", sh->devices[d].id);
        // the merge device holds every shard's block; another device its own shards' (peer) or everybody's (all-gather)
        const size_t blocks = (d == 0 || rccl) ? G : sh->devices[d].n_local;
        int rc = slot_grow(&L.d_recv[d], &L.recv_cap[d], stride * blocks);
        if (rc == 0 && d != 0) rc = slot_grow(&L.d_q[d], &L.q_cap[d], qb);
        if (rc) return rc;
    }
    L.nq = nq;
    L.k = num_neighbors;
    L.stride = stride;

    DeviceGuard gm(sh->merge_device);
    HIP_TRY(hipEventRecord(L.ready, stream));
    // where shard s writes: its place in the gather buffer of ITS device (the all-gather is in place: a device's
    // contribution is the slice of its own receive buffer that the collective would put there anyway)
    auto block_of = [&](uint32_t s) -> uint8_t* {
        const auto& S = sh->shards[s];
        if (S.dev == 0 || rccl) return L.d_recv[S.dev] + stride * s;
        return L.d_recv[S.dev] + stride * S.local;
    };
    std::vector<char> q_sent(nd, 0);
    for (uint32_t s = 0; s < G; ++s) {
        auto& S = sh->shards[s];
        DeviceGuard g(S.ix->device);
        if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", S.ix->device);
        HIP_TRY(hipStreamWaitEvent(S.stream, L.ready, 0));
        if (L.used) { // the slot's previous batch has left these buffers
            HIP_TRY(hipStreamWaitEvent(S.stream, L.merged, 0));
            HIP_TRY(hipStreamWaitEvent(S.stream, L.xdone[S.dev], 0));
        }
        const void* q_here = d_queries;
        if (S.dev != 0) {
            if (!q_sent[S.dev]) { // the first shard of a device fetches the queries for all of them
                HIP_TRY(hipMemcpyPeerAsync(L.d_q[S.dev], S.ix->device, d_queries, sh->merge_device, qb, S.stream));
                HIP_TRY(hipEventRecord(L.q_there[S.dev], S.stream));
                q_sent[S.dev] = 1;
            } else {
                HIP_TRY(hipStreamWaitEvent(S.stream, L.q_there[S.dev], 0));
            }
            q_here = L.d_q[S.dev];
        }
        uint8_t* blk = block_of(s);
        HIP_TRY(hipMemsetAsync(blk + pb, 0, SHARD_STATUS_BYTES, S.stream));
        int rc = granne_hip_search_batch_packed_device(S.ix, q_here, nq, max_search, num_neighbors, blk, (uint32_t*)(blk + pb), S.stream);
        if (rc) return rc;
        if (!rccl && S.dev != 0)
            HIP_TRY(hipMemcpyPeerAsync(L.d_recv[0] + stride * s, sh->merge_device, blk, S.ix->device, stride, S.stream));
        HIP_TRY(hipEventRecord(L.searched[s], S.stream));
    }
    if (rccl) {
        RcclApi* api = rccl_api();
        for (uint32_t s = 0; s < G; ++s) {
            const auto& S = sh->shards[s];
            DeviceGuard g(S.ix->device);
            HIP_TRY(hipStreamWaitEvent(sh->devices[S.dev].xstream, L.searched[s], 0));
        }
        RCCL_TRY(api, api->GroupStart());
        int bad = 0;
        for (uint32_t d = 0; d < nd; ++d) {
            DeviceGuard g(sh->devices[d].id);
            const size_t part = stride * sh->devices[d].n_local;
            const int r = api->AllGather(L.d_recv[d] + part * d, L.d_recv[d], part, RCCL_UINT8, sh->devices[d].comm, sh->devices[d].xstream);
            if (r && !bad) bad = r;
        }
        const int ge = api->GroupEnd();
        if (bad || ge) return fail(GRANNE_HIP_ERR_HIP, "ncclAllGather failed: %s", api->GetErrorString(bad ? bad : ge));
        for (uint32_t d = 1; d < nd; ++d) {
            DeviceGuard g(sh->devices[d].id);
            HIP_TRY(hipEventRecord(L.xdone[d], sh->devices[d].xstream));
        }
    } else {
        for (uint32_t s = 0; s < G; ++s) HIP_TRY(hipStreamWaitEvent(sh->merge_stream, L.searched[s], 0));
    }
    uint64_t offs[64];
    for (uint32_t s = 0; s < G; ++s) offs[s] = sh->shards[s].offset;
    int rc = granne_hip_merge_topk_packed_strided_device(L.d_recv[0], stride, offs, G, nq, num_neighbors, d_out_ids, d_out_dists,
                                                         d_out_counts, sh->merge_device, sh->merge_stream);
    if (rc) return rc;
    if (d_status) {
        hipLaunchKernelGGL(fold_status_kernel, dim3(1), dim3(64), 0, sh->merge_stream, (const uint8_t*)L.d_recv[0], (uint64_t)stride,
                           (uint64_t)pb, G, d_status);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(L.merged, sh->merge_stream));
    HIP_TRY(hipEventRecord(L.xdone[0], sh->merge_stream));
    if (!rccl)
        for (uint32_t d = 1; d < nd; ++d) { // peer mode: a device's buffers are free once its shards' copies are out
            DeviceGuard g(sh->devices[d].id);
            for (uint32_t s = 0; s < G; ++s)
                if (sh->shards[s].dev == d && sh->shards[s].local + 1 == sh->devices[d].n_local)
                    HIP_TRY(hipEventRecord(L.xdone[d], sh->shards[s].stream));
        }
    L.used = true;
    L.busy = true;
    L.seq = sh->next_seq++;
    sh->next_slot = (si + 1) % sh->depth;
    *out_ticket = (L.seq << 8) | si;
    return GRANNE_HIP_OK;
}

static int sharded_check_args(granne_hip_sharded* sh, uint32_t max_search, uint32_t num_neighbors) {
    if (!sh) return fail(GRANNE_HIP_ERR_INVALID, "sharded index is null");
    if (max_search == 0) return fail(GRANNE_HIP_ERR_INVALID, "max_search must be > 0 (the reference panics, src/index/mod.rs:1019)");
    if ((uint64_t)sh->shards.size() * num_neighbors > 4096) return fail(GRANNE_HIP_ERR_INVALID, "n_shards * num_neighbors must be <= 4096");
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_sharded_begin_device(granne_hip_sharded* sh, const void* d_queries, uint32_t nq, uint32_t max_search,
                                               uint32_t num_neighbors, uint64_t* d_out_ids, float* d_out_dists,
                                               uint32_t* d_out_counts, uint32_t* d_status, void* stream, uint64_t* out_ticket) {
    int rc = sharded_check_args(sh, max_search, num_neighbors);
    if (rc) return rc;
    if (!out_ticket) return fail(GRANNE_HIP_ERR_INVALID, "out_ticket is null");
    if (nq == 0 || num_neighbors == 0) return fail(GRANNE_HIP_ERR_INVALID, "nq and num_neighbors must be > 0 for a batch in flight");
    if (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    std::lock_guard<std::mutex> lk(sh->mu);
    rc = sharded_begin_locked(sh, d_queries, nq, max_search, num_neighbors, d_out_ids, d_out_dists, d_out_counts, d_status,
                              (hipStream_t)stream, out_ticket);
    if (rc) sharded_quiesce(sh); // an error return leaves nothing running on the caller's buffers
    return rc;
}

extern "C" int granne_hip_sharded_end_device(granne_hip_sharded* sh, uint64_t ticket, void* stream) {
    if (!sh) return fail(GRANNE_HIP_ERR_INVALID, "sharded index is null");
    std::lock_guard<std::mutex> lk(sh->mu);
    const uint32_t si = (uint32_t)(ticket & 0xFF);
    if (si >= sh->slots.size() || !sh->slots[si].busy || sh->slots[si].seq != (ticket >> 8))
        return fail(GRANNE_HIP_ERR_INVALID, "no batch in flight has this ticket");
    auto& L = sh->slots[si];
    DeviceGuard g(sh->merge_device);
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, L.merged, 0));
    L.busy = false;
    return GRANNE_HIP_OK;
}

// Granne::search on every shard + the merge, device buffers (on the merge device = shard 0's), stream-ordered.
extern "C" int granne_hip_sharded_search_batch_device(granne_hip_sharded* sh, const void* d_queries, uint32_t nq,
                                                      uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                                      float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_status,
                                                      void* stream) {
    int rc = sharded_check_args(sh, max_search, num_neighbors);
    if (rc) return rc;
    if (nq == 0) return GRANNE_HIP_OK;
    if (num_neighbors == 0) { // .take(0), src/index/mod.rs:974-977
        if (!d_out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
        DeviceGuard g(sh->merge_device);
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, (size_t)nq * 4, (hipStream_t)stream));
        return GRANNE_HIP_OK;
    }
    uint64_t ticket = 0;
    rc = granne_hip_sharded_begin_device(sh, d_queries, nq, max_search, num_neighbors, d_out_ids, d_out_dists, d_out_counts,
                                         d_status, stream, &ticket);
    if (rc) return rc;
    return granne_hip_sharded_end_device(sh, ticket, stream);
}

// ---- host pointers: pinned staging, `depth` batches in flight --------------------------------------------------
// queries: [n_batches][nq][dim] (dense, prepared like the elements); outputs [n_batches][nq][num_neighbors] / [n_batches][nq].
extern "C" int granne_hip_sharded_search_batches(granne_hip_sharded* sh, const void* queries, uint32_t n_batches, uint32_t nq,
                                                 uint32_t max_search, uint32_t num_neighbors, uint64_t* out_ids,
                                                 float* out_dists, uint32_t* out_counts) {
    int rc = sharded_check_args(sh, max_search, num_neighbors);
    if (rc) return rc;
    if (nq == 0 || n_batches == 0) return GRANNE_HIP_OK;
    if (num_neighbors == 0) {
        if (!out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
        memset(out_counts, 0, (size_t)n_batches * nq * 4);
        return GRANNE_HIP_OK;
    }
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    std::lock_guard<std::mutex> hk(sh->host_mu);
    const size_t k = num_neighbors;
    const size_t qb = (size_t)nq * sh->dim * elem_size(sh->dtype);
    const size_t o_ids = (qb + 255) & ~(size_t)255;
    const size_t o_d = o_ids + (size_t)nq * k * 8;
    const size_t o_c = o_d + (size_t)nq * k * 4;
    const size_t o_st = (o_c + (size_t)nq * 4 + 15) & ~(size_t)15;
    const size_t total = o_st + 16;
    struct Pending {
        uint32_t batch;
        uint32_t io; // which of the handle's host_io sets carries the batch
        uint64_t ticket;
    };
    std::vector<Pending> pending;
    struct Quiesce { // whatever way this call ends, nothing it enqueued is still running when it returns
        granne_hip_sharded* sh;
        std::vector<Pending>* mine;
        bool armed = true;
        ~Quiesce() {
            if (!armed) return;
            sharded_quiesce(sh);
            // the places of THIS call's batches become free; batches other threads have begun through the device-pointer
            // entries keep theirs (their end_device finds the ticket it was given)
            std::lock_guard<std::mutex> lk(sh->mu);
            for (const auto& P : *mine) {
                const uint32_t si = (uint32_t)(P.ticket & 0xFF);
                if (si < sh->slots.size() && sh->slots[si].busy && sh->slots[si].seq == (P.ticket >> 8)) sh->slots[si].busy = false;
            }
        }
    } quiesce{sh, &pending};
    DeviceGuard g(sh->merge_device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", sh->merge_device);
    auto finish = [&](const Pending& P) -> int {
        auto& H = sh->host_io[P.io];
        rc = granne_hip_sharded_end_device(sh, P.ticket, sh->io_stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(H.h_pin + o_ids, H.d_io + o_ids, total - o_ids, hipMemcpyDeviceToHost, sh->io_stream));
        HIP_TRY(hipEventRecord(H.h_done, sh->io_stream));
        HIP_TRY(hipEventSynchronize(H.h_done));
        const uint32_t* st = (const uint32_t*)(H.h_pin + o_st);
        if (st[0]) return fail(GRANNE_HIP_ERR_OVERFLOW, "a shard's exact-search scratch is exhausted (raise GRANNE_HIP_OPT_SLOW_SLOTS)");
        memcpy(out_ids + (size_t)P.batch * nq * k, H.h_pin + o_ids, (size_t)nq * k * 8);
        memcpy(out_dists + (size_t)P.batch * nq * k, H.h_pin + o_d, (size_t)nq * k * 4);
        memcpy(out_counts + (size_t)P.batch * nq, H.h_pin + o_c, (size_t)nq * 4);
        return GRANNE_HIP_OK;
    };
    const uint32_t depth = sh->depth;
    for (uint32_t b = 0; b < n_batches; ++b) {
        if (pending.size() >= depth) { // the oldest batch comes home first: its buffers (and its slot) are the next ones
            rc = finish(pending.front());
            if (rc) return rc;
            pending.erase(pending.begin());
        }
        const uint32_t io = b % depth;
        auto& H = sh->host_io[io];
        if (!H.h_done) HIP_TRY(hipEventCreateWithFlags(&H.h_done, hipEventDisableTiming));
        if (H.h_cap < total) {
            if (H.h_pin) (void)hipHostFree(H.h_pin);
            H.h_pin = nullptr;
            H.h_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&H.h_pin, total + (total >> 2), hipHostMallocDefault));
            H.h_cap = total + (total >> 2);
        }
        rc = slot_grow(&H.d_io, &H.io_cap, total);
        if (rc) return rc;
        memcpy(H.h_pin, (const uint8_t*)queries + (size_t)b * qb, qb);
        HIP_TRY(hipMemcpyAsync(H.d_io, H.h_pin, qb, hipMemcpyHostToDevice, sh->io_stream));
        HIP_TRY(hipMemsetAsync(H.d_io + o_st, 0, 16, sh->io_stream));
        Pending P{b, io, 0};
        for (;;) {
            rc = granne_hip_sharded_begin_device(sh, H.d_io, nq, max_search, num_neighbors, (uint64_t*)(H.d_io + o_ids),
                                                 (float*)(H.d_io + o_d), (uint32_t*)(H.d_io + o_c), (uint32_t*)(H.d_io + o_st),
                                                 sh->io_stream, &P.ticket);
            // every place taken although this call holds fewer than `depth`: other threads have batches in flight through the
            // device-pointer entries. This call's oldest batch comes home and its place is taken; with none of its own in
            // flight the error stands (the caller shares the handle's depth with those threads).
            if (rc != GRANNE_HIP_ERR_INVALID || pending.empty()) break;
            const int rf = finish(pending.front());
            if (rf) return rf;
            pending.erase(pending.begin());
        }
        if (rc) return rc;
        pending.push_back(P);
    }
    for (const auto& P : pending) {
        rc = finish(P);
        if (rc) return rc;
    }
    quiesce.armed = false; // everything has been waited for
    return GRANNE_HIP_OK;
}

// queries: host, dense [nq][dim], prepared like the elements. Outputs: host, global ids.
extern "C" int granne_hip_sharded_search_batch(granne_hip_sharded* sh, const void* queries, uint32_t nq, uint32_t max_search,
                                               uint32_t num_neighbors, uint64_t* out_ids, float* out_dists,
                                               uint32_t* out_counts) {
    return granne_hip_sharded_search_batches(sh, queries, 1, nq, max_search, num_neighbors, out_ids, out_dists, out_counts);
}

// Granne::search on a partitioned index: one query
extern "C" int granne_hip_sharded_search(granne_hip_sharded* sh, const void* query, uint32_t max_search,
                                         uint32_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_count) {
    if (!out_count) return fail(GRANNE_HIP_ERR_INVALID, "out_count is null");
    return granne_hip_sharded_search_batch(sh, query, 1, max_search, num_neighbors, out_ids, out_dists, out_count);
}
