// granne_hip.hip -- host runtime + C ABI of libgranne_hip.so (see include/granne_hip.h).
// Compiled by hipcc for gfx950 only; there is no CPU code path for search in this library.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/granne_hip.h"
#include "search_kernel.h"
#include "walk_fast.h"
#include "slow_kernel.h"
#include "util_kernels.h"
#include "brute_force.h"

using namespace granne_hip;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? GRANNE_HIP_ERR_NO_DEVICE \
                                                                            : GRANNE_HIP_ERR_HIP,  \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

extern "C" const char* granne_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int granne_hip_abi_version(void) { return GRANNE_HIP_ABI_VERSION; }
extern "C" int granne_hip_device_count(int* out_count) {
    if (!out_count) return fail(GRANNE_HIP_ERR_INVALID, "out_count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return fail(GRANNE_HIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out_count = n;
    return GRANNE_HIP_OK;
}

// experiment knobs, read once per process
struct EnvKnobs {
    int visited_cap = 0, front_eighths = 0, maxc = 0, lds_pad = 0, visited = 0, tail_blocks = -1, touch_max = -1, inline_tails = 1, seen_min = -1, bf_b16 = 1, bf_ring = 1;
    EnvKnobs() {
        auto geti = [](const char* name, int dflt) {
            const char* e = getenv(name);
            return e ? atoi(e) : dflt;
        };
        visited_cap = geti("GRANNE_HIP_VISITED_CAP", 0);
        front_eighths = geti("GRANNE_HIP_FRONT_EIGHTHS", 0);
        maxc = geti("GRANNE_HIP_MAXC", 0);
        lds_pad = geti("GRANNE_HIP_LDS_PAD", 0);
        visited = geti("GRANNE_HIP_VISITED", 0); // GRANNE_HIP_OPT_VISITED16's values, where the option says auto
        inline_tails = geti("GRANNE_HIP_INLINE_TAILS", 1); // 0: no index keeps LayerDev::adjx (experiments: the layout before round 6)
        bf_ring = geti("GRANNE_HIP_BF_RING", 1); // 0: the exact scan of 128-byte int8 rows stages its tiles through registers (round 5's bf_i8_kernel)
        bf_b16 = geti("GRANNE_HIP_BF_B16", 1); // 0: the exact scan of f32 rows scores on the f32 matrix path (round 5's: 5 x slower, scores to the last bits)
        seen_min = geti("GRANNE_HIP_SEEN_MIN", -1); // launches of at least this many walks skip revisits before their rows are fetched (-1: default)
        touch_max = geti("GRANNE_HIP_TOUCH_MAX", -1); // launches of up to this many queries touch rows ahead (-1: default)
    }
};
static const EnvKnobs& knobs() {
    static const EnvKnobs k;
    return k;
}

// ------------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------------
// search_launch's scratch blocks, one per stream that has searched (see scratch_for)
struct ScratchCache {
    struct Block {
        hipStream_t stream;
        uint8_t* p;
        size_t cap;
        uint64_t last_use;   // `uses` at the block's last launch
        uint32_t small_runs; // consecutive launches that needed less than a quarter of an oversized block
        std::chrono::steady_clock::time_point last_time; // wall clock of the block's last launch
    };
    std::mutex mu;
    std::vector<Block> blocks;
    uint64_t uses = 0;
    // blocks of streams that gave their place up: freed by search_launch AFTER it has released `mu` (hipFree waits for
    // the device; every concurrent searcher of the index would wait with it under the lock)
    std::vector<uint8_t*> graveyard;
    void free_all() {
        for (auto& b : blocks)
            if (b.p) (void)hipFree(b.p);
        blocks.clear();
        for (auto* p : graveyard) (void)hipFree(p);
        graveyard.clear();
    }
};

struct LayerHost {
    uint64_t len = 0;
    uint32_t width = 0;     // caller's row width
    uint32_t dev_width = 0; // multiple of 32
    uint32_t* d_adj = nullptr;
    uint8_t* d_adjx = nullptr; // the register walker's copy: ids + the neighbors' row tails (LayerDev::adjx), or null
    uint32_t adjx_stride = 0;
};

struct granne_hip_index {
    int device = 0;
    uint32_t dim = 0;
    int dtype = 0;
    uint64_t n_elements = 0;
    uint32_t row_bytes = 0;  // data bytes of a device row (zero padded to 16)
    uint32_t row_stride = 0; // bytes from one device row to the next
    uint8_t* d_elements = nullptr;
    std::vector<LayerHost> layers;
    LayerDev* d_layers = nullptr;
    uint64_t hbm_bytes = 0;
    uint32_t max_dev_width = 32;
    // options
    uint64_t opt_visited_slots = 0;
    uint64_t opt_force_slow = 0;
    uint64_t opt_slow_slots = 1u << 18;
    uint64_t opt_slow_blocks = 16;
    uint64_t opt_overflow_slots = 0; // 0 auto, 1 off, else slots per overflow table
    uint64_t opt_visited16 = 0;      // 0 auto, 1 off, 2 always the 20-bit entries (tests)
    uint64_t opt_visited16_lg = 0;   // 0 auto, else log2(buckets)
    uint64_t opt_inline_tails = 1;   // GRANNE_HIP_OPT_INLINE_TAILS: keep LayerDev::adjx for the shapes that have one
    uint64_t opt_seen_min = 2048;    // GRANNE_HIP_OPT_SEEN_MIN: launches of at least this many walks skip revisits before their rows are fetched
    std::atomic<uint64_t> last_slow_count{0};
    std::atomic<uint64_t> last_walker{0}; // GRANNE_HIP_OPT_LAST_WALKER
    // the exact scan of int8 rows (brute_force.h): 1 / |x| per row, made at the first scan
    std::mutex norm_mu;
    float* d_inv_norm = nullptr;
    // host-pointer searches (granne_hip_search / _search_batch): a stream, a device buffer and a pinned
    // staging buffer per concurrent caller, kept for the life of the index -- the reference's API is one
    // query per call (src/index/mod.rs:140-150), so a call must not pay stream creation and hipMalloc
    struct HostCall {
        hipStream_t stream = nullptr;
        uint8_t* d_buf = nullptr;
        size_t d_cap = 0;
        uint8_t* h_pin = nullptr;
        size_t h_cap = 0;
    };
    std::mutex call_mu;
    std::vector<HostCall*> call_free;
    ScratchCache scratch;
    // granne_hip_search_begin_device / _end_device: batches in flight on streams of the index's own
    struct Flight {
        hipStream_t stream = nullptr;
        hipEvent_t ready = nullptr, done = nullptr;
        bool busy = false;
        uint64_t seq = 0;
    };
    std::mutex flight_mu;
    Flight flights[GRANNE_HIP_SEARCH_DEPTH_MAX];
    uint32_t depth = GRANNE_HIP_SEARCH_DEPTH; // GRANNE_HIP_OPT_SEARCH_DEPTH
    uint32_t next_flight = 0;
    uint64_t next_seq = 1;
};

static inline uint32_t elem_size(int dtype) { return dtype == GRANNE_HIP_F32 ? 4u : 1u; }

static uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// device row stride: f32 rows are padded to 16 bytes; i8 rows to a power of two (<= 1024) or a
// multiple of 1024 so that a row is split over a power-of-two number of lanes and never
// straddles more 128-byte lines than it has to (SURVEY 7 "unaligned rows").
static uint32_t device_row_bytes(uint32_t dim, int dtype) {
    if (dtype == GRANNE_HIP_F32) return (dim * 4u + 15u) & ~15u;
    if (dim <= 1024) return next_pow2(dim < 16 ? 16 : dim);
    return (dim + 1023u) & ~1023u;
}

// Bytes from one device row to the next. f32 rows of 256 bytes and more start on a 128-byte line: the register walker
// reads a row's 32-float chunks as whole lines, and when the layers carry the neighbors' tails next to their ids
// (LayerDev::adjx) that is all it reads of a row -- a 100-d row is three lines, not the four a 400-byte stride touches.
static uint32_t device_row_stride(uint32_t dim, int dtype) {
    const uint32_t rb = device_row_bytes(dim, dtype);
    if (dtype == GRANNE_HIP_F32 && rb >= 256u) return (rb + 127u) & ~127u;
    return rb;
}
// 16-byte units of a row's tail (the dim % 32 last floats) that LayerDev::adjx carries per neighbor; 0: the shape has no
// such copy. The unrolled f32 walkers (dims 100 and 200) read it; their layers are 32 ids wide.
static uint32_t inline_tail_units(uint32_t dim, int dtype) {
    if (dtype != GRANNE_HIP_F32 || (dim != 100u && dim != 200u)) return 0u;
    return (dim % 32u) / 4u;
}
// the scan's per-row norms: inv_norm [n_pad], then inv_gmax [n_pad / 32][2] (brute_force.h)
static inline uint64_t inv_norm_bytes(uint64_t n) {
    const uint64_t n_pad = (n + 31u) & ~31ull;
    return (n_pad + n_pad / 16 + 1 + 16) * 4; // (+16: bf_i8_ring_kernel reads a tile's eight inv_gmax entries in one scalar load, past the last block's too)
}

static int grid_for(uint64_t work, int block) {
    uint64_t g = (work + block - 1) / block;
    if (g > 256ull * 32) g = 256ull * 32;
    if (g < 1) g = 1;
    return (int)g;
}

static int validate_common(granne_hip_index** out, uint64_t n_elements, uint32_t dim, int dtype, uint32_t n_layers,
                           const uint64_t* layer_len) {
    if (!out) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (dtype != GRANNE_HIP_F32 && dtype != GRANNE_HIP_I8) return fail(GRANNE_HIP_ERR_INVALID, "unknown dtype %d", dtype);
    if (dim == 0) return fail(GRANNE_HIP_ERR_INVALID, "dim must be > 0");
    if (n_elements >= 0xFFFFFFFFull)
        return fail(GRANNE_HIP_ERR_INVALID, "too many elements (reference limit, src/index/mod.rs:420)");
    if (n_layers && !layer_len) return fail(GRANNE_HIP_ERR_INVALID, "layer_len is null");
    for (uint32_t l = 0; l < n_layers; ++l) {
        if (layer_len[l] > n_elements)
            return fail(GRANNE_HIP_ERR_INVALID, "layer %u has %llu nodes but only %llu elements", l,
                        (unsigned long long)layer_len[l], (unsigned long long)n_elements);
        if (l > 0 && layer_len[l] < layer_len[l - 1])
            return fail(GRANNE_HIP_ERR_INVALID, "layers must be prefix-nested (layer %u shrinks)", l);
        if (layer_len[l] == 0) return fail(GRANNE_HIP_ERR_INVALID, "layer %u is empty", l);
    }
    return GRANNE_HIP_OK;
}

static void destroy_index(granne_hip_index* ix) {
    if (!ix) return;
    DeviceGuard g(ix->device);
    if (ix->d_elements) (void)hipFree(ix->d_elements);
    for (auto& L : ix->layers) {
        if (L.d_adj) (void)hipFree(L.d_adj);
        if (L.d_adjx) (void)hipFree(L.d_adjx);
    }
    if (ix->d_layers) (void)hipFree(ix->d_layers);
    if (ix->d_inv_norm) (void)hipFree(ix->d_inv_norm);
    for (auto* c : ix->call_free) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->d_buf) (void)hipFree(c->d_buf);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        delete c;
    }
    for (auto& f : ix->flights) {
        if (f.stream) (void)hipStreamSynchronize(f.stream);
        if (f.ready) (void)hipEventDestroy(f.ready);
        if (f.done) (void)hipEventDestroy(f.done);
        if (f.stream) (void)hipStreamDestroy(f.stream);
    }
    ix->scratch.free_all();
    delete ix;
}

// src is a device pointer to dense rows; fills ix->d_elements
static int upload_elements_from_device(granne_hip_index* ix, const void* d_src, hipStream_t s) {
    uint64_t n = ix->n_elements;
    uint32_t dense = ix->dim * elem_size(ix->dtype);
    size_t bytes = (size_t)n * ix->row_stride;
    HIP_TRY(hipMalloc((void**)&ix->d_elements, bytes ? bytes : 16));
    ix->hbm_bytes += bytes;
    if (n == 0) return GRANNE_HIP_OK;
    if (dense == ix->row_stride) {
        HIP_TRY(hipMemcpyAsync(ix->d_elements, d_src, bytes, hipMemcpyDeviceToDevice, s));
    } else {
        uint64_t units = n * (ix->row_stride / 16);
        hipLaunchKernelGGL(relayout_rows_kernel, dim3(grid_for(units, 256)), dim3(256), 0, s, (const uint8_t*)d_src,
                           ix->d_elements, n, dense, ix->row_stride);
        HIP_TRY(hipGetLastError());
    }
    return GRANNE_HIP_OK;
}

// (Re)makes the register walker's copies of the layers (LayerDev::adjx) -- or drops them when the option is off or the
// shape has none. The layers and the elements are final when this runs (finish_layers).
static int make_inline_tails(granne_hip_index* ix, hipStream_t s) {
    for (auto& L : ix->layers) {
        if (L.d_adjx) {
            (void)hipFree(L.d_adjx);
            ix->hbm_bytes -= (uint64_t)L.len * L.adjx_stride;
            L.d_adjx = nullptr;
            L.adjx_stride = 0;
        }
    }
    const uint32_t tu = inline_tail_units(ix->dim, ix->dtype);
    if (!tu || !ix->opt_inline_tails || !knobs().inline_tails) return GRANNE_HIP_OK;
    for (auto& L : ix->layers)
        if (L.dev_width != 32u) return GRANNE_HIP_OK; // layers of up to 64 ids (WIDE) read their tails from the rows
    for (auto& L : ix->layers) {
        L.adjx_stride = 128u + 32u * tu * 16u;
        const size_t bytes = (size_t)L.len * L.adjx_stride;
        if (hipMalloc((void**)&L.d_adjx, bytes ? bytes : 16) != hipSuccess) {
            // The copy is an accelerator, not part of the index: an index that fits without it (a 200M x 100-d shard: 128 GB
            // of copy) is made without it -- the walkers then read the tails from the rows' own last line.
            (void)hipGetLastError();
            L.d_adjx = nullptr;
            for (auto& M : ix->layers) {
                if (M.d_adjx) {
                    (void)hipStreamSynchronize(s);
                    (void)hipFree(M.d_adjx);
                    ix->hbm_bytes -= (uint64_t)M.len * M.adjx_stride;
                    M.d_adjx = nullptr;
                }
                M.adjx_stride = 0;
            }
            return GRANNE_HIP_OK;
        }
        ix->hbm_bytes += bytes;
        if (L.len)
            hipLaunchKernelGGL(inline_tails_kernel, dim3(grid_for(L.len * 32u * (1u + tu), 256)), dim3(256), 0, s, L.d_adj,
                               L.len, ix->d_elements, ix->row_stride, (ix->dim / 32u) * 128u, tu, L.d_adjx, L.adjx_stride);
        HIP_TRY(hipGetLastError());
    }
    return GRANNE_HIP_OK;
}

// The exact scan of int8 rows (brute_force.h) reads 1 / |x| per row and the largest of them per half block of 32: made
// here, when the index is made (and again after reorder), on the creating stream -- a scan never allocates, never
// synchronises its caller's stream and can be captured (round 5 made them lazily inside the first scan, under norm_mu).
static int make_scan_norms(granne_hip_index* ix, hipStream_t s) {
    if (ix->dtype != GRANNE_HIP_I8 || ix->n_elements == 0) return GRANNE_HIP_OK;
    std::lock_guard<std::mutex> lk(ix->norm_mu);
    if (ix->d_inv_norm) return GRANNE_HIP_OK;
    const uint64_t n = ix->n_elements, n_pad = (n + 31u) & ~31ull;
    float* dn = nullptr;
    HIP_TRY(hipMalloc((void**)&dn, (size_t)inv_norm_bytes(n)));
    hipLaunchKernelGGL(inv_norm_rows_kernel, dim3(grid_for(n * 8, 256)), dim3(256), 0, s, ix->d_elements, n, ix->row_stride, dn);
    hipLaunchKernelGGL(inv_gmax_kernel, dim3(grid_for(n_pad / 16 + 1, 256)), dim3(256), 0, s, (const float*)dn, n, dn + n_pad);
    if (hipGetLastError() != hipSuccess) {
        (void)hipFree(dn);
        return fail(GRANNE_HIP_ERR_HIP, "inv_norm_rows_kernel failed");
    }
    ix->d_inv_norm = dn; // (finish_layers synchronises s before the index is handed out)
    ix->hbm_bytes += inv_norm_bytes(n);
    return GRANNE_HIP_OK;
}

static int finish_layers(granne_hip_index* ix, hipStream_t s) {
    std::vector<LayerDev> h(ix->layers.size());
    ix->max_dev_width = 32;
    {
        int r = make_inline_tails(ix, s);
        if (r == 0) r = make_scan_norms(ix, s);
        if (r) return r;
    }
    // rows that name a neighbor twice (LAYER_TWIN_ROWS, walk_fast.h): looked for once, here, on the device rows
    uint32_t* d_found = nullptr;
    std::vector<uint32_t> found(h.size(), 0u);
    if (!h.empty()) {
        HIP_TRY(hipMalloc((void**)&d_found, h.size() * 4));
        if (hipMemsetAsync(d_found, 0, h.size() * 4, s) != hipSuccess) {
            (void)hipFree(d_found);
            return fail(GRANNE_HIP_ERR_HIP, "hipMemsetAsync failed");
        }
    }
    for (size_t l = 0; l < ix->layers.size(); ++l) {
        h[l].adj = ix->layers[l].d_adj;
        h[l].len = ix->layers[l].len;
        h[l].width = ix->layers[l].dev_width;
        h[l].flags = 0;
        h[l].adjx = ix->layers[l].d_adjx;
        h[l].adjx_stride = ix->layers[l].adjx_stride;
        h[l].reserved = 0;
        if (ix->layers[l].dev_width > ix->max_dev_width) ix->max_dev_width = ix->layers[l].dev_width;
        if (h[l].len && h[l].width <= 64)
            hipLaunchKernelGGL(twin_rows_kernel, dim3(grid_for(h[l].len * 64, 256)), dim3(256), 0, s, h[l].adj, h[l].len,
                               h[l].width, d_found + l);
    }
    if (!h.empty()) {
        hipError_t e = hipMemcpyAsync(found.data(), d_found, h.size() * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d_found);
        if (e != hipSuccess) return fail(GRANNE_HIP_ERR_HIP, "twin_rows_kernel: %s", hipGetErrorString(e));
        for (size_t l = 0; l < h.size(); ++l)
            if (found[l]) h[l].flags |= LAYER_TWIN_ROWS;
    }
    size_t bytes = sizeof(LayerDev) * (h.size() ? h.size() : 1);
    if (ix->d_layers) (void)hipFree(ix->d_layers);
    ix->d_layers = nullptr;
    HIP_TRY(hipMalloc((void**)&ix->d_layers, bytes));
    if (!h.empty()) HIP_TRY(hipMemcpyAsync(ix->d_layers, h.data(), sizeof(LayerDev) * h.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GRANNE_HIP_OK;
}

static int add_layer_from_device_rows(granne_hip_index* ix, uint64_t len, uint32_t width, const uint32_t* d_rows,
                                      hipStream_t s) {
    LayerHost L;
    L.len = len;
    L.width = width;
    L.dev_width = (width + 31u) & ~31u;
    if (L.dev_width == 0) L.dev_width = 32;
    size_t bytes = (size_t)len * L.dev_width * 4;
    HIP_TRY(hipMalloc((void**)&L.d_adj, bytes ? bytes : 16));
    ix->hbm_bytes += bytes;
    ix->layers.push_back(L);
    if (width == 0) {
        HIP_TRY(hipMemsetAsync(L.d_adj, 0xFF, bytes, s));
    } else {
        hipLaunchKernelGGL(relayout_adj_kernel, dim3(grid_for(len * L.dev_width, 256)), dim3(256), 0, s, d_rows,
                           L.d_adj, len, width, L.dev_width);
        HIP_TRY(hipGetLastError());
    }
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_index_create_device(granne_hip_index** out, const void* d_elements, uint64_t n_elements,
                                              uint32_t dim, int dtype, uint32_t n_layers, const uint64_t* layer_len,
                                              const uint32_t* const* d_layer_rows, const uint32_t* layer_width,
                                              int device_id, void* stream) {
    int rc = validate_common(out, n_elements, dim, dtype, n_layers, layer_len);
    if (rc) return rc;
    if (n_elements && !d_elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    if (n_layers && (!d_layer_rows || !layer_width)) return fail(GRANNE_HIP_ERR_INVALID, "layer arrays are null");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    hipStream_t s = (hipStream_t)stream;
    granne_hip_index* ix = new granne_hip_index();
    ix->device = device_id;
    ix->dim = dim;
    ix->dtype = dtype;
    ix->n_elements = n_elements;
    ix->row_bytes = device_row_bytes(dim, dtype);
    ix->row_stride = device_row_stride(dim, dtype);
    rc = upload_elements_from_device(ix, d_elements, s);
    uint32_t* d_bad = nullptr; // neighbor ids outside their layer (the file loader checks the same on the host)
    if (rc == 0 && hipMalloc((void**)&d_bad, 4) != hipSuccess) rc = fail(GRANNE_HIP_ERR_HIP, "hipMalloc failed");
    if (rc == 0 && hipMemsetAsync(d_bad, 0, 4, s) != hipSuccess) rc = fail(GRANNE_HIP_ERR_HIP, "hipMemsetAsync failed");
    for (uint32_t l = 0; rc == 0 && l < n_layers; ++l) {
        rc = add_layer_from_device_rows(ix, layer_len[l], layer_width[l], d_layer_rows[l], s);
        const uint64_t total = layer_len[l] * layer_width[l];
        if (rc == 0 && total)
            hipLaunchKernelGGL(check_adj_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, d_layer_rows[l], total,
                               layer_len[l], d_bad);
    }
    if (rc == 0) rc = finish_layers(ix, s); // synchronises s
    uint32_t bad = 0;
    if (rc == 0 && hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GRANNE_HIP_ERR_HIP, "hipMemcpy failed");
    if (d_bad) (void)hipFree(d_bad);
    if (rc == 0 && bad) rc = fail(GRANNE_HIP_ERR_INVALID, "%u neighbor ids lie outside their layer", bad);
    if (rc) {
        destroy_index(ix);
        return rc;
    }
    *out = ix;
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_index_create(granne_hip_index** out, const void* elements, uint64_t n_elements, uint32_t dim,
                                       int dtype, uint32_t n_layers, const uint64_t* layer_len,
                                       const uint32_t* const* layer_rows, const uint32_t* layer_width, int device_id) {
    int rc = validate_common(out, n_elements, dim, dtype, n_layers, layer_len);
    if (rc) return rc;
    if (n_elements && !elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    if (n_layers && (!layer_rows || !layer_width)) return fail(GRANNE_HIP_ERR_INVALID, "layer arrays are null");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);

    // stage the host buffers on the device, then share the device path
    void* d_el = nullptr;
    size_t el_bytes = (size_t)n_elements * dim * elem_size(dtype);
    std::vector<uint32_t*> d_rows(n_layers, nullptr);
    auto cleanup = [&]() {
        if (d_el) (void)hipFree(d_el);
        for (auto p : d_rows)
            if (p) (void)hipFree(p);
    };
    hipError_t e = hipMalloc(&d_el, el_bytes ? el_bytes : 16);
    if (e == hipSuccess && el_bytes) e = hipMemcpy(d_el, elements, el_bytes, hipMemcpyHostToDevice);
    for (uint32_t l = 0; e == hipSuccess && l < n_layers; ++l) {
        size_t b = (size_t)layer_len[l] * layer_width[l] * 4;
        e = hipMalloc((void**)&d_rows[l], b ? b : 16);
        if (e == hipSuccess && b) e = hipMemcpy(d_rows[l], layer_rows[l], b, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        cleanup();
        return fail(e == hipErrorNoDevice ? GRANNE_HIP_ERR_NO_DEVICE : GRANNE_HIP_ERR_HIP, "staging upload failed: %s",
                    hipGetErrorString(e));
    }
    rc = granne_hip_index_create_device(out, d_el, n_elements, dim, dtype, n_layers, layer_len,
                                        (const uint32_t* const*)d_rows.data(), layer_width, device_id, nullptr);
    cleanup();
    return rc;
}

extern "C" int granne_hip_index_create_csr(granne_hip_index** out, const void* elements, uint64_t n_elements,
                                           uint32_t dim, int dtype, uint32_t n_layers, const uint64_t* layer_len,
                                           const uint64_t* const* layer_offsets, const uint32_t* const* layer_ids,
                                           int device_id) {
    int rc = validate_common(out, n_elements, dim, dtype, n_layers, layer_len);
    if (rc) return rc;
    if (n_elements && !elements) return fail(GRANNE_HIP_ERR_INVALID, "elements is null");
    if (n_layers && (!layer_offsets || !layer_ids)) return fail(GRANNE_HIP_ERR_INVALID, "layer arrays are null");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);

    granne_hip_index* ix = new granne_hip_index();
    ix->device = device_id;
    ix->dim = dim;
    ix->dtype = dtype;
    ix->n_elements = n_elements;
    ix->row_bytes = device_row_bytes(dim, dtype);
    ix->row_stride = device_row_stride(dim, dtype);
    void* d_el = nullptr;
    size_t el_bytes = (size_t)n_elements * dim * elem_size(dtype);
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&d_el, el_bytes ? el_bytes : 16));
        if (el_bytes) HIP_TRY(hipMemcpy(d_el, elements, el_bytes, hipMemcpyHostToDevice));
        int r = upload_elements_from_device(ix, d_el, nullptr);
        if (r) return r;
        for (uint32_t l = 0; l < n_layers; ++l) {
            uint64_t len = layer_len[l];
            const uint64_t* off = layer_offsets[l];
            uint32_t maxdeg = 0;
            for (uint64_t i = 0; i < len; ++i) {
                if (off[i + 1] < off[i]) return fail(GRANNE_HIP_ERR_INVALID, "layer %u: offsets not monotone", l);
                uint64_t d = off[i + 1] - off[i];
                if (d > 0xFFFF) return fail(GRANNE_HIP_ERR_INVALID, "layer %u: degree too large", l);
                if (d > maxdeg) maxdeg = (uint32_t)d;
            }
            for (uint64_t t = 0; t < off[len]; ++t)
                if (layer_ids[l][t] >= len) return fail(GRANNE_HIP_ERR_INVALID, "layer %u: neighbor id outside the layer", l);
            LayerHost L;
            L.len = len;
            L.width = maxdeg;
            L.dev_width = ((maxdeg ? maxdeg : 1) + 31u) & ~31u;
            size_t bytes = (size_t)len * L.dev_width * 4;
            HIP_TRY(hipMalloc((void**)&L.d_adj, bytes ? bytes : 16));
            ix->hbm_bytes += bytes;
            ix->layers.push_back(L);
            uint64_t* d_off = nullptr;
            uint32_t* d_ids = nullptr;
            size_t nids = (size_t)off[len];
            HIP_TRY(hipMalloc((void**)&d_off, (len + 1) * 8));
            HIP_TRY(hipMalloc((void**)&d_ids, nids ? nids * 4 : 16));
            hipError_t e1 = hipMemcpy(d_off, off, (len + 1) * 8, hipMemcpyHostToDevice);
            hipError_t e2 = nids ? hipMemcpy(d_ids, layer_ids[l], nids * 4, hipMemcpyHostToDevice) : hipSuccess;
            if (e1 == hipSuccess && e2 == hipSuccess) {
                hipLaunchKernelGGL(csr_to_adj_kernel, dim3(grid_for(len * L.dev_width, 256)), dim3(256), 0, nullptr,
                                   d_off, d_ids, L.d_adj, len, L.dev_width);
                e1 = hipDeviceSynchronize();
            }
            (void)hipFree(d_off);
            (void)hipFree(d_ids);
            if (e1 != hipSuccess || e2 != hipSuccess) return fail(GRANNE_HIP_ERR_HIP, "CSR upload failed");
        }
        return finish_layers(ix, nullptr);
    };
    rc = body();
    if (d_el) (void)hipFree(d_el);
    if (rc) {
        destroy_index(ix);
        return rc;
    }
    *out = ix;
    return GRANNE_HIP_OK;
}

extern "C" void granne_hip_index_destroy(granne_hip_index* index) { destroy_index(index); }

extern "C" uint64_t granne_hip_index_len(const granne_hip_index* ix) {
    return (ix && !ix->layers.empty()) ? ix->layers.back().len : 0; // src/index/mod.rs:76-83
}
extern "C" uint32_t granne_hip_index_num_layers(const granne_hip_index* ix) { return ix ? (uint32_t)ix->layers.size() : 0; }
extern "C" uint64_t granne_hip_index_layer_len(const granne_hip_index* ix, uint32_t layer) {
    return (ix && layer < ix->layers.size()) ? ix->layers[layer].len : 0;
}
extern "C" uint32_t granne_hip_index_dim(const granne_hip_index* ix) { return ix ? ix->dim : 0; }
extern "C" int granne_hip_index_dtype(const granne_hip_index* ix) { return ix ? ix->dtype : -1; }
extern "C" int granne_hip_index_device(const granne_hip_index* ix) { return ix ? ix->device : -1; }
extern "C" uint64_t granne_hip_index_hbm_bytes(const granne_hip_index* ix) { return ix ? ix->hbm_bytes : 0; }

extern "C" int granne_hip_index_get_neighbors(const granne_hip_index* ix, uint64_t node, uint32_t layer,
                                              uint32_t* out_ids, uint32_t cap, uint32_t* out_count) {
    if (!ix || !out_count) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (layer >= ix->layers.size()) return fail(GRANNE_HIP_ERR_INVALID, "layer %u out of range", layer);
    const LayerHost& L = ix->layers[layer];
    if (node >= L.len) return fail(GRANNE_HIP_ERR_INVALID, "node out of range");
    DeviceGuard g(ix->device);
    std::vector<uint32_t> row(L.dev_width);
    HIP_TRY(hipMemcpy(row.data(), L.d_adj + node * L.dev_width, (size_t)L.dev_width * 4, hipMemcpyDeviceToHost));
    uint32_t n = 0;
    while (n < L.dev_width && row[n] != GRANNE_HIP_UNUSED) ++n;
    *out_count = n;
    for (uint32_t i = 0; i < n && i < cap && out_ids; ++i) out_ids[i] = row[i];
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_index_get_element(const granne_hip_index* ix, uint64_t idx, void* out) {
    if (!ix || !out) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    if (idx >= ix->n_elements) return fail(GRANNE_HIP_ERR_INVALID, "element index out of range");
    DeviceGuard g(ix->device);
    HIP_TRY(hipMemcpy(out, ix->d_elements + idx * ix->row_stride, (size_t)ix->dim * elem_size(ix->dtype),
                      hipMemcpyDeviceToHost));
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_index_set_option(granne_hip_index* ix, int option, uint64_t value) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    switch (option) {
    case GRANNE_HIP_OPT_VISITED_SLOTS:
        {
            const uint64_t odd = value ? value >> __builtin_ctzll(value) : 1; // 2^k or 3 * 2^k
            if (value != 0 && (value < 256 || value > 32768 || (odd != 1 && odd != 3)))
                return fail(GRANNE_HIP_ERR_INVALID, "visited slots must be 0, 2^k or 3 * 2^k in [256, 32768]");
        }
        ix->opt_visited_slots = value;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_FORCE_SLOW:
        ix->opt_force_slow = value ? 1 : 0;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SLOW_SLOTS:
        if (value < 256 || value > (1ull << 30) || (value & (value - 1)))
            return fail(GRANNE_HIP_ERR_INVALID, "slow slots must be a power of two in [256, 2^30]");
        ix->opt_slow_slots = value;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SLOW_BLOCKS:
        if (value < 1 || value > 1024) return fail(GRANNE_HIP_ERR_INVALID, "slow blocks must be in [1, 1024]");
        ix->opt_slow_blocks = value;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_OVERFLOW_SLOTS:
        if (value > 1 && (value < 512 || value > (1ull << 20) || (value & (value - 1))))
            return fail(GRANNE_HIP_ERR_INVALID, "overflow slots must be 0 (auto), 1 (off) or a power of two in [512, 2^20]");
        ix->opt_overflow_slots = value;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_VISITED16:
        if (value > 4) return fail(GRANNE_HIP_ERR_INVALID, "the visited-set option must be 0 (auto = none), 1..3 (the exact set) or 4 (none)");
        ix->opt_visited16 = value;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SEARCH_DEPTH: {
        if (value < 1 || value > GRANNE_HIP_SEARCH_DEPTH_MAX) return fail(GRANNE_HIP_ERR_INVALID, "search depth must be in [1, %d]", GRANNE_HIP_SEARCH_DEPTH_MAX);
        std::lock_guard<std::mutex> lk(ix->flight_mu);
        for (auto& f : ix->flights)
            if (f.busy) return fail(GRANNE_HIP_ERR_INVALID, "the depth cannot change while a batch is in flight");
        ix->depth = (uint32_t)value;
        ix->next_flight = 0;
        return GRANNE_HIP_OK;
    }
    case GRANNE_HIP_OPT_INLINE_TAILS: {
        if (value > 1) return fail(GRANNE_HIP_ERR_INVALID, "the inline-tails option is 0 or 1");
        DeviceGuard g(ix->device);
        if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
        HIP_TRY(hipDeviceSynchronize()); // (searches in flight read the copy and the layer table this replaces)
        ix->opt_inline_tails = value;
        return finish_layers(ix, nullptr);
    }
    case GRANNE_HIP_OPT_SEEN_MIN:
        ix->opt_seen_min = value > 0xFFFFFFFFull ? 0xFFFFFFFFull : value;
        return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_VISITED16_LG: // (retired with the bucket tables it sized: accepted, ignored)
        if (value > 12) return fail(GRANNE_HIP_ERR_INVALID, "value out of range");
        ix->opt_visited16_lg = value;
        return GRANNE_HIP_OK;
    default:
        return fail(GRANNE_HIP_ERR_INVALID, "unknown option %d", option);
    }
}

extern "C" int granne_hip_index_get_option(const granne_hip_index* ix, int option, uint64_t* value) {
    if (!ix || !value) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    switch (option) {
    case GRANNE_HIP_OPT_VISITED_SLOTS: *value = ix->opt_visited_slots; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_FORCE_SLOW: *value = ix->opt_force_slow; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SLOW_SLOTS: *value = ix->opt_slow_slots; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SLOW_BLOCKS: *value = ix->opt_slow_blocks; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_OVERFLOW_SLOTS: *value = ix->opt_overflow_slots; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_VISITED16: *value = ix->opt_visited16; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_VISITED16_LG: *value = ix->opt_visited16_lg; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_LAST_WALKER: *value = ix->last_walker.load(); return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SEARCH_DEPTH: *value = ix->depth; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_INLINE_TAILS: *value = (!ix->layers.empty() && ix->layers.back().d_adjx) ? 1 : 0; return GRANNE_HIP_OK;
    case GRANNE_HIP_OPT_SEEN_MIN: *value = ix->opt_seen_min; return GRANNE_HIP_OK;
    default: return fail(GRANNE_HIP_ERR_INVALID, "unknown option %d", option);
    }
}

extern "C" uint64_t granne_hip_index_last_slow_count(const granne_hip_index* ix) {
    return ix ? ix->last_slow_count.load() : 0;
}

// ------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------
// what a search launch needs to know about the graph it walks (an index, or a builder's layers)
struct SearchTarget {
    int device;
    const uint8_t* d_elements;
    uint64_t n_elements;
    uint32_t dim;
    int dtype;
    uint32_t row_bytes, row_stride;
    const LayerDev* d_layers;
    uint32_t n_layers;
    uint32_t max_dev_width;
    uint64_t opt_visited_slots, opt_force_slow, opt_slow_slots, opt_slow_blocks, opt_overflow_slots;
    uint64_t opt_visited16 = 0, opt_visited16_lg = 0;
    uint64_t opt_seen_min = 0xFFFFFFFFull; // (a builder's searches: never -- its layers change between launches, its batches are its own)
    ScratchCache* scratch; // search_launch's per-stream scratch blocks
    std::atomic<uint64_t>* last_walker = nullptr; // which kernel the last launch took (an index's read-only option)
};

static SearchTarget target_of(const granne_hip_index* ix) {
    SearchTarget T;
    T.device = ix->device;
    T.d_elements = ix->d_elements;
    T.n_elements = ix->n_elements;
    T.dim = ix->dim;
    T.dtype = ix->dtype;
    T.row_bytes = ix->row_bytes;
    T.row_stride = ix->row_stride;
    T.d_layers = ix->d_layers;
    T.n_layers = (uint32_t)ix->layers.size();
    T.max_dev_width = ix->max_dev_width;
    T.opt_visited_slots = ix->opt_visited_slots;
    T.opt_force_slow = ix->opt_force_slow;
    T.opt_slow_slots = ix->opt_slow_slots;
    T.opt_slow_blocks = ix->opt_slow_blocks;
    T.opt_overflow_slots = ix->opt_overflow_slots;
    T.opt_visited16 = ix->opt_visited16;
    T.opt_visited16_lg = ix->opt_visited16_lg;
    T.opt_seen_min = ix->opt_seen_min;
    T.scratch = &const_cast<granne_hip_index*>(ix)->scratch;
    T.last_walker = &const_cast<granne_hip_index*>(ix)->last_walker;
    return T;
}

typedef void (*search_fn)(const SlowParams);


template <int DT, int DIM>
static search_fn pick_s(uint32_t ef) {
    if (ef <= 64) return search_kernel<DT, DIM, 1>;
    if (ef <= 128) return search_kernel<DT, DIM, 2>;
    return search_kernel<DT, DIM, 4>;
}

// the general walker (search_kernel.h): run-time dims, any row width
static search_fn pick_trail_kernel(int dtype) { // Granne::reorder's trail walks (max_search 1)
    return dtype == GRANNE_HIP_I8 ? (search_fn)search_kernel<DT_I8, 0, 1, true> : (search_fn)search_kernel<DT_F32, 0, 1, true>;
}
static search_fn pick_kernel(int dtype, uint32_t ef) {
    return dtype == GRANNE_HIP_I8 ? pick_s<DT_I8, 0>(ef) : pick_s<DT_F32, 0>(ef);
}

// the walker of the common shapes (walk_fast.h): every layer 32 ids wide on the device, ids within 31
// bits, f32 rows of an instantiated dim or int8 rows of 128 bytes. The list holds 64*S keys and must
// hold max_search of them plus a few spare places: two distances of a walk tie surprisingly often (4000
// candidates share 2^23 float values), and a tie between entry max_search-1 and an entry pushed off the
// end hands the walk over -- with spare places that takes a run of ties.
constexpr uint32_t FAST_MAX_SEARCH = 8192; // f32 rows of 100 / 200 dims and int8 rows of 128 bytes: two-level lists of up to 129 x 64 keys
static uint32_t fast_list_slots(uint32_t ef) {
    return ef <= 60 ? 1u : ef <= 124 ? 2u : ef <= 252 ? 4u : ef <= 508 ? 8u : ef <= 1024 ? 17u : ef <= 2048 ? 33u : ef <= 4096 ? 65u : 129u;
}
// v16: the form of the visited set (FastWalker's V16): 0 = the exact 32-bit table, 3 = none, 4 = none + rows touched ahead
template <int DT, int DIM, int S>
static search_fn pick_fast_v(int v16) {
    if constexpr (S == 1 && !(DT == DT_F32 && DIM == 0) && !(DT == DT_I8 && DIM >= 256)) {
        if (v16 == 4) return fast_kernel<DT, DIM, S, false, 4>; // no visited set + rows touched ahead (few queries)
    }
    if constexpr (DT == DT_F32 && S <= 4) {
        if (v16 == 5) return fast_kernel<DT, DIM, S, false, 5>; // no visited set + revisits skipped before their rows are fetched (many walks)
    }
    if (v16 >= 3) return fast_kernel<DT, DIM, S, false, 3>;
    return fast_kernel<DT, DIM, S>;
}
template <int DT, int DIM>
static search_fn pick_fast_s(uint32_t S, bool trail, int v16) {
    if (trail) return fast_kernel<DT, DIM, 1, true>;
    switch (S) {
    case 1: return pick_fast_v<DT, DIM, 1>(v16);
    case 2: return pick_fast_v<DT, DIM, 2>(v16);
    case 4: return pick_fast_v<DT, DIM, 4>(v16);
    case 8: return v16 >= 3 ? fast_kernel<DT, DIM, 8, false, 3> : fast_kernel<DT, DIM, 8>;
    default:
        if constexpr ((DT == DT_F32 && DIM == 0) || (DT == DT_I8 && DIM >= 256)) {
            // streamed f32 dims and int8 rows of 256 / 512 bytes: lists of up to 17 x 64 keys (max_search 1024)
            if constexpr (walk_list_is_long(17, false)) return fast_kernel<DT, DIM, 17, false, 3>; // (a two-level list: no exact set)
            else return v16 >= 3 ? fast_kernel<DT, DIM, 17, false, 3> : fast_kernel<DT, DIM, 17>;
        } else {
            // lists of 33 / 65 slots (max_search up to 2048 / 4096) exist without a visited set only: plan_launch sends
            // such a search there whatever the option says (an exact set of ~40 x max_search ids fits no LDS)
            if (S == 33) return fast_kernel<DT, DIM, 33, false, 3>;
            if (S == 65) return fast_kernel<DT, DIM, 65, false, 3>;
            if (S == 129) return fast_kernel<DT, DIM, 129, false, 3>;
            if constexpr (walk_list_is_long(17, false)) return fast_kernel<DT, DIM, 17, false, 3>;
            else return v16 >= 3 ? fast_kernel<DT, DIM, 17, false, 3> : fast_kernel<DT, DIM, 17>;
        }
    }
}
// Layers of up to 64 ids per node (graphs with num_neighbors 33..63: the GPU builder makes them, BuildConfig::num_neighbors
// src/index/mod.rs:242) are walked in two passes of 32 pairs per expansion (FastWalker's WIDE) -- instantiated for lists of
// up to 17 x 64 keys, without a visited set, not for Granne::reorder's trail walks, and for int8 rows of 128 bytes only.
static bool fast_wide(const SearchTarget* ix) { return ix->max_dev_width == 64; }
static bool fast_shape(const SearchTarget* ix) {
    // (ids: 31 bits. A 2^31-element index needs 275 GB for its bottom layer's 128-byte adjacency rows alone, so the
    //  reference's 2^32 - 2 capacity, src/index/mod.rs:27-28, is out of one device's reach whatever the key layout)
    if (ix->max_dev_width > 64 || ix->n_elements > WALK_MAX_ELEMENTS) return false;
    if (ix->dtype == GRANNE_HIP_I8)
        return ix->row_bytes == 128 || (!fast_wide(ix) && (ix->row_bytes == 256 || ix->row_bytes == 512)); // dims <= 512
    return true; // f32: 100 and 200 fully unrolled, any other dim streamed (dims below 32: the tail alone)
}
static bool fast_generic(const SearchTarget* ix) { return ix->dtype == GRANNE_HIP_F32 && ix->dim != 100 && ix->dim != 200; }
// the longest max_search the register walker is instantiated for, by shape
static uint32_t fast_max_search(const SearchTarget* ix) {
    if (fast_wide(ix)) return 1024u;                                                        // layers of 64 ids: lists of up to 17 x 64 keys
    if (ix->dtype == GRANNE_HIP_I8) return ix->row_bytes == 128 ? FAST_MAX_SEARCH : 1024u; // wide int8 rows: lists of up to 17 x 64 keys
    return fast_generic(ix) ? 1024u : FAST_MAX_SEARCH;                                      // streamed f32 dims: up to 17 x 64 keys
}
// int8 rows of 256 / 512 bytes (dims 129..512, e.g. the 200- and 300-d rows of benches/distance_computation.rs:29-39)
template <int ROWB>
static search_fn pick_fast_i8_wide(uint32_t S, bool trail, int v16) {
    return pick_fast_s<DT_I8, ROWB>(S, trail, v16);
}
template <int DT, int DIM>
static search_fn pick_fast_wide(uint32_t S) {
    switch (S) {
    case 1: return fast_kernel<DT, DIM, 1, false, 3, true>;
    case 2: return fast_kernel<DT, DIM, 2, false, 3, true>;
    case 4: return fast_kernel<DT, DIM, 4, false, 3, true>;
    case 8: return fast_kernel<DT, DIM, 8, false, 3, true>;
    default: return fast_kernel<DT, DIM, 17, false, 3, true>;
    }
}
static search_fn pick_fast_kernel(const SearchTarget* ix, uint32_t S, bool trail, int v16) {
    if (fast_wide(ix)) { // (search_launch sends trail walks and longer lists of such graphs to the general walker)
        if (ix->dtype == GRANNE_HIP_I8) return pick_fast_wide<DT_I8, 0>(S);
        if (ix->dim == 100) return pick_fast_wide<DT_F32, 100>(S);
        if (ix->dim == 200) return pick_fast_wide<DT_F32, 200>(S);
        return pick_fast_wide<DT_F32, 0>(S);
    }
    if (ix->dtype == GRANNE_HIP_I8 && ix->row_bytes == 256) return pick_fast_i8_wide<256>(S, trail, v16);
    if (ix->dtype == GRANNE_HIP_I8 && ix->row_bytes == 512) return pick_fast_i8_wide<512>(S, trail, v16);
    if (ix->dtype == GRANNE_HIP_I8) return pick_fast_s<DT_I8, 0>(S, trail, v16);
    if (ix->dim == 100) return pick_fast_s<DT_F32, 100>(S, trail, v16);
    if (ix->dim == 200) return pick_fast_s<DT_F32, 200>(S, trail, v16);
    return pick_fast_s<DT_F32, 0>(S, trail, v16);
}

struct LaunchPlan {
    uint32_t visited_slots, upper_slots, maxc, lrow_bytes, stage_bytes, adjspec_bytes, lds_bytes;
    int v16; // FastWalker's V16: 0 = the exact 32-bit table, 3 = no visited set, 4 = none + rows touched ahead
};

// LDS plan. The visited table dominates; the f32 stage gets what keeps four walkers per CU
// (160 KiB / 4) when that leaves it at least 16 rows, else up to 32 rows within 64 KiB, else
// whatever fits in the CU's 160 KiB. GRANNE_HIP_MAXC overrides the stage rows (experiments).
static LaunchPlan plan_launch(const SearchTarget* ix, uint32_t ef, uint32_t nq, uint32_t fastS /* 0: general walker */,
                              bool trail) {
    LaunchPlan P;
    P.v16 = 0;
    // The register walkers walk without a visited set by default (VisitedNone, wave_prims.h: the list itself is searched for
    // a candidate's id): no table in LDS, only the query's staging area and what a tail block needs, so the registers
    // alone bound the walkers per CU. GRANNE_HIP_OPT_VISITED16 = 1..3 switches the exact set on (a 32-bit open-addressing
    // table in LDS + a global overflow table): n_dist is then the reference's count of distinct evaluated nodes.
    const int vmode = ix->opt_visited16 ? ix->opt_visited16 : knobs().visited;
    const bool none = vmode == 4 || vmode == 0;
    const bool longest = walk_list_is_long((int)fastS, fast_wide(ix)) || fast_wide(ix); // (lists of 33 / 65 slots, and 64-id layers: instantiated without a set only, whatever the options say)
    if (fastS >= 1 && !trail && ((none && !ix->opt_visited_slots) || longest) && ix->n_elements <= WALK_MAX_ELEMENTS) { // (fast_shape's bound)
        // a launch of a few queries leaves the chip idle: its walkers touch the next node's rows ahead (walk_fast.h, TOUCH)
        const uint32_t touch_max = knobs().touch_max >= 0 ? (uint32_t)knobs().touch_max : 256u; // (round 5: +4 % at 256 queries, -10 % at 1024: profiles/r5_touch.txt)
        const bool touch_shape = fastS == 1 && !fast_generic(ix) && !(ix->dtype == GRANNE_HIP_I8 && ix->row_bytes != 128);
        P.v16 = (touch_shape && nq <= touch_max) ? 4 : 3;
        // launches of many walks are bound by bandwidth: their walkers skip revisits BEFORE the rows are fetched (walk_fast.h, SEEN)
        const uint64_t seen_min = knobs().seen_min >= 0 ? (uint64_t)knobs().seen_min : ix->opt_seen_min; // (GRANNE_HIP_SEEN_MIN overrides the option: experiments)
        // (f32 rows: the walkers per CU are bound by registers there, 8 KB of cache each fit; int8 walkers are four times as many
        //  and lose more to the look-up than the few revisits of their rows cost: measured, profiles/r6_seen_ab.txt)
        if (fastS <= 4 && !fast_wide(ix) && ix->dtype == GRANNE_HIP_F32 && nq >= seen_min) P.v16 = 5; // (every f32 dim: unrolled and streamed)
        P.visited_slots = P.upper_slots = 0;
        P.maxc = 0;
        P.lrow_bytes = 16;
        P.stage_bytes = 0;
        P.adjspec_bytes = 0;
        P.lds_bytes = fast_lds_bytes(ix->dtype == GRANNE_HIP_I8, fast_generic(ix), ix->dim, ix->row_bytes, fastS, 0u, P.v16 == 5, fast_wide(ix));
        const uint32_t least = lds_query_bytes(ix->row_bytes) + 64u * 8u; // int8 query staging; a tail block (slow_kernel.h)
        if (P.lds_bytes < least) P.lds_bytes = least;
        return P;
    }
    // The front table must hold the walk's visited ids (~40 x max_search on 10M uniform points) below its 7/8
    // load limit: 4096 slots at max_search 50. (Tables of 3 * 2^k slots are accepted as an option; 3072 slots --
    // 12 KB, twelve walkers per CU instead of nine -- measured no faster: 0.5 % of the walks spill to the
    // overflow table and a launch lasts as long as its slowest walk.)
    uint32_t want = ix->opt_visited_slots ? (uint32_t)ix->opt_visited_slots : next_pow2(ef * 64u);
    if (!ix->opt_visited_slots) {
        if (want < 1024) want = 1024;
        // larger walks spill to the global overflow table. A launch with more walkers than the chip
        // holds is better off with small tables (more walkers per CU; measured on the 10M build:
        // 37.7 s vs 45.4 s), one batch of a thousand queries with fewer spills (ef 200: 508k vs 421k q/s)
        uint32_t cap = nq >= 4096 ? 4096u : 8192u;
        // long lists (max_search > 252) walk ~20x max_search nodes: most inserts land in the overflow table
        // whatever the front table's size, and a 32 KB front table beside the list's LDS mirror leaves room
        // for only three walkers per CU (768 of a batch of 1024 resident: the batch runs in two rounds)
        if (fastS >= 8) cap = 4096u;
        if (knobs().visited_cap) cap = next_pow2((uint32_t)knobs().visited_cap); // experiments
        if (cap < 1024) cap = 1024;
        if (cap > 32768) cap = 32768;
        if (want > cap) want = cap;
    }
    P.visited_slots = want;
    P.upper_slots = want < 1024 ? want : 1024;
    if (fastS) { // walk_fast.h: [query][S >= 8: 64*S keys][visited front table]
        P.maxc = 0;
        P.lrow_bytes = 16;
        P.stage_bytes = 0;
        P.adjspec_bytes = 0;
        P.lds_bytes = fast_lds_bytes(ix->dtype == GRANNE_HIP_I8, fast_generic(ix), ix->dim, ix->row_bytes, fastS, P.visited_slots, false, fast_wide(ix));
        return P;
    }
    // the general walker (search_kernel.h): int8 keeps its speculative adjacency rows in registers (Walker::REGSPEC),
    // run-time-dim f32 parks them in LDS
    P.adjspec_bytes = ix->dtype == GRANNE_HIP_I8 ? 0u : LDS_ADJSPEC_BYTES;
    uint32_t fixed = lds_query_bytes(ix->row_bytes) + LDS_FIXED_BYTES + P.adjspec_bytes;
    if (ix->dtype == GRANNE_HIP_F32) {
        uint32_t row16 = ix->row_bytes / 16;
        P.lrow_bytes = (row16 | 1u) * 16u; // odd number of 16-byte units: conflict-free ds_read_b128
        uint32_t wmax = ix->max_dev_width < 64 ? ix->max_dev_width : 64;
        uint32_t used = fixed + P.visited_slots * 4u;
        auto rows_in = [&](uint32_t budget) { return budget > used ? (budget - used) / P.lrow_bytes : 0u; };
        uint32_t maxc = rows_in(40u * 1024u);
        if (maxc < 16) maxc = rows_in(64u * 1024u) < 32u ? rows_in(64u * 1024u) : 32u;
        if (maxc < 16) maxc = rows_in(160u * 1024u) < 32u ? rows_in(160u * 1024u) : 32u;
        if (maxc > wmax) maxc = wmax;
        if (knobs().maxc >= 1 && knobs().maxc <= 64) maxc = (uint32_t)knobs().maxc;
        if (maxc < 1) maxc = 1;
        P.maxc = maxc;
        P.stage_bytes = P.maxc * P.lrow_bytes;
    } else {
        P.maxc = 0;
        P.lrow_bytes = 16;
        P.stage_bytes = 0;
    }
    P.lds_bytes = fixed + P.stage_bytes + P.visited_slots * 4u;
    return P;
}

// A search's scratch block (control words, overflow-region states, hand-over list, overflow tables, the exact
// walker's containers) is kept per (index, stream) for the life of the index: launches on one stream run in
// order, so they can share a block, and the kernels leave its control words and region states zeroed -- no
// allocator call and no memset per search.
constexpr size_t SLOW_SCRATCH_BUDGET = (size_t)1 << 30;               // bytes of exact-walker containers per scratch block
constexpr uint32_t SCRATCH_MAX_REGIONS = 16384;                       // 256 CUs x 32 waves x 2
constexpr size_t SCRATCH_STATE_OFF = CTL_WORDS * 4;                   // region states follow the control words
constexpr size_t SCRATCH_FIXED = SCRATCH_STATE_OFF + (size_t)SCRATCH_MAX_REGIONS * 4; // zero between launches

constexpr size_t SCRATCH_CACHE_STREAMS = 64;      // streams that keep a block for the life of the index
constexpr int SCRATCH_IDLE_SECONDS = 10;          // a cached stream idle this long gives its place to a newcomer when the cache is full
constexpr size_t SCRATCH_SHRINK_ABOVE = 256u << 20; // a cached block this large is let go when a launch needs < 1/4 of it

// `*transient` is set when the block was taken from the stream-ordered allocator for this launch alone (the cache is
// full: a caller with more live streams than SCRATCH_CACHE_STREAMS): the caller frees it with hipFreeAsync after its
// last use on `s`. No device-wide synchronisation anywhere.
static int scratch_for(ScratchCache* cache, hipStream_t s, size_t total, uint8_t** out, bool* transient) {
    // (the caller holds cache->mu for the enqueue)
    *transient = false;
    ScratchCache::Block* b = nullptr;
    cache->uses += 1;
    for (auto& x : cache->blocks)
        if (x.stream == s) b = &x;
    if (!b) {
        if (cache->blocks.size() >= SCRATCH_CACHE_STREAMS) {
            // a stream that has not searched for a long while (destroyed, most likely) gives its place up; while every
            // cached stream is in use the newcomer takes a block from the stream-ordered allocator for this launch alone
            ScratchCache::Block* lru = &cache->blocks[0];
            for (auto& x : cache->blocks)
                if (x.last_use < lru->last_use) lru = &x;
            // idle for SCRATCH_IDLE_SECONDS of wall clock (not "for so many launches": a live stream that pauses while others
            // search keeps its block)
            const auto idle = std::chrono::steady_clock::now() - lru->last_time;
            if (idle > std::chrono::seconds(SCRATCH_IDLE_SECONDS)) {
                if (lru->p) cache->graveyard.push_back(lru->p); // freed once the lock is released (search_launch)
                *lru = {s, nullptr, 0, cache->uses, 0, std::chrono::steady_clock::now()};
                b = lru;
            } else {
                HIP_TRY(hipMallocAsync((void**)out, total, s));
                HIP_TRY(hipMemsetAsync(*out, 0, SCRATCH_FIXED, s));
                *transient = true;
                return GRANNE_HIP_OK;
            }
        } else {
            cache->blocks.push_back({s, nullptr, 0, cache->uses, 0, std::chrono::steady_clock::now()});
            b = &cache->blocks.back();
        }
    }
    b->last_use = cache->uses;
    b->last_time = std::chrono::steady_clock::now();
    // a block left oversized by one exact-walker batch is let go once eight launches in a row needed less than a quarter
    // of it (not at the first: a stream that alternates the two kinds of batches keeps its block)
    if (b->cap > SCRATCH_SHRINK_ABOVE && total < b->cap / 4) b->small_runs += 1;
    else b->small_runs = 0;
    const bool oversized = b->small_runs >= 8;
    if (b->cap < total || oversized) {
        if (b->p) {
            HIP_TRY(hipStreamSynchronize(s)); // earlier launches on this stream still use the old block
            (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
        b->small_runs = 0;
        size_t want = total + (total >> 2); // some headroom: batch sizes vary
        HIP_TRY(hipMalloc((void**)&b->p, want));
        b->cap = want;
        HIP_TRY(hipMemsetAsync(b->p, 0, SCRATCH_FIXED, s));
    }
    *out = b->p;
    return GRANNE_HIP_OK;
}

static int search_launch(const SearchTarget* ix, const void* d_queries, int64_t q_stride, uint32_t nq, uint32_t ef,
                         uint32_t k, uint64_t* d_ids, float* d_dists, uint32_t* d_counts, uint64_t* d_stats,
                         uint32_t* d_status, hipStream_t s, uint32_t* h_slow_count /* optional, syncs */,
                         uint32_t* d_trail = nullptr /* [nq][8]: trail mode, no search outputs */,
                         uint32_t trail_layers = 0, hipEvent_t ev_before = nullptr, hipEvent_t ev_after = nullptr,
                         uint32_t* host_status = nullptr /* u32[2], host-mapped: hand-over count and exhaustion flag, plain stores */,
                         const BatchIO* batches = nullptr, uint32_t n_batches = 0 /* > 0: `nq` queries in EACH of these, one launch */) {
    if (ef == 0) return fail(GRANNE_HIP_ERR_INVALID, "max_search must be > 0 (the reference panics, src/index/mod.rs:1019)");
    if (nq == 0) return GRANNE_HIP_OK;
    if (n_batches > MAX_LAUNCH_BATCHES) return fail(GRANNE_HIP_ERR_INVALID, "at most %u batches per launch", MAX_LAUNCH_BATCHES);
    if (n_batches && (uint64_t)n_batches * nq > 0x7FFFFFFFull) return fail(GRANNE_HIP_ERR_INVALID, "too many queries in one launch");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    const uint32_t batch_nq = nq;
    if (n_batches) {
        for (uint32_t b = 0; b < n_batches; ++b)
            if (!batches[b].queries || !batches[b].out_counts || (k && (!batches[b].out_ids || !batches[b].out_dists)))
                return fail(GRANNE_HIP_ERR_INVALID, "null buffer (batch %u)", b);
        if (k == 0) {
            for (uint32_t b = 0; b < n_batches; ++b) HIP_TRY(hipMemsetAsync(batches[b].out_counts, 0, (size_t)nq * 4, s));
            return GRANNE_HIP_OK;
        }
        d_queries = batches[0].queries;
        d_ids = batches[0].out_ids;
        d_dists = batches[0].out_dists;
        d_counts = batches[0].out_counts;
        d_stats = batches[0].out_stats;
        nq *= n_batches; // walkers of the launch
    }
    if (k == 0 && !d_trail) { // .take(0): every result is empty (src/index/mod.rs:974-977)
        if (!d_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
        HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nq * 4, s));
        return GRANNE_HIP_OK;
    }
    if (!d_queries || (!d_trail && (!d_ids || !d_dists || !d_counts))) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");

    // (the streamed run-time-dim walker is instantiated up to 8 x 64 keys: beyond that the exact walker)
    const bool fast = fast_shape(ix) && ef <= fast_max_search(ix) && !(fast_wide(ix) && d_trail);
    const uint32_t fastS = fast ? fast_list_slots(ef) : 0u;
    const uint32_t ef_walk = fast ? ef : (ef > 256 ? 256 : ef); // what the register/LDS walker is sized for
    const bool all_slow = ix->opt_force_slow || (!fast && ef > 256);
    LaunchPlan plan = plan_launch(ix, ef_walk, nq, fastS, d_trail != nullptr);
    if (plan.lds_bytes > 160u * 1024u) return fail(GRANNE_HIP_ERR_INVALID, "dimension too large for the LDS stage");

    // visited-set overflow pool: one table per walker that can be resident at once (bounded by
    // LDS: 160 KiB per CU, and by 32 waves per CU), at most one per query
    uint32_t ovf_slots = 0, ovf_regions = 0;
    if (!all_slow && ix->opt_overflow_slots != 1 && plan.v16 < 3) { // (no visited set, no overflow)
        ovf_slots = ix->opt_overflow_slots ? next_pow2((uint32_t)ix->opt_overflow_slots) : next_pow2(ef_walk * 64u);
        if (!ix->opt_overflow_slots && ovf_slots < 4096) ovf_slots = 4096;
        if (ovf_slots < 512) ovf_slots = 512;
        if (ovf_slots > (1u << 20)) ovf_slots = 1u << 20;
        uint32_t per_cu = (160u * 1024u) / (plan.lds_bytes ? plan.lds_bytes : 1u);
        if (per_cu > 32) per_cu = 32;
        if (per_cu < 1) per_cu = 1;
        ovf_regions = 256u * per_cu * 2u; // 2x the residency bound keeps the region probe short
        if (ovf_regions > nq) ovf_regions = nq;
        if (ovf_regions > SCRATCH_MAX_REGIONS) ovf_regions = SCRATCH_MAX_REGIONS;
    }

    // scratch: [control words][region states] (zero between launches) [hand-over list][overflow tables][exact walker]
    // The exact walker's blocks: a few as the tail of a register-walker launch (hand-overs are rare), many when the whole
    // batch is its to walk (max_search beyond the register lists, GRANNE_HIP_OPT_FORCE_SLOW): one block per query up to
    // 32x the option (512 at its default of 16) -- each block owns 12 bytes x slow_slots of global scratch.
    // Every block owns 12 bytes x slow_slots + 8 x max_search of global scratch (3 MB at the default 2^18 slots), and the
    // scratch block is kept per (index, stream): the many-block form is bounded by SLOW_SCRATCH_BUDGET bytes (never below
    // the option itself), so that a handful of streams searching beyond the register lists hold a few GB, not tens.
    uint32_t slow_blocks = (uint32_t)ix->opt_slow_blocks;
    const uint32_t slots = (uint32_t)ix->opt_slow_slots;
    if (all_slow) {
        uint32_t most = slow_blocks * 32u;
        const size_t per_block = (size_t)slots * 12 + (size_t)ef * 8;
        const size_t fit = SLOW_SCRATCH_BUDGET / per_block;
        if (most > fit) most = fit > slow_blocks ? (uint32_t)fit : slow_blocks;
        slow_blocks = nq < most ? (nq > slow_blocks ? nq : slow_blocks) : most;
    }
    const uint32_t n_tail = all_slow ? 0u : (slow_blocks < nq ? slow_blocks : nq);
    const size_t list_bytes = ((size_t)nq * 4 + 15) & ~(size_t)15;
    size_t off_list = SCRATCH_FIXED;
    size_t off_ovf = off_list + list_bytes;
    const uint32_t ovf_stride = ovf_slots;
    size_t off_vis = off_ovf + (size_t)ovf_regions * ovf_stride * 4;
    size_t off_pq = off_vis + (size_t)slow_blocks * slots * 4;
    size_t off_res = off_pq + (size_t)slow_blocks * slots * 8;
    size_t total = off_res + (size_t)slow_blocks * ef * 8;
    // The cache's mutex covers finding (or growing) this stream's block and the enqueue -- host work of microseconds.
    // Whatever waits for the GPU (the synchronisation behind h_slow_count) happens after it is released: host threads
    // searching one index on streams of their own do not queue behind each other's kernels.
    struct Bury { // (declared before the lock: destroyed after it is released)
        std::vector<uint8_t*> blocks;
        ~Bury() {
            for (auto* b : blocks) (void)hipFree(b); // waits for the device: whatever still used an evicted block is over
        }
    } bury;
    std::unique_lock<std::mutex> cache_lock(ix->scratch->mu);
    uint8_t* scratch = nullptr;
    bool transient = false;
    {
        int r = scratch_for(ix->scratch, s, total, &scratch, &transient);
        bury.blocks.swap(ix->scratch->graveyard);
        if (r) return r;
    }
    struct FreeTransient { // a block of the stream-ordered allocator goes back after the launch's last use of it
        uint8_t* p;
        hipStream_t s;
        ~FreeTransient() {
            if (p) (void)hipFreeAsync(p, s);
        }
    } free_transient{transient ? scratch : nullptr, s};
    uint32_t* ctl = (uint32_t*)scratch;

    SlowParams sp;
    SearchParams& p = sp.sp;
    p.elements = ix->d_elements;
    p.n_elements = ix->n_elements;
    p.dim = ix->dim;
    p.row_bytes = ix->row_bytes;
    p.row_stride = ix->row_stride;
    p.layers = ix->d_layers;
    p.n_layers = ix->n_layers;
    p.queries = (const uint8_t*)d_queries;
    p.q_stride = q_stride;
    p.nq = nq;
    p.ef = ef;
    p.k = k;
    p.out_ids = d_ids;
    p.out_dists = d_dists;
    p.out_counts = d_counts;
    p.out_stats = d_stats;
    p.visited_slots = plan.visited_slots;
    p.upper_slots = plan.upper_slots;
    // A walk that will outgrow the front table anyway (it visits ~40 x max_search nodes) freezes it at 5/8 load
    // instead of 7/8: every later lookup of a new id runs to an empty slot of the frozen table, 8 probes on average
    // at 7/8 load with the wave waiting for its slowest lane, 2.7 at 5/8 (C5-like int8 walk at max_search 200:
    // launch 2.70 -> 2.44 ms; f32 at 800: 7.75 -> 6.5 ms).
    p.front_eighths = (!plan.v16 && (uint64_t)ef * 40u > (uint64_t)plan.visited_slots) ? 5u : 7u;
    if (knobs().front_eighths >= 1 && knobs().front_eighths <= 7) p.front_eighths = (uint32_t)knobs().front_eighths;
    p.maxc = plan.maxc;
    p.lrow_bytes = plan.lrow_bytes;
    p.stage_bytes = plan.stage_bytes;
    p.adjspec_bytes = plan.adjspec_bytes;
    p.slow_count = ctl + CTL_SLOW_COUNT;
    p.slow_list = (uint32_t*)(scratch + off_list);
    p.force_slow = 0;
    p.spec = 1;
    p.ovf.tables = (uint32_t*)(scratch + off_ovf);
    p.ovf.state = (uint32_t*)(scratch + SCRATCH_STATE_OFF);
    p.ovf.slots = ovf_slots;
    p.ovf.stride = ovf_stride;
    p.ovf.regions = ovf_regions;
    p.ovf.spilled = d_status ? d_status + 2 : ctl + CTL_SPILLED;
    p.trail_out = d_trail;
    p.trail_layers = trail_layers;
    p.n_batches = n_batches;
    p.batch_nq = batch_nq;
    for (uint32_t b = 0; b < n_batches; ++b) p.batch[b] = batches[b];
    sp.ctl = ctl;
    sp.vis = (uint32_t*)(scratch + off_vis);
    sp.pq = (uint64_t*)(scratch + off_pq);
    sp.res = (uint64_t*)(scratch + off_res);
    sp.slots = slots;
    sp.all = all_slow ? 1u : 0u;
    sp.status2 = d_status;
    sp.host_status = host_status;

    if (ix->last_walker)
        ix->last_walker->store(all_slow ? GRANNE_HIP_WALKER_EXACT
                                        : fast ? (fast_wide(ix) ? GRANNE_HIP_WALKER_REGISTER_WIDE : GRANNE_HIP_WALKER_REGISTER)
                                               : GRANNE_HIP_WALKER_GENERAL);
    const uint32_t slow_lds = lds_query_bytes(ix->row_bytes) + 64 * 8;
    if (all_slow) { // every query on the exact walker: its kernel alone
        if (ev_before) HIP_TRY(hipEventRecord(ev_before, s));
        if (ix->dtype == GRANNE_HIP_F32)
            hipLaunchKernelGGL(slow_kernel<DT_F32>, dim3(slow_blocks), dim3(64), slow_lds, s, sp);
        else
            hipLaunchKernelGGL(slow_kernel<DT_I8>, dim3(slow_blocks), dim3(64), slow_lds, s, sp);
        HIP_TRY(hipGetLastError());
        if (ev_after) HIP_TRY(hipEventRecord(ev_after, s));
    } else {
        search_fn fn = fast ? pick_fast_kernel(ix, fastS, d_trail != nullptr, plan.v16)
                            : (d_trail ? pick_trail_kernel(ix->dtype) : pick_kernel(ix->dtype, ef_walk));
        uint32_t lds = plan.lds_bytes > slow_lds ? plan.lds_bytes : slow_lds; // the tail blocks stage a query too
        lds += (uint32_t)knobs().lds_pad; // occupancy experiments
        if (lds > 160u * 1024u) return fail(GRANNE_HIP_ERR_INVALID, "dimension too large for the LDS stage");
        if (lds > 32u * 1024u)
            HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (ev_before) HIP_TRY(hipEventRecord(ev_before, s));
        hipLaunchKernelGGL(fn, dim3(nq + n_tail), dim3(64), lds, s, sp);
        HIP_TRY(hipGetLastError());
        if (ev_after) HIP_TRY(hipEventRecord(ev_after, s));
    }

    if (h_slow_count) {
        uint32_t hs[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(hs, ctl + CTL_LAST_SLOW, 8, hipMemcpyDeviceToHost, s)); // enqueued under the lock, ...
        cache_lock.unlock();
        HIP_TRY(hipStreamSynchronize(s));                                               // ... waited for outside it
        h_slow_count[0] = hs[0]; // queries served by the global-memory walker
        h_slow_count[1] = hs[1]; // its containers ran out
    }
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_search_batch_device(const granne_hip_index* ix, const void* d_queries, uint32_t nq,
                                              uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                              float* d_out_dists, uint32_t* d_out_counts, uint64_t* d_out_stats,
                                              uint32_t* d_status, void* stream) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    SearchTarget T = target_of(ix);
    return search_launch(&T, d_queries, (int64_t)ix->dim * elem_size(ix->dtype), nq, max_search, num_neighbors,
                         d_out_ids, d_out_dists, d_out_counts, d_out_stats, d_status, (hipStream_t)stream, nullptr);
}

extern "C" int granne_hip_search_batch_device_timed(const granne_hip_index* ix, const void* d_queries, uint32_t nq,
                                                    uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                                    float* d_out_dists, uint32_t* d_out_counts, uint64_t* d_out_stats,
                                                    uint32_t* d_status, void* stream, void* ev_before, void* ev_after) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    SearchTarget T = target_of(ix);
    return search_launch(&T, d_queries, (int64_t)ix->dim * elem_size(ix->dtype), nq, max_search, num_neighbors,
                         d_out_ids, d_out_dists, d_out_counts, d_out_stats, d_status, (hipStream_t)stream, nullptr,
                         nullptr, 0, (hipEvent_t)ev_before, (hipEvent_t)ev_after);
}

// A batch in flight beside the caller's stream. begin: the search is ordered after what `stream` holds (an event), runs on
// one of the index's own streams, and the call returns a ticket at once; end: `stream` waits for that search. Between
// the two the caller begins further batches -- the walks of up to GRANNE_HIP_SEARCH_DEPTH batches share the chip, which
// is what a host with a stream of independent batch-sized requests needs (one launch of 1024 walks is one wave per SIMD).
extern "C" int granne_hip_search_begin_device(const granne_hip_index* cix, const void* d_queries, uint32_t nq,
                                              uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                              float* d_out_dists, uint32_t* d_out_counts, uint64_t* d_out_stats,
                                              uint32_t* d_status, void* stream, uint64_t* out_ticket) {
    if (!cix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (!out_ticket) return fail(GRANNE_HIP_ERR_INVALID, "out_ticket is null");
    if (max_search == 0) return fail(GRANNE_HIP_ERR_INVALID, "max_search must be > 0 (the reference panics, src/index/mod.rs:1019)");
    granne_hip_index* ix = const_cast<granne_hip_index*>(cix);
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    std::lock_guard<std::mutex> lk(ix->flight_mu);
    // any flight that is free, looked for from the one after the last begun (tickets may be ended in any order)
    uint32_t fi = ix->depth;
    for (uint32_t t = 0; t < ix->depth; ++t) {
        const uint32_t c = (ix->next_flight + t) % ix->depth;
        if (!ix->flights[c].busy) {
            fi = c;
            break;
        }
    }
    if (fi == ix->depth)
        return fail(GRANNE_HIP_ERR_INVALID, "%u batches are in flight already (GRANNE_HIP_OPT_SEARCH_DEPTH): end one first", ix->depth);
    auto& F = ix->flights[fi];
    if (!F.done) { // created together or not at all: a failure leaves the flight as it was
        hipStream_t st = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&e0, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&e1, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (e1) (void)hipEventDestroy(e1);
            if (e0) (void)hipEventDestroy(e0);
            if (st) (void)hipStreamDestroy(st);
            return fail(GRANNE_HIP_ERR_HIP, "a stream for a batch in flight: %s", hipGetErrorString(e));
        }
        F.stream = st;
        F.ready = e0;
        F.done = e1;
    }
    HIP_TRY(hipEventRecord(F.ready, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(F.stream, F.ready, 0));
    SearchTarget T = target_of(ix);
    int rc = search_launch(&T, d_queries, (int64_t)ix->dim * elem_size(ix->dtype), nq, max_search, num_neighbors, d_out_ids,
                           d_out_dists, d_out_counts, d_out_stats, d_status, F.stream, nullptr);
    if (rc) {
        (void)hipStreamSynchronize(F.stream); // an error return leaves nothing running
        return rc;
    }
    HIP_TRY(hipEventRecord(F.done, F.stream));
    F.busy = true;
    F.seq = ix->next_seq++;
    ix->next_flight = (fi + 1) % ix->depth;
    *out_ticket = (F.seq << 8) | fi;
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_search_end_device(const granne_hip_index* cix, uint64_t ticket, void* stream) {
    if (!cix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    granne_hip_index* ix = const_cast<granne_hip_index*>(cix);
    std::lock_guard<std::mutex> lk(ix->flight_mu);
    const uint32_t fi = (uint32_t)(ticket & 0xFF);
    if (fi >= GRANNE_HIP_SEARCH_DEPTH_MAX || !ix->flights[fi].busy || ix->flights[fi].seq != (ticket >> 8))
        return fail(GRANNE_HIP_ERR_INVALID, "no batch in flight has this ticket");
    DeviceGuard g(ix->device);
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ix->flights[fi].done, 0));
    ix->flights[fi].busy = false;
    return GRANNE_HIP_OK;
}

// Several batches of `nq` queries in ONE launch: a grid of n_batches x nq walkers, which the dispatcher refills from as
// walks finish -- the in-flight depth a host otherwise has to provide with streams of its own (a launch of 1024 walks is
// one wave per SIMD and lasts as long as its slowest walk). More than MAX_LAUNCH_BATCHES batches go out as several
// launches on the same stream.
extern "C" int granne_hip_search_batches_device(const granne_hip_index* ix, uint32_t n_batches, const void* const* d_queries,
                                                uint32_t nq, uint32_t max_search, uint32_t num_neighbors,
                                                uint64_t* const* d_out_ids, float* const* d_out_dists,
                                                uint32_t* const* d_out_counts, uint64_t* const* d_out_stats,
                                                uint32_t* d_status, void* stream) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (max_search == 0) return fail(GRANNE_HIP_ERR_INVALID, "max_search must be > 0 (the reference panics, src/index/mod.rs:1019)");
    if (n_batches == 0 || nq == 0) return GRANNE_HIP_OK;
    if (!d_queries || !d_out_counts || (num_neighbors && (!d_out_ids || !d_out_dists)))
        return fail(GRANNE_HIP_ERR_INVALID, "null pointer array");
    SearchTarget T = target_of(ix);
    // as many batches per launch as keep the launch's walker count within 31 bits (and the kernel argument table)
    uint32_t per = MAX_LAUNCH_BATCHES;
    while (per > 1 && (uint64_t)per * nq > 0x7FFFFFFFull) per >>= 1;
    for (uint32_t b0 = 0; b0 < n_batches; b0 += per) {
        const uint32_t nb = n_batches - b0 < per ? n_batches - b0 : per;
        BatchIO io[MAX_LAUNCH_BATCHES];
        for (uint32_t b = 0; b < nb; ++b) {
            io[b].queries = (const uint8_t*)d_queries[b0 + b];
            io[b].out_ids = num_neighbors ? d_out_ids[b0 + b] : nullptr;
            io[b].out_dists = num_neighbors ? d_out_dists[b0 + b] : nullptr;
            io[b].out_counts = d_out_counts[b0 + b];
            io[b].out_stats = d_out_stats ? d_out_stats[b0 + b] : nullptr;
        }
        int rc = search_launch(&T, nullptr, (int64_t)ix->dim * elem_size(ix->dtype), nq, max_search, num_neighbors, nullptr,
                               nullptr, nullptr, nullptr, d_status, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, nullptr,
                               nullptr, io, nb);
        if (rc) return rc;
    }
    return GRANNE_HIP_OK;
}

#if GRANNE_HIP_PHASE_TIMERS
// diagnostics build only (tools/phase_probe.py): the per-walk phase clocks of the last launches
extern "C" int granne_hip_debug_phases(uint64_t* out, uint32_t nq) {
    if (!out || nq > PHASE_QUERIES) return fail(GRANNE_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), (size_t)nq * PHASE_SLOTS * 8));
    return GRANNE_HIP_OK;
}
#endif

extern "C" int granne_hip_device_malloc(void** out_ptr, uint64_t bytes, int device_id) {
    if (!out_ptr) return fail(GRANNE_HIP_ERR_INVALID, "out_ptr is null");
    *out_ptr = nullptr;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    HIP_TRY(hipMalloc(out_ptr, bytes ? bytes : 16));
    return GRANNE_HIP_OK;
}
extern "C" int granne_hip_device_free(void* ptr, int device_id) {
    if (!ptr) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    HIP_TRY(hipFree(ptr));
    return GRANNE_HIP_OK;
}
extern "C" int granne_hip_copy_to_device(void* d_dst, const void* src, uint64_t bytes, int device_id, void* stream) {
    if (bytes == 0) return GRANNE_HIP_OK;
    if (!d_dst || !src) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return GRANNE_HIP_OK;
}
extern "C" int granne_hip_copy_to_host(void* dst, const void* d_src, uint64_t bytes, int device_id, void* stream) {
    if (bytes == 0) return GRANNE_HIP_OK;
    if (!dst || !d_src) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return GRANNE_HIP_OK;
}
extern "C" int granne_hip_stream_create(void** out_stream, int device_id) {
    if (!out_stream) return fail(GRANNE_HIP_ERR_INVALID, "out_stream is null");
    *out_stream = nullptr;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = (void*)s;
    return GRANNE_HIP_OK;
}
extern "C" int granne_hip_stream_destroy(void* stream, int device_id) {
    if (!stream) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return GRANNE_HIP_OK;
}
extern "C" int granne_hip_stream_synchronize(void* stream, int device_id) {
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_event_create(void** out_event) {
    if (!out_event) return fail(GRANNE_HIP_ERR_INVALID, "out_event is null");
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    *out_event = (void*)e;
    return GRANNE_HIP_OK;
}
extern "C" void granne_hip_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}
extern "C" int granne_hip_event_elapsed_ms(void* before, void* after, float* out_ms) {
    if (!before || !after || !out_ms) return fail(GRANNE_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipEventElapsedTime(out_ms, (hipEvent_t)before, (hipEvent_t)after));
    return GRANNE_HIP_OK;
}

// borrow / return a host-call context (see granne_hip_index::HostCall)
static granne_hip_index::HostCall* host_call_acquire(granne_hip_index* ix) {
    {
        std::lock_guard<std::mutex> lk(ix->call_mu);
        if (!ix->call_free.empty()) {
            auto* c = ix->call_free.back();
            ix->call_free.pop_back();
            return c;
        }
    }
    auto* c = new granne_hip_index::HostCall();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return nullptr;
    }
    return c;
}
static void host_call_release(granne_hip_index* ix, granne_hip_index::HostCall* c) {
    std::lock_guard<std::mutex> lk(ix->call_mu);
    ix->call_free.push_back(c);
}

extern "C" int granne_hip_search_batch(const granne_hip_index* cix, const void* queries, uint32_t nq, uint32_t max_search,
                                       uint32_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                                       uint64_t* out_stats) {
    if (!cix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    granne_hip_index* ix = const_cast<granne_hip_index*>(cix);
    if (max_search == 0) return fail(GRANNE_HIP_ERR_INVALID, "max_search must be > 0 (the reference panics, src/index/mod.rs:1019)");
    if (nq == 0) return GRANNE_HIP_OK;
    if (num_neighbors == 0) { // .take(0), src/index/mod.rs:974-977
        if (!out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
        memset(out_counts, 0, (size_t)nq * 4);
        return GRANNE_HIP_OK;
    }
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);

    // device layout: [queries][ids][dists][counts][stats]; the four outputs are contiguous so that a
    // small batch comes back in one copy
    const size_t k = num_neighbors;
    const size_t qb = (size_t)nq * ix->dim * elem_size(ix->dtype);
    const size_t o_ids = (qb + 255) & ~(size_t)255;
    const size_t o_d = o_ids + (size_t)nq * k * 8;
    const size_t o_c = o_d + (((size_t)nq * k * 4 + 15) & ~(size_t)15);
    const size_t o_s = o_c + (((size_t)nq * 4 + 15) & ~(size_t)15);
    const size_t total = o_s + (size_t)nq * 24;
    const bool staged = total <= (256u << 10); // small calls go through the pinned buffer: two DMA copies per call

    granne_hip_index::HostCall* c = host_call_acquire(ix);
    if (!c) return fail(GRANNE_HIP_ERR_HIP, "cannot create a stream");
    struct Release { // the context goes back to the pool only once its stream is idle: on an error return kernels and
        granne_hip_index* ix; // copies of this call may still be running on it, reading and writing the caller's buffers
        granne_hip_index::HostCall* c;
        ~Release() {
            (void)hipStreamSynchronize(c->stream);
            host_call_release(ix, c);
        }
    } release{ix, c};
    if (staged && c->h_cap < total) {
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->h_pin = nullptr;
        c->h_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&c->h_pin, (256u << 10) + 64, hipHostMallocMapped));
        c->h_cap = 256u << 10;
    }
    hipStream_t s = c->stream;
    SearchTarget T = target_of(ix);
    uint32_t slow[2] = {0, 0};
    if (staged) {
        // Small calls (one query per call is the reference's own shape, src/index/mod.rs:140-150): no copy
        // engine at all. The pinned block is mapped into the device's address space: the walker reads the
        // queries from it and writes results and status words into it over PCIe (a few hundred bytes), the
        // scratch block is the caller context's own. What is left on the stream: one memset of the scratch
        // header, the walker, the (normally empty) exact walker, one synchronisation.
        void* dev_pin = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dev_pin, c->h_pin, 0));
        uint8_t* dp = (uint8_t*)dev_pin;
        uint32_t* hst = (uint32_t*)(c->h_pin + total);
        memcpy(c->h_pin, queries, qb);
        hst[0] = hst[1] = hst[2] = hst[3] = 0;
        int r = search_launch(&T, dp, (int64_t)ix->dim * elem_size(ix->dtype), nq, max_search, num_neighbors,
                              (uint64_t*)(dp + o_ids), (float*)(dp + o_d), (uint32_t*)(dp + o_c), (uint64_t*)(dp + o_s),
                              nullptr, s, nullptr, nullptr, 0, nullptr, nullptr, (uint32_t*)(dp + total));
        hipError_t e = hipStreamSynchronize(s);
        if (r) return r;
        if (e != hipSuccess) return fail(GRANNE_HIP_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
        slow[0] = hst[1]; // queries served by the exact global-memory walker
        slow[1] = hst[0]; // its scratch ran out
        ix->last_slow_count.store(slow[0]);
        if (slow[1]) return fail(GRANNE_HIP_ERR_OVERFLOW, "exact-search scratch exhausted (raise GRANNE_HIP_OPT_SLOW_SLOTS)");
        memcpy(out_ids, c->h_pin + o_ids, (size_t)nq * k * 8);
        memcpy(out_dists, c->h_pin + o_d, (size_t)nq * k * 4);
        memcpy(out_counts, c->h_pin + o_c, (size_t)nq * 4);
        if (out_stats) memcpy(out_stats, c->h_pin + o_s, (size_t)nq * 24);
        return GRANNE_HIP_OK;
    }
    if (c->d_cap < total) {
        if (c->d_buf) (void)hipFree(c->d_buf);
        c->d_buf = nullptr;
        c->d_cap = 0;
        HIP_TRY(hipMalloc((void**)&c->d_buf, total));
        c->d_cap = total;
    }
    uint8_t* buf = c->d_buf;
    HIP_TRY(hipMemcpyAsync(buf, queries, qb, hipMemcpyHostToDevice, s));
    int r = search_launch(&T, buf, (int64_t)ix->dim * elem_size(ix->dtype), nq, max_search, num_neighbors,
                          (uint64_t*)(buf + o_ids), (float*)(buf + o_d), (uint32_t*)(buf + o_c), (uint64_t*)(buf + o_s),
                          nullptr, s, slow);
    if (r) {
        (void)hipStreamSynchronize(s);
        return r;
    }
    ix->last_slow_count.store(slow[0]);
    if (slow[1]) return fail(GRANNE_HIP_ERR_OVERFLOW, "exact-search scratch exhausted (raise GRANNE_HIP_OPT_SLOW_SLOTS)");
    HIP_TRY(hipMemcpyAsync(out_ids, buf + o_ids, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_dists, buf + o_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_counts, buf + o_c, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    if (out_stats) HIP_TRY(hipMemcpyAsync(out_stats, buf + o_s, (size_t)nq * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_search(const granne_hip_index* ix, const void* query, uint32_t max_search,
                                 uint32_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_count) {
    if (!out_count) return fail(GRANNE_HIP_ERR_INVALID, "out_count is null");
    return granne_hip_search_batch(ix, query, 1, max_search, num_neighbors, out_ids, out_dists, out_count, nullptr);
}

// ------------------------------------------------------------------------------------------------
// element preparation / Dist operator / synthetic data
// ------------------------------------------------------------------------------------------------
extern "C" int granne_hip_normalize_f32_device(float* d_rows, uint64_t n, uint32_t dim, int device_id, void* stream) {
    if (!d_rows && n) return fail(GRANNE_HIP_ERR_INVALID, "rows is null");
    if (dim == 0) return fail(GRANNE_HIP_ERR_INVALID, "dim must be > 0");
    if (n == 0) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    uint32_t lstride = dim | 1u;
    uint32_t rpb = (60u * 1024u) / (lstride * 4u);
    if (rpb > 256) rpb = 256;
    if (rpb < 1) return fail(GRANNE_HIP_ERR_INVALID, "dim too large");
    uint64_t blocks = (n + rpb - 1) / rpb;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((uint32_t)blocks), dim3(256), rpb * lstride * 4, (hipStream_t)stream,
                       d_rows, n, dim, rpb, lstride);
    HIP_TRY(hipGetLastError());
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_quantize_f32_device(const float* d_rows, int8_t* d_out, uint64_t n, uint32_t dim,
                                              int device_id, void* stream) {
    if ((!d_rows || !d_out) && n) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (n == 0) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    hipLaunchKernelGGL(quantize_rows_kernel, dim3(grid_for(n * 64, 256)), dim3(256), 0, (hipStream_t)stream, d_rows,
                       d_out, n, dim);
    HIP_TRY(hipGetLastError());
    return GRANNE_HIP_OK;
}

static int dists_launch(const granne_hip_index* ix, const void* d_queries, const uint32_t* d_qidx, uint32_t m,
                        const uint32_t* d_ids, uint64_t n_pairs, float* d_out, uint32_t* d_status, void* stream) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (n_pairs == 0) return GRANNE_HIP_OK;
    if (!d_queries || !d_ids || !d_out) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    uint32_t* st = d_status; // optional: count of out-of-range ids
    // eight lanes per pair, 32 pairs per 256-thread block; enough blocks to cover the chip many times
    uint64_t blocks = (n_pairs + 31) / 32;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    if (ix->dtype == GRANNE_HIP_F32)
        hipLaunchKernelGGL(dists_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, ix->d_elements,
                           ix->n_elements, ix->row_bytes, ix->row_stride, ix->dim, (const uint8_t*)d_queries, d_qidx, m, d_ids, n_pairs,
                           d_out, st);
    else
        hipLaunchKernelGGL(dists_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, ix->d_elements,
                           ix->n_elements, ix->row_bytes, ix->row_stride, ix->dim, (const uint8_t*)d_queries, d_qidx, m, d_ids, n_pairs,
                           d_out, st);
    HIP_TRY(hipGetLastError());
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_dist_pairs_device(const granne_hip_index* ix, const void* d_queries, const uint32_t* d_qidx,
                                            const uint32_t* d_ids, uint64_t n_pairs, float* d_out, void* stream) {
    if (n_pairs && !d_qidx) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    return dists_launch(ix, d_queries, d_qidx, 1, d_ids, n_pairs, d_out, nullptr, stream);
}

extern "C" int granne_hip_dists_device(const granne_hip_index* ix, const void* d_queries, uint32_t nq,
                                       const uint32_t* d_ids, uint32_t m, float* d_out, uint32_t* d_status,
                                       void* stream) {
    if (nq && m == 0) return GRANNE_HIP_OK;
    return dists_launch(ix, d_queries, nullptr, m ? m : 1, d_ids, (uint64_t)nq * m, d_out, d_status, stream);
}

static int merge_launch(const uint8_t* ids, const uint8_t* dists, const uint8_t* counts, uint64_t ids_stride,
                        uint64_t dists_stride, uint64_t counts_stride, const uint64_t* shard_offsets, uint32_t n_shards,
                        uint32_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                        int device_id, void* stream);

// exact k nearest elements by a scan of all of them on the matrix cores (brute_force.h)
extern "C" int granne_hip_brute_force_device(const granne_hip_index* ix, const void* d_queries, uint32_t nq, uint32_t k,
                                             uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, void* stream) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (nq == 0) return GRANNE_HIP_OK;
    if (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (k == 0 || k > BF_KMAX) return fail(GRANNE_HIP_ERR_INVALID, "k must be in [1, %u]", BF_KMAX);
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t n = ix->n_elements;
    const uint32_t kk = k + BF_EXTRA < BF_KMAX ? k + BF_EXTRA : BF_KMAX;
    // element ranges: the lists of up to 64 ranges are merged; a range is a whole number of tiles
    uint32_t R = 4, lds = 0, qt = BF_QT, threads = BF_THREADS;
    uint32_t prime_lds = 0, prime_qt = 0, prime_threads = 0; // 0: the scan's own
    uint64_t max_ranges = 64;
    void (*fn)(const BruteParams) = nullptr;
    void (*fn_prime)(const BruteParams) = nullptr; // the same scan keeping only the best score per (range, query)
    if (ix->dtype == GRANNE_HIP_I8) {
        R = 4;
        fn = bf_i8_kernel<4>;
        fn_prime = bf_i8_kernel<4, true>;
        lds = BF_I8_SUB * (32u * R * (128u + 16u) + 32u * R * 4u + 2u * R * 4u);
        if (ix->row_bytes > 128u) { // rows of any length: in chunks of 128 bytes (brute_force.h, bf_i8_chunked_kernel)
            fn = bf_i8_chunked_kernel<4>;
            fn_prime = bf_i8_chunked_kernel<4, true>;
            lds = 32u * R * (128u + 16u) + 32u * R * 4u + 2u * R * 4u;
        } else if (knobs().bf_ring && ix->row_bytes == 128u && ix->row_stride == 128u && n < (1ull << 29)) {
            // tiles by LDS-DMA into a ring, 64 queries per wave (brute_force.h, bf_i8_ring_kernel); the priming pass stays
            prime_lds = lds, prime_qt = qt, prime_threads = threads;
            fn = bf_i8_ring_kernel;
            lds = BF_RING_LDS;
            qt = BF_RING_QT, threads = BF_RING_THREADS;
            // 512 queries per block: half as many blocks per range, so twice the ranges fill the chip (merged in two steps)
            if ((uint64_t)((nq + qt - 1) / qt) * 64u < 256u) max_ranges = 128;
        }
    } else if (knobs().bf_b16 && ix->dim <= 112) { // f32 rows on the bf16 matrix path, three instructions per product (brute_force.h)
        R = 4;
        fn = bf_b16_kernel<7, 4>;
        fn_prime = bf_b16_kernel<7, 4, true>;
        lds = 2u * 32u * R * (2u * 16u * 7u + 16u);
        qt = BF_B16_QT, threads = BF_B16_THREADS;
    } else if (knobs().bf_b16 && ix->dim <= 208) {
        R = 2;
        fn = bf_b16_kernel<13, 2>;
        fn_prime = bf_b16_kernel<13, 2, true>;
        lds = 2u * 32u * R * (2u * 16u * 13u + 16u);
        qt = BF_B16_QT, threads = BF_B16_THREADS;
    } else if (ix->dim > 256) { // rows of any length: the vector in chunks of 128 components (brute_force.h, bf_b16_chunked_kernel)
        R = 2; // (tiles of 64 rows: 128 take 256 registers and spill)
        fn = bf_b16_chunked_kernel<2>;
        fn_prime = bf_b16_chunked_kernel<2, true>;
        lds = 2u * 32u * R * (2u * 16u * 8u + 16u);
        qt = BF_B16_QT, threads = BF_B16_THREADS;
    } else if (knobs().bf_b16) {
        R = 1;
        fn = bf_b16_kernel<16, 1>;
        fn_prime = bf_b16_kernel<16, 1, true>;
        lds = 2u * 32u * R * (2u * 16u * 16u + 16u);
        qt = BF_B16_QT, threads = BF_B16_THREADS;
    } else if (ix->dim <= 104) {
        R = 4;
        fn = bf_f32_kernel<52, 4>;
        fn_prime = bf_f32_kernel<52, 4, true>;
        lds = 32u * R * (2u * 52u + 4u) * 4u;
    } else if (ix->dim <= 200) {
        R = 2;
        fn = bf_f32_kernel<100, 2>;
        fn_prime = bf_f32_kernel<100, 2, true>;
        lds = 32u * R * (2u * 100u + 4u) * 4u;
    } else {
        R = 1;
        fn = bf_f32_kernel<128, 1>;
        fn_prime = bf_f32_kernel<128, 1, true>;
        lds = 32u * R * (2u * 128u + 4u) * 4u;
    }
    const uint64_t tile = 32ull * R;
    uint64_t G = (n + tile - 1) / tile;
    if (G > max_ranges) G = max_ranges;
    if (G < 1) G = 1;
    uint64_t per_range = (n + G - 1) / G;
    per_range = (per_range + tile - 1) / tile * tile;
    if (per_range < tile) per_range = tile;
    const size_t lists = (size_t)G * nq;
    const size_t o_pid = 0, o_pd = o_pid + lists * kk * 8, o_pc = o_pd + lists * kk * 4;
    const size_t o_mid = (o_pc + lists * 4 + 15) & ~(size_t)15, o_md = o_mid + (size_t)nq * kk * 8, o_mc = o_md + (size_t)nq * kk * 4;
    const size_t o_cand = (o_mc + (size_t)nq * 4 + 15) & ~(size_t)15, o_ex = o_cand + (size_t)nq * kk * 4;
    const size_t o_tau = (o_ex + (size_t)nq * kk * 4 + 15) & ~(size_t)15;
    const size_t o_share = (o_tau + (size_t)nq * 4 + 15) & ~(size_t)15; // what the ranges of a query have seen, per score bucket
    const size_t share_bytes = (size_t)nq * BF_SHARE_BUCKETS * 4;
    const size_t o_qpad = (o_share + share_bytes + 15) & ~(size_t)15; // int8 rows of more than 128 bytes: the queries, zero padded
    const size_t qpad_bytes = (ix->dtype == GRANNE_HIP_I8 && ix->row_bytes > 128u) ? (size_t)nq * ix->row_bytes : 0;
    const size_t total = o_qpad + qpad_bytes;
    uint8_t* scratch = nullptr;
    HIP_TRY(hipMallocAsync((void**)&scratch, total, s));
    struct Release {
        void* p;
        hipStream_t s;
        ~Release() { (void)hipFreeAsync(p, s); }
    } release{scratch, s};
    BruteParams P;
    P.inv_norm = nullptr;
    P.inv_gmax = nullptr;
    const uint64_t n_pad = (n + 31u) & ~31ull; // inv_norm [n_pad], then inv_gmax [n_pad / 32][2]
    if (ix->dtype == GRANNE_HIP_I8) {
        granne_hip_index* mix = const_cast<granne_hip_index*>(ix);
        std::lock_guard<std::mutex> lk(mix->norm_mu);
        if (!mix->d_inv_norm) { // (made with the index, make_scan_norms: this is the path of an index that has no layers yet)
            float* dn = nullptr;
            HIP_TRY(hipMalloc((void**)&dn, (size_t)inv_norm_bytes(n)));
            hipLaunchKernelGGL(inv_norm_rows_kernel, dim3(grid_for(n * 8, 256)), dim3(256), 0, s, ix->d_elements, n, ix->row_stride, dn);
            hipLaunchKernelGGL(inv_gmax_kernel, dim3(grid_for(n_pad / 16 + 1, 256)), dim3(256), 0, s, (const float*)dn, n, dn + n_pad);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                (void)hipFree(dn);
                return fail(GRANNE_HIP_ERR_HIP, "inv_norm_rows_kernel failed");
            }
            mix->d_inv_norm = dn;
            mix->hbm_bytes += inv_norm_bytes(n);
        }
        P.inv_norm = mix->d_inv_norm;
        P.inv_gmax = mix->d_inv_norm + n_pad;
    }
    P.elements = ix->d_elements;
    P.n = n;
    P.row_bytes = ix->row_bytes;
    P.row_stride = ix->row_stride;
    P.dim = ix->dim;
    P.queries = (const uint8_t*)d_queries;
    P.nq = nq;
    P.kk = kk;
    P.per_range = per_range;
    P.part_ids = (uint64_t*)(scratch + o_pid);
    P.part_d = (float*)(scratch + o_pd);
    P.part_c = (uint32_t*)(scratch + o_pc);
    if (lds > 64u * 1024u) HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    P.tau_in = nullptr;
    P.share_hist = nullptr;
    P.qpad = nullptr;
    if (qpad_bytes) {
        hipLaunchKernelGGL(bf_pad_queries_kernel, dim3(grid_for(qpad_bytes, 256)), dim3(256), 0, s, (const uint8_t*)d_queries, nq, ix->dim,
                           ix->row_bytes, scratch + o_qpad);
        HIP_TRY(hipGetLastError());
        P.qpad = scratch + o_qpad;
    }
    uint64_t zeros[64];
    memset(zeros, 0, sizeof(zeros)); // the lists hold global ids already
    if (G >= 32) {
        // the priming pass (brute_force.h, bf_tau_kernel): the first range alone, cut into up to 64 sub-ranges so that the
        // whole chip scans it (1/64 of the scan proper, without its lists: the best score per sub-range and query)
        BruteParams Q = P;
        Q.n = per_range < n ? per_range : n;
        uint64_t Gs = (Q.n + tile - 1) / tile;
        if (Gs > 64) Gs = 64;
        Q.per_range = ((Q.n + Gs - 1) / Gs + tile - 1) / tile * tile;
        Gs = (Q.n + Q.per_range - 1) / Q.per_range;
        if (Gs >= kk) {
            const uint32_t plds = prime_qt ? prime_lds : lds, pqt = prime_qt ? prime_qt : qt, pthreads = prime_qt ? prime_threads : threads;
            if (plds > 64u * 1024u) HIP_TRY(hipFuncSetAttribute((const void*)fn_prime, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
            hipLaunchKernelGGL(fn_prime, dim3((nq + pqt - 1) / pqt, (uint32_t)Gs), dim3(pthreads), plds, s, Q);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(bf_tau_kernel, dim3(nq), dim3(64), 0, s, (const float*)P.part_d, (uint32_t)Gs, nq, kk,
                               (float*)(scratch + o_tau));
            HIP_TRY(hipGetLastError());
            P.tau_in = (const float*)(scratch + o_tau);
            // the ranges of a query tell each other what they have seen (brute_force.h, BfShare)
            HIP_TRY(hipMemsetAsync(scratch + o_share, 0, share_bytes, s));
            P.share_hist = (uint32_t*)(scratch + o_share);
        }
    }
    hipLaunchKernelGGL(fn, dim3((nq + qt - 1) / qt, (uint32_t)G), dim3(threads), lds, s, P);
    HIP_TRY(hipGetLastError());
    int rc = 0;
    if (G > 64) {
        // more lists than the merge has lanes: ranges 0..63 and 64.. are merged into two lists of their own, then those two
        const uint8_t* pi = (const uint8_t*)P.part_ids;
        const uint8_t* pd = (const uint8_t*)P.part_d;
        const uint8_t* pc = (const uint8_t*)P.part_c;
        const uint64_t si = (uint64_t)nq * kk * 8, sd = (uint64_t)nq * kk * 4, sc = (uint64_t)nq * 4;
        uint8_t* half = nullptr; // [2] x (ids, dists, counts)
        const size_t hi = 0, hd = 2 * si, hc = hd + 2 * sd, hbytes = hc + 2 * sc;
        HIP_TRY(hipMallocAsync((void**)&half, hbytes, s));
        Release release_half{half, s};
        for (uint32_t part = 0; part < 2 && rc == 0; ++part) {
            const uint32_t first = part * 64u, count = part == 0 ? 64u : (uint32_t)G - 64u;
            rc = merge_launch(pi + first * si, pd + first * sd, pc + first * sc, si, sd, sc, zeros, count, nq, kk,
                              (uint64_t*)(half + hi + part * si), (float*)(half + hd + part * sd),
                              (uint32_t*)(half + hc + part * sc), ix->device, stream);
        }
        if (rc == 0)
            rc = merge_launch(half + hi, half + hd, half + hc, si, sd, sc, zeros, 2, nq, kk, (uint64_t*)(scratch + o_mid),
                              (float*)(scratch + o_md), (uint32_t*)(scratch + o_mc), ix->device, stream);
    } else {
        rc = merge_launch((const uint8_t*)P.part_ids, (const uint8_t*)P.part_d, (const uint8_t*)P.part_c, (uint64_t)nq * kk * 8,
                          (uint64_t)nq * kk * 4, (uint64_t)nq * 4, zeros, (uint32_t)G, nq, kk, (uint64_t*)(scratch + o_mid),
                          (float*)(scratch + o_md), (uint32_t*)(scratch + o_mc), ix->device, stream);
    }
    if (rc) return rc;
    // the candidates' distances in the reference's own arithmetic, then the k best by (distance, id)
    const uint32_t pairs = nq * kk;
    hipLaunchKernelGGL(bf_narrow_ids_kernel, dim3((pairs + 255) / 256), dim3(256), 0, s, (const uint64_t*)(scratch + o_mid),
                       (const uint32_t*)(scratch + o_mc), nq, kk, (uint32_t*)(scratch + o_cand));
    HIP_TRY(hipGetLastError());
    rc = dists_launch(ix, d_queries, nullptr, kk, (const uint32_t*)(scratch + o_cand), (uint64_t)pairs, (float*)(scratch + o_ex),
                      nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(bf_final_kernel, dim3((pairs + 255) / 256), dim3(256), 0, s, (const uint32_t*)(scratch + o_cand),
                       (const float*)(scratch + o_ex), nq, kk, k, d_out_ids, d_out_dists, d_out_counts);
    HIP_TRY(hipGetLastError());
    return GRANNE_HIP_OK;
}

// host convenience over the scan: host buffers in and out
extern "C" int granne_hip_brute_force(const granne_hip_index* ix, const void* queries, uint32_t nq, uint32_t k,
                                      uint64_t* out_ids, float* out_dists, uint32_t* out_counts) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (nq == 0) return GRANNE_HIP_OK;
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (k == 0 || k > BF_KMAX) return fail(GRANNE_HIP_ERR_INVALID, "k must be in [1, %u]", BF_KMAX);
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    const size_t qb = (size_t)nq * ix->dim * elem_size(ix->dtype);
    const size_t o_ids = (qb + 255) & ~(size_t)255, o_d = o_ids + (size_t)nq * k * 8, o_c = o_d + (((size_t)nq * k * 4 + 15) & ~(size_t)15);
    const size_t total = o_c + (size_t)nq * 4;
    uint8_t* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, total));
    auto body = [&]() -> int {
        HIP_TRY(hipMemcpy(buf, queries, qb, hipMemcpyHostToDevice));
        int rc = granne_hip_brute_force_device(ix, buf, nq, k, (uint64_t*)(buf + o_ids), (float*)(buf + o_d), (uint32_t*)(buf + o_c), nullptr);
        if (rc) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out_ids, buf + o_ids, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_dists, buf + o_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_counts, buf + o_c, (size_t)nq * 4, hipMemcpyDeviceToHost));
        return GRANNE_HIP_OK;
    };
    const int rc = body();
    (void)hipDeviceSynchronize();
    (void)hipFree(buf);
    return rc;
}

extern "C" int granne_hip_synth_rows_device(float* d_out, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim,
                                            int device_id, void* stream) {
    if (!d_out && n) return fail(GRANNE_HIP_ERR_INVALID, "out is null");
    if (n == 0) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    hipLaunchKernelGGL(synth_rows_kernel, dim3(grid_for(n * dim, 256)), dim3(256), 0, (hipStream_t)stream, d_out, seed,
                       row0, n, dim);
    HIP_TRY(hipGetLastError());
    return GRANNE_HIP_OK;
}

// the per-shard top-k of a batch in ONE buffer: [nq*k u64 ids][nq*k f32 dists][nq u32 counts], padded to
// 16 bytes -- what one rank contributes to the all-gather of the partitioned mode
static inline size_t packed_dists_off(uint32_t nq, uint32_t k) { return (size_t)nq * k * 8; }
static inline size_t packed_counts_off(uint32_t nq, uint32_t k) { return (size_t)nq * k * 12; }
extern "C" uint64_t granne_hip_packed_topk_bytes(uint32_t nq, uint32_t k) {
    return ((uint64_t)nq * k * 12 + (uint64_t)nq * 4 + 15) & ~(uint64_t)15;
}

static int merge_launch(const uint8_t* ids, const uint8_t* dists, const uint8_t* counts, uint64_t ids_stride,
                        uint64_t dists_stride, uint64_t counts_stride, const uint64_t* shard_offsets, uint32_t n_shards,
                        uint32_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                        int device_id, void* stream) {
    if (nq == 0) return GRANNE_HIP_OK;
    if (!ids || !dists || !counts || !shard_offsets || !d_out_ids || !d_out_dists || !d_out_counts)
        return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (n_shards == 0 || n_shards > 64) return fail(GRANNE_HIP_ERR_INVALID, "n_shards must be in [1, 64]");
    if (k == 0 || (uint64_t)n_shards * k > 4096) return fail(GRANNE_HIP_ERR_INVALID, "n_shards * k must be in [1, 4096]");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    MergeParams P;
    P.ids = ids;
    P.dists = dists;
    P.counts = counts;
    P.ids_stride = ids_stride;
    P.dists_stride = dists_stride;
    P.counts_stride = counts_stride;
    for (uint32_t s = 0; s < 64; ++s) P.offsets[s] = s < n_shards ? shard_offsets[s] : 0;
    P.n_shards = n_shards;
    P.nq = nq;
    P.k = k;
    P.out_ids = d_out_ids;
    P.out_dists = d_out_dists;
    P.out_counts = d_out_counts;
    uint32_t C = n_shards * k;
    uint32_t lds = ((C * 4 + 7) & ~7u) + C * 8;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(64), lds, (hipStream_t)stream, P);
    HIP_TRY(hipGetLastError());
    return GRANNE_HIP_OK;
}

extern "C" int granne_hip_merge_topk_device(const uint64_t* d_ids, const float* d_dists, const uint32_t* d_counts,
                                            const uint64_t* shard_offsets, uint32_t n_shards, uint32_t nq, uint32_t k,
                                            uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                                            int device_id, void* stream) {
    return merge_launch((const uint8_t*)d_ids, (const uint8_t*)d_dists, (const uint8_t*)d_counts, (uint64_t)nq * k * 8,
                        (uint64_t)nq * k * 4, (uint64_t)nq * 4, shard_offsets, n_shards, nq, k, d_out_ids, d_out_dists,
                        d_out_counts, device_id, stream);
}

extern "C" int granne_hip_merge_topk_packed_device(const void* d_packed, const uint64_t* shard_offsets, uint32_t n_shards,
                                                   uint32_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_dists,
                                                   uint32_t* d_out_counts, int device_id, void* stream) {
    const uint8_t* base = (const uint8_t*)d_packed;
    const uint64_t stride = granne_hip_packed_topk_bytes(nq, k);
    if (!base && nq) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    return merge_launch(base, base + packed_dists_off(nq, k), base + packed_counts_off(nq, k), stride, stride, stride,
                        shard_offsets, n_shards, nq, k, d_out_ids, d_out_dists, d_out_counts, device_id, stream);
}

extern "C" int granne_hip_merge_topk_packed_strided_device(const void* d_packed, uint64_t stride_bytes,
                                                           const uint64_t* shard_offsets, uint32_t n_shards, uint32_t nq,
                                                           uint32_t k, uint64_t* d_out_ids, float* d_out_dists,
                                                           uint32_t* d_out_counts, int device_id, void* stream) {
    const uint8_t* base = (const uint8_t*)d_packed;
    if (!base && nq) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (stride_bytes < granne_hip_packed_topk_bytes(nq, k) || (stride_bytes & 3))
        return fail(GRANNE_HIP_ERR_INVALID, "stride must be a multiple of 4 and at least granne_hip_packed_topk_bytes");
    return merge_launch(base, base + packed_dists_off(nq, k), base + packed_counts_off(nq, k), stride_bytes, stride_bytes,
                        stride_bytes, shard_offsets, n_shards, nq, k, d_out_ids, d_out_dists, d_out_counts, device_id, stream);
}

extern "C" int granne_hip_search_batch_packed_device(const granne_hip_index* ix, const void* d_queries, uint32_t nq,
                                                     uint32_t max_search, uint32_t num_neighbors, void* d_packed,
                                                     uint32_t* d_status, void* stream) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (!d_packed && nq) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (num_neighbors == 0) return fail(GRANNE_HIP_ERR_INVALID, "num_neighbors must be > 0 for a packed result");
    uint8_t* base = (uint8_t*)d_packed;
    return granne_hip_search_batch_device(ix, d_queries, nq, max_search, num_neighbors, (uint64_t*)base,
                                          (float*)(base + packed_dists_off(nq, num_neighbors)),
                                          (uint32_t*)(base + packed_counts_off(nq, num_neighbors)), nullptr, d_status, stream);
}

// host conveniences -------------------------------------------------------------------------------
extern "C" int granne_hip_normalize_f32(float* rows, uint64_t n, uint32_t dim, int device_id) {
    if (!rows && n) return fail(GRANNE_HIP_ERR_INVALID, "rows is null");
    if (n == 0) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    float* d = nullptr;
    size_t bytes = (size_t)n * dim * 4;
    HIP_TRY(hipMalloc((void**)&d, bytes));
    int rc = GRANNE_HIP_OK;
    hipError_t e = hipMemcpy(d, rows, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = granne_hip_normalize_f32_device(d, n, dim, device_id, nullptr);
    if (e == hipSuccess && rc == 0) e = hipMemcpy(rows, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(GRANNE_HIP_ERR_HIP, "normalize: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int granne_hip_quantize_f32(const float* rows, int8_t* out, uint64_t n, uint32_t dim, int device_id) {
    if ((!rows || !out) && n) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    if (n == 0) return GRANNE_HIP_OK;
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", device_id);
    float* d = nullptr;
    int8_t* o = nullptr;
    size_t bytes = (size_t)n * dim * 4;
    HIP_TRY(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMalloc((void**)&o, (size_t)n * dim);
    int rc = GRANNE_HIP_OK;
    if (e == hipSuccess) e = hipMemcpy(d, rows, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = granne_hip_quantize_f32_device(d, o, n, dim, device_id, nullptr);
    if (e == hipSuccess && rc == 0) e = hipMemcpy(out, o, (size_t)n * dim, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (o) (void)hipFree(o);
    if (e != hipSuccess) return fail(GRANNE_HIP_ERR_HIP, "quantize: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int granne_hip_dist_pairs(const granne_hip_index* ix, const void* queries, uint32_t nq, const uint32_t* qidx,
                                     const uint32_t* ids, uint64_t n_pairs, float* out) {
    if (!ix) return fail(GRANNE_HIP_ERR_INVALID, "index is null");
    if (n_pairs == 0) return GRANNE_HIP_OK;
    if (!queries || !qidx || !ids || !out) return fail(GRANNE_HIP_ERR_INVALID, "null buffer");
    for (uint64_t i = 0; i < n_pairs; ++i) {
        if (qidx[i] >= nq) return fail(GRANNE_HIP_ERR_INVALID, "query index out of range");
        if (ids[i] >= ix->n_elements) return fail(GRANNE_HIP_ERR_INVALID, "element id out of range");
    }
    DeviceGuard g(ix->device);
    if (!g.ok) return fail(GRANNE_HIP_ERR_NO_DEVICE, "cannot select HIP device %d", ix->device);
    size_t qb = (size_t)nq * ix->dim * elem_size(ix->dtype);
    uint8_t *dq = nullptr, *dqi = nullptr, *did = nullptr, *dout = nullptr;
    auto freeall = [&]() {
        if (dq) (void)hipFree(dq);
        if (dqi) (void)hipFree(dqi);
        if (did) (void)hipFree(did);
        if (dout) (void)hipFree(dout);
    };
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&dq, qb));
        HIP_TRY(hipMalloc((void**)&dqi, n_pairs * 4));
        HIP_TRY(hipMalloc((void**)&did, n_pairs * 4));
        HIP_TRY(hipMalloc((void**)&dout, n_pairs * 4));
        HIP_TRY(hipMemcpy(dq, queries, qb, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dqi, qidx, n_pairs * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(did, ids, n_pairs * 4, hipMemcpyHostToDevice));
        int r = granne_hip_dist_pairs_device(ix, dq, (const uint32_t*)dqi, (const uint32_t*)did, n_pairs, (float*)dout, nullptr);
        if (r) return r;
        HIP_TRY(hipMemcpy(out, dout, n_pairs * 4, hipMemcpyDeviceToHost));
        return GRANNE_HIP_OK;
    };
    int rc = body();
    freeall();
    return rc;
}

// ------------------------------------------------------------------------------------------------
// GranneBuilder on the GPU
// ------------------------------------------------------------------------------------------------
#include "builder_host.h"

// ------------------------------------------------------------------------------------------------
// Granne::reorder on the GPU
// ------------------------------------------------------------------------------------------------
#include "reorder_host.h"

// ------------------------------------------------------------------------------------------------
// granne's file formats
// ------------------------------------------------------------------------------------------------
#include "fileformat_host.h"

// ------------------------------------------------------------------------------------------------
// partitioned indexes driven by one host process
// ------------------------------------------------------------------------------------------------
#include "sharded_host.h"
