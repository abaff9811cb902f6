/*
 * granne_hip.h -- C ABI of libgranne_hip.so, the MI355X (gfx950) search path for granne.
 *
 * This is the drop-in boundary for ONE path of the reference: `Granne::search`
 * (/root/reference/src/index/mod.rs:140-150) and everything below it (find_entrypoint :984-997,
 * search_for_neighbors :999-1037, MaxSizeHeap src/max_size_heap.rs:5-45, the angular distances
 * src/elements/angular.rs:63-74 / angular_int.rs:47-60 over src/math.rs:5-52,59-89). The
 * reference has no FFI for this path (it is generic Rust over the ElementContainer/Dist traits,
 * src/elements/mod.rs:17-70); these entry points are what a `GpuGranne` Rust wrapper binds --
 * INTEGRATION.md shows the `extern "C"` block and the wrapper.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns GRANNE_HIP_OK (0) or a negative error code and never throws or
 *     aborts; granne_hip_last_error() returns a thread-local message for the last failure;
 *   - the reference's panics on this path become error codes (max_search == 0,
 *     src/index/mod.rs:1019 -> GRANNE_HIP_ERR_INVALID);
 *   - results are bit-identical to the reference's CPU search on the same index and queries:
 *     ids exact, distances exact (same f32 operation order), order ascending by (dist, id);
 *   - "host" functions take host pointers and copy; "_device" functions take device pointers
 *     (hipMalloc'd on the index's device) plus a hipStream_t passed as void* and are
 *     asynchronous with respect to the host.
 */
#ifndef GRANNE_HIP_H
#define GRANNE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRANNE_HIP_ABI_VERSION 3

/* element scalar types: granne::angular::Vectors (f32, rows normalised) and
 * granne::angular_int::Vectors (i8, rows quantised) -- src/elements/angular.rs:53,
 * src/elements/angular_int.rs:17 */
enum { GRANNE_HIP_F32 = 0, GRANNE_HIP_I8 = 1 };

enum {
    GRANNE_HIP_OK = 0,
    GRANNE_HIP_ERR_INVALID = -1,  /* bad argument (incl. the reference's panic conditions) */
    GRANNE_HIP_ERR_HIP = -2,      /* a HIP runtime call failed; see granne_hip_last_error() */
    GRANNE_HIP_ERR_NO_DEVICE = -3,/* no gfx950 device / kernels not loadable: never falls back to CPU */
    GRANNE_HIP_ERR_OVERFLOW = -4, /* exact-search scratch exhausted; raise GRANNE_HIP_OPT_SLOW_* */
    GRANNE_HIP_ERR_IO = -5        /* malformed index / elements file */
};

/* NeighborId padding value of fixed-width adjacency rows (src/index/mod.rs:27-28) */
#define GRANNE_HIP_UNUSED 0xFFFFFFFFu

typedef struct granne_hip_index granne_hip_index; /* opaque; owns HBM copies of elements + graph */

const char* granne_hip_last_error(void);
int granne_hip_abi_version(void);
int granne_hip_device_count(int* out_count);

/* ---- index lifetime -------------------------------------------------------------------------
 * granne_hip_index_create: replaces building a `Granne` from a builder's layers
 * (GranneBuilder::get_index, src/index/mod.rs:483-488 -> Layers::FixWidth). Layers are
 * prefix-nested (layer i holds nodes 0..layer_len[i]); layer_rows[i] is a row-major
 * [layer_len[i]][layer_width[i]] u32 matrix, valid ids first, padded with GRANNE_HIP_UNUSED
 * (FixedWidthSliceVector<u32>, src/slice_vector/mod.rs:42-45, 344-356). `elements` is the
 * payload of an angular(_int)::Vectors file: row-major [n_elements][dim] scalars
 * (src/slice_vector/mod.rs:213-221). All inputs are host memory and are copied to HBM; the
 * caller may free/unmap them afterwards. n_layers == 0 is a valid (empty) index.
 * Capacity: the reference allows 2^32 - 2 elements per index (src/index/mod.rs:27-28, 420); the
 * register walkers' list keys carry 31-bit ids, so an index of more than 2^31 elements is walked
 * by the exact (slower) walker. That is a limit PER INDEX: the shards of a partitioned index
 * (granne_hip_sharded_*) walk local ids and add their offsets afterwards -- and 2^31 rows of 100
 * int8 dimensions are 275 GB before their graph, more than one device holds.               */
int granne_hip_index_create(granne_hip_index** out, const void* elements, uint64_t n_elements,
                            uint32_t dim, int dtype, uint32_t n_layers, const uint64_t* layer_len,
                            const uint32_t* const* layer_rows, const uint32_t* layer_width,
                            int device_id);

/* Same, from the decoded form of the on-disk layers (MultiSetVector,
 * src/slice_vector/set_vector.rs:8-115; Layers::Compressed, src/index/io.rs:72-87): per layer a
 * CSR pair offsets[layer_len+1] (u64) / ids[offsets[layer_len]] (u32, any order).          */
int granne_hip_index_create_csr(granne_hip_index** out, const void* elements, uint64_t n_elements,
                                uint32_t dim, int dtype, uint32_t n_layers, const uint64_t* layer_len,
                                const uint64_t* const* layer_offsets, const uint32_t* const* layer_ids,
                                int device_id);

/* Same as granne_hip_index_create but `elements` and every layer_rows[i] are DEVICE pointers on
 * device_id (e.g. produced on the GPU); they are re-laid-out into the index's own HBM buffers. */
int granne_hip_index_create_device(granne_hip_index** out, const void* d_elements, uint64_t n_elements,
                                   uint32_t dim, int dtype, uint32_t n_layers, const uint64_t* layer_len,
                                   const uint32_t* const* d_layer_rows, const uint32_t* layer_width,
                                   int device_id, void* stream);

void granne_hip_index_destroy(granne_hip_index* index);

/* Index trait (src/index/mod.rs:54-71) */
uint64_t granne_hip_index_len(const granne_hip_index* index);        /* bottom layer length */
uint32_t granne_hip_index_num_layers(const granne_hip_index* index);
uint64_t granne_hip_index_layer_len(const granne_hip_index* index, uint32_t layer);
uint32_t granne_hip_index_dim(const granne_hip_index* index);
int granne_hip_index_dtype(const granne_hip_index* index);
int granne_hip_index_device(const granne_hip_index* index);
uint64_t granne_hip_index_hbm_bytes(const granne_hip_index* index);
/* Index::get_neighbors(index, layer) (src/index/mod.rs:64): copies up to cap ids, returns count
 * in *out_count. */
int granne_hip_index_get_neighbors(const granne_hip_index* index, uint64_t node, uint32_t layer,
                                   uint32_t* out_ids, uint32_t cap, uint32_t* out_count);
/* Granne::get_element (src/index/mod.rs:153-155): copies dim scalars of element `idx`. */
int granne_hip_index_get_element(const granne_hip_index* index, uint64_t idx, void* out);

/* ---- search -----------------------------------------------------------------------------------
 * granne_hip_search_batch: `nq` independent `Granne::search(&query, max_search, num_neighbors)`
 * calls (src/index/mod.rs:140-150). queries: row-major [nq][dim] scalars of the index's dtype,
 * ALREADY prepared the way Vector::from does it (normalised f32: src/elements/angular.rs:55-61;
 * quantised i8: src/elements/angular_int.rs:19-45) -- granne_hip_normalize_f32 /
 * granne_hip_quantize_f32 below do that on the device, bit-exactly.
 * Outputs (row-major [nq][num_neighbors], may be NULL where noted):
 *   out_ids    u64 (usize)  neighbor ids, ascending by (dist, id); unused slots = UINT64_MAX
 *   out_dists  f32          the distances; unused slots = +inf
 *   out_counts u32 [nq]     results per query = min(num_neighbors, max_search, #reachable)
 *   out_stats  u64 [nq][3]  (optional) per query: n_dist, n_expand, n_adj -- the counters the
 *                           roofline uses. [1] n_expand = get_neighbors calls
 *                           (src/index/mod.rs:1025) and [2] n_adj = neighbor ids those calls returned
 *                           are the reference's counts in every mode. [0] n_dist = element rows
 *                           EVALUATED: by default the walkers keep no visited set and evaluate a
 *                           neighbor they have seen before again (same result, ~3 % more rows on
 *                           uniform data), so [0] >= the reference's dist_to_element calls
 *                           (mod.rs:1012,1027); with the exact visited set switched on
 *                           (GRANNE_HIP_OPT_VISITED16 = 1..3) [0] IS the reference's count.
 * Thread-safe on a shared index.                                                              */
int granne_hip_search_batch(const granne_hip_index* index, const void* queries, uint32_t nq,
                            uint32_t max_search, uint32_t num_neighbors, uint64_t* out_ids,
                            float* out_dists, uint32_t* out_counts, uint64_t* out_stats);

/* Device-resident variant: all pointers are device memory on the index's device; the work is
 * enqueued on `stream` (a hipStream_t) and NOT synchronised. d_status (u32[4], optional, zeroed by
 * the caller): [0] is set to 1 if any query exhausted the exact-search scratch
 * (GRANNE_HIP_ERR_OVERFLOW); [1] accumulates the queries served by the exact global-memory
 * walker; [2] accumulates the walks whose LDS visited table filled and that continued with a
 * global overflow table (GRANNE_HIP_OPT_OVERFLOW_SLOTS).                                          */
int granne_hip_search_batch_device(const granne_hip_index* index, const void* d_queries, uint32_t nq,
                                   uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                   float* d_out_dists, uint32_t* d_out_counts, uint64_t* d_out_stats,
                                   uint32_t* d_status, void* stream);

/* n_batches batches of nq queries each through ONE kernel launch (a grid of n_batches x nq walkers): the
 * caller-side loop / par_iter over Granne::search (src/index/mod.rs:140-150) for a host that has several
 * batches ready. One launch of 1024 walks occupies one of a SIMD's wave slots and lasts as long as its
 * slowest walk; here walks of later batches take the place of finished ones inside the same launch, so ONE
 * stream and default HIP settings reach what otherwise takes 5-10 streams in flight.
 * d_queries / d_out_*: HOST arrays of n_batches DEVICE pointers (batch b: [nq][dim] queries, [nq][k] ids
 * and dists, [nq] counts, [nq][3] stats); d_out_stats may be NULL, and so may its entries. d_status as in
 * granne_hip_search_batch_device (shared by all batches). Results are those of n_batches separate calls,
 * bit for bit. More than 32 batches go out as several launches on `stream`. Asynchronous.              */
int granne_hip_search_batches_device(const granne_hip_index* index, uint32_t n_batches,
                                     const void* const* d_queries, uint32_t nq, uint32_t max_search,
                                     uint32_t num_neighbors, uint64_t* const* d_out_ids,
                                     float* const* d_out_dists, uint32_t* const* d_out_counts,
                                     uint64_t* const* d_out_stats, uint32_t* d_status, void* stream);

/* granne_hip_search_batch_device in two halves, for a host whose batches arrive one at a time: begin orders the
 * search after what `stream` holds, runs it on one of the index's own streams and returns a ticket at once; end
 * makes `stream` wait for that search. Up to GRANNE_HIP_OPT_SEARCH_DEPTH batches of an index (default
 * GRANNE_HIP_SEARCH_DEPTH, at most GRANNE_HIP_SEARCH_DEPTH_MAX) may be begun and not yet ended; tickets are ended in any
 * order and a begin takes any free place; their walks share the chip. (HIP maps streams onto GPU_MAX_HW_QUEUES = 4
 * hardware queues unless the process sets that variable before its first HIP call: a depth beyond 3 pays with more.) Arguments as granne_hip_search_batch_device;
 * the buffers of a batch must stay untouched between its begin and the completion of what follows its end.    */
#define GRANNE_HIP_SEARCH_DEPTH 3      /* the default depth: HIP maps streams onto 4 hardware queues by default, the caller's + these */
#define GRANNE_HIP_SEARCH_DEPTH_MAX 16 /* GRANNE_HIP_OPT_SEARCH_DEPTH goes up to this */
int granne_hip_search_begin_device(const granne_hip_index* index, const void* d_queries, uint32_t nq,
                                   uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                   float* d_out_dists, uint32_t* d_out_counts, uint64_t* d_out_stats,
                                   uint32_t* d_status, void* stream, uint64_t* out_ticket);
int granne_hip_search_end_device(const granne_hip_index* index, uint64_t ticket, void* stream);

/* granne_hip_search_batch_device, and two optional hipEvent_t recorded on `stream` immediately
 * before and after the dispatch of the search kernel itself (the call also enqueues a small
 * scratch memset before it and the slow-path kernel after it): lets a caller time the dominant
 * kernel alone, the way rocprofv3 --kernel-trace sees it. No reference counterpart (measurement). */
int granne_hip_search_batch_device_timed(const granne_hip_index* index, const void* d_queries, uint32_t nq,
                                         uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                         float* d_out_dists, uint32_t* d_out_counts, uint64_t* d_out_stats,
                                         uint32_t* d_status, void* stream, void* ev_before, void* ev_after);

/* hipEvent_t helpers for the call above, so that a host language without HIP bindings can time it:
 * create / destroy an event, milliseconds between two recorded and completed events. */
int granne_hip_event_create(void** out_event);
void granne_hip_event_destroy(void* event);
int granne_hip_event_elapsed_ms(void* before, void* after, float* out_ms);

/* Device memory and streams for a host language without HIP bindings of its own (INTEGRATION.md's safe Rust wrappers of
 * the _device entry points are built on these): allocate / free on a device, copy host <-> device ordered on a stream
 * (pageable host memory: the call returns when the copy is staged, the data is there when the stream reaches it), create
 * / destroy / wait for a hipStream_t (returned as void*: what every `stream` parameter of this header takes). */
int granne_hip_device_malloc(void** out_ptr, uint64_t bytes, int device_id);
int granne_hip_device_free(void* ptr, int device_id);
int granne_hip_copy_to_device(void* d_dst, const void* src, uint64_t bytes, int device_id, void* stream);
int granne_hip_copy_to_host(void* dst, const void* d_src, uint64_t bytes, int device_id, void* stream);
int granne_hip_stream_create(void** out_stream, int device_id);
int granne_hip_stream_destroy(void* stream, int device_id);
int granne_hip_stream_synchronize(void* stream, int device_id);

/* Granne::search for one query (host pointers); *out_count results written. */
int granne_hip_search(const granne_hip_index* index, const void* query, uint32_t max_search,
                      uint32_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_count);

/* ---- Granne::reorder (src/index/reorder.rs) ----------------------------------------------------- */
/* Granne::reorder(&mut self) -> Vec<usize> (reorder.rs:59-85): computes the entry-point-trail order
 * (compute_order :135-175, find_entrypoint_trail :180-208) on the device, rewrites every layer
 * through the reverse mapping (reorder_layers :210-281; neighbor sets come out sorted, as the
 * reference's MultiSetVector stores them) and permutes the elements. out_order (optional, host,
 * [len]) receives the permutation: out_order[i] == j means the element with idx j moved to idx i.
 * The index must not be searched concurrently (the reference takes &mut self). Fewer than two
 * layers, or len() != number of elements, are the reference's panics: GRANNE_HIP_ERR_INVALID. */
int granne_hip_index_reorder(granne_hip_index* index, uint64_t* out_order);
/* Granne::reorder_by_keys(&mut self, keys) (reorder.rs:88-133) for u64 keys (host, [len]): a
 * layer-preserving sort by (key, idx). */
int granne_hip_index_reorder_by_keys(granne_hip_index* index, const uint64_t* keys, uint64_t* out_order);

/* ---- element preparation and the Dist / ElementContainer operators on device ------------------ */
/* angular::Vector::from(Vec<f32>) over n rows, in place (src/math.rs:123-150). Device pointers. */
int granne_hip_normalize_f32_device(float* d_rows, uint64_t n, uint32_t dim, int device_id, void* stream);
/* angular_int::Vector::quantize over n rows (src/elements/angular_int.rs:27-45). */
int granne_hip_quantize_f32_device(const float* d_rows, int8_t* d_out, uint64_t n, uint32_t dim,
                                   int device_id, void* stream);
/* ElementContainer::dist_to_element for explicit (query, element id) pairs
 * (src/elements/dense_vector.rs:149-151): d_out[i] = dist(elements[d_ids[i]], queries[d_qidx[i]]). */
int granne_hip_dist_pairs_device(const granne_hip_index* index, const void* d_queries,
                                 const uint32_t* d_qidx, const uint32_t* d_ids, uint64_t n_pairs,
                                 float* d_out, void* stream);
/* ElementContainer::dists(&self, element, indices) -> Vec<f32> (src/elements/mod.rs:35-39,
 * src/elements/dense_vector.rs:157-163), batched over nq elements with m indices each:
 * d_out[q*m + j] = dist(elements[d_ids[q*m + j]], queries[q]). An id >= len yields +inf and is
 * counted in *d_status (optional device u32, zeroed by the caller). Same arithmetic, bit for bit,
 * as the walk. */
int granne_hip_dists_device(const granne_hip_index* index, const void* d_queries, uint32_t nq,
                            const uint32_t* d_ids, uint32_t m, float* d_out, uint32_t* d_status,
                            void* stream);
/* Host conveniences over the ones above (copy in, run, copy out). */
int granne_hip_normalize_f32(float* rows, uint64_t n, uint32_t dim, int device_id);
int granne_hip_quantize_f32(const float* rows, int8_t* out, uint64_t n, uint32_t dim, int device_id);
int granne_hip_dist_pairs(const granne_hip_index* index, const void* queries, uint32_t nq,
                          const uint32_t* qidx, const uint32_t* ids, uint64_t n_pairs, float* out);

/* Synthetic element rows: component (row, col) = uniform [-0.5, 0.5) from a
 * splitmix64 counter, the distribution of src/test_helper.rs:3-6. Device pointer out [n][dim]. */
int granne_hip_synth_rows_device(float* d_out, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim,
                                 int device_id, void* stream);

/* ---- granne's files ------------------------------------------------------------------------------
 * granne_hip_index_load: Granne::from_bytes(index, Vectors::from_bytes(elements))
 * (src/index/mod.rs:106-113, src/elements/dense_vector.rs:49-51). `index_bytes` is an index file
 * as written by Index::write_index (src/index/io.rs:11-70: 1 KiB "granne"+JSON header, then one
 * compressed MultiSetVector blob per layer); `elements_bytes` is a Vectors file ([u64 dim][raw
 * scalars], src/slice_vector/mod.rs:213-221, 460-466). Every neighbor list is decoded once on
 * the host (stream-vbyte / raw, delta, src/slice_vector/set_vector.rs:91-162) and uploaded; the
 * buffers may be unmapped afterwards. _load_files maps the two files itself
 * (Granne::from_file, src/index/mod.rs:122-135).                                              */
int granne_hip_index_load(granne_hip_index** out, const void* index_bytes, uint64_t index_len,
                          const void* elements_bytes, uint64_t elements_len, int dtype, int device_id);
int granne_hip_index_load_files(granne_hip_index** out, const char* index_path, const char* elements_path,
                                int dtype, int device_id);
/* Index::write_index for fixed-width layers (what a builder holds) and Writeable::write for
 * Vectors: files the reference (and granne_hip_index_load) reads. */
int granne_hip_write_index_file(const char* path, uint32_t n_layers, const uint64_t* layer_len,
                                const uint32_t* const* layer_rows, const uint32_t* layer_width);
int granne_hip_write_elements_file(const char* path, const void* elements, uint64_t n_elements, uint32_t dim,
                                   int dtype);
/* save_index / save_elements of a device-resident index (py/src/lib.rs:318-343); either path may
 * be NULL. */
int granne_hip_index_save(const granne_hip_index* index, const char* index_path, const char* elements_path);
/* Index::write_index<B: Write + Seek> (src/index/mod.rs:67-70, src/index/io.rs:11-70) for a device-resident index: the
 * bytes of the index file in a buffer of the library's (released with granne_hip_bytes_free); a Rust host hands them to
 * its writer. */
int granne_hip_index_encode(const granne_hip_index* index, void** out_bytes, uint64_t* out_len);
void granne_hip_bytes_free(void* bytes);
/* Host-only inspection of an index file (no device involved): layer count, nodes and ids per
 * layer (arrays of `cap` entries), and the decoded CSR form of one layer
 * (out_offsets[layer_len + 1], out_ids[ids of that layer], ascending within a node). */
int granne_hip_index_file_info(const void* index_bytes, uint64_t index_len, uint32_t* out_n_layers,
                               uint64_t* out_layer_len, uint64_t* out_layer_ids, uint32_t cap);
int granne_hip_index_file_decode_layer(const void* index_bytes, uint64_t index_len, uint32_t layer,
                                       uint64_t* out_offsets, uint32_t* out_ids);

/* ---- partitioned indexes: merge of per-shard results --------------------------------------------
 * The element set partitions into independent indexes (how the reference's own shard helper is
 * meant to be used, src/elements/embeddings/parsing.rs:63-100). Every shard answers the same
 * query batch; this call merges the n_shards x k candidates of each query into the k best by
 * (dist, global id). d_ids/d_dists: [n_shards][nq][k] (the layout an all-gather of per-rank
 * [nq][k] results produces), d_counts: [n_shards][nq]; shard_offsets (HOST array, n_shards
 * entries) are added to the local ids. n_shards <= 64, n_shards * k <= 4096.
 * A shard's list is normally what a search returned: ascending by (dist, id), its count[q] valid entries first. Lists in
 * any other order are accepted and give the same answer (the kernel checks each list and sorts one that is out of
 * order before its k-way merge) -- sorted lists are just the fast case.                                            */
int granne_hip_merge_topk_device(const uint64_t* d_ids, const float* d_dists, const uint32_t* d_counts,
                                 const uint64_t* shard_offsets, uint32_t n_shards, uint32_t nq, uint32_t k,
                                 uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                                 int device_id, void* stream);

/* Exact k nearest elements of every query by a scan of ALL elements on the matrix cores: the many-to-many form of
 * ElementContainer::dists (src/elements/mod.rs:35-39, src/elements/dense_vector.rs:157-163) -- the recall ground truth
 * next to the graph walk, and the one contraction-shaped piece of the element side (f32 rows as two bf16 pieces per
 * component on v_mfma_f32_32x32x16_bf16, int8 rows on v_mfma_i32_32x32x32_i8; granne_amd/csrc/brute_force.h).
 * Candidates are SELECTED by the MFMA score (f32: to ~4e-5 of a unit-vector dot); the returned
 * distances are recomputed in the reference's arithmetic (bit-exact for the returned ids) and the results are ordered
 * ascending by (distance, id). The id set can differ from a scalar scan only between elements whose distances to the
 * query are within the score's rounding of each other at the boundary of the selection: min(k + 6, 16)
 * candidates per query are selected and re-ranked, so k <= 10 has six spare candidates, k = 16 none. k <= 16; f32 rows
 * and int8 rows of any dimension (f32 beyond 256 dims and int8 beyond 128 are walked in chunks of 128 components /
 * bytes, at about half the rate per operation). queries: dense [nq][dim], prepared like the elements (the Python and
 * C++ wrappers check the width). Asynchronous on `stream`. */
int granne_hip_brute_force_device(const granne_hip_index* index, const void* d_queries, uint32_t nq, uint32_t k,
                                  uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, void* stream);
/* the same with host buffers in and out (synchronous) */
int granne_hip_brute_force(const granne_hip_index* index, const void* queries, uint32_t nq, uint32_t k,
                           uint64_t* out_ids, float* out_dists, uint32_t* out_counts);

/* The per-shard top-k of one batch as ONE buffer -- what a rank contributes to the single exchange
 * step of the partitioned mode (one all-gather, or one peer copy):
 *   [nq*k u64 local ids][nq*k f32 dists][nq u32 counts], padded to 16 bytes.                      */
uint64_t granne_hip_packed_topk_bytes(uint32_t nq, uint32_t k);
/* granne_hip_search_batch_device writing that buffer (d_packed: granne_hip_packed_topk_bytes). */
int granne_hip_search_batch_packed_device(const granne_hip_index* index, const void* d_queries, uint32_t nq,
                                          uint32_t max_search, uint32_t num_neighbors, void* d_packed,
                                          uint32_t* d_status, void* stream);
/* granne_hip_merge_topk_device over n_shards such buffers laid end to end (the all-gather's output). */
int granne_hip_merge_topk_packed_device(const void* d_packed, const uint64_t* shard_offsets, uint32_t n_shards,
                                        uint32_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_dists,
                                        uint32_t* d_out_counts, int device_id, void* stream);
/* The same with shard s's buffer at d_packed + s * stride_bytes (stride_bytes >= granne_hip_packed_topk_bytes, a
 * multiple of 4): a rank can append words of its own to each packed buffer -- granne_amd/sharded.py sends every
 * shard's four status words through the same all-gather, so that all ranks see a shard that ran out of scratch. */
int granne_hip_merge_topk_packed_strided_device(const void* d_packed, uint64_t stride_bytes,
                                                const uint64_t* shard_offsets, uint32_t n_shards, uint32_t nq,
                                                uint32_t k, uint64_t* d_out_ids, float* d_out_dists,
                                                uint32_t* d_out_counts, int device_id, void* stream);

/* ---- a partitioned index driven by one host process ----------------------------------------------
 * SURVEY.md 8b's `device_ids / n_devices / partitioned`: shard s is a granne_hip_index of its own
 * (created on whatever device it should live on: granne_hip_index_create*, _load*, a builder's
 * get_index) over the elements [id_offsets[s], id_offsets[s] + len_s) of the whole set, with local ids
 * (the split of src/elements/embeddings/parsing.rs:63-100). A search = Granne::search on every shard +
 * the merge: each shard searches the same batch on its own device and stream and writes its packed
 * top-k + status words into the gather buffer on shard 0's device -- directly when it lives there, by
 * a peer copy over xGMI otherwise, or (GRANNE_HIP_SHARDED_OPT_EXCHANGE) through ONE ncclAllGather over
 * the shard devices -- then merge by (dist, global id). Results equal the per-shard searches merged on
 * the host, bit for bit. The handle BORROWS the shard indexes (destroy them after it).
 * n_shards <= 64, n_shards * num_neighbors <= 4096.                                               */
typedef struct granne_hip_sharded granne_hip_sharded;
int granne_hip_sharded_create(granne_hip_sharded** out, granne_hip_index* const* shards,
                              const uint64_t* id_offsets, uint32_t n_shards);
/* The same with the EXCHANGE GROUP of every shard spelled out (groups[s]: any labels; NULL = one group per HIP device =
 * granne_hip_sharded_create). A group is what the exchange step treats as one device: its first shard fetches a batch's
 * queries for all of them, its results travel to the merge device together, one event releases its buffers. All shards
 * of a group live on one device; several groups MAY share a device -- every branch of the multi-device exchange then
 * runs on a single GPU (tests/test_gpu_sharded.py), with device-local copies. The all-gather exchange needs one group
 * per device. */
int granne_hip_sharded_create_grouped(granne_hip_sharded** out, granne_hip_index* const* shards,
                                      const uint64_t* id_offsets, uint32_t n_shards, const uint32_t* groups);
void granne_hip_sharded_destroy(granne_hip_sharded* sharded);
uint32_t granne_hip_sharded_num_shards(const granne_hip_sharded* sharded);
/* shard s of the handle (borrowed) and the first global id it holds */
granne_hip_index* granne_hip_sharded_shard(const granne_hip_sharded* sharded, uint32_t shard);
uint64_t granne_hip_sharded_shard_offset(const granne_hip_sharded* sharded, uint32_t shard);
uint64_t granne_hip_sharded_len(const granne_hip_sharded* sharded);
int granne_hip_sharded_device(const granne_hip_sharded* sharded); /* where queries / results of the _device calls live */

enum {
    GRANNE_HIP_SHARDED_OPT_DEPTH = 1,    /* batches that can be in flight at once (begin without end): 1..8, default 2 */
    GRANNE_HIP_SHARDED_OPT_EXCHANGE = 2  /* how the per-shard top-k reach the merge device */
};
enum {
    GRANNE_HIP_SHARDED_EXCHANGE_PEER = 0, /* hipMemcpyPeerAsync per remote shard (default; nothing at all for local shards) */
    GRANNE_HIP_SHARDED_EXCHANGE_RCCL = 1  /* one grouped, in-place ncclAllGather over a communicator of the shard devices;
                                             librccl is loaded with dlopen (an already loaded one first, then
                                             $GRANNE_HIP_RCCL_LIB, then librccl.so); needs the shards in device order, the
                                             same number on every device */
};
int granne_hip_sharded_set_option(granne_hip_sharded* sharded, int option, uint64_t value);
int granne_hip_sharded_get_option(const granne_hip_sharded* sharded, int option, uint64_t* value);

/* Device buffers (on granne_hip_sharded_device), stream-ordered, not synchronised: the batch is ordered after
 * what `stream` holds, `stream` continues after the merged result is written. d_status (u32[4], optional,
 * zeroed by the caller) receives the shards' status words folded: [0] |= a shard ran out of exact-search
 * scratch, [1] += queries served by the exact walker, [2] += walks that borrowed an overflow table.        */
int granne_hip_sharded_search_batch_device(granne_hip_sharded* sharded, const void* d_queries, uint32_t nq,
                                           uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                           float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_status,
                                           void* stream);
/* The same in two halves, so that batches overlap: begin enqueues the shard searches, the exchange and the
 * merge on the handle's own streams (ordered after what `stream` holds now) and returns a ticket; end makes
 * `stream` wait for that batch's merged result. Up to GRANNE_HIP_SHARDED_OPT_DEPTH batches may be begun and
 * not yet ended: batch b+1 is searched while batch b is exchanged and merged. The buffers of a batch must
 * stay untouched between its begin and the completion of what follows its end on `stream`.               */
int granne_hip_sharded_begin_device(granne_hip_sharded* sharded, const void* d_queries, uint32_t nq,
                                    uint32_t max_search, uint32_t num_neighbors, uint64_t* d_out_ids,
                                    float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_status,
                                    void* stream, uint64_t* out_ticket);
int granne_hip_sharded_end_device(granne_hip_sharded* sharded, uint64_t ticket, void* stream);

/* HOST buffers (synchronous; pinned staging inside): queries [n_batches][nq][dim], out_ids (global ids,
 * ascending (dist, id)) / out_dists [n_batches][nq][num_neighbors], out_counts [n_batches][nq]; batches
 * are pipelined GRANNE_HIP_SHARDED_OPT_DEPTH deep (batch b+1's upload and search overlap batch b's
 * exchange, merge and download). The host-pointer calls of one handle run one at a time, and share the
 * handle's depth with whatever device-pointer batches other threads have in flight.                      */
int granne_hip_sharded_search_batches(granne_hip_sharded* sharded, const void* queries, uint32_t n_batches,
                                      uint32_t nq, uint32_t max_search, uint32_t num_neighbors,
                                      uint64_t* out_ids, float* out_dists, uint32_t* out_counts);
int granne_hip_sharded_search_batch(granne_hip_sharded* sharded, const void* queries, uint32_t nq,
                                    uint32_t max_search, uint32_t num_neighbors, uint64_t* out_ids,
                                    float* out_dists, uint32_t* out_counts);
int granne_hip_sharded_search(granne_hip_sharded* sharded, const void* query, uint32_t max_search,
                              uint32_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_count);

/* ---- GranneBuilder on the GPU ------------------------------------------------------------------
 * Mirrors GranneBuilder / BuildConfig / Builder (src/index/mod.rs:198-531): the same layer
 * pyramid (compute_num_elements_in_layer :634-643), the same per-element work (index_element
 * :805-846 = entry search + search_for_neighbors at the build max_search -- the SAME kernel
 * Granne::search runs -- + select_neighbors :849-883 + connect_nodes/add_and_limit_neighbors
 * :898-959 + the final per-row limit :795-797), with every distance bit-exact. What differs is
 * the SCHEDULE: the reference inserts with a rayon par_iter over per-node RwLocks (:757-782,
 * nondeterministic); this builder inserts in batches -- every member of a batch searches the
 * graph as it stood when the batch began, then the batch's link updates are applied in element
 * order. Batch = clamp(nodes_in_graph / batch_div, 1, batch_max). The result is deterministic. */
typedef struct granne_hip_builder granne_hip_builder;

typedef struct {
    float layer_multiplier;         /* 15.0  (BuildConfig::default, src/index/mod.rs:220-231) */
    uint64_t expected_num_elements; /* 0 = None */
    uint32_t num_neighbors;         /* 30; at most 63 on the GPU */
    uint32_t max_search;            /* 200; at most 256 stays on the fast search path */
    int reinsert_elements;          /* 1 */
    int show_progress;              /* 0 */
    uint32_t batch_max;             /* 0 = default (65536); at most 2^20 */
    uint32_t batch_div;             /* 0 = default (8) */
} granne_hip_build_config;

void granne_hip_build_config_default(granne_hip_build_config* config);

/* GranneBuilder::new(config, elements): elements are host rows [n][dim], prepared like a
 * Vectors file (normalised f32 / quantised i8); copied to HBM. */
int granne_hip_builder_create(granne_hip_builder** out, const granne_hip_build_config* config,
                              const void* elements, uint64_t n_elements, uint32_t dim, int dtype,
                              int device_id);
/* Same with elements already in device memory (dense [n][dim]). */
int granne_hip_builder_create_device(granne_hip_builder** out, const granne_hip_build_config* config,
                                     const void* d_elements, uint64_t n_elements, uint32_t dim, int dtype,
                                     int device_id, void* stream);
/* Builder::push (src/index/mod.rs:303-315) for n_new prepared rows (host, dense [n_new][dim]): the
 * builder's element container grows; nothing is indexed until the next granne_hip_builder_build
 * (the reference's append_elements flow, src/index/tests.rs:502-566). */
int granne_hip_builder_append(granne_hip_builder* builder, const void* elements, uint64_t n_new);

/* GranneBuilder::from_bytes(config, buffer, elements) (src/index/mod.rs:430-461): a builder that has
 * not built anything yet adopts the layers of a written index (granne's index file format); every
 * neighbor list is resized to config.num_neighbors -- truncated, or padded with UNUSED (:448).
 * granne_hip_builder_build then continues from len() as the reference does. */
int granne_hip_builder_load_index(granne_hip_builder* builder, const void* index_bytes, uint64_t index_len);

/* Builder::build_partial(num_elements) (src/index/mod.rs:374-402); num_elements == 0 returns at once
 * as the reference does (:375); GRANNE_HIP_BUILD_ALL is Builder::build() = all elements. Synchronous. */
#define GRANNE_HIP_BUILD_ALL UINT64_MAX
int granne_hip_builder_build(granne_hip_builder* builder, uint64_t num_elements);
uint64_t granne_hip_builder_len(const granne_hip_builder* builder);          /* indexed elements */
uint64_t granne_hip_builder_num_elements(const granne_hip_builder* builder);
uint32_t granne_hip_builder_num_layers(const granne_hip_builder* builder);
uint64_t granne_hip_builder_layer_len(const granne_hip_builder* builder, uint32_t layer);
/* copies layer `layer` as a host [layer_len][num_neighbors] u32 matrix, UNUSED padded: the
 * FixedWidthSliceVector<u32> the reference's builder holds (src/index/mod.rs:300) */
int granne_hip_builder_get_layer(const granne_hip_builder* builder, uint32_t layer, uint32_t* out_rows);
/* GranneBuilder::get_index (src/index/mod.rs:483-488): a searchable index over the current
 * layers (device-to-device copy; the builder stays usable). */
int granne_hip_builder_get_index(const granne_hip_builder* builder, granne_hip_index** out);
void granne_hip_builder_destroy(granne_hip_builder* builder);

/* SURVEY.md 8b's `index_create(..., device_ids, n_devices, partitioned)` in one call: the whole element set (host rows,
 * prepared like a Vectors file) is split into n_shards id ranges of ceil(n / n_shards) elements
 * (src/elements/embeddings/parsing.rs:72-98), shard s is built with the GPU builder under `config`
 * (GranneBuilder::new(config, shard).build()) on device_ids[s / ceil(n_shards / n_devices)] -- one host thread per entry
 * of device_ids, so the devices build at the same time -- and the searchable partitioned index is returned; every entry
 * of device_ids is an exchange group of its own (granne_hip_sharded_create_grouped), also when two name the same device.
 * The handle OWNS its shard indexes (granne_hip_sharded_destroy releases them).                                       */
int granne_hip_sharded_build(granne_hip_sharded** out, const granne_hip_build_config* config, const void* elements,
                             uint64_t n_elements, uint32_t dim, int dtype, uint32_t n_shards,
                             const int* device_ids, uint32_t n_devices);

/* ---- options (per index) ---------------------------------------------------------------------- */
enum {
    GRANNE_HIP_OPT_VISITED_SLOTS = 1, /* LDS visited-table slots per query: 2^k or 3 * 2^k in [256, 32768]; 0 = auto */
    GRANNE_HIP_OPT_FORCE_SLOW = 2,    /* 1: route every query through the exact global-memory path */
    GRANNE_HIP_OPT_SLOW_SLOTS = 3,    /* global visited/queue slots per slow-path query (pow2)      */
    GRANNE_HIP_OPT_SLOW_BLOCKS = 4,   /* concurrent slow-path walkers serving hand-overs; a batch that is the exact walker's
                                         as a whole runs up to 32x as many, within 1 GB of scratch
                                         (12 bytes x SLOW_SLOTS per walker; the block is kept per (index, stream)) */
    GRANNE_HIP_OPT_OVERFLOW_SLOTS = 5,/* global overflow slots per walk for a full LDS visited table:
                                         0 = auto, 1 = off (such walks go to the slow path), else pow2 */
    GRANNE_HIP_OPT_VISITED16 = 6,     /* the visited set of the register walkers (unless VISITED_SLOTS is set):
                                         0 = auto = 4 = NONE -- every neighbor is evaluated and the list itself is
                                         searched for a candidate's id; the results are the reference's, the n_dist
                                         counter counts evaluations instead of distinct nodes
                                         (granne_amd/csrc/wave_prims.h, VisitedNone). 1, 2, 3 = the EXACT set (a 32-bit
                                         open-addressing table in LDS + a global overflow table): n_dist is then the
                                         reference's count. (Rounds 3a-3 distinguished three forms of the exact set; one
                                         is left.) Lists beyond 1024 keys and 64-id layers always walk without a set */
    GRANNE_HIP_OPT_VISITED16_LG = 7,  /* retired with the bucket tables it sized: accepted (0..12), ignored */
    GRANNE_HIP_OPT_LAST_WALKER = 8,   /* read-only (get_option): which kernel the index's last search launch took */
    GRANNE_HIP_OPT_SEARCH_DEPTH = 9,  /* batches that granne_hip_search_begin_device may have in flight: 1..GRANNE_HIP_SEARCH_DEPTH_MAX
                                         [GRANNE_HIP_SEARCH_DEPTH]; cannot change while one is */
    GRANNE_HIP_OPT_INLINE_TAILS = 10, /* f32 indexes of 100 or 200 dimensions whose layers are 32 ids wide keep, for the
                                         register walker, a second copy of every layer in which a node's 32 neighbor ids are
                                         followed by the TAILS of those neighbors' rows (the dim % 32 last components, which
                                         src/math.rs:32-39 adds after the ordered sum): an expansion then reads whole 128-byte
                                         lines only -- three per 100-d neighbor instead of four. Costs 128 + 32 x tail bytes
                                         per node and layer (6.9 GB at 10M x 100-d) of HBM; results are the same bits.
                                         1 = keep it [default], 0 = drop it (the tails are read from the rows). Setting it
                                         (re)makes or frees the copy: not while a search of the index is running. An index
                                         whose copy does not fit in HBM is made without it (no error).
                                         get_option returns 1 only when the index actually holds the copy */
    GRANNE_HIP_OPT_SEEN_MIN = 11      /* f32 walks (every dim) of max_search up to 252 on layers of 32 ids: launches of at least this many
                                         walks (queries x batches) consult a cache of the ids the walk has EVALUATED before
                                         they fetch a neighbor's row, and skip a hit -- the reference's `!visited.insert(n)`
                                         (src/index/mod.rs:1026) for the recent part of the visited set; a miss means nothing
                                         (the row is evaluated, again if need be), so results do not depend on it. It saves the
                                         rows of revisits (3.6 % of a walk's rows on i.i.d.-uniform data, 40-70 % on clustered
                                         data) and costs an LDS round trip before the row loads, which only pays where the
                                         launch is bound by bandwidth: [2048]; 0 = every launch, 0xFFFFFFFF = never */
};
enum {
    GRANNE_HIP_WALKER_NONE = 0,          /* no search yet */
    GRANNE_HIP_WALKER_REGISTER = 1,      /* walk_fast.h: layers of up to 32 ids, max_search up to 8192 (1024 for f32 dims other
                                            than 100 / 200 and for int8 rows of 256 / 512 bytes) */
    GRANNE_HIP_WALKER_REGISTER_WIDE = 2, /* walk_fast.h, two passes per expansion: layers of up to 64 ids, max_search up to 1024 */
    GRANNE_HIP_WALKER_GENERAL = 3,       /* search_kernel.h: everything else up to max_search 256 */
    GRANNE_HIP_WALKER_EXACT = 4          /* slow_kernel.h: max_search beyond those, GRANNE_HIP_OPT_FORCE_SLOW */
};
int granne_hip_index_set_option(granne_hip_index* index, int option, uint64_t value);
int granne_hip_index_get_option(const granne_hip_index* index, int option, uint64_t* value);
/* number of queries of the last search_batch (host variant) that took the slow exact path */
uint64_t granne_hip_index_last_slow_count(const granne_hip_index* index);

#ifdef __cplusplus
}
#endif
#endif
