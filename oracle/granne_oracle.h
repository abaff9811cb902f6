/*
 * granne_oracle.h -- CPU restatement of granne's search path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle and the CPU baseline for the MI355X search path. It is NOT part
 * of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it. The product (granne_amd/, libgranne_hip.so) never links or calls it.
 *
 * Every function cites the reference lines (relative to /root/reference) it restates.
 *
 * Pinning status: the reference (Rust, v0.5.2) cannot be compiled or run here (no cargo /
 * rustc, no network), and its tests use an unseeded RNG, so it holds NO golden search
 * outputs. The oracle is pinned by
 *   - the reference's known-answer tests: layer sizes (src/index/tests.rs:304-335),
 *     delta encoding (src/slice_vector/set_vector.rs:231-237);
 *   - the reference's property tests restated in tests/ (math.rs:183-196,
 *     angular.rs:99-126, index/tests.rs:50-62,114-132);
 *   - a second, independent numpy/Python restatement (oracle/pyref.py: arithmetic, search, the
 *     whole single-threaded builder, reorder) diffed against it bit for bit / id for id.
 * Search-output parity against a *running* reference is therefore "unpinned"; see DESIGN.md.
 */
#ifndef GRANNE_ORACLE_H
#define GRANNE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRO_F32 0
#define GRO_I8 1
#define GRO_UNUSED 0xFFFFFFFFu /* src/index/mod.rs:27-28 */

/* ---- src/math.rs ------------------------------------------------------------------- */
float gro_dot_f32(const float* x, const float* y, size_t n);                 /* :5-52   */
void gro_dot_i8(const int8_t* x, const int8_t* y, size_t n, int32_t* r, int32_t* dx,
                int32_t* dy);                                                /* :59-89  */
void gro_normalize_f32(float* x, size_t n);                                  /* :123-150 */

/* ---- src/elements/angular.rs, angular_int.rs -------------------------------------------- */
float gro_dist_f32(const float* x, const float* y, size_t n);                /* angular.rs:63-74 */
float gro_reference_dist_f32(const float* x, const float* y, size_t n);      /* angular.rs:78-90 */
void gro_quantize(const float* s, size_t n, int8_t* out);                    /* angular_int.rs:27-45 */
float gro_dist_i8(const int8_t* x, const int8_t* y, size_t n);               /* angular_int.rs:47-60 */

/* ---- src/index/mod.rs:634-643 ---------------------------------------------------------- */
uint64_t gro_num_elements_in_layer(uint64_t total, float layer_multiplier, uint64_t layer_idx);

/* An index in "builder / FixWidth" form (src/index/mod.rs:483-488, 540-552): prefix-nested
 * layers, each a row-major u32 matrix [layer_len][width] padded with GRO_UNUSED. */
typedef struct {
    const void* elements; /* row-major [n_elements][dim] f32 (normalised) or i8 (quantised) */
    uint64_t n_elements;
    uint32_t dim;
    int dtype; /* GRO_F32 | GRO_I8 */
    uint32_t n_layers;
    const uint64_t* layer_len;
    const uint32_t* const* layer_rows;
    const uint32_t* layer_width;
} gro_index;

typedef struct {
    uint64_t n_dist;   /* calls to dist_to_element   (src/index/mod.rs:1012,1027) */
    uint64_t n_expand; /* calls to get_neighbors     (src/index/mod.rs:1025)      */
    uint64_t n_adj;    /* valid neighbor ids returned by those calls              */
} gro_counters;

/* search_for_neighbors, src/index/mod.rs:999-1037. Writes up to max_search (id,dist) pairs in
 * ascending (dist,id) order; returns the count. */
size_t gro_search_for_neighbors(const gro_index* ix, uint32_t layer, uint64_t entrypoint,
                                const void* goal, size_t max_search, uint64_t* out_ids,
                                float* out_dists, gro_counters* ctr);

/* Granne::search, src/index/mod.rs:140-150, 962-997. Returns the count (<= num_neighbors).
 * max_search == 0 is the reference's panic (mod.rs:1019): returns (size_t)-1. */
size_t gro_search(const gro_index* ix, const void* query, size_t max_search,
                  size_t num_neighbors, uint64_t* out_ids, float* out_dists, gro_counters* ctr);

/* Caller-side par_iter over queries (the reference has no batch API). out_* are
 * [nq][num_neighbors]; out_counts [nq]; ctr (optional) [nq]. Returns 0, or -1 on panic. */
int gro_search_batch(const gro_index* ix, const void* queries, size_t nq, size_t max_search,
                     size_t num_neighbors, uint64_t* out_ids, float* out_dists,
                     uint32_t* out_counts, gro_counters* ctr, int n_threads);

/* bench.py's CPU-baseline clock: one untimed pass, then `repeats` timed passes over the same batch inside one
 * parallel region (threads and their scratch stay warm). Returns wall seconds of the timed passes, < 0 on panic. */
double gro_search_batch_timed(const gro_index* ix, const void* queries, size_t nq, size_t max_search,
                              size_t num_neighbors, uint64_t* out_ids, float* out_dists,
                              uint32_t* out_counts, int n_threads, int repeats);

/* Exact k nearest elements by a scan of all elements (ElementContainer::dists over every index + a sort by
 * (distance, id)): the checker and CPU baseline of granne_hip_brute_force_device. Returns wall seconds. */
double gro_scan_topk(const gro_index* ix, const void* queries, size_t nq, size_t k, uint64_t* out_ids, float* out_dists,
                     int n_threads);

/* ---- build half, src/index/mod.rs:364-402, 645-960 ------------------------------------- */
typedef struct {
    float layer_multiplier;     /* 15.0 */
    uint64_t expected_num_elements; /* 0 = None */
    uint32_t num_neighbors;     /* 30 */
    uint32_t max_search;        /* 200 */
    int reinsert_elements;      /* 1 */
    int n_threads;              /* 1 = feature "singlethreaded" order (deterministic) */
    /* batch_max > 0 selects the BATCHED insertion schedule of the GPU builder
     * (granne_amd/csrc/builder.hip) instead of the reference's sequential / rayon schedule:
     * ids are taken in the reference's order in batches of clamp(n_in_graph / batch_div, 1,
     * batch_max); every member of a batch runs index_element's search + select_neighbors
     * against the graph as it was when the batch started, then the link updates
     * (initialize_node / connect_nodes) are applied in batch order. Deterministic for any
     * n_threads. The per-element arithmetic is the reference's, only the schedule differs
     * (the reference's own rayon schedule is nondeterministic, src/index/mod.rs:771-782). */
    uint32_t batch_max;
    uint32_t batch_div;
} gro_build_config;

void gro_build_config_default(gro_build_config* cfg); /* src/index/mod.rs:220-231 */

typedef struct gro_builder gro_builder;
/* GranneBuilder::new + build(): elements are borrowed (must outlive the builder). */
gro_builder* gro_builder_create(const gro_build_config* cfg, const void* elements,
                                uint64_t n_elements, uint32_t dim, int dtype);
void gro_builder_build_partial(gro_builder* b, uint64_t num_elements); /* mod.rs:374-402 */
/* GranneBuilder::from_bytes (mod.rs:430-461): adopt the layers of a written index; sets[l] is
 * [len[l]][set_width[l]], GRO_UNUSED padded, each list resized to cfg.num_neighbors. */
int gro_builder_load_layers(gro_builder* b, uint32_t n_layers, const uint64_t* len,
                            const uint32_t* const* sets, const uint32_t* set_width);
uint32_t gro_builder_num_layers(const gro_builder* b);
uint64_t gro_builder_layer_len(const gro_builder* b, uint32_t layer);
uint32_t gro_builder_layer_width(const gro_builder* b, uint32_t layer);
const uint32_t* gro_builder_layer_rows(const gro_builder* b, uint32_t layer);
void gro_builder_destroy(gro_builder* b);

/* GranneBuilder::select_neighbors (mod.rs:849-883), exposed for the reference's own test
 * (index/tests.rs:11-39). cand_* sorted ascending by distance. Returns the number kept. */
size_t gro_select_neighbors(const void* elements, uint32_t dim, int dtype, const uint64_t* cand_ids,
                            const float* cand_dists, size_t n_cand, size_t max_neighbors,
                            uint64_t* out_ids, float* out_dists);

/* ---- adjacency set codec, src/slice_vector/set_vector.rs:91-162 + stream-vbyte 0.3.2 ---- */
void gro_delta_encode(uint32_t* data, size_t n);  /* :150-155 */
void gro_delta_decode(uint32_t* data, size_t n);  /* :157-162 */
/* set_encode (:117-148): data must be sorted; out must hold 1 + 5*max(4,n) bytes. */
size_t gro_set_encode(const uint32_t* sorted, size_t n, uint8_t* out);
/* decode_into (:91-115): returns count; out must hold max(4,count) u32. */
size_t gro_set_decode(const uint8_t* enc, size_t enc_len, uint32_t* out);

/* ---- Granne::reorder, src/index/reorder.rs ------------------------------------------------ */
/* compute_order (:135-175) with find_entrypoint_trail (:180-208). order[i] = j means the
 * element with idx j moves to idx i. Needs n_layers >= 2 (the reference underflows
 * num_layers() - 2 and panics otherwise): returns -1 then, 0 on success. */
int gro_compute_order(const gro_index* ix, uint64_t* order, int n_threads);
/* reorder_by_keys (:88-133): layer-preserving sort by (key, idx). */
int gro_order_by_keys(const gro_index* ix, const uint64_t* keys, uint64_t* order);
/* reorder_layer (:230-281) for one FixWidth layer: row i of the result holds the neighbors of
 * order[i] mapped through the reverse mapping, SORTED ascending (MultiSetVector::push sorts,
 * src/slice_vector/set_vector.rs:41-47) and GRO_UNUSED padded to `width`. */
void gro_reorder_layer(const uint32_t* rows, uint64_t len, uint32_t width, const uint64_t* order,
                       uint64_t n_order, uint32_t* out_rows);

/* ---- synthetic data: counter-based, identical on every box ------------------ */
/* component i of row r: uniform [-0.5, 0.5) with 24 random bits, like rand 0.7's
 * gen::<f32>() - 0.5 (src/test_helper.rs:3-6). */
float gro_synth_component(uint64_t seed, uint64_t row, uint32_t col, uint32_t dim);
void gro_synth_rows(uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dim, float* out);

int gro_max_threads(void);
void gro_parallel_copy(void* dst, const void* src, size_t bytes, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
