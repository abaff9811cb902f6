/*
 * granne_oracle.c -- CPU restatement of granne's search (and build) path.
 * TEST INFRASTRUCTURE ONLY: see granne_oracle.h for who may load this and for the pinning
 * status. All file:line citations are relative to /root/reference (granne v0.5.2).
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp). With
 * -ffp-contract=off and explicit fmaf() the f32 arithmetic below is, operation for
 * operation, the arithmetic of the reference's default (non-BLAS) build: Rust never
 * re-associates or contracts floating point, and f32::mul_add is a fused multiply-add.
 */
#include "granne_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================================
 * src/math.rs
 * ====================================================================================== */

/* dot_product_f32, src/math.rs:5-52 (dot_product_fallback :16-42; the AVX2 clone :9-14 is the
 * same body). 32 independent fused accumulators over 32-wide chunks, then an ORDERED scalar
 * sum of the 32 accumulators starting from 0.0, then a fused tail. */
float gro_dot_f32(const float* x, const float* y, size_t n) {
    enum { CHUNK = 32 };
    float chunk[CHUNK];
    for (int i = 0; i < CHUNK; ++i) chunk[i] = 0.0f;

    size_t n_chunks = n / CHUNK;
    for (size_t c = 0; c < n_chunks; ++c) {
        const float* a = x + c * CHUNK;
        const float* b = y + c * CHUNK;
        for (int i = 0; i < CHUNK; ++i) chunk[i] = fmaf(a[i], b[i], chunk[i]); /* :22-24 */
    }

    float r = 0.0f;
    for (int i = 0; i < CHUNK; ++i) r += chunk[i]; /* :27-30 */

    for (size_t i = n_chunks * CHUNK; i < n; ++i) r = fmaf(x[i], y[i], r); /* :32-39 */
    return r;
}

/* dot_product_and_squared_norms_i8, src/math.rs:59-89: three exact i32 sums. */
void gro_dot_i8(const int8_t* x, const int8_t* y, size_t n, int32_t* r, int32_t* dx, int32_t* dy) {
    int32_t sr = 0, sx = 0, sy = 0;
    for (size_t i = 0; i < n; ++i) {
        int32_t xi = x[i], yi = y[i];
        sr += xi * yi;
        sx += xi * xi;
        sy += yi * yi;
    }
    *r = sr;
    *dx = sx;
    *dy = sy;
}

/* normalize_f32, src/math.rs:123-150: norm = sqrt(dot(x,x)); if norm > 0, x[i] /= norm. */
void gro_normalize_f32(float* x, size_t n) {
    float norm = sqrtf(gro_dot_f32(x, x, n));
    if (norm > 0.0f) {
        for (size_t i = 0; i < n; ++i) x[i] /= norm;
    }
}

/* ======================================================================================
 * src/elements/angular.rs, src/elements/angular_int.rs
 * ====================================================================================== */

/* impl Dist for angular::Vector, angular.rs:63-74: d = 1 - dot; max(0, d). The reference
 * panics on NaN (NotNan::new(..).unwrap()); NaN inputs are outside the contract. */
float gro_dist_f32(const float* x, const float* y, size_t n) {
    float r = gro_dot_f32(x, y, n);
    float d = 1.0f - r;
    /* cmp::max(0.0, d) on NotNan: returns d when 0.0 <= d, else 0.0 (angular.rs:72) */
    return (0.0f <= d) ? d : 0.0f;
}

/* angular_reference_dist, angular.rs:78-90 (naive form used by the reference's own tests). */
float gro_reference_dist_f32(const float* x, const float* y, size_t n) {
    float r = 0.0f, dx = 0.0f, dy = 0.0f;
    for (size_t i = 0; i < n; ++i) r += x[i] * y[i];
    for (size_t i = 0; i < n; ++i) dx += x[i] * x[i];
    for (size_t i = 0; i < n; ++i) dy += y[i] * y[i];
    float d = 1.0f - (r / (sqrtf(dx) * sqrtf(dy)));
    return (0.0f <= d) ? d : 0.0f;
}

/* Rust `f32 as i8`: truncate toward zero, saturate, NaN -> 0. */
static int8_t f32_as_i8(float v) {
    if (v != v) return 0;
    if (v <= -128.0f) return -128;
    if (v >= 127.0f) return 127;
    return (int8_t)(int32_t)v; /* C conversion truncates toward zero */
}

/* angular_int::Vector::quantize, angular_int.rs:27-45. */
void gro_quantize(const float* s, size_t n, int8_t* out) {
    const float MAX_QVALUE = 127.0f;
    float max_value = MAX_QVALUE; /* unwrap_or_else for the empty slice, :33 */
    if (n > 0) {
        max_value = fabsf(s[0]);
        for (size_t i = 1; i < n; ++i) {
            float a = fabsf(s[i]);
            if (a > max_value) max_value = a;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        float vi = s[i] * MAX_QVALUE / max_value; /* left to right, :38 */
        out[i] = f32_as_i8(vi);                   /* :40 */
    }
}

/* impl Dist for angular_int::Vector, angular_int.rs:47-60. */
float gro_dist_i8(const int8_t* x, const int8_t* y, size_t n) {
    int32_t ri, dxi, dyi;
    gro_dot_i8(x, y, n, &ri, &dxi, &dyi);
    float r = (float)ri, dx = (float)dxi, dy = (float)dyi;
    float q = r / (sqrtf(dx) * sqrtf(dy));
    if (q != q) q = 0.0f; /* NotNan::new(..).unwrap_or_else(|_| 0.0), :55 */
    float d = 1.0f - q;
    return (0.0f <= d) ? d : 0.0f;
}

/* ======================================================================================
 * src/index/mod.rs:634-643  compute_num_elements_in_layer
 * ====================================================================================== */
uint64_t gro_num_elements_in_layer(uint64_t total, float layer_multiplier_f32, uint64_t layer_idx) {
    double m = (double)layer_multiplier_f32; /* `layer_multiplier as f64`, :635 */
    double t = (double)total;
    /* f64::log(self, base) = self.ln() / base.ln() */
    double e = floor(log(t) / log(m)) - (double)layer_idx;
    double v = ceil(t / pow(m, e));
    uint64_t r;
    if (!(v >= 0.0)) r = 0; /* Rust `as usize` saturates; NaN -> 0 */
    else if (v >= 18446744073709551615.0) r = UINT64_MAX;
    else r = (uint64_t)v;
    return r < total ? r : total;
}

/* ======================================================================================
 * generic element access (ElementContainer for Vectors, dense_vector.rs:110-112,149-151)
 * ====================================================================================== */
static inline size_t elem_size(int dtype) { return dtype == GRO_F32 ? 4 : 1; }

static inline const void* row_ptr(const void* elements, uint32_t dim, int dtype, uint64_t idx) {
    return (const char*)elements + idx * (uint64_t)dim * elem_size(dtype);
}

static inline float dist_rows(int dtype, const void* a, const void* b, uint32_t dim) {
    return dtype == GRO_F32 ? gro_dist_f32((const float*)a, (const float*)b, dim)
                            : gro_dist_i8((const int8_t*)a, (const int8_t*)b, dim);
}

/* ======================================================================================
 * containers used by search_for_neighbors
 * ====================================================================================== */

/* (NotNan<f32>, usize) tuple with the derived lexicographic Ord. Distances are never NaN and
 * never -0.0 (1.0 - r cannot produce -0.0; the clamp returns +0.0), so `<` on floats is the
 * NotNan order. */
typedef struct {
    float d;
    uint64_t id;
} entry_t;

static inline int entry_lt(entry_t a, entry_t b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }

typedef struct {
    entry_t* v;
    size_t len, cap;
} heap_t;

static void heap_reserve(heap_t* h, size_t cap) {
    if (cap > h->cap) {
        size_t nc = h->cap ? h->cap : 64;
        while (nc < cap) nc *= 2;
        h->v = (entry_t*)realloc(h->v, nc * sizeof(entry_t));
        h->cap = nc;
    }
}

/* max-heap (std BinaryHeap<T>) */
static void maxheap_push(heap_t* h, entry_t e) {
    heap_reserve(h, h->len + 1);
    size_t i = h->len++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!entry_lt(h->v[p], e)) break;
        h->v[i] = h->v[p];
        i = p;
    }
    h->v[i] = e;
}
static void maxheap_pop(heap_t* h) {
    entry_t e = h->v[--h->len];
    size_t i = 0, n = h->len;
    if (n == 0) return;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && entry_lt(h->v[c], h->v[c + 1])) ++c;
        if (!entry_lt(e, h->v[c])) break;
        h->v[i] = h->v[c];
        i = c;
    }
    h->v[i] = e;
}
/* min-heap (BinaryHeap<Reverse<T>>) */
static void minheap_push(heap_t* h, entry_t e) {
    heap_reserve(h, h->len + 1);
    size_t i = h->len++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!entry_lt(e, h->v[p])) break;
        h->v[i] = h->v[p];
        i = p;
    }
    h->v[i] = e;
}
static entry_t minheap_pop(heap_t* h) {
    entry_t top = h->v[0];
    entry_t e = h->v[--h->len];
    size_t i = 0, n = h->len;
    if (n > 0) {
        for (;;) {
            size_t c = 2 * i + 1;
            if (c >= n) break;
            if (c + 1 < n && entry_lt(h->v[c + 1], h->v[c])) ++c;
            if (!entry_lt(h->v[c], e)) break;
            h->v[i] = h->v[c];
            i = c;
        }
        h->v[i] = e;
    }
    return top;
}

/* exact visited set (HashSet<usize, Fx>, mod.rs:1009-1010): open addressing, grows. */
typedef struct {
    uint64_t* keys; /* key + 1, 0 = empty */
    size_t cap, len;
} set_t;

static inline size_t set_hash(uint64_t k, size_t mask) {
    return (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & mask;
}
static void set_init(set_t* s, size_t want) {
    size_t cap = 64;
    while (cap < want * 2) cap *= 2;
    if (cap > s->cap) {
        free(s->keys);
        s->keys = (uint64_t*)malloc(cap * sizeof(uint64_t));
        s->cap = cap;
    } else {
        /* keep allocation; shrink logical capacity so clearing stays cheap */
        cap = s->cap;
    }
    memset(s->keys, 0, s->cap * sizeof(uint64_t));
    s->len = 0;
}
static void set_grow(set_t* s) {
    size_t ocap = s->cap;
    uint64_t* old = s->keys;
    s->cap = ocap * 2;
    s->keys = (uint64_t*)calloc(s->cap, sizeof(uint64_t));
    size_t mask = s->cap - 1;
    for (size_t i = 0; i < ocap; ++i) {
        if (old[i]) {
            size_t h = set_hash(old[i] - 1, mask);
            while (s->keys[h]) h = (h + 1) & mask;
            s->keys[h] = old[i];
        }
    }
    free(old);
}
/* returns 1 if newly inserted (HashSet::insert) */
static inline int set_insert(set_t* s, uint64_t k) {
    size_t mask = s->cap - 1;
    size_t h = set_hash(k, mask);
    while (s->keys[h]) {
        if (s->keys[h] == k + 1) return 0;
        h = (h + 1) & mask;
    }
    s->keys[h] = k + 1;
    if (++s->len * 2 > s->cap) set_grow(s);
    return 1;
}

typedef struct {
    heap_t res, pq;
    set_t visited;
    uint64_t* tmp_ids;
    float* tmp_d;
    size_t tmp_cap;
} scratch_t;

static void scratch_free(scratch_t* s) {
    free(s->res.v);
    free(s->pq.v);
    free(s->visited.keys);
    free(s->tmp_ids);
    free(s->tmp_d);
    memset(s, 0, sizeof(*s));
}

/* ======================================================================================
 * Graph trait (mod.rs:535-577): a layer is either a plain fixed-width matrix or, while
 * building in parallel, the same matrix guarded by per-node locks.
 * ====================================================================================== */
typedef struct {
    const uint32_t* rows;
    uint64_t len;
    uint32_t width;
    volatile unsigned char* locks; /* NULL when not building in parallel */
} layer_view;

static inline void node_lock(volatile unsigned char* l) {
    while (__atomic_test_and_set((void*)l, __ATOMIC_ACQUIRE)) {
        while (__atomic_load_n(l, __ATOMIC_RELAXED)) { /* spin */ }
    }
}
static inline void node_unlock(volatile unsigned char* l) { __atomic_clear((void*)l, __ATOMIC_RELEASE); }

/* get_neighbors (mod.rs:540-552 / 564-577): row prefix until UNUSED. Returns count. */
static inline uint32_t layer_get_neighbors(const layer_view* L, uint64_t idx, uint32_t* out) {
    const uint32_t* row = L->rows + idx * L->width;
    uint32_t n = 0;
    if (L->locks) {
        node_lock(&L->locks[idx]);
        while (n < L->width && row[n] != GRO_UNUSED) { out[n] = row[n]; ++n; }
        node_unlock(&L->locks[idx]);
    } else {
        while (n < L->width && row[n] != GRO_UNUSED) { out[n] = row[n]; ++n; }
    }
    return n;
}

/* ======================================================================================
 * search_for_neighbors, src/index/mod.rs:999-1037
 * ====================================================================================== */
static size_t search_for_neighbors_impl(const layer_view* L, uint64_t entrypoint, const void* elements,
                                        uint32_t dim, int dtype, const void* goal, size_t max_search,
                                        scratch_t* S, gro_counters* ctr) {
    heap_t* res = &S->res; /* MaxSizeHeap<(NotNan<f32>, usize)>(max_search), :1006 */
    heap_t* pq = &S->pq;   /* BinaryHeap<Reverse<..>>, :1007 */
    res->len = 0;
    pq->len = 0;
    set_init(&S->visited, max_search * 20); /* :1009-1010 */

    uint32_t nbuf_static[256];
    uint32_t* nbuf = nbuf_static;
    uint32_t* nbuf_heap = NULL;
    if (L->width > 256) nbuf = nbuf_heap = (uint32_t*)malloc(sizeof(uint32_t) * L->width);

    uint64_t n_dist = 0, n_expand = 0, n_adj = 0;

    entry_t e0;
    e0.d = dist_rows(dtype, row_ptr(elements, dim, dtype, entrypoint), goal, dim); /* :1012 */
    e0.id = entrypoint;
    ++n_dist;
    minheap_push(pq, e0);               /* :1014 */
    set_insert(&S->visited, entrypoint); /* :1016 */

    while (pq->len > 0) { /* :1018 */
        entry_t top = minheap_pop(pq);
        int full = res->len >= max_search;
        if (full && top.d > res->v[0].d) break; /* :1019-1021; max_search==0 panics upstream */

        /* res.push, max_size_heap.rs:18-32 */
        if (!full) {
            maxheap_push(res, top);
        } else if (entry_lt(top, res->v[0])) {
            maxheap_pop(res);
            maxheap_push(res, top);
        }

        uint32_t nn = layer_get_neighbors(L, top.id, nbuf); /* :1025 */
        ++n_expand;
        n_adj += nn;
        for (uint32_t k = 0; k < nn; ++k) {
            uint64_t nb = nbuf[k];
            if (set_insert(&S->visited, nb)) { /* :1026 */
                entry_t e;
                e.d = dist_rows(dtype, row_ptr(elements, dim, dtype, nb), goal, dim); /* :1027 */
                e.id = nb;
                ++n_dist;
                if (!(res->len >= max_search) || e.d < res->v[0].d) minheap_push(pq, e); /* :1029-1031 */
            }
        }
    }
    free(nbuf_heap);

    if (ctr) {
        ctr->n_dist += n_dist;
        ctr->n_expand += n_expand;
        ctr->n_adj += n_adj;
    }

    /* into_sorted_vec, :1036: ascending (dist, id). Heap-sort in place. */
    size_t n = res->len;
    heap_t tmp = *res;
    for (size_t i = n; i > 0; --i) {
        entry_t mx = tmp.v[0];
        maxheap_pop(&tmp); /* shrinks tmp.len to i-1, slot i-1 now free */
        tmp.v[i - 1] = mx;
    }
    res->len = n;
    return n;
}

static void make_layer_view(const gro_index* ix, uint32_t layer, layer_view* L) {
    L->rows = ix->layer_rows[layer];
    L->len = ix->layer_len[layer];
    L->width = ix->layer_width[layer];
    L->locks = NULL;
}

size_t gro_search_for_neighbors(const gro_index* ix, uint32_t layer, uint64_t entrypoint, const void* goal,
                                size_t max_search, uint64_t* out_ids, float* out_dists, gro_counters* ctr) {
    if (max_search == 0) return (size_t)-1;
    scratch_t S;
    memset(&S, 0, sizeof(S));
    layer_view L;
    make_layer_view(ix, layer, &L);
    size_t n = search_for_neighbors_impl(&L, entrypoint, ix->elements, ix->dim, ix->dtype, goal, max_search, &S,
                                         ctr);
    for (size_t i = 0; i < n; ++i) {
        out_ids[i] = S.res.v[i].id;
        out_dists[i] = S.res.v[i].d;
    }
    scratch_free(&S);
    return n;
}

/* Granne::search -> search_internal -> find_entrypoint, mod.rs:140-150, 963-997 */
static size_t search_impl(const gro_index* ix, uint32_t n_layers, const void* query, size_t max_search,
                          size_t num_neighbors, uint64_t* out_ids, float* out_dists, scratch_t* S,
                          gro_counters* ctr) {
    if (n_layers == 0) return 0; /* :978-980 */
    uint64_t entrypoint = 0;     /* :989 */
    layer_view L;
    for (uint32_t l = 0; l + 1 < n_layers; ++l) { /* top_layers, :990-994 */
        make_layer_view(ix, l, &L);
        search_for_neighbors_impl(&L, entrypoint, ix->elements, ix->dim, ix->dtype, query, 1, S, ctr);
        entrypoint = S->res.v[0].id; /* res[0].0, :993 */
    }
    make_layer_view(ix, n_layers - 1, &L);
    size_t n = search_for_neighbors_impl(&L, entrypoint, ix->elements, ix->dim, ix->dtype, query, max_search, S,
                                         ctr); /* :973 */
    if (n > num_neighbors) n = num_neighbors;  /* .take(num_neighbors), :975 */
    for (size_t i = 0; i < n; ++i) {
        out_ids[i] = S->res.v[i].id;
        out_dists[i] = S->res.v[i].d;
    }
    return n;
}

size_t gro_search(const gro_index* ix, const void* query, size_t max_search, size_t num_neighbors,
                  uint64_t* out_ids, float* out_dists, gro_counters* ctr) {
    if (max_search == 0) return (size_t)-1; /* reference panics, mod.rs:1019 */
    scratch_t S;
    memset(&S, 0, sizeof(S));
    size_t n = search_impl(ix, ix->n_layers, query, max_search, num_neighbors, out_ids, out_dists, &S, ctr);
    scratch_free(&S);
    return n;
}

/* ---- Granne::reorder, src/index/reorder.rs ------------------------------------------------ */
#define TRAIL_LAYERS 8 /* NUM_LAYERS, reorder.rs:177 */
typedef struct {
    uint32_t eps[TRAIL_LAYERS];
    uint64_t idx;
} trail_key;

static int trail_key_cmp(const void* a, const void* b) { /* (eps, idx) tuple order, reorder.rs:161 */
    const trail_key* x = (const trail_key*)a;
    const trail_key* y = (const trail_key*)b;
    for (int i = 0; i < TRAIL_LAYERS; ++i)
        if (x->eps[i] != y->eps[i]) return x->eps[i] < y->eps[i] ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* find_entrypoint_trail, reorder.rs:180-208. `let ep = if i == 0 { 0 } else { eps[i] }` reads the
 * slot that is still 0, so every layer's walk starts at node 0 -- restated as written. */
static void find_entrypoint_trail(const gro_index* ix, uint32_t max_layer, const void* element, scratch_t* S,
                                  uint32_t* eps) {
    for (int i = 0; i < TRAIL_LAYERS; ++i) eps[i] = 0;
    uint32_t take = max_layer < TRAIL_LAYERS ? max_layer : TRAIL_LAYERS;
    if (take > ix->n_layers) take = ix->n_layers;
    for (uint32_t i = 0; i < take; ++i) {
        uint64_t ep = (i == 0) ? 0 : eps[i];
        layer_view L;
        make_layer_view(ix, i, &L);
        search_for_neighbors_impl(&L, ep, ix->elements, ix->dim, ix->dtype, element, 1, S, NULL);
        eps[i] = (uint32_t)S->res.v[0].id;
    }
}

int gro_compute_order(const gro_index* ix, uint64_t* order, int n_threads) {
    if (ix->n_layers < 2) return -1; /* num_layers() - 2 underflows, reorder.rs:137 */
    const size_t esz = ix->dtype == GRO_F32 ? 4 : 1;
    const uint64_t len0 = ix->layer_len[0];
    for (uint64_t i = 0; i < len0; ++i) order[i] = i;            /* :136 */
    const uint64_t inv_len = ix->layer_len[ix->n_layers - 2];
    uint64_t* order_inv = (uint64_t*)calloc(inv_len ? inv_len : 1, sizeof(uint64_t)); /* zeros, :137 */
    (void)n_threads;
    for (uint32_t layer = 1; layer < ix->n_layers; ++layer) {   /* :149 */
        const uint64_t lo = ix->layer_len[layer - 1], hi = ix->layer_len[layer];
        trail_key* keys = (trail_key*)malloc(sizeof(trail_key) * (size_t)(hi > lo ? hi - lo : 1));
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
        {
            scratch_t S;
            memset(&S, 0, sizeof(S));
#pragma omp for schedule(dynamic, 256)
            for (int64_t idx = (int64_t)lo; idx < (int64_t)hi; ++idx) {
                trail_key* k = &keys[idx - (int64_t)lo];
                find_entrypoint_trail(ix, layer, (const uint8_t*)ix->elements + (size_t)idx * ix->dim * esz, &S,
                                      k->eps);
                for (int i = 0; i < TRAIL_LAYERS; ++i) k->eps[i] = (uint32_t)order_inv[k->eps[i]]; /* :159 */
                k->idx = (uint64_t)idx;
            }
            scratch_free(&S);
        }
        qsort(keys, (size_t)(hi - lo), sizeof(trail_key), trail_key_cmp); /* :164 (total order: idx is unique) */
        for (uint64_t i = lo; i < hi; ++i) order[i] = keys[i - lo].idx;   /* :165 */
        free(keys);
        if (layer < ix->n_layers - 1)                                     /* :167-171 */
            for (uint64_t i = lo; i < hi; ++i) order_inv[order[i]] = i;
    }
    free(order_inv);
    return 0;
}

typedef struct {
    uint64_t key, idx;
} key_idx;
static int key_idx_cmp(const void* a, const void* b) {
    const key_idx* x = (const key_idx*)a;
    const key_idx* y = (const key_idx*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int gro_order_by_keys(const gro_index* ix, const uint64_t* keys, uint64_t* order) { /* reorder.rs:88-110 */
    for (uint32_t layer = 0; layer < ix->n_layers; ++layer) {
        const uint64_t lo = layer ? ix->layer_len[layer - 1] : 0, hi = ix->layer_len[layer];
        key_idx* v = (key_idx*)malloc(sizeof(key_idx) * (size_t)(hi > lo ? hi - lo : 1));
        for (uint64_t l = lo; l < hi; ++l) {
            v[l - lo].key = keys[l];
            v[l - lo].idx = l;
        }
        qsort(v, (size_t)(hi - lo), sizeof(key_idx), key_idx_cmp);
        for (uint64_t l = lo; l < hi; ++l) order[l] = v[l - lo].idx;
        free(v);
    }
    return 0;
}

static int u32_cmp(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

void gro_reorder_layer(const uint32_t* rows, uint64_t len, uint32_t width, const uint64_t* order, uint64_t n_order,
                       uint32_t* out_rows) {
    uint64_t* rev = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n_order ? n_order : 1)); /* get_reverse_mapping, :283-292 */
    for (uint64_t i = 0; i < n_order; ++i) rev[order[i]] = i;
    for (uint64_t i = 0; i < len; ++i) { /* mapping[..layer.len()], :243 */
        const uint32_t* src = rows + (size_t)order[i] * width;
        uint32_t* dst = out_rows + (size_t)i * width;
        uint32_t n = 0;
        for (uint32_t c = 0; c < width && src[c] != GRO_UNUSED; ++c) dst[n++] = (uint32_t)rev[src[c]]; /* :250-256 */
        qsort(dst, n, sizeof(uint32_t), u32_cmp); /* MultiSetVector::push sorts */
        for (uint32_t c = n; c < width; ++c) dst[c] = GRO_UNUSED;
    }
    free(rev);
}

int gro_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* dst[0..bytes) = src[0..bytes), copied by an OpenMP team in static chunks of 2 MB: the pages of `dst` are first
 * touched by the threads of the team that will read them (spread over the NUMA nodes the team spans) -- how a host that
 * tunes for its own machine would place an element file it gathers from at random. bench.py's CPU baseline only. */
void gro_parallel_copy(void* dst, const void* src, size_t bytes, int n_threads) {
    if (n_threads <= 0) n_threads = gro_max_threads();
    const size_t chunk = (size_t)2 << 20;
    const long long n_chunks = (long long)((bytes + chunk - 1) / chunk);
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (long long c = 0; c < n_chunks; ++c) {
        const size_t lo = (size_t)c * chunk, hi = lo + chunk < bytes ? lo + chunk : bytes;
        memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
    }
}

static double gro_wtime(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return (double)clock() / CLOCKS_PER_SEC;
#endif
}

int gro_search_batch(const gro_index* ix, const void* queries, size_t nq, size_t max_search,
                     size_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                     gro_counters* ctr, int n_threads) {
    if (max_search == 0) return -1;
    if (n_threads <= 0) n_threads = gro_max_threads();
    size_t qstride = (size_t)ix->dim * elem_size(ix->dtype);
#pragma omp parallel num_threads(n_threads)
    {
        scratch_t S;
        memset(&S, 0, sizeof(S));
#pragma omp for schedule(dynamic, 1)
        for (long long q = 0; q < (long long)nq; ++q) {
            gro_counters c = {0, 0, 0};
            size_t n = search_impl(ix, ix->n_layers, (const char*)queries + (size_t)q * qstride, max_search,
                                   num_neighbors, out_ids + (size_t)q * num_neighbors,
                                   out_dists + (size_t)q * num_neighbors, &S, ctr ? &c : NULL);
            out_counts[q] = (uint32_t)n;
            if (ctr) ctr[q] = c;
        }
        scratch_free(&S);
    }
    return 0;
}

/* The CPU baseline's clock (bench.py): `repeats` passes over the same batch inside ONE parallel region -- the
 * threads, their scratch (visited set, heaps) and the index pages stay warm between passes, as they would in a
 * service that answers queries all day -- after one untimed pass. Returns the wall seconds of the timed passes
 * (omp_get_wtime around them, barriers on both sides), or a negative number on the reference's panic. */
double gro_search_batch_timed(const gro_index* ix, const void* queries, size_t nq, size_t max_search,
                              size_t num_neighbors, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                              int n_threads, int repeats) {
    if (max_search == 0) return -1.0;
    if (n_threads <= 0) n_threads = gro_max_threads();
    if (repeats < 1) repeats = 1;
    size_t qstride = (size_t)ix->dim * elem_size(ix->dtype);
    double t0 = 0.0, t1 = 0.0;
#pragma omp parallel num_threads(n_threads)
    {
        scratch_t S;
        memset(&S, 0, sizeof(S));
        for (int rep = 0; rep <= repeats; ++rep) {
            if (rep == 1) {
#pragma omp barrier
#pragma omp master
                t0 = gro_wtime();
            }
#pragma omp for schedule(dynamic, 1)
            for (long long q = 0; q < (long long)nq; ++q) {
                size_t n = search_impl(ix, ix->n_layers, (const char*)queries + (size_t)q * qstride, max_search,
                                       num_neighbors, out_ids + (size_t)q * num_neighbors,
                                       out_dists + (size_t)q * num_neighbors, &S, NULL);
                out_counts[q] = (uint32_t)n;
            }
        }
#pragma omp barrier
#pragma omp master
        t1 = gro_wtime();
        scratch_free(&S);
    }
    return t1 - t0;
}

/* Exact k nearest elements of every query by a scan of ALL elements: what a caller gets from
 * ElementContainer::dists over every index (src/elements/mod.rs:35-39) followed by a sort by (distance, id) -- the
 * checker and the CPU baseline of granne_hip_brute_force_device. Threads split the elements; a thread walks its rows
 * once and evaluates every query against a row while the row is in L1 (queries are few: they stay cached), keeping
 * the k best per query; the per-thread lists are merged at the end. Distances are the reference's own (gro_dist_*).
 * Returns wall seconds of the scan. out_ids / out_dists: [nq][k], padded with ~0 / +inf when n < k. */
typedef struct { float d; uint64_t id; } scan_ent;
static int scan_less(float da, uint64_t ia, float db, uint64_t ib) { return da < db || (da == db && ia < ib); }
double gro_scan_topk(const gro_index* ix, const void* queries, size_t nq, size_t k, uint64_t* out_ids, float* out_dists,
                     int n_threads) {
    if (n_threads <= 0) n_threads = gro_max_threads();
    const size_t esz = elem_size(ix->dtype), rowb = (size_t)ix->dim * esz;
    const uint64_t n = ix->n_elements;
    scan_ent* all = (scan_ent*)malloc(sizeof(scan_ent) * (size_t)n_threads * nq * k);
    for (size_t i = 0; i < (size_t)n_threads * nq * k; ++i) { all[i].d = INFINITY; all[i].id = ~(uint64_t)0; }
    const double t0 = gro_wtime();
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        scan_ent* mine = all + (size_t)t * nq * k;
        const uint64_t lo = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T;
        for (uint64_t e = lo; e < hi; ++e) {
            const char* row = (const char*)ix->elements + (size_t)e * rowb;
            for (size_t q = 0; q < nq; ++q) {
                const char* qp = (const char*)queries + q * rowb;
                const float d = ix->dtype == GRO_F32 ? gro_dist_f32((const float*)row, (const float*)qp, ix->dim)
                                                     : gro_dist_i8((const int8_t*)row, (const int8_t*)qp, ix->dim);
                scan_ent* L = mine + q * k;
                if (scan_less(d, e, L[k - 1].d, L[k - 1].id)) { /* insertion into the sorted k-list */
                    size_t p = k - 1;
                    while (p > 0 && scan_less(d, e, L[p - 1].d, L[p - 1].id)) { L[p] = L[p - 1]; --p; }
                    L[p].d = d;
                    L[p].id = e;
                }
            }
        }
    }
    for (size_t q = 0; q < nq; ++q) { /* merge the threads' lists */
        for (size_t j = 0; j < k; ++j) { out_ids[q * k + j] = ~(uint64_t)0; out_dists[q * k + j] = INFINITY; }
        for (int t = 0; t < n_threads; ++t) {
            const scan_ent* L = all + ((size_t)t * nq + q) * k;
            for (size_t j = 0; j < k && L[j].id != ~(uint64_t)0; ++j) {
                if (!scan_less(L[j].d, L[j].id, out_dists[q * k + k - 1], out_ids[q * k + k - 1])) break;
                size_t p = k - 1;
                while (p > 0 && scan_less(L[j].d, L[j].id, out_dists[q * k + p - 1], out_ids[q * k + p - 1])) {
                    out_dists[q * k + p] = out_dists[q * k + p - 1];
                    out_ids[q * k + p] = out_ids[q * k + p - 1];
                    --p;
                }
                out_dists[q * k + p] = L[j].d;
                out_ids[q * k + p] = L[j].id;
            }
        }
    }
    const double dt = gro_wtime() - t0;
    free(all);
    return dt;
}

/* ======================================================================================
 * build half, src/index/mod.rs:364-402, 645-960
 * ====================================================================================== */
struct gro_builder {
    gro_build_config cfg;
    const void* elements;
    uint64_t n_elements;
    uint32_t dim;
    int dtype;
    uint32_t n_layers;
    uint32_t cap_layers;
    uint32_t** rows;   /* per layer, [len][width] */
    uint64_t* len;     /* per layer */
    uint32_t* width;   /* per layer (= cfg.num_neighbors for every layer, mod.rs:393-398) */
};

void gro_build_config_default(gro_build_config* cfg) { /* mod.rs:220-231 */
    cfg->layer_multiplier = 15.0f;
    cfg->expected_num_elements = 0;
    cfg->num_neighbors = 30;
    cfg->max_search = 200;
    cfg->reinsert_elements = 1;
    cfg->n_threads = 1;
    cfg->batch_max = 0;
    cfg->batch_div = 8;
}

gro_builder* gro_builder_create(const gro_build_config* cfg, const void* elements, uint64_t n_elements,
                                uint32_t dim, int dtype) {
    gro_builder* b = (gro_builder*)calloc(1, sizeof(gro_builder));
    b->cfg = *cfg;
    b->elements = elements;
    b->n_elements = n_elements;
    b->dim = dim;
    b->dtype = dtype;
    return b;
}

void gro_builder_destroy(gro_builder* b) {
    if (!b) return;
    for (uint32_t l = 0; l < b->n_layers; ++l) free(b->rows[l]);
    free(b->rows);
    free(b->len);
    free(b->width);
    free(b);
}

uint32_t gro_builder_num_layers(const gro_builder* b) { return b->n_layers; }
uint64_t gro_builder_layer_len(const gro_builder* b, uint32_t l) { return b->len[l]; }
uint32_t gro_builder_layer_width(const gro_builder* b, uint32_t l) { return b->width[l]; }
const uint32_t* gro_builder_layer_rows(const gro_builder* b, uint32_t l) { return b->rows[l]; }

static inline float bdist(const gro_builder* b, uint64_t i, uint64_t j) {
    return dist_rows(b->dtype, row_ptr(b->elements, b->dim, b->dtype, i), row_ptr(b->elements, b->dim, b->dtype, j),
                     b->dim);
}

/* select_neighbors, mod.rs:849-883. cand sorted ascending by distance. */
static size_t select_neighbors_impl(const void* elements, uint32_t dim, int dtype, const uint64_t* cid,
                                    const float* cd, size_t n_cand, size_t max_neighbors, uint64_t* oid,
                                    float* od) {
    if (n_cand <= max_neighbors) { /* :854-856 */
        for (size_t i = 0; i < n_cand; ++i) {
            oid[i] = cid[i];
            od[i] = cd[i];
        }
        return n_cand;
    }
    size_t n = 0;
    for (size_t c = 0; c < n_cand; ++c) {
        if (n >= max_neighbors) break; /* :867-869 */
        uint64_t j = cid[c];
        float d = cd[c];
        const void* ej = row_ptr(elements, dim, dtype, j);
        int ok = 1;
        for (size_t k = 0; k < n; ++k) { /* :874-877: d <= dist(n_k, j) for all selected */
            float dk = dist_rows(dtype, row_ptr(elements, dim, dtype, oid[k]), ej, dim);
            if (!(d <= dk)) {
                ok = 0;
                break;
            }
        }
        if (ok) {
            oid[n] = j;
            od[n] = d;
            ++n;
        }
    }
    return n;
}

size_t gro_select_neighbors(const void* elements, uint32_t dim, int dtype, const uint64_t* cand_ids,
                            const float* cand_dists, size_t n_cand, size_t max_neighbors, uint64_t* out_ids,
                            float* out_dists) {
    return select_neighbors_impl(elements, dim, dtype, cand_ids, cand_dists, n_cand, max_neighbors, out_ids,
                                 out_dists);
}

typedef struct {
    float d;
    uint64_t id;
} cand_t;

/* sort_unstable_by_key(|&(_, d)| d), mod.rs:943. The reference's unstable sort leaves the
 * order of equal distances unspecified; we break ties by id to stay deterministic. */
static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a;
    const cand_t* y = (const cand_t*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* add_and_limit_neighbors, mod.rs:923-959 (caller holds the node's write lock). */
static void add_and_limit_neighbors(const gro_builder* b, uint32_t* node, uint32_t node_width, uint64_t node_id,
                                    const cand_t* extra, size_t n_extra, size_t num_neighbors) {
    size_t nn = 0;
    while (nn < node_width && node[nn] != GRO_UNUSED) ++nn; /* :932-936 */
    size_t nc = nn + n_extra;
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (nc ? nc : 1));
    for (size_t k = 0; k < nn; ++k) { /* elements.dists(node_id, &neighbors), :938 */
        c[k].id = node[k];
        c[k].d = bdist(b, node_id, node[k]);
    }
    for (size_t k = 0; k < n_extra; ++k) c[nn + k] = extra[k]; /* :941-943 */
    qsort(c, nc, sizeof(cand_t), cand_cmp);

    uint64_t* cid = (uint64_t*)malloc(sizeof(uint64_t) * (nc ? nc : 1) * 2);
    uint64_t* oid = cid + (nc ? nc : 1);
    float* cd = (float*)malloc(sizeof(float) * (nc ? nc : 1) * 2);
    float* od = cd + (nc ? nc : 1);
    for (size_t k = 0; k < nc; ++k) {
        cid[k] = c[k].id;
        cd[k] = c[k].d;
    }
    size_t ns = select_neighbors_impl(b->elements, b->dim, b->dtype, cid, cd, nc, num_neighbors, oid, od); /* :947 */
    for (size_t k = 0; k < node_width; ++k) node[k] = (k < ns) ? (uint32_t)oid[k] : GRO_UNUSED; /* :950-958 */
    free(c);
    free(cid);
    free(cd);
}

/* connect_nodes, mod.rs:898-921 */
static void connect_nodes(const gro_builder* b, uint32_t* rows, uint32_t width, volatile unsigned char* locks,
                          uint64_t i, uint64_t j, float d) {
    if (i == j) return;
    uint32_t* node = rows + i * width;
    if (locks) node_lock(&locks[i]);
    uint32_t jid = (uint32_t)j;
    uint32_t pos = 0;
    while (pos < width && !(node[pos] == GRO_UNUSED || node[pos] == jid)) ++pos; /* :912 */
    if (pos < width) {
        node[pos] = jid;
    } else {
        cand_t e;
        e.id = j;
        e.d = d;
        add_and_limit_neighbors(b, node, width, i, &e, 1, width); /* num_neighbors = node.len(), :916-917 */
    }
    if (locks) node_unlock(&locks[i]);
}

/* index_element, mod.rs:805-846, split in two so that the batched schedule can reuse it:
 * phase A = everything up to the dead-node rule (reads the graph), phase B = the link updates. */
static size_t index_element_select(const gro_builder* b, const gro_build_config* config, const gro_index* prev,
                                   uint32_t* rows, uint32_t width, uint64_t layer_len,
                                   volatile unsigned char* locks, uint64_t idx, scratch_t* S, uint64_t** out_ids,
                                   float** out_d) {
    const float EPS100 = 100.0f * 1.1920929e-07f; /* 100.0 * f32::EPSILON */
    if (bdist(b, idx, idx) > EPS100) return 0;   /* zero vectors, :813-815 */
    const void* element = row_ptr(b->elements, b->dim, b->dtype, idx);

    /* prev_layers.search(&element, 1, 1).first().map_or(0, |r| r.0), :819 */
    uint64_t entrypoint = 0;
    {
        uint64_t id1;
        float d1;
        size_t n = search_impl(prev, prev->n_layers, element, 1, 1, &id1, &d1, S, NULL);
        if (n > 0) entrypoint = id1;
    }

    layer_view L;
    L.rows = rows;
    L.len = layer_len;
    L.width = width;
    L.locks = locks;
    size_t nc = search_for_neighbors_impl(&L, entrypoint, b->elements, b->dim, b->dtype, element,
                                          config->max_search, S, NULL); /* :820 */
    if (S->tmp_cap < nc + 1) {
        S->tmp_cap = nc + 64;
        S->tmp_ids = (uint64_t*)realloc(S->tmp_ids, sizeof(uint64_t) * S->tmp_cap * 2);
        S->tmp_d = (float*)realloc(S->tmp_d, sizeof(float) * S->tmp_cap * 2);
    }
    uint64_t* cid = S->tmp_ids;
    float* cd = S->tmp_d;
    uint64_t* nid = S->tmp_ids + S->tmp_cap;
    float* nd = S->tmp_d + S->tmp_cap;
    size_t m = 0;
    for (size_t k = 0; k < nc; ++k) { /* filter id != idx, :822 */
        if (S->res.v[k].id != idx) {
            cid[m] = S->res.v[k].id;
            cd[m] = S->res.v[k].d;
            ++m;
        }
    }
    size_t nsel = select_neighbors_impl(b->elements, b->dim, b->dtype, cid, cd, m, config->num_neighbors, nid,
                                        nd); /* :824 */

    /* duplicate ("dead node") rule, :828-832 */
    size_t half = config->num_neighbors / 2;
    if (half < nsel && nd[half] < EPS100) return 0;
    *out_ids = nid;
    *out_d = nd;
    return nsel;
}

static void index_element_apply(const gro_builder* b, uint32_t* rows, uint32_t width, volatile unsigned char* locks,
                                uint64_t idx, const uint64_t* nid, const float* nd, size_t nsel) {
    if (nsel == 0) {
        /* the reference still runs the (empty) loops; an empty selection leaves the row as is,
         * except that an empty row is "initialised" with nothing. Either way: no change. */
        return;
    }
    uint32_t* node = rows + idx * width;
    int empty;
    if (locks) node_lock(&locks[idx]);
    empty = node[0] == GRO_UNUSED; /* :835 */
    if (empty) {                   /* initialize_node, :886-896 */
        for (size_t k = 0; k < nsel && k < width; ++k) node[k] = (uint32_t)nid[k];
    }
    if (locks) node_unlock(&locks[idx]);
    if (!empty) {
        for (size_t k = 0; k < nsel; ++k) connect_nodes(b, rows, width, locks, idx, nid[k], nd[k]); /* :838-840 */
    }
    for (size_t k = 0; k < nsel; ++k) connect_nodes(b, rows, width, locks, nid[k], idx, nd[k]); /* :843-845 */
}

static void index_element(const gro_builder* b, const gro_build_config* config, const gro_index* prev,
                          uint32_t* rows, uint32_t width, uint64_t layer_len, volatile unsigned char* locks,
                          uint64_t idx, scratch_t* S) {
    uint64_t* nid = NULL;
    float* nd = NULL;
    size_t nsel = index_element_select(b, config, prev, rows, width, layer_len, locks, idx, S, &nid, &nd);
    index_element_apply(b, rows, width, locks, idx, nid, nd, nsel);
}

/* The batched schedule (see gro_build_config.batch_max). */
static void index_elements_batched(gro_builder* b, const gro_build_config* config, const gro_index* prev,
                                   uint32_t* rows, uint32_t width, uint64_t layer_len, uint64_t already_indexed,
                                   int reinsert) {
    int nt = config->n_threads <= 0 ? gro_max_threads() : config->n_threads;
    uint64_t total = reinsert ? layer_len : layer_len - already_indexed;
    uint32_t nn = config->num_neighbors;
    uint64_t bmax = config->batch_max;
    uint64_t* sel_ids = (uint64_t*)malloc(sizeof(uint64_t) * bmax * nn);
    float* sel_d = (float*)malloc(sizeof(float) * bmax * nn);
    uint32_t* sel_n = (uint32_t*)malloc(sizeof(uint32_t) * bmax);
    uint64_t pos = 0;
    while (pos < total) {
        uint64_t n_in_graph = reinsert ? layer_len : already_indexed + pos;
        uint64_t B = n_in_graph / (config->batch_div ? config->batch_div : 1);
        if (B < 1) B = 1;
        if (B > bmax) B = bmax;
        if (B > total - pos) B = total - pos;
        /* phase A: graph frozen */
#pragma omp parallel num_threads(nt)
        {
            scratch_t S;
            memset(&S, 0, sizeof(S));
#pragma omp for schedule(dynamic, 16)
            for (long long t = 0; t < (long long)B; ++t) {
                uint64_t idx = reinsert ? (layer_len - 1 - (pos + (uint64_t)t)) : (already_indexed + pos + (uint64_t)t);
                uint64_t* nid = NULL;
                float* nd = NULL;
                size_t ns = index_element_select(b, config, prev, rows, width, layer_len, NULL, idx, &S, &nid, &nd);
                sel_n[t] = (uint32_t)ns;
                for (size_t k = 0; k < ns; ++k) {
                    sel_ids[(uint64_t)t * nn + k] = nid[k];
                    sel_d[(uint64_t)t * nn + k] = nd[k];
                }
            }
            scratch_free(&S);
        }
        /* phase B: apply in batch order */
        for (uint64_t t = 0; t < B; ++t) {
            uint64_t idx = reinsert ? (layer_len - 1 - (pos + t)) : (already_indexed + pos + t);
            index_element_apply(b, rows, width, NULL, idx, sel_ids + t * nn, sel_d + t * nn, sel_n[t]);
        }
        pos += B;
    }
    free(sel_ids);
    free(sel_d);
    free(sel_n);
}

/* index_elements, mod.rs:715-802 */
static void index_elements(gro_builder* b, const gro_build_config* config, uint64_t num_elements,
                           const gro_index* prev, uint32_t** rows_p, uint64_t* len_p, uint32_t width,
                           int reinsert) {
    uint64_t already_indexed = *len_p;
    if (reinsert) {
        already_indexed = 0;
    } else { /* layer.resize(num_elements, UNUSED), :730 */
        *rows_p = (uint32_t*)realloc(*rows_p, sizeof(uint32_t) * (size_t)num_elements * width);
        for (uint64_t i = (uint64_t)(*len_p) * width; i < num_elements * width; ++i) (*rows_p)[i] = GRO_UNUSED;
        *len_p = num_elements;
    }
    uint32_t* rows = *rows_p;
    uint64_t layer_len = *len_p;
    int nt = config->n_threads <= 0 ? gro_max_threads() : config->n_threads;

    if (config->batch_max > 0) {
        index_elements_batched(b, config, prev, rows, width, layer_len, already_indexed, reinsert);
#pragma omp parallel for num_threads(nt) schedule(dynamic, 256)
        for (long long t = 0; t < (long long)layer_len; ++t) /* :795-797 */
            add_and_limit_neighbors(b, rows + (uint64_t)t * width, width, (uint64_t)t, NULL, 0,
                                    config->num_neighbors);
    } else if (nt == 1) { /* feature "singlethreaded": sequential, deterministic, :771-782 */
        scratch_t S;
        memset(&S, 0, sizeof(S));
        if (reinsert) {
            for (uint64_t i = layer_len; i > 0; --i) index_element(b, config, prev, rows, width, layer_len, NULL, i - 1, &S);
        } else {
            for (uint64_t i = already_indexed; i < layer_len; ++i) index_element(b, config, prev, rows, width, layer_len, NULL, i, &S);
        }
        for (uint64_t i = 0; i < layer_len; ++i) /* :795-797 */
            add_and_limit_neighbors(b, rows + i * width, width, i, NULL, 0, config->num_neighbors);
        scratch_free(&S);
    } else { /* rayon par_iter with per-node RwLocks, :757-782 */
        volatile unsigned char* locks = (volatile unsigned char*)calloc((size_t)layer_len, 1);
        long long lo = (long long)already_indexed, hi = (long long)layer_len;
#pragma omp parallel num_threads(nt)
        {
            scratch_t S;
            memset(&S, 0, sizeof(S));
#pragma omp for schedule(dynamic, 64)
            for (long long t = lo; t < hi; ++t) {
                uint64_t i = reinsert ? (uint64_t)(hi - 1 - (t - lo)) : (uint64_t)t;
                index_element(b, config, prev, rows, width, layer_len, locks, i, &S);
            }
#pragma omp for schedule(dynamic, 256)
            for (long long t = 0; t < hi; ++t)
                add_and_limit_neighbors(b, rows + (uint64_t)t * width, width, (uint64_t)t, NULL, 0,
                                        config->num_neighbors);
            scratch_free(&S);
        }
        free((void*)locks);
    }
}

/* index_elements_in_last_layer, mod.rs:646-713 */
static void index_elements_in_last_layer(gro_builder* b, uint64_t max_num_elements) {
    uint64_t total = b->cfg.expected_num_elements ? b->cfg.expected_num_elements : b->n_elements;
    uint64_t t2 = total > b->n_elements ? total : b->n_elements;
    uint32_t last = b->n_layers - 1;
    uint64_t ideal = gro_num_elements_in_layer(t2, b->cfg.layer_multiplier, last);
    if (ideal <= b->len[last]) return; /* :654-657 */
    uint64_t num_in_layer = max_num_elements < ideal ? max_num_elements : ideal;

    gro_build_config config = b->cfg;
    if (ideal < total) { /* not last layer: half num_neighbors, :665-668 */
        config.num_neighbors = config.num_neighbors / 2 > 1 ? config.num_neighbors / 2 : 1;
    }

    /* prev_layers = self.get_index() after popping the last layer, :670-674 */
    gro_index prev;
    prev.elements = b->elements;
    prev.n_elements = b->n_elements;
    prev.dim = b->dim;
    prev.dtype = b->dtype;
    prev.n_layers = last;
    prev.layer_len = b->len;
    prev.layer_rows = (const uint32_t* const*)b->rows;
    prev.layer_width = b->width;

    index_elements(b, &config, num_in_layer, &prev, &b->rows[last], &b->len[last], b->width[last], 0);
    if (b->cfg.reinsert_elements) { /* :692-710 */
        config.max_search = config.max_search / 2 > 1 ? config.max_search / 2 : 1;
        index_elements(b, &config, num_in_layer, &prev, &b->rows[last], &b->len[last], b->width[last], 1);
    }
}

/* build_partial, mod.rs:374-402 */
/* GranneBuilder::from_bytes, mod.rs:430-461: the layers of a written index become the builder's
 * layers; every neighbor list is resized to config.num_neighbors (:448 -- truncated or padded
 * with UNUSED). sets[l] is a [len[l]][set_width[l]] matrix, UNUSED padded. Returns 0, -1 if the
 * builder already has layers. */
int gro_builder_load_layers(gro_builder* b, uint32_t n_layers, const uint64_t* len, const uint32_t* const* sets,
                            const uint32_t* set_width) {
    if (b->n_layers) return -1;
    b->cap_layers = n_layers > 8 ? n_layers : 8;
    b->rows = (uint32_t**)realloc(b->rows, sizeof(uint32_t*) * b->cap_layers);
    b->len = (uint64_t*)realloc(b->len, sizeof(uint64_t) * b->cap_layers);
    b->width = (uint32_t*)realloc(b->width, sizeof(uint32_t) * b->cap_layers);
    const uint32_t w = b->cfg.num_neighbors;
    for (uint32_t l = 0; l < n_layers; ++l) {
        size_t bytes = sizeof(uint32_t) * (size_t)len[l] * w;
        b->rows[l] = (uint32_t*)malloc(bytes ? bytes : 4);
        b->len[l] = len[l];
        b->width[l] = w;
        for (uint64_t i = 0; i < len[l]; ++i) {
            const uint32_t* src = sets[l] + (size_t)i * set_width[l];
            uint32_t* dst = b->rows[l] + (size_t)i * w;
            uint32_t n = 0;
            while (n < set_width[l] && src[n] != GRO_UNUSED) ++n; /* layer.get_into(i, ..) */
            for (uint32_t c = 0; c < w; ++c) dst[c] = c < n ? src[c] : GRO_UNUSED; /* neighbors.resize(..) */
        }
    }
    b->n_layers = n_layers;
    return 0;
}

void gro_builder_build_partial(gro_builder* b, uint64_t num_elements) {
    if (num_elements == 0) return;
    if (num_elements > b->n_elements) num_elements = b->n_elements; /* reference asserts */
    if (b->n_layers > 0) index_elements_in_last_layer(b, num_elements);
    while ((b->n_layers ? b->len[b->n_layers - 1] : 0) < num_elements) {
        if (b->n_layers == b->cap_layers) {
            b->cap_layers = b->cap_layers ? b->cap_layers * 2 : 8;
            b->rows = (uint32_t**)realloc(b->rows, sizeof(uint32_t*) * b->cap_layers);
            b->len = (uint64_t*)realloc(b->len, sizeof(uint64_t) * b->cap_layers);
            b->width = (uint32_t*)realloc(b->width, sizeof(uint32_t) * b->cap_layers);
        }
        uint32_t l = b->n_layers;
        if (l == 0) { /* FixedWidthSliceVector::with_width(num_neighbors), :394 */
            b->rows[l] = NULL;
            b->len[l] = 0;
            b->width[l] = b->cfg.num_neighbors;
        } else { /* prev_layer.clone(), :395 */
            size_t bytes = sizeof(uint32_t) * (size_t)b->len[l - 1] * b->width[l - 1];
            b->rows[l] = (uint32_t*)malloc(bytes ? bytes : 4);
            memcpy(b->rows[l], b->rows[l - 1], bytes);
            b->len[l] = b->len[l - 1];
            b->width[l] = b->width[l - 1];
        }
        b->n_layers = l + 1;
        index_elements_in_last_layer(b, num_elements);
    }
}

/* ======================================================================================
 * adjacency set codec: src/slice_vector/set_vector.rs:91-162 over stream-vbyte 0.3.2
 * (Scalar codec; Lemire's layout: ceil(n/4) control bytes, 2 bits per number = byte length
 * minus one, first number in the low bits; then the numbers' little-endian bytes).
 * ====================================================================================== */
void gro_delta_encode(uint32_t* data, size_t n) {
    for (size_t i = n; i > 1; --i) data[i - 1] -= data[i - 2];
}
void gro_delta_decode(uint32_t* data, size_t n) {
    for (size_t i = 1; i < n; ++i) data[i] += data[i - 1];
}

static size_t svb_encode(const uint32_t* in, size_t n, uint8_t* out) {
    size_t n_ctrl = (n + 3) / 4;
    memset(out, 0, n_ctrl);
    uint8_t* data = out + n_ctrl;
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t v = in[i];
        unsigned len = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
        out[i / 4] |= (uint8_t)((len - 1) << (2 * (i % 4)));
        for (unsigned k = 0; k < len; ++k) data[w++] = (uint8_t)(v >> (8 * k));
    }
    return n_ctrl + w;
}

static size_t svb_decode(const uint8_t* in, size_t n, uint32_t* out) {
    size_t n_ctrl = (n + 3) / 4;
    const uint8_t* data = in + n_ctrl;
    size_t r = 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned len = ((in[i / 4] >> (2 * (i % 4))) & 3u) + 1;
        uint32_t v = 0;
        for (unsigned k = 0; k < len; ++k) v |= (uint32_t)data[r++] << (8 * k);
        out[i] = v;
    }
    return n_ctrl + r;
}

size_t gro_set_encode(const uint32_t* sorted, size_t n, uint8_t* out) { /* :117-148 */
    if (n >= 255) n = 255;
    uint32_t buf[256];
    memcpy(buf, sorted, n * sizeof(uint32_t));
    gro_delta_encode(buf, n);
    size_t count = n;
    size_t m = n < 4 ? 4 : n; /* MIN_NUMBERS_TO_ENCODE, :125-127 */
    for (size_t i = n; i < m; ++i) buf[i] = 0;
    size_t enc = svb_encode(buf, m, out + 1);
    if (enc >= 4 * count) { /* only use compression if it makes the data smaller, :137-143 */
        for (size_t i = 0; i < count; ++i) {
            uint32_t v = buf[i];
            out[1 + 4 * i + 0] = (uint8_t)v;
            out[1 + 4 * i + 1] = (uint8_t)(v >> 8);
            out[1 + 4 * i + 2] = (uint8_t)(v >> 16);
            out[1 + 4 * i + 3] = (uint8_t)(v >> 24);
        }
        enc = 4 * count;
    }
    out[0] = (uint8_t)count; /* :145 */
    return enc + 1;
}

size_t gro_set_decode(const uint8_t* enc, size_t enc_len, uint32_t* out) { /* :91-115 */
    size_t count = enc[0];
    const uint8_t* p = enc + 1;
    size_t len = enc_len - 1;
    if (len != count * 4) {
        size_t m = count < 4 ? 4 : count;
        svb_decode(p, m, out);
    } else {
        for (size_t i = 0; i < count; ++i)
            out[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 |
                     (uint32_t)p[4 * i + 3] << 24;
    }
    gro_delta_decode(out, count);
    return count;
}

/* ======================================================================================
 * synthetic data. splitmix64 over a (seed,row,col) counter; 24 random bits
 * -> [0,1) -> minus 0.5, mirroring `rng.gen::<f32>() - 0.5` (src/test_helper.rs:3-6).
 * ====================================================================================== */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float gro_synth_component(uint64_t seed, uint64_t row, uint32_t col, uint32_t dim) {
    uint64_t ctr = row * (uint64_t)dim + col;
    uint64_t z = splitmix64(splitmix64(seed) ^ ctr);
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f) - 0.5f;
}

void gro_synth_rows(uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dim, float* out) {
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < (long long)n_rows; ++r)
        for (uint32_t c = 0; c < dim; ++c)
            out[(uint64_t)r * dim + c] = gro_synth_component(seed, row0 + (uint64_t)r, c, dim);
}
