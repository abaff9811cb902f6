// wave_prims.h -- wave64 building blocks of the on-device HNSW walk (gfx950 / CDNA4).
//
// One wavefront (64 lanes) owns one query. The reference's three containers
// (/root/reference/src/index/mod.rs:1006-1010) map to:
//   MaxSizeHeap `res`        -> SortedList<S>: ascending keys held in VGPRs, entry e in
//   BinaryHeap  `pq`         -> SortedList<S>   (slot e/64, lane e%64); insert = ballot-rank +
//                               one-lane shift (DPP wave_shr), pop-min = shift the other way
//   HashSet     `visited`    -> VisitedSet: exact open-addressing table in LDS, ds_cmpst CAS
// A key packs the reference's (NotNan<f32>, usize) tuple into one u64: distance bits in the
// high word, id in the low word. Distances are >= +0.0 (clamped, angular.rs:72), never NaN
// and never -0.0, so unsigned integer order on the key IS the tuple's lexicographic Ord.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef GRANNE_HIP_PHASE_TIMERS
#define GRANNE_HIP_PHASE_TIMERS 0 // diagnostics build, see walk_fast.h
#endif

namespace granne_hip {

// Pointers that were themselves loaded from memory (e.g. LayerDev::adj) reach the compiler as
// generic pointers and compile to flat_load; these typedefs keep them in the global address
// space (global_load: vmcnt only, no LDS aperture check).
typedef const uint32_t __attribute__((address_space(1))) * gptr_u32;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const u32x4_t __attribute__((address_space(1))) * gptr_v4;
__device__ __forceinline__ uint4 load_global_u4(gptr_u32 p) { // one global_load_dwordx4
    u32x4_t t = *(gptr_v4)p;
    return make_uint4(t.x, t.y, t.z, t.w);
}

constexpr uint64_t KEY_INF = ~0ull;      // sorts after every real key (dist bits <= 0x40000000)
constexpr uint32_t ID_EMPTY = 0xFFFFFFFFu; // == UNUSED (src/index/mod.rs:27-28): never a node id

__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) {
    return ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)id;
}
__device__ __forceinline__ float key_dist(uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)k; }

__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __ballot(p); }

// v_readlane with a wave-uniform lane index
__device__ __forceinline__ uint32_t readlane32(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t lane) {
    uint32_t lo = readlane32((uint32_t)v, lane);
    uint32_t hi = readlane32((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// lane i receives lane i-1's value (lane 0: unspecified)
__device__ __forceinline__ uint32_t shift_up1(uint32_t v) {
#if GRANNE_HIP_USE_DPP
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
    return (uint32_t)__shfl_up((int)v, 1, 64);
#endif
}
// lane i receives lane i+1's value (lane 63: unspecified)
__device__ __forceinline__ uint32_t shift_down1(uint32_t v) {
#if GRANNE_HIP_USE_DPP
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
#else
    return (uint32_t)__shfl_down((int)v, 1, 64);
#endif
}
__device__ __forceinline__ uint64_t shift_up1(uint64_t v) {
    return ((uint64_t)shift_up1((uint32_t)(v >> 32)) << 32) | shift_up1((uint32_t)v);
}
__device__ __forceinline__ uint64_t shift_down1(uint64_t v) {
    return ((uint64_t)shift_down1((uint32_t)(v >> 32)) << 32) | shift_down1((uint32_t)v);
}

// Ascending sorted list of up to 64*S keys spread over the wave's registers.
template <int S>
struct SortedList {
    uint64_t key[S];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < S; ++s) key[s] = KEY_INF;
    }
    // number of entries strictly smaller than c (wave-uniform)
    __device__ __forceinline__ uint32_t rank(uint64_t c) const {
        uint32_t r = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) r += (uint32_t)__popcll(wave_ballot(key[s] < c));
        return r;
    }
    // entry e (wave-uniform index)
    __device__ __forceinline__ uint64_t get(uint32_t e) const {
        uint64_t v = readlane64(key[0], e & 63u);
#pragma unroll
        for (int s = 1; s < S; ++s) {
            uint64_t t = readlane64(key[s], e & 63u);
            if ((e >> 6) == (uint32_t)s) v = t;
        }
        return v;
    }
    // insert c at position r (< 64*S): entries at >= r move up one, the last one falls off
    __device__ __forceinline__ void insert_at(uint32_t r, uint64_t c, uint32_t lane) {
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            uint64_t up = shift_up1(key[s]);
            if (s > 0) {
                uint64_t carry = readlane64(key[s - 1], 63);
                if (lane == 0) up = carry;
            }
            uint32_t e = (uint32_t)s * 64u + lane;
            key[s] = (e > r) ? up : ((e == r) ? c : key[s]);
        }
    }
    // remove entry 0
    __device__ __forceinline__ void pop_front(uint32_t lane) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            uint64_t dn = shift_down1(key[s]);
            uint64_t carry = KEY_INF;
            if (s + 1 < S) carry = readlane64(key[s + 1], 0);
            key[s] = (lane == 63) ? carry : dn;
        }
    }
    // overwrite entry e with KEY_INF (used to cap `res` at max_search entries)
    __device__ __forceinline__ void clear_at(uint32_t e, uint32_t lane) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if ((uint32_t)s * 64u + lane == e) key[s] = KEY_INF;
    }
};

// Exact visited set (HashSet<usize>, src/index/mod.rs:1009-1010,1016,1026): open addressing with
// double hashing (no primary clustering, so a table can run at 7/8 load), in two levels. The
// front table lives in LDS. When it reaches its load limit it is frozen (lookups only) and new
// ids go to an overflow table in global memory that the walker borrows from a per-launch pool --
// the walk continues where it is, nothing is recomputed. The set stays exact: an id is in the set
// iff it is in one of the two tables.
struct OverflowPool {
    uint32_t* tables;  // [regions][stride] u32, global: a region = [overflow table: slots][id mirror: stride - slots]
    uint32_t* state;   // [regions] 0 = free, 1 = taken (all zero between launches: every walker gives its region back)
    uint32_t slots;    // overflow table of a region, power of two; 0 = no overflow: a full front table bails
    uint32_t stride;   // u32 words per region (>= slots)
    uint32_t regions;
    uint32_t* spilled; // optional statistics: += 1 per walk that spilled
};

struct VisitedSet {
    uint32_t* tab;   // LDS front table
    uint32_t size;   // slots: 2^k or 3 * 2^k (3072 slots hold a max_search-50 walk in 12 KB instead of 16)
    bool size_div3;  // wave-uniform
    uint32_t count;  // wave-uniform number of ids in the front table
    uint32_t limit;  // front table load limit
    // overflow state, kept to two wave-uniform words: the borrowed region (or NONE) and the number
    // of ids in its table (NONE while the front table still accepts inserts)
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t region;
    uint32_t ocount;
#if GRANNE_HIP_PHASE_TIMERS
    uint32_t pt_rounds = 0; // probe rounds of the wave (the slowest lane's) in the front table
#endif

    __device__ __forceinline__ void init_walker() {
        region = NONE;
        ocount = NONE;
    }
    __device__ __forceinline__ bool frozen() const { return ocount != NONE; }
    // `eighths`: the front table freezes at eighths/8 load (7: a walk that is expected to fit; lower for walks
    // that will spill anyway -- a lookup in a frozen table at 7/8 load probes 8 slots on average and the wave
    // waits for its slowest lane)
    __device__ __forceinline__ void reset(uint32_t* lds, uint32_t slots, uint32_t lane, uint32_t eighths = 7u) {
        tab = lds;
        size = slots;
        size_div3 = (slots % 3u) == 0u;
        count = 0;
        // 87.5 % load, and never fewer than 72 free slots: one expansion adds up to 64 ids before
        // the limit is checked, so probing always terminates
        limit = (slots >> 3) * eighths;
        if (limit + 72u > slots) limit = slots - 72u;
        uint4 e = make_uint4(ID_EMPTY, ID_EMPTY, ID_EMPTY, ID_EMPTY);
        uint4* t4 = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = lane; i < (slots >> 2); i += 64) t4[i] = e;
        ocount = NONE; // a borrowed region is kept for the next layer and wiped when it is used again
    }
    __device__ __forceinline__ static uint32_t hash(uint32_t id) { return (id * 0x9E3779B1u) >> 7; }
    __device__ __forceinline__ static uint32_t step(uint32_t id) { return ((id * 0x85EBCA6Bu) >> 9) | 1u; }
    // front table (any size): home slot = floor(hash32 * size / 2^32); the probe step is odd, below size and,
    // when 3 divides size, not a multiple of 3 -- coprime with size, so a probe sequence visits every slot
    __device__ __forceinline__ uint32_t home(uint32_t id) const { return __umulhi(id * 0x9E3779B1u, size); }
    __device__ __forceinline__ uint32_t stride(uint32_t id) const {
        uint32_t st = __umulhi(id * 0x85EBCA6Bu, size >> 1) * 2u + 1u; // odd, <= size - 1
        if (size_div3) {
            const uint32_t r = st - 3u * __umulhi(st, 0x55555556u); // st % 3
            if (r == 0u) st = (st + 2u < size) ? st + 2u : st - 2u;
        }
        return st;
    }
    __device__ __forceinline__ uint32_t next(uint32_t slot, uint32_t st) const {
        slot += st;
        return slot >= size ? slot - size : slot;
    }

    // HashSet::insert: true iff id was not present. Lanes with active==false do nothing.
    __device__ __forceinline__ bool insert(uint32_t id, bool active, const OverflowPool& pool) {
        bool fresh = false;
        const uint32_t st = stride(id);
#if GRANNE_HIP_PHASE_TIMERS
        uint32_t r_ = 0;
#endif
        if (!frozen()) {
            if (active) {
                uint32_t slot = home(id);
                for (;;) {
#if GRANNE_HIP_PHASE_TIMERS
                    r_ += 1;
#endif
                    uint32_t old = atomicCAS(&tab[slot], ID_EMPTY, id);
                    if (old == ID_EMPTY) { fresh = true; break; }
                    if (old == id) break;
                    slot = next(slot, st);
                }
            }
#if GRANNE_HIP_PHASE_TIMERS
            for (int o_ = 32; o_ > 0; o_ >>= 1) r_ = max(r_, (uint32_t)__shfl_xor((int)r_, o_, 64));
            pt_rounds += r_;
#endif
        } else {
            bool absent = false;
            if (active) { // the frozen front table: lookup only
                uint32_t slot = home(id);
                for (;;) {
                    uint32_t v = tab[slot];
                    if (v == ID_EMPTY) { absent = true; break; }
                    if (v == id) break;
                    slot = next(slot, st);
                }
            }
            if (absent) {
                uint32_t* otab = pool.tables + (size_t)region * pool.stride;
                const uint32_t omask = pool.slots - 1; // the overflow tables are powers of two
                const uint32_t ost = step(id);         // odd
                uint32_t slot = (hash(id) >> 3) & omask;
                for (;;) {
                    uint32_t old = atomicCAS(&otab[slot], ID_EMPTY, id);
                    if (old == ID_EMPTY) { fresh = true; break; }
                    if (old == id) break;
                    slot = (slot + ost) & omask;
                }
            }
        }
        return fresh;
    }
    // m ids were inserted (wave-uniform)
    __device__ __forceinline__ void added(uint32_t m) {
        if (frozen()) ocount += m; else count += m;
    }
    // after an expansion: spill to the overflow table when the front table is full. Returns false
    // when the walk cannot go on in this kernel (no overflow configured, pool exhausted, or the
    // overflow table itself is full): the caller hands the query to the global-memory walker.
    __device__ __forceinline__ bool make_room(const OverflowPool& pool, uint32_t lane) {
        if (frozen()) return ocount <= pool.slots - (pool.slots >> 2) - 72u; // 75 % load
        if (count <= limit) return true;
        if (pool.slots == 0) return false;
        if (region == NONE) {
            uint32_t got = NONE;
            if (lane == 0) {
                uint32_t r = (blockIdx.x * 0x9E3779B1u) % pool.regions;
                for (uint32_t tries = 0; tries < pool.regions; ++tries) {
                    if (atomicCAS(&pool.state[r], 0u, 1u) == 0u) { got = r; break; }
                    r = (r + 1 == pool.regions) ? 0u : r + 1;
                }
                if (got != NONE && pool.spilled) atomicAdd(pool.spilled, 1u);
            }
            region = (uint32_t)__shfl((int)got, 0, 64);
            if (region == NONE) return false;
        }
        uint4 e = make_uint4(ID_EMPTY, ID_EMPTY, ID_EMPTY, ID_EMPTY);
        uint4* t4 = reinterpret_cast<uint4*>(pool.tables + (size_t)region * pool.stride);
        for (uint32_t i = lane; i < (pool.slots >> 2); i += 64) t4[i] = e;
        __threadfence(); // the wipe is complete before any lane's atomicCAS on the table
        ocount = 0;
        return true;
    }
    // end of the walk: give the region back
    __device__ __forceinline__ void release(const OverflowPool& pool, uint32_t lane) {
        if (region != NONE) {
            __threadfence();
            if (lane == 0) atomicExch(&pool.state[region], 0u);
            region = NONE;
        }
    }
};

// ---- the same exact set in HALF the LDS: 16-bit entries, two-choice buckets ----------------------------------
// For id spaces of at most 32767 * nb ids (nb = number of buckets, a power of two): 10M ids fit nb = 512, i.e. an
// 8 KB table instead of 16 KB -- LDS is what bounds the walkers per CU.
//   id = q * nb + r.   tag = q + 1 (1..32767, so 0 can mean "empty").   Home bucket b1 = r ^ scramble(q), second
//   bucket b2 = b1 ^ g(q) with g odd (never b1).   (bucket, choice bit, tag) names the id exactly: q from the tag, r
//   from the bucket and q -- nothing is hashed away, the set is exact.
//   A bucket is 8 entries = 16 bytes = one ds_read_b128; entry = choice << 15 | tag; buckets fill front to back.
// The two lanes of a pair (walk_fast.h: both hold the same neighbor id) take one bucket each: one read, a packed
// 16-bit compare of the eight entries, the fill count; one DPP exchange decides present / which bucket is emptier
// (ties: b1); the lane that owns the chosen bucket claims the first free entry with ONE ds_cmpst on its 32-bit word.
// A claim fails only when another lane of the same expansion took that word in the same round (two ids sharing a
// bucket: ~0.4 pairs per expansion); those pairs go round again. Two-choice placement keeps every bucket below 8
// entries up to ~0.72 load (2,930 ids in 512 buckets, simulated: tools/model_visited16.py); an id that finds both
// of its buckets full goes to the walk's overflow table in global memory, exactly as with the 32-bit table, and is
// looked up there by every later pair that finds both of its buckets full. Buckets never lose entries, so an id is
// in the set iff it is in b1, in b2, or (both full) in the overflow table.
//
// Id spaces beyond 32767 * nb (125M ids would need 4096 buckets = 64 KB) keep the 32-bit table. Measured and dropped
// (round 3): 15-bit tags as a filter with the entries' full ids mirrored in the walk's global region and read back on a
// tag match. 97 % of a walk's lookups find no matching tag and never leave LDS, but every insert then costs a scattered
// 4-byte store (a read-modify-write at the memory side: as many transactions as the int8 row gather itself) whose
// acknowledgement every later vmcnt wait also waits for: 0.28 ms per launch against 0.185 ms with the 32-bit table on
// 40M int8 rows at max_search 50, no better than it at max_search 200.
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min1_u16(uint32_t a) { // min(each half, 1): 1 where the half is non-zero
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
// The lane id, computed where it is used. (Table wipes run once per layer; the compiler otherwise keeps their loop
// counters -- lane, lane + 64 -- alive through the whole walk, and under the f32 walkers' register pressure spills them.)
__device__ __forceinline__ uint32_t lane_id_here() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l));
    return l;
}
__device__ __forceinline__ uint32_t dpp_pair_swap(uint32_t v) { // the other lane of the pair (lane ^ 1)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
}

constexpr uint32_t V16_TAG_MAX = 32767u;   // 16-bit entries: tags 1..32767
constexpr uint32_t V20_TAG_MAX = 524287u;  // 20-bit entries: tags 1..524287
constexpr uint32_t V16_MIN_LG = 6;         // smallest table: 64 buckets = 1 KB (upper layers)
__host__ __device__ inline uint32_t v16_lg_for_ids(uint64_t n_ids, uint32_t tag_max = V16_TAG_MAX) { // smallest log2(nb) whose tags hold n_ids ids
    uint32_t lg = V16_MIN_LG;
    while (((uint64_t)tag_max << lg) < n_ids && lg < 31) ++lg;
    return lg;
}

// TB = bits of an entry: 16 (8 entries per bucket, tags of 15 bits: up to 32767 ids per bucket) or 20 (6 entries per bucket
// -- three per 64-bit half, claimed with a 64-bit ds_cmpst -- tags of 19 bits: up to 524286 ids per bucket, which puts
// the 125M-id shards of BASELINE.json's configs[4] into a 16 KB table). Everything but the probe is shared.
template <int TB>
struct VisitedSetB {
    static_assert(TB == 16 || TB == 20, "entries of 16 or 20 bits");
    static constexpr uint32_t PER_BUCKET = TB == 16 ? 8u : 6u;
    uint32_t* tab;     // LDS: nb buckets of 4 words
    uint32_t lg;       // log2(nb)
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t region;   // overflow region borrowed from the pool, or NONE
    uint32_t ocount;   // ids in the overflow table
    uint32_t count;    // statistics only
#if GRANNE_HIP_PHASE_TIMERS
    uint32_t pt_rounds = 0;
#endif

    __device__ __forceinline__ void init_walker() {
        region = NONE;
        ocount = 0;
    }
    __device__ __forceinline__ uint32_t acquire(const OverflowPool& pool, uint32_t lane) {
        uint32_t got = NONE;
        if (pool.slots != 0 && lane == 0) {
            uint32_t r = (blockIdx.x * 0x9E3779B1u) % pool.regions;
            for (uint32_t tries = 0; tries < pool.regions; ++tries) {
                if (atomicCAS(&pool.state[r], 0u, 1u) == 0u) { got = r; break; }
                r = (r + 1 == pool.regions) ? 0u : r + 1;
            }
        }
        return (uint32_t)__shfl((int)got, 0, 64);
    }
    __device__ __forceinline__ void reset(uint32_t* lds, uint32_t lg_, uint32_t lane) {
        tab = lds;
        lg = lg_;
        count = 0;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        uint4* t4 = reinterpret_cast<uint4*>(lds);
        (void)lane;
        for (uint32_t i = lane_id_here(); i < (1u << lg_); i += 64) t4[i] = z;
        if (region != NONE && ocount != 0) ocount = NONE; // a borrowed region is wiped before it is used again
    }
    __device__ __forceinline__ void added(uint32_t) {}

    // One probe of the pair's two buckets, written without branches (bit operations and selects; only the claim itself
    // runs under a mask): every lane executes it, lanes without `pending` change nothing. A lane with `pending` looks the
    // id up and, when it is absent and a bucket has room, claims the bucket's first free entry. Outcome per pair (the
    // same in both of its lanes):
    //   fresh     the id was inserted (it was not in the set)
    //   both_full it is in neither bucket and both are full: the overflow table decides
    // `pending` stays set for a pair that lost its entry to another pair of the same expansion; it goes round again (and
    // finds the id present if that other pair held the same id: a row that lists a neighbor twice).
    __device__ __forceinline__ void probe16(uint32_t h, uint4* slot4, uint32_t entry, bool& pending, bool& fresh, bool& both_full) {
        const uint4 w = *slot4;
        const uint32_t pat = entry | (entry << 16);
        const uint32_t m = pk_min_u16(pk_min_u16(w.x ^ pat, w.y ^ pat), pk_min_u16(w.z ^ pat, w.w ^ pat));
        const uint32_t match = (uint32_t)((m & 0xFFFFu) == 0u) | (uint32_t)(m < 0x10000u);
        const uint32_t c2 = pk_add_u16(pk_add_u16(pk_min1_u16(w.x), pk_min1_u16(w.y)),
                                       pk_add_u16(pk_min1_u16(w.z), pk_min1_u16(w.w)));
        const uint32_t cnt = (c2 & 0xFFFFu) + (c2 >> 16);
        // one exchange: fill count in the low bits, "the tag is here" above them
        const uint32_t other = dpp_pair_swap(cnt | (match << 8));
        const uint32_t cnt_o = other & 0xFFu;
        const bool present = (match | (other >> 8)) != 0u; // the tag names the id: it is in the set
        const bool full2 = (cnt >= 8u) & (cnt_o >= 8u);
        // this lane's bucket is the emptier one (ties: b1): it claims the bucket's first free entry
        const bool mine_emptier = h ? (cnt < cnt_o) : (cnt <= cnt_o);
        const bool claim = pending & !present & !full2 & mine_emptier;
        const uint32_t wi = cnt >> 1;
        const uint32_t old = wi == 0u ? w.x : wi == 1u ? w.y : wi == 2u ? w.z : w.w;
        uint32_t ok = 0u;
        if (claim) {
            const uint32_t got = atomicCAS(reinterpret_cast<uint32_t*>(slot4) + (wi & 3u), old, old | (entry << ((cnt & 1u) * 16u)));
            ok = got == old ? 1u : 0u;
        }
        const bool okp = (ok | dpp_pair_swap(ok)) != 0u;
        fresh = fresh | (pending & okp);
        both_full = both_full | (pending & !present & full2);
        pending = pending & !present & !full2 & !okp;
    }

    // the same with 20-bit entries: a bucket is two 64-bit halves of three entries each (bits 0-19, 20-39, 40-59)
    __device__ __forceinline__ void probe20(uint32_t h, uint4* slot4, uint32_t entry, bool& pending, bool& fresh, bool& both_full) {
        const uint4 w = *slot4;
        const uint32_t M = 0xFFFFFu;
        const uint32_t f0 = w.x & M, f1 = __builtin_amdgcn_alignbit(w.y, w.x, 20) & M, f2 = (w.y >> 8) & M;
        const uint32_t f3 = w.z & M, f4 = __builtin_amdgcn_alignbit(w.w, w.z, 20) & M, f5 = (w.w >> 8) & M;
        const uint32_t match = (uint32_t)(f0 == entry) | (uint32_t)(f1 == entry) | (uint32_t)(f2 == entry) |
                               (uint32_t)(f3 == entry) | (uint32_t)(f4 == entry) | (uint32_t)(f5 == entry);
        const uint32_t cnt = (uint32_t)(f0 != 0u) + (uint32_t)(f1 != 0u) + (uint32_t)(f2 != 0u) + (uint32_t)(f3 != 0u) +
                             (uint32_t)(f4 != 0u) + (uint32_t)(f5 != 0u); // entries fill front to back
        const uint32_t other = dpp_pair_swap(cnt | (match << 8));
        const uint32_t cnt_o = other & 0xFFu;
        const bool present = (match | (other >> 8)) != 0u;
        const bool full2 = (cnt >= 6u) & (cnt_o >= 6u);
        const bool mine_emptier = h ? (cnt < cnt_o) : (cnt <= cnt_o);
        const bool claim = pending & !present & !full2 & mine_emptier;
        const bool hi = cnt >= 3u;
        const uint32_t pos = hi ? cnt - 3u : cnt;
        const uint64_t old = hi ? (((uint64_t)w.w << 32) | w.z) : (((uint64_t)w.y << 32) | w.x);
        uint32_t ok = 0u;
        if (claim) {
            unsigned long long* half = reinterpret_cast<unsigned long long*>(slot4) + (hi ? 1 : 0);
            const unsigned long long got = atomicCAS(half, (unsigned long long)old, (unsigned long long)(old | ((uint64_t)entry << (pos * 20u))));
            ok = got == old ? 1u : 0u;
        }
        const bool okp = (ok | dpp_pair_swap(ok)) != 0u;
        fresh = fresh | (pending & okp);
        both_full = both_full | (pending & !present & full2);
        pending = pending & !present & !full2 & !okp;
    }
    __device__ __forceinline__ void probe(uint32_t h, uint4* slot4, uint32_t entry, bool& pending, bool& fresh, bool& both_full) {
        if constexpr (TB == 16) probe16(h, slot4, entry, pending, fresh, both_full);
        else probe20(h, slot4, entry, pending, fresh, both_full);
    }

    // HashSet::insert for the id both lanes of a pair hold (h = lane & 1). `active` is the same in both lanes.
    // Returns true in BOTH lanes iff the id was not present.
    __device__ __forceinline__ bool insert(uint32_t id, bool active, uint32_t h, const OverflowPool& pool, uint32_t lane,
                                           bool& bail) {
        const uint32_t sh = 32u - lg;
        const uint32_t q = id >> lg;
        const uint32_t b1 = (id ^ ((q * 0x9E3779B1u) >> sh)) & ((1u << lg) - 1u);
        const uint32_t mine = h ? (b1 ^ (((q * 0x85EBCA6Bu) >> sh) | 1u)) : b1;
        // tag: q + 1 (never 0: 0 is "empty"); the host sizes the table so that it fits the entry's tag bits
        const uint32_t entry = (q + 1u) | (h << (TB - 1));
        uint4* slot4 = reinterpret_cast<uint4*>(tab) + mine;
        bool pending = active, fresh = false, both_full = false;
#if GRANNE_HIP_PHASE_TIMERS
        uint32_t r_ = 0;
#endif
        while (wave_ballot(pending)) {
#if GRANNE_HIP_PHASE_TIMERS
            r_ += 1;
#endif
            probe(h, slot4, entry, pending, fresh, both_full);
        }
#if GRANNE_HIP_PHASE_TIMERS
        pt_rounds += r_;
#endif
        if (wave_ballot(both_full)) { // rare: the walk outgrew its table's two-choice capacity
            if (region == NONE) {
                region = acquire(pool, lane);
                ocount = NONE;
            }
            if (region == NONE) {
                bail = true; // no overflow configured or none left: the exact global-memory walker takes the query
                return false;
            }
            uint32_t* otab = pool.tables + (size_t)region * pool.stride;
            if (ocount == NONE) { // first use (in this layer): wipe
                if (lane == 0 && pool.spilled) atomicAdd(pool.spilled, 1u);
                const uint4 e = make_uint4(ID_EMPTY, ID_EMPTY, ID_EMPTY, ID_EMPTY);
                uint4* t4 = reinterpret_cast<uint4*>(otab);
                for (uint32_t i = lane_id_here(); i < (pool.slots >> 2); i += 64) t4[i] = e;
                __threadfence();
                ocount = 0;
            }
            bool ofresh = false;
            if (both_full && h == 0u) {
                const uint32_t omask = pool.slots - 1;
                const uint32_t ost = VisitedSet::step(id);
                uint32_t slot = (VisitedSet::hash(id) >> 3) & omask;
                for (;;) {
                    const uint32_t old = atomicCAS(&otab[slot], ID_EMPTY, id);
                    if (old == ID_EMPTY) { ofresh = true; break; }
                    if (old == id) break;
                    slot = (slot + ost) & omask;
                }
            }
            const uint64_t om = wave_ballot(ofresh);
            ocount += (uint32_t)__popcll(om);
            if (((om | (om << 1)) >> lane) & 1ull) fresh = true; // both lanes of the pair
        }
        return fresh;
    }
    // after an expansion: false when the overflow table itself is full (75 % load)
    __device__ __forceinline__ bool make_room(const OverflowPool& pool, uint32_t) {
        return region == NONE || ocount == NONE || ocount <= pool.slots - (pool.slots >> 2) - 72u;
    }
    __device__ __forceinline__ void release(const OverflowPool& pool, uint32_t lane) {
        if (region != NONE) {
            __threadfence();
            if (lane == 0) atomicExch(&pool.state[region], 0u);
            region = NONE;
        }
    }
};

typedef VisitedSetB<16> VisitedSet16;
typedef VisitedSetB<20> VisitedSet20;

// No visited set at all (the register walkers with lists of up to 256 keys: FastWalker<.., V16 = 3>).
// The reference's HashSet (mod.rs:1008,1016,1026) keeps a node from being evaluated twice; in a walk whose lanes
// evaluate a whole adjacency row at once -- rows requested before anything is known about their ids -- a second
// evaluation costs nothing that was not already spent, and what it must not do is enter the list twice. It cannot:
//  * a node rejected by `distance < res.peek()` (mod.rs:1029) is rejected again, res.peek() only decreases;
//  * a node that is in the list (expanded or not, including the entries past max_search) is found there by id;
//  * a node that was pushed off the list's end has CAP keys below its own and the same key again: it falls off again.
// So the results are the reference's, bit for bit, and the set's LDS, its instructions, its overflow tables and its
// hand-overs are gone; WalkStats::n_dist counts evaluations (revisits included), no longer distinct nodes.
struct VisitedNone {
    uint32_t count = 0;
    uint32_t pt_rounds = 0;
    __device__ __forceinline__ void init_walker() {}
    __device__ __forceinline__ void added(uint32_t) {}
    __device__ __forceinline__ bool make_room(const OverflowPool&, uint32_t) { return true; }
    __device__ __forceinline__ void release(const OverflowPool&, uint32_t) {}
};

} // namespace granne_hip
