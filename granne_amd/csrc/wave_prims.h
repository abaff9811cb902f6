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

__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); } // (__ballot takes an int: a 0/1 select and a second compare)

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

// (Rounds 3a-3 also carried two-choice bucket tables of 16- and 20-bit entries -- half the LDS per id, exact through
//  (bucket, choice bit, tag) -- as a second and third form of this set. They bought residency while a set was the default;
//  with no set at all as the default (VisitedNone below) the only job left for an exact set is to count the reference's
//  n_dist for bench.py and the tests, which the 32-bit table above does for every list length and id range. Retired in
//  round 4; profiles/r3a_* keep their measurements.)

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
