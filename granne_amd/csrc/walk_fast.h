// walk_fast.h -- the walker of the common shapes (every layer 32 ids wide on the device; f32 rows
// of a compile-time or streamed dim, or int8 rows of 128 / 256 / 512 bytes): one wavefront per query, all layers in one
// launch (Granne::search -> search_internal -> find_entrypoint -> search_for_neighbors,
// /root/reference/src/index/mod.rs:140-150, 963-1037).
//
// Lane layout: lane = 2*R + h. R (0..31) is a neighbor slot of the expanded node, the two lanes of a
// pair hold that neighbor's element row, h selecting the 64-byte half of every 128-byte block.
//  f32: lane (R,h) owns accumulators 16h..16h+15 of the reference's 32 (src/math.rs:17-26) and
//       applies the 32-wide chunks in order with explicit fmas; the ordered sum
//       ((0 + acc[0]) + acc[1]) + ... + acc[31] (math.rs:27-30) is 16 adds in the even lane, one
//       DPP row_shr:1 hand-over, 16 adds in the odd lane; the tail fmas (math.rs:32-39) follow in
//       the odd lane. 33 VALU ops for the sum of 32 rows, no LDS, no shuffle network.
//  i8:  v_dot4_i32_i8 partial sums of r and dx per half row, one quad_perm DPP exchange each, then
//       the reference's float tail (angular_int.rs:52-58) once per lane; sqrt(dy) is the query's
//       and is taken once per query (same operation on the same input: same bits).
// All row loads of an expansion are issued from the neighbor ids alone. The walker keeps NO visited set by default
// (V16 = 3; wave_prims.h VisitedNone says why the results stay the reference's): every neighbor is evaluated and a
// candidate that passed the filter is looked up in the list before it is inserted. With the exact set switched on
// (V16 = 0: LDS front table + global overflow) its look-ups run under the row loads.
//
// The reference's two heaps (`res`: the max_search best popped nodes, `pq`: the unbounded
// candidate queue, mod.rs:1006-1007) are ONE ascending list of 64*S keys in registers, each key
// (dist bits | id | expanded flag):
//   next node to expand = the first entry whose flag is clear (= pq.pop(): the smallest queued key);
//   break  <=>  #{entries with dist < d_x} >= max_search. Every such entry precedes x and is
//               therefore expanded, so this is `res.len() == max_search && d_x > res.peek().dist`
//               (mod.rs:1019): the max_search-th smallest popped distance is strictly smaller;
//   res.push(x) = set the flag: `res` IS the first max_search flagged entries of the list;
//   the enqueue filter `res.len() < max_search || d < res.peek().dist` (mod.rs:1029) reads the
//               max_search-th flagged entry when the list holds that many;
//   a candidate with max_search entries strictly closer can never be expanded (when it would reach the
//               head, max_search closer nodes have been popped and the loop breaks), so it is not
//               inserted; an entry pushed off the end of the list is dead for the same reason unless
//               its distance TIES with entry max_search-1 -- then the walk is abandoned and the
//               query handed, untouched, to the exact global-memory walker (slow_kernel.h).
// tools/model_unified.py replays this list against the two heaps on tie-heavy random graphs.
#pragma once

#include <type_traits>

#include "slow_kernel.h"

namespace granne_hip {

// Diagnostics build (-DGRANNE_HIP_PHASE_TIMERS=1, tools/phase_probe.py): s_memtime stamps around the phases of
// an expansion, summed per walk into a device array the host reads back. Never part of the shipped library.
#ifndef GRANNE_HIP_PHASE_TIMERS
#define GRANNE_HIP_PHASE_TIMERS 0
#endif
#if GRANNE_HIP_PHASE_TIMERS
constexpr uint32_t PHASE_SLOTS = 48, PHASE_QUERIES = 4096;
__device__ uint64_t g_phase[PHASE_QUERIES * PHASE_SLOTS];
#define PT_MARK(i)                                                                   \
    do {                                                                             \
        asm volatile("" ::: "memory");                                               \
        __builtin_amdgcn_sched_barrier(0);                                           \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();                          \
        const uint64_t dt_ = now_ - pt_last;                                         \
        if (pt_bottom) pt_b[i] += dt_; else pt_u[i] += dt_;                          \
        pt_last = now_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("" ::: "memory");                                               \
    } while (0)
#define PT_RESET()                                                                   \
    do {                                                                             \
        asm volatile("" ::: "memory");                                               \
        pt_last = __builtin_amdgcn_s_memtime();                                      \
        asm volatile("" ::: "memory");                                               \
    } while (0)
#define PT_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PT_COUNT() do { if (pt_bottom) pt_nb += 1; else pt_nu += 1; } while (0)
#define PT_PIN(v) asm volatile("" ::"v"(v))
#define PT_ADD(i, n) do { if (pt_bottom) pt_cnt[i] += (n); } while (0)
#else
#define PT_MARK(i) do {} while (0)
#define PT_RESET() do {} while (0)
#define PT_WAIT_VM() do {} while (0)
#define PT_COUNT() do {} while (0)
#define PT_PIN(v) do {} while (0)
#define PT_ADD(i, n) do {} while (0)
#endif

// ---- list keys: dist bits (32) | id (31) | expanded (1) ----------------------------------------
__device__ __forceinline__ uint64_t wkey(float d, uint32_t id) {
    return ((uint64_t)__float_as_uint(d) << 32) | ((uint64_t)id << 1);
}
__device__ __forceinline__ uint32_t wkey_hi(uint64_t k) { return (uint32_t)(k >> 32); }
__device__ __forceinline__ uint32_t wkey_id(uint64_t k) { return ((uint32_t)k) >> 1; }
__device__ __forceinline__ float wkey_dist(uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); }
constexpr uint32_t WPOS_NONE = 0xFFFFFFFFu;
constexpr uint32_t VCACHE_SLOTS = 512; // FastWalker::vcache (power of two, 2 KB of LDS per walker)
constexpr uint32_t VCACHE_SLOTS_SEEN = 2048; // ... of the walkers that skip revisits before their rows are fetched (SEEN: 8 KB; f32 rows -- 12 walkers per CU)
constexpr uint64_t WALK_MAX_ELEMENTS = 1ull << 31; // ids must fit 31 bits

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) { // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// value of the other lane of the pair (lane ^ 1)
__device__ __forceinline__ int pair_swap(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
}
// lane i receives lane i-1's value within its row of 16 (odd lanes: their even partner)
__device__ __forceinline__ float from_lower_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

// lane i receives lane i-1's value, lane 0 receives 0
__device__ __forceinline__ uint32_t from_prev_lane_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}

// From how many slots on a list is the two-level list of FastWalker::search_layer_long (M in LDS only, F in registers).
#ifndef GRANNE_HIP_LONG_MIN
#define GRANNE_HIP_LONG_MIN 33
#endif
constexpr bool walk_list_is_long(int S, bool wide) { return S >= GRANNE_HIP_LONG_MIN && !wide; } // (layers of 64 ids: two passes per expansion, the short lists only)

template <int S, bool LONGF = walk_list_is_long(S, false)>
struct WalkList : SortedList<S> {
    using SortedList<S>::key;
    static constexpr uint32_t CAP = 64u * S;

    static constexpr bool LONG = LONGF; // the two-level lists live in LDS only (FastWalker::search_layer_long): `key` stays unused
    // position of the first entry whose expanded flag is clear (KEY_INF has it set)
    __device__ __forceinline__ bool first_unexpanded(uint32_t& pos) const {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint64_t um = wave_ballot((((uint32_t)key[s]) & 1u) == 0u);
            if (um) {
                pos = (uint32_t)s * 64u + (uint32_t)__builtin_ctzll(um);
                return true;
            }
        }
        return false;
    }
    // number of entries whose distance is strictly smaller than dbits (wave-uniform)
    __device__ __forceinline__ uint32_t count_closer(uint32_t dbits) const {
        uint32_t c = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) c += (uint32_t)__popcll(wave_ballot(wkey_hi(key[s]) < dbits));
        return c;
    }
    // Every list has an image in LDS (`mir`): lists of up to 17 slots merge through it (FastWalker::merge_ranked: CAP keys +
    // 32 places for what an expansion's candidates push off the end) and it holds exactly the list after every merge and
    // every flag change; the long lists' bulk merge scatters through it too. Reading entry e (wave-uniform index) out of
    // the registers means selecting a register by a run-time slot: short lists do it with a tree of uniform branches (two
    // readlanes at the leaf), lists of S >= 8 read the image: one broadcast ds_read instead of 4*S scalar selects.
    static constexpr bool MIRROR = S >= 8;
    static constexpr uint32_t IMAGE_KEYS = CAP + (LONG ? 0u : 32u);
    uint64_t* mir;
    template <int LO, int HI>
    __device__ __forceinline__ uint64_t get_rec(uint32_t slot, uint32_t l) const {
        if constexpr (HI - LO == 1) {
            return readlane64(key[LO], l);
        } else {
            constexpr int MID = (LO + HI) / 2;
            return slot < (uint32_t)MID ? get_rec<LO, MID>(slot, l) : get_rec<MID, HI>(slot, l);
        }
    }
    __device__ __forceinline__ uint64_t at(uint32_t e) const {
        if constexpr (MIRROR) {
            const uint64_t v = mir[e];
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
            return ((uint64_t)hi << 32) | lo;
        } else {
            return get_rec<0, S>(e >> 6, e & 63u);
        }
    }
    // set the expanded flag of entry pos, whose key is x
    __device__ __forceinline__ void mark_expanded(uint32_t pos, uint64_t x, uint32_t lane) {
#pragma unroll
        for (int s = 0; s < S; ++s) key[s] |= ((uint32_t)s * 64u + lane == pos) ? 1ull : 0ull;
        if constexpr (MIRROR) {
            if (lane == 0) mir[pos] = x | 1ull;
        }
    }
    __device__ __forceinline__ void init_list(uint64_t* mirror, uint32_t lane) {
        mir = mirror;
        if constexpr (LONG) {
#pragma unroll 1
            for (uint32_t s = 0; s < (uint32_t)S; ++s) mir[s * 64u + lane] = KEY_INF;
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                key[s] = KEY_INF;
                mir[(uint32_t)s * 64u + lane] = KEY_INF; // the image holds the list from the start
            }
        }
    }
    // the first entry of an empty list
    __device__ __forceinline__ void set_first(uint64_t k0, uint32_t lane) {
        if (lane == 0) {
            if constexpr (!LONG) key[0] = k0;
            mir[0] = k0;
        }
    }
    // position of the n-th (n >= 1) expanded real entry, WPOS_NONE when there are fewer
    __device__ __forceinline__ uint32_t nth_expanded(uint32_t n) const {
        uint32_t before = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool f = (((uint32_t)key[s]) & 1u) && wkey_hi(key[s]) != 0xFFFFFFFFu;
            const uint64_t em = wave_ballot(f);
            const uint32_t c = (uint32_t)__popcll(em);
            if (before + c >= n) {
                const uint64_t hit = wave_ballot(f && mbcnt64(em) == n - 1u - before);
                return (uint32_t)s * 64u + (uint32_t)__builtin_ctzll(hit);
            }
            before += c;
        }
        return WPOS_NONE;
    }
};

#ifndef GRANNE_HIP_QUERY_IN_LDS
#define GRANNE_HIP_QUERY_IN_LDS 0 // experiments: 1 = the short list reads an f32 query from LDS too
#endif
// A query that lives in registers needs no LDS of its own: int8 rows (64 bytes per lane; staged once through the
// start of the walker's LDS) and, for the short list, f32 rows of the unrolled
// dims (this lane's pieces in VGPRs; longer lists need the registers and read the query from LDS, 13 ds_read_b128
// per expansion at 100-d, issued under the row loads). Without a visited table the walker's LDS is that staging area
// (and the mirror of lists of S >= 8): the registers bound the walkers per CU.
__host__ __device__ constexpr bool fast_query_in_regs(bool i8, bool gen, uint32_t dim, uint32_t S) {
    if (i8) return true;
    if (gen || S != 1u || GRANNE_HIP_QUERY_IN_LDS) return false;
    return (dim / 32u) * 16u + ((dim % 32u) / 4u) * 4u <= 64u;
}
// LDS bytes of the query. f32 rows of a compile-time dim: the row itself. Run-time f32 dims (DIM == 0): full
// 32-float chunks padded with zero chunks to a whole number of groups of three, then one 128-byte tail block.
constexpr uint32_t GEN_GROUP = 3; // chunks whose loads are in flight together (12 x 16 bytes per lane)
__host__ __device__ inline uint32_t fast_query_bytes(bool i8, bool gen, uint32_t dim, uint32_t row_bytes, uint32_t S) {
    if (fast_query_in_regs(i8, gen, dim, S)) return 0u;
    if (!gen) return lds_query_bytes(row_bytes);
    const uint32_t nbk = dim / 32u;
    const uint32_t ngroups = (nbk + GEN_GROUP - 1u) / GEN_GROUP;
    return ngroups * GEN_GROUP * 128u + 128u;
}

// V16: the form of the visited set. 3 = none (the default; 4 = none + rows touched ahead, for launches of a few
// queries); 0 = the exact set: 32-bit open addressing in LDS + a global overflow table (VisitedSet, wave_prims.h) --
// kept so that n_dist can be counted the way the reference counts it. The host picks (plan_launch, granne_hip.hip).
// WIDE: layers of up to 64 ids per node (graphs built with num_neighbors 33..63): an expansion takes the row's ids in two
// passes of 32 pairs -- rows, distances, filter, insert for ids 0..31, then for ids 32..63 when the row goes that far.
// `res` does not change within an expansion (mod.rs:1025-1033 pushes to pq only), so the reference's filter gives every
// neighbor of the row the same answer whichever pass it is in; the list's own bookkeeping (dead candidates beyond entry
// max_search-1, the tie test after a pass's last insert) is per insert group and does not care where a group ends.
template <int DT, int DIM, int S, int V16 = 0, bool WIDE = false>
struct FastWalker {
    static constexpr bool F32 = (DT == DT_F32);
    // DIM == 0: any f32 dim, known at run time (below 32 the row is its tail only). The chunks stream through the registers in groups of
    // GEN_GROUP (the first group's loads are the ones issued ahead), the query is read from LDS.
    static constexpr bool GEN = F32 && DIM == 0;
    static constexpr int NB = F32 ? (GEN ? (int)GEN_GROUP : DIM / 32) : 0; // full 32-float chunks (GEN: per group)
    static constexpr int TU = F32 ? (DIM % 32) / 4 : 0;  // 16-byte units of the tail
    static constexpr bool QREG = F32 && fast_query_in_regs(false, GEN, (uint32_t)DIM, (uint32_t)S);
    static constexpr int NBI = F32 ? 1 : (DIM ? DIM / 128 : 1); // int8: 128-byte blocks per row (DIM = row bytes; 0 = 128)
    static constexpr uint32_t ROWB = F32 ? (uint32_t)DIM * 4u : 128u * NBI;
    // bytes from one row to the next (device_row_stride, granne_hip.hip): f32 rows of 256 bytes and more start on a line
    static constexpr uint32_t RSTRIDE = (F32 && ROWB >= 256u) ? ((ROWB + 127u) & ~127u) : ROWB;
    // XT: the shape whose layers may carry the neighbors' tails next to their ids (LayerDev::adjx): the unrolled f32 dims
    // with a tail, layers of 32 ids. Then an expansion reads the 32-float chunks of a neighbor from its row (whole lines)
    // and its tail from the expanded node's own record, XTAILB bytes per neighbor slot behind the 32 ids.
    static constexpr bool XT = F32 && !GEN && TU > 0 && !WIDE;
    static constexpr uint32_t XTAILB = (uint32_t)TU * 16u;
    static_assert(F32 || DIM == 0 || DIM == 256 || DIM == 512, "fast int8 rows: 128, 256 or 512 bytes");
    static constexpr uint32_t CAP = 64u * S;
    using List = WalkList<S, walk_list_is_long(S, WIDE)>;
    static constexpr bool LONG_LIST = List::LONG;
    static constexpr int NT = ROWB ? (int)((ROWB + 255u) / 256u) : 1; // TOUCH: lines of a row per lane of its pair
    static_assert(!F32 || GEN || (DIM % 4 == 0 && DIM >= 32), "fast f32 rows: dim a multiple of 4, at least one chunk");

    const SearchParams& p;
    uint32_t lane, h, R;
    uint8_t* lds_q;       // the query (f32 without QREG: read per expansion; i8: staging, shared with the visited table)
    uint64_t* mslot;      // lists with an LDS mirror (S >= 8): CAP keys, the mirror = the scatter space of the bulk merge
    uint32_t* vis_tab;
    uint32_t* vcache;     // NOVIS, lists of up to 17 slots: ids that entered the list (a direct-mapped cache, see seen_before)
    float qh[QREG ? NB * 16 : 1];
    float qt[(QREG && TU) ? TU * 4 : 1];
    uint4 qi8[F32 ? 1 : 4 * NBI]; // i8: bytes 64h..64h+63 of every 128-byte block of the query
    float sy;               // i8: sqrt(sum of squares of the query) as f32
    uint32_t g_nbk, g_ngroups, g_tu; // GEN: full chunks, groups of them, 16-byte units of the (zero padded) tail
    static constexpr bool NOVIS = V16 >= 3; // no visited set: the list is searched for a candidate's id (VisitedNone, wave_prims.h)
    // V16 == 4, launches of a few queries on an otherwise idle chip (one query per call is the reference's own shape):
    // the rows of the neighbors of the node that is first in line are touched one expansion early -- one load per
    // 128-byte line, data dropped -- so that, when that node is expanded next (half of the time), its rows come from L2.
    // With a thousand walks in a launch the same touches cost throughput and are not compiled in.
    static constexpr bool TOUCH = V16 == 4;
    // V16 == 5, launches of many walks (the bandwidth-bound shapes): the cache of ids is consulted BEFORE the row loads and
    // holds every id the walk EVALUATED (not only the ones that entered the list): a hit is the reference's
    // `!visited.insert(n)` (mod.rs:1026) -- the neighbor is skipped, its row is not fetched. A miss (never seen, or pushed
    // out of its slot) means nothing: the row is evaluated (again). On i.i.d.-uniform data 3.6 % of the rows a walk
    // fetches are revisits, on clustered data 40 %; the look-up's LDS round trip sits between the adjacency row and the
    // row loads, which a lone wave per SIMD would pay for on every expansion -- hence only for the launches that are
    // bound by bandwidth, not by one wave's latency.
    static constexpr bool SEEN = V16 == 5;
    static_assert(!WIDE || V16 == 3, "layers of 64 ids: walked without a visited set only");
    static_assert(V16 == 0 || V16 == 3 || V16 == 4 || V16 == 5, "the exact 32-bit table, or no visited set (4: + rows touched ahead, 5: + revisits skipped before their rows are fetched)");
    typename std::conditional<V16 == 0, VisitedSet, VisitedNone>::type vis;
    List L;
    WalkStats st;
    bool bail;
    uint32_t theta; // distance bits of list entry max_search-1 (0xFFFFFFFF while the list is shorter): kept by insert()
#if GRANNE_HIP_PHASE_TIMERS
    uint64_t pt_b[16] = {}, pt_u[16] = {}, pt_last = 0, pt_t0 = 0;
    uint32_t pt_nb = 0, pt_nu = 0, pt_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool pt_bottom = false;
#endif

    __device__ __forceinline__ FastWalker(const SearchParams& p_, uint8_t* smem) : p(p_) {
        lane = threadIdx.x;
        h = lane & 1u;
        R = lane >> 1;
        const uint32_t qb = fast_query_bytes(!F32, GEN, p.dim, p.row_bytes, (uint32_t)S);
        g_nbk = p.dim / 32u;
        g_ngroups = (g_nbk + GEN_GROUP - 1u) / GEN_GROUP;
        g_tu = ((p.dim & 31u) + 3u) / 4u;
        lds_q = smem;
        mslot = reinterpret_cast<uint64_t*>(smem + qb);
        // [query][the list's image (lists beyond 1024 keys: M)][those lists: F's image][the cache of entered ids][visited]
        fimg = reinterpret_cast<uint64_t*>(smem + qb + List::IMAGE_KEYS * 8u);
        vcache = reinterpret_cast<uint32_t*>(smem + qb + List::IMAGE_KEYS * 8u + (LONG_LIST ? FIMG_KEYS * 8u : 0u));
        fkey = KEY_INF;
        nF = nM = 0;
        m_un = 0;
        m_un_key = KEY_INF;
        lost_bits = 0xFFFFFFFFu;
        vis_tab = reinterpret_cast<uint32_t*>(smem + qb + List::IMAGE_KEYS * 8u + (LONG_LIST ? FIMG_KEYS * 8u : 0u) + VSLOTS * 4u);
        st.n_dist = st.n_expand = st.n_adj = 0;
        bail = false;
        sy = 0.0f;
        vis.init_walker();
    }

    __device__ __forceinline__ void load_query(uint32_t qi) {
        if constexpr (F32) {
            const float* q = reinterpret_cast<const float*>(query_io(p, qi).q);
            if constexpr (QREG) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int j = 0; j < 16; ++j) qh[b * 16 + j] = q[b * 32 + h * 16 + j];
#pragma unroll
                for (int j = 0; j < TU * 4; ++j) qt[j] = q[NB * 32 + j];
            } else if constexpr (GEN) {
                float* l = reinterpret_cast<float*>(lds_q);
                const uint32_t full = g_nbk * 32u, padded = g_ngroups * GEN_GROUP * 32u;
                for (uint32_t i = lane; i < padded; i += 64) l[i] = (i < full) ? q[i] : 0.0f;
                if (lane < 32u) l[padded + lane] = (full + lane < p.dim) ? q[full + lane] : 0.0f;
                __syncthreads();
            } else {
                float* l = reinterpret_cast<float*>(lds_q);
                for (uint32_t i = lane; i < (uint32_t)DIM; i += 64) l[i] = q[i];
                __syncthreads();
            }
        } else {
            const int8_t* q = reinterpret_cast<const int8_t*>(query_io(p, qi).q);
            int8_t* l = reinterpret_cast<int8_t*>(lds_q);
            int part = 0;
            for (uint32_t i = lane; i < ROWB; i += 64) {
                const int v = (i < p.dim) ? (int)q[i] : 0;
                l[i] = (int8_t)v;
                part += v * v;
            }
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            sy = __builtin_sqrtf((float)part); // sqrt(dy), angular_int.rs:53
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NBI; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) qi8[b * 4 + k] = *reinterpret_cast<const uint4*>(lds_q + b * 128u + h * 64u + k * 16u);
        }
    }

    // The element rows of one expansion in flight: lane (R,h) holds half h of row R.
    struct RowRegs {
        float4 v[NB ? NB : 1][4];
        float4 vt[GEN ? 8 : (TU ? TU : 1)]; // GEN: the (zero padded) tail, up to 31 floats
        uint4 x[4 * NBI];
        const uint8_t* row; // GEN: the row, for the groups and the tail that finish_rows loads itself
    };

    // GEN: the loads of chunks 3g..3g+2 of the row (chunks past the row's last re-read the last one; their
    // query chunk in LDS is zero, so they add +-0 to accumulators whose zero sign never reaches the result)
    __device__ __forceinline__ void load_group(RowRegs& rr, uint32_t g) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t blk = min(g * GEN_GROUP + (uint32_t)b, g_nbk - 1u);
            const uint8_t* e = rr.row + (size_t)blk * 128u + h * 64u;
#pragma unroll
            for (int k = 0; k < 4; ++k) rr.v[b][k] = *reinterpret_cast<const float4*>(e + k * 16);
        }
    }

    // Issue every load of the rows idl (lanes of a pair pass the same idl). Nothing is waited for.
    // tails: where this lane's row keeps its tail when that is not the row itself (XT: the expanded node's record in
    // LayerDev::adjx; null = the row).
    __device__ __forceinline__ void issue_rows(uint32_t idl, RowRegs& rr, [[maybe_unused]] const uint8_t* tails = nullptr) {
        if constexpr (GEN) {
            rr.row = p.elements + (size_t)idl * p.row_stride;
            if (g_nbk) load_group(rr, 0u); // (dims below 32 have no full chunk: the row is its tail)
            // the tail (dim % 32 floats, zero padded to 16-byte units) is read by every lane, used by the odd one
            const uint8_t* tailp = rr.row + (size_t)g_nbk * 128u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if ((uint32_t)u < g_tu) rr.vt[u] = *reinterpret_cast<const float4*>(tailp + u * 16);
        } else if constexpr (F32) {
            const uint8_t* row = p.elements + (size_t)idl * RSTRIDE;
            const uint8_t* e = row + h * 64u;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) rr.v[b][k] = *reinterpret_cast<const float4*>(e + b * 128 + k * 16);
            if constexpr (XT) {
                const uint8_t* tp = tails ? tails : row + NB * 128;
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const uint4 t = load_global_u4((gptr_u32)(tp + u * 16));
                    rr.vt[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
                }
            } else {
#pragma unroll
                for (int u = 0; u < TU; ++u) rr.vt[u] = *reinterpret_cast<const float4*>(row + NB * 128 + u * 16);
            }
        } else {
            const uint8_t* e = p.elements + (size_t)idl * RSTRIDE + h * 64u;
#pragma unroll
            for (int b = 0; b < NBI; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) rr.x[b * 4 + k] = *reinterpret_cast<const uint4*>(e + b * 128 + k * 16);
        }
        asm volatile("" ::: "memory"); // the loads are issued here, whatever follows runs under them
    }

    // Distances of the rows in rr to the query: valid in ODD lanes.
    __device__ __forceinline__ float finish_rows(RowRegs& rr) {
        float d;
        if constexpr (GEN) {
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
            for (uint32_t g = 0; g < g_ngroups; ++g) {
                if (g > 0) load_group(rr, g); // group 0 was issued by issue_rows
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 qq = *reinterpret_cast<const float4*>(lds_q + (size_t)(g * GEN_GROUP + (uint32_t)b) * 128u + h * 64u + k * 16);
                        acc[k * 4 + 0] = __builtin_fmaf(rr.v[b][k].x, qq.x, acc[k * 4 + 0]);
                        acc[k * 4 + 1] = __builtin_fmaf(rr.v[b][k].y, qq.y, acc[k * 4 + 1]);
                        acc[k * 4 + 2] = __builtin_fmaf(rr.v[b][k].z, qq.z, acc[k * 4 + 2]);
                        acc[k * 4 + 3] = __builtin_fmaf(rr.v[b][k].w, qq.w, acc[k * 4 + 3]);
                    }
                }
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s = s + acc[j];
            float r = from_lower_f(s);
#pragma unroll
            for (int j = 0; j < 16; ++j) r = r + acc[j];
            const uint8_t* qtail = lds_q + (size_t)g_ngroups * GEN_GROUP * 128u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if ((uint32_t)u < g_tu) {
                    const float4 qq = *reinterpret_cast<const float4*>(qtail + u * 16);
                    r = __builtin_fmaf(rr.vt[u].x, qq.x, r);
                    r = __builtin_fmaf(rr.vt[u].y, qq.y, r);
                    r = __builtin_fmaf(rr.vt[u].z, qq.z, r);
                    r = __builtin_fmaf(rr.vt[u].w, qq.w, r);
                }
            }
            d = angular_from_dot(r);
        } else if constexpr (F32) {
            // the 16 accumulators of this lane, two to a register pair: v_pk_fma_f32 applies a chunk's 16 fused
            // multiply-adds in 8 instructions -- each half is the same IEEE fma as the scalar form (math.rs:20-25)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 acc2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc2[j] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float q0, q1, q2, q3;
                    if constexpr (QREG) {
                        q0 = qh[b * 16 + k * 4 + 0]; q1 = qh[b * 16 + k * 4 + 1];
                        q2 = qh[b * 16 + k * 4 + 2]; q3 = qh[b * 16 + k * 4 + 3];
                    } else {
                        const float4 qq = *reinterpret_cast<const float4*>(lds_q + b * 128 + h * 64u + k * 16);
                        q0 = qq.x; q1 = qq.y; q2 = qq.z; q3 = qq.w;
                    }
                    acc2[k * 2 + 0] = __builtin_elementwise_fma(f32x2{rr.v[b][k].x, rr.v[b][k].y}, f32x2{q0, q1}, acc2[k * 2 + 0]);
                    acc2[k * 2 + 1] = __builtin_elementwise_fma(f32x2{rr.v[b][k].z, rr.v[b][k].w}, f32x2{q2, q3}, acc2[k * 2 + 1]);
                }
            }
            float acc[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[2 * j] = acc2[j].x; acc[2 * j + 1] = acc2[j].y; }
            // ordered sum: even lane 0.0 + acc[0] + ... + acc[15]; odd lane continues with its 16
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s = s + acc[j];
            float r = from_lower_f(s);
#pragma unroll
            for (int j = 0; j < 16; ++j) r = r + acc[j];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                float q0, q1, q2, q3;
                if constexpr (QREG) {
                    q0 = qt[u * 4 + 0]; q1 = qt[u * 4 + 1]; q2 = qt[u * 4 + 2]; q3 = qt[u * 4 + 3];
                } else {
                    const float4 qq = *reinterpret_cast<const float4*>(lds_q + NB * 128 + u * 16);
                    q0 = qq.x; q1 = qq.y; q2 = qq.z; q3 = qq.w;
                }
                r = __builtin_fmaf(rr.vt[u].x, q0, r);
                r = __builtin_fmaf(rr.vt[u].y, q1, r);
                r = __builtin_fmaf(rr.vt[u].z, q2, r);
                r = __builtin_fmaf(rr.vt[u].w, q3, r);
            }
            d = angular_from_dot(r);
        } else {
            int r = 0, dx = 0;
#pragma unroll
            for (int k = 0; k < 4 * NBI; ++k) {
                r = dot4_i8(rr.x[k].x, qi8[k].x, r); r = dot4_i8(rr.x[k].y, qi8[k].y, r);
                r = dot4_i8(rr.x[k].z, qi8[k].z, r); r = dot4_i8(rr.x[k].w, qi8[k].w, r);
                dx = dot4_i8(rr.x[k].x, rr.x[k].x, dx); dx = dot4_i8(rr.x[k].y, rr.x[k].y, dx);
                dx = dot4_i8(rr.x[k].z, rr.x[k].z, dx); dx = dot4_i8(rr.x[k].w, rr.x[k].w, dx);
            }
            r += pair_swap(r);
            dx += pair_swap(dx);
            // angular_int.rs:52-58 with sqrt(dy) hoisted
            float q = (float)r / (__builtin_sqrtf((float)dx) * sy);
            if (q != q) q = 0.0f;
            const float t = 1.0f - q;
            d = (0.0f <= t) ? t : 0.0f;
        }
        return d;
    }

    // ---- lists of 33 / 65 / 129 slots (max_search 1025 .. 8192): the TWO-LEVEL list ---------------------------------------
    // Round 4 kept one sorted array of 64 S keys (registers + an LDS image) and merged every expansion's candidates into
    // it: O(S) per expansion with a large constant (13.7 k of 25.4 k clocks per expansion at max_search 4096,
    // profiles/r6_phase_f32_ef4096_before.txt). Now (tools/model_twolevel.py replays this against the reference's two heaps):
    //   M  the main sorted array, CAP keys, in LDS ONLY (no register copy: 2 S registers free);
    //   F  the "fresh" list: up to 63 keys, sorted, ONE register pair per lane (+ a 96-key image in LDS for its scatter) --
    //      an expansion's candidates enter here, by the one-slot ranked merge of the short lists;
    //   the logical list is M u F: keys are unique across both (a candidate is looked up in both before it enters).
    //   theta = distance of the union's entry max_search-1: one merge-path split of the two sorted arrays, the 64 lanes
    //           trying the 64 possible splits at once (update_theta);
    //   next  = the smaller of the two first unexpanded entries, flagged where it stands;
    //   break <=> theta < d_next  (#{entries with dist < d_x} >= max_search  <=>  entry max_search-1 is strictly closer);
    //   flush : when F cannot take an expansion's candidates it is merged into M -- the O(S) step, now once per several
    //           expansions: every F entry's rank in M by binary search (all lanes at once), M's windows of 64 moved up IN
    //           PLACE from the top down (a window's shift is uniform except where an F entry's rank falls into it), F's
    //           entries written last. What falls off M's end is dead unless the closest of it ties with theta after the
    //           expansion's inserts: then the walk is handed to the exact walker, as with the short lists.
    static constexpr bool LONG = List::LONG;
    static constexpr int LONG_STEPS = CAP >= 4096u ? 7 : 6; // steps of the 4-ary lower bound over CAP + 1 outcomes (worst case, by simulation: 2112 -> 6, 4160 / 8256 -> 7)
    static constexpr uint32_t FCAP = 63u;       // keys F may hold (a split of the union takes 0..63 of them: one per lane)
    static constexpr uint32_t FIMG_KEYS = 128u; // F's image: 64 entries + up to 32 candidates, padded
    uint64_t* fimg;     // LONG: F's image (behind M's)
    uint64_t fkey;      // LONG: lane j holds F[j] (KEY_INF beyond nF)
    uint32_t nF, nM;    // LONG: real entries of F and of M
    uint32_t m_un;      // LONG: position of M's first unexpanded entry (CAP: none)
    uint64_t m_un_key;  // LONG: its key (KEY_INF: none)
    uint32_t lost_bits; // LONG: smallest distance bits among what flushes pushed off M's end since the last tie test

    // number of M's entries below k (k differs per lane; every lane runs the same steps). A dependent LDS read is ~64
    // clocks + the compare: three pivots per step (their reads in flight together) halve the steps of a binary search.
    __device__ __forceinline__ uint32_t m_lower_bound(uint64_t k) const {
        uint32_t lo = 0, hi = CAP;
#pragma unroll 1
        for (int t = 0; t < LONG_STEPS; ++t) {
            const uint32_t len = hi - lo, last = hi ? hi - 1u : 0u;
            const uint32_t p1 = min(last, lo + (len >> 2)), p2 = min(last, lo + (len >> 1)), p3 = min(last, lo + ((3u * len) >> 2));
            const uint64_t v1 = mslot[p1], v2 = mslot[p2], v3 = mslot[p3];
            const bool go = lo < hi, l1 = v1 < k, l2 = v2 < k, l3 = v3 < k; // (ascending: l3 implies l2 implies l1)
            const uint32_t nlo = l3 ? p3 + 1u : l2 ? p2 + 1u : l1 ? p1 + 1u : lo;
            const uint32_t nhi = l3 ? hi : l2 ? p3 : l1 ? p2 : p1;
            lo = go ? nlo : lo;
            hi = go ? nhi : hi;
        }
        return lo;
    }
    // M's first unexpanded entry at or after `from`
    __device__ __forceinline__ void m_scan_unexpanded(uint32_t from) {
        for (;;) {
            if (from >= nM) {
                m_un = CAP;
                m_un_key = KEY_INF;
                return;
            }
            const uint32_t e = from + lane;
            const uint64_t v = mslot[e < CAP ? e : CAP - 1u];
            const uint64_t um = wave_ballot(e < nM && (((uint32_t)v) & 1u) == 0u);
            if (um) {
                const uint32_t l = (uint32_t)__builtin_ctzll(um);
                m_un = from + l;
                m_un_key = readlane64(v, l);
                return;
            }
            from += 64u;
        }
    }
    // position of M's n-th (n >= 1) expanded entry, WPOS_NONE when there are fewer (the tie path of the filter)
    __device__ __forceinline__ uint32_t m_nth_expanded(uint32_t n) const {
        uint32_t before = 0;
        for (uint32_t w0 = 0; w0 < nM; w0 += 64u) {
            const uint32_t e = w0 + lane;
            const uint64_t v = mslot[e < CAP ? e : CAP - 1u];
            const bool f = e < nM && (((uint32_t)v) & 1u) != 0u;
            const uint64_t em = wave_ballot(f);
            const uint32_t c = (uint32_t)__popcll(em);
            if (before + c >= n) {
                const uint64_t hit = wave_ballot(f && mbcnt64(em) == n - 1u - before);
                return w0 + (uint32_t)__builtin_ctzll(hit);
            }
            before += c;
        }
        return WPOS_NONE;
    }
    __device__ __forceinline__ void long_init(uint32_t xkey_lo_unused) {
        (void)xkey_lo_unused;
        fkey = KEY_INF;
        nF = 0;
        lost_bits = 0xFFFFFFFFu;
        fimg[lane] = KEY_INF;
        fimg[64u + lane] = KEY_INF;
    }
    // F into M. Afterwards F is empty and the union is M.
    __device__ __forceinline__ void flush() {
        if (nF == 0u) return;
        const bool live = lane < nF;
        uint32_t r = CAP; // F is ascending, so are the ranks
        if (live) r = m_lower_bound(fkey);
        uint32_t lost = 0xFFFFFFFFu;
        if (nM) {
            const uint32_t w_first = readlane32(r, 0) >> 6, w_top = (nM - 1u) >> 6;
            uint32_t a_hi = nF; // F entries whose rank lies below the end of the window in hand (all of them, at the top)
            uint64_t vnext = mslot[w_top * 64u + lane]; // (w <= S - 1: inside the image)
            for (uint32_t w = w_top + 1u; w-- > w_first;) {
                const uint32_t a_lo = (uint32_t)__popcll(wave_ballot(live && r < w * 64u));
                const uint32_t e = w * 64u + lane;
                const uint64_t v = vnext;
                if (w > w_first) vnext = mslot[e - 64u]; // the window below travels while this one is placed (it lies wholly below what this one writes)
                uint32_t c = a_lo;
                for (uint32_t j = a_lo; j < a_hi; ++j) c += (e >= readlane32(r, j)) ? 1u : 0u; // F entries ranked inside this window
                const uint32_t dest = e + c;
                if (e < nM) {
                    if (dest < CAP) mslot[dest] = v;
                    else lost = min(lost, wkey_hi(v));
                }
                a_hi = a_lo;
            }
        }
        const uint32_t dest = r + lane; // F[j] lands behind the r_j entries of M below it and the j entries of F below it
        if (live) {
            if (dest < CAP) mslot[dest] = fkey;
            else lost = min(lost, wkey_hi(fkey));
        }
        // M's first unexpanded entry: where it stood or later, or the first unexpanded entry that came from F
        const uint64_t fun = wave_ballot(live && (((uint32_t)fkey) & 1u) == 0u);
        uint32_t lb = m_un < CAP ? m_un : nM;
        if (fun) lb = min(lb, readlane32(dest, (uint32_t)__builtin_ctzll(fun)));
        for (int o = 32; o > 0; o >>= 1) lost = min(lost, (uint32_t)__shfl_xor((int)lost, o, 64));
        lost_bits = min(lost_bits, lost);
        nM = min(CAP, nM + nF);
        nF = 0;
        fkey = KEY_INF;
        fimg[lane] = KEY_INF;
        asm volatile("" ::: "memory"); // one wave: LDS executes its accesses in program order
        m_scan_unexpanded(lb);
    }
    // theta = distance bits of the union's entry ef-1 (0xFFFFFFFF while the union is shorter). Lane j tries the split "j keys
    // of F and ef - j of M are the union's first ef": right iff F[j-1] < M[ef-j] and M[ef-j-1] < F[j]; keys are unique, so
    // exactly one lane is right, and the entry wanted is the larger of F[j-1] and M[ef-j-1].
    __device__ __forceinline__ void update_theta(uint32_t ef) {
        if (nM + nF < ef) {
            theta = 0xFFFFFFFFu;
            return;
        }
        const uint32_t j = lane;
        const bool in_range = j <= nF && j <= ef && ef - j <= nM;
        const uint32_t i = in_range ? ef - j : 0u;
        const uint64_t Mi = mslot[i < CAP ? i : CAP - 1u]; // (i <= ef <= CAP - 64; at nM and beyond: KEY_INF)
        const uint64_t Mi1 = mslot[i ? i - 1u : 0u];
        const uint64_t Fj1 = shift_up1(fkey);
        const bool ok = in_range && (j == 0u || Fj1 < Mi) && (i == 0u || Mi1 < fkey);
        const uint64_t last = (j == 0u) ? Mi1 : (i == 0u) ? Fj1 : (Fj1 > Mi1 ? Fj1 : Mi1);
        const uint64_t okm = wave_ballot(ok);
        if (okm == 0) { // (cannot happen while the two arrays are sorted and their keys distinct: never walk on on a wrong theta)
            bail = true;
            return;
        }
        theta = readlane32(wkey_hi(last), (uint32_t)__builtin_ctzll(okm));
    }
    // mod.rs:1029 on the union: a candidate beyond theta is dead; one that ties with it needs res.peek() -- the ef-th
    // EXPANDED entry: rare, and taken on M alone after a flush
    __device__ __forceinline__ uint64_t filter_mask_long(const uint64_t candm, const uint32_t dbits, const uint32_t ef) {
        uint64_t passm = candm & wave_ballot(dbits <= theta);
        const uint64_t tiem = passm & wave_ballot(dbits == theta);
        if (tiem) {
            flush();
            const uint32_t w = m_nth_expanded(ef);
            if (w != WPOS_NONE) {
                const uint64_t wk = mslot[w];
                const uint32_t worst = (uint32_t)__builtin_amdgcn_readfirstlane((int)wkey_hi(wk));
                passm &= ~(tiem & ~wave_ballot(dbits < worst));
            }
        }
        return passm;
    }
    // The candidates of passm (odd lanes; ck = their keys) enter F. Before: the ones M or F holds already leave, and so does
    // the second lane of a row that names a node twice. Returns the lanes that entered; `cmin` = the smallest entered key.
    __device__ __forceinline__ uint64_t insert_fresh(uint64_t passm, const uint64_t ck, const bool twin_rows, uint64_t& cmin) {
        cmin = KEY_INF;
        if (passm == 0) return 0;
        {   // in M already? the lower bound stops at the candidate's own node (same key up to the flag). Only the candidates'
            // lanes search: a few lanes' reads of the image do not collide in the LDS banks the way 64 random ones do
            bool in_m = false;
            if (__builtin_amdgcn_inverse_ballot_w64(passm)) {
                const uint32_t rm = m_lower_bound(ck);
                const uint64_t at = mslot[rm < CAP ? rm : CAP - 1u];
                in_m = rm < CAP && (at | 1ull) == (ck | 1ull);
            }
            passm &= ~wave_ballot(in_m);
        }
        if (twin_rows) {
            for (uint64_t it = passm; it; it &= it - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(it);
                const uint64_t K = readlane64(ck, j);
                const uint64_t twin = wave_ballot(ck == K && lane > j) & it;
                passm &= ~twin;
                it &= ~twin;
            }
        }
        const uint32_t* fimg_lo = reinterpret_cast<const uint32_t*>(fimg);
        const uint32_t me = (uint32_t)ck | 1u;
        uint32_t shift, rankv, below;
        for (;;) {
            if (passm == 0) return 0;
            shift = 0;
            rankv = 0;
            below = 0;
            for (uint64_t it = passm; it; it &= it - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(it);
                const uint64_t K = readlane64(ck, j);
                const bool g = fkey > K;
                shift += g ? 1u : 0u;
                const uint32_t above = (uint32_t)__popcll(wave_ballot(g));
                rankv = (lane == j) ? 64u - above : rankv; // F's entries below K
                below += (ck > K) ? 1u : 0u;
            }
            // in F already? it stands at `rank` when expanded (K|1 is the smallest key above K), at rank - 1 when not
            const uint32_t r0 = rankv < 64u ? rankv : 63u, r1 = rankv ? rankv - 1u : 0u;
            const uint32_t e0 = fimg_lo[2u * r0], e1 = fimg_lo[2u * r1];
            const uint64_t known = passm & ((wave_ballot((e0 | 1u) == me) & wave_ballot(rankv < 64u)) | wave_ballot((e1 | 1u) == me));
            if (known == 0) break;
            passm &= ~known;
        }
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(passm);
        fimg[lane + shift] = fkey;
        if (mine) fimg[rankv + below] = ck;
        asm volatile("" ::: "memory");
        fkey = fimg[lane];
        asm volatile("" ::: "memory");
        nF += (uint32_t)__popcll(passm);
        const uint64_t firstm = passm & wave_ballot(below == 0u);
        cmin = readlane64(ck, (uint32_t)__builtin_ctzll(firstm));
        return passm;
    }

    // ---- lists of up to 17 slots: every candidate of an expansion ranked at once, ONE pass through the LDS image -----------
    // pq.push (mod.rs:1030) of an expansion's candidates and the choice of the node expanded next, without one insert per
    // candidate. The loop below visits the candidates that passed the filter once each (K = a candidate's key, wave-uniform):
    //   * the list holds K's node already (a revisit: same node, same distance, so the same key up to the flag) -> it leaves;
    //   * every list entry above K counts it (`shift`), K's own rank is the number of entries below it;
    //   * every other candidate above K counts it (`below`);
    // nothing in the loop waits for a scalar decision except the look-up's branch. Then entry e goes to place e + shift,
    // candidate c to rank + below, all through one scatter into the image and one read back; what lands on place CAP is
    // the smallest key pushed off the end (the tie test of the file comment reads its distance, entry max_search-1 is read
    // from the same image). The candidate with below == 0 is the smallest: it is expanded next iff its rank is at most the
    // position of the first unexpanded entry y (#{entries < K} <= pos(y)  <=>  K < y) -- known BEFORE the scatter, so its
    // adjacency row is requested under the merge, and it enters the list with its flag already set.
    // Without a visited set about half of the candidates that pass the filter are revisits: nodes the list holds already,
    // found there by rank_one at the price of a loop trip each. A direct-mapped cache of the ids that entered the list
    // (VCACHE_SLOTS words of LDS, one look-up for all 32 neighbors under the row loads) takes most of them out before the
    // loop: a hit means "this node entered the list earlier in this walk of the layer", i.e. the reference's visited set
    // holds it and skips it (mod.rs:1026); a miss (never entered, or evicted by a colliding id) means nothing -- the
    // candidate goes the exact way. Results do not depend on the cache.
    static constexpr uint32_t VSLOTS = (V16 == 5) ? VCACHE_SLOTS_SEEN : VCACHE_SLOTS;
    __device__ __forceinline__ static uint32_t vcache_slot(uint32_t id) {
        if constexpr (V16 == 5) return (id ^ (id >> 11) ^ (id >> 22)) & (VSLOTS - 1u);
        else return (id ^ (id >> 9)) & (VSLOTS - 1u);
    }
    __device__ __forceinline__ void vcache_reset(uint32_t first_id) {
        if constexpr (NOVIS) {
            uint4* t4 = reinterpret_cast<uint4*>(vcache);
            const uint4 e = make_uint4(ID_EMPTY, ID_EMPTY, ID_EMPTY, ID_EMPTY);
            for (uint32_t i = lane; i < VSLOTS / 4u; i += 64) t4[i] = e;
            asm volatile("" ::: "memory");
            if (lane == 0) vcache[vcache_slot(first_id)] = first_id;
        }
    }

    // Where a candidate's rank (the entries above its key) is counted. One slot (max_search up to 60: the shape of one
    // query per call and of one batch per launch, where a walker has its SIMD to itself): on the VECTOR side -- v_bcnt of
    // the compare's mask, because a scalar instruction that reads a vector result makes a lone wave wait ~16 clocks. Longer
    // lists run many walkers per SIMD (max_search 61 and up: 4-6 waves) and are bound by the VALU's issue slots instead
    // (C5's shard: 70 k vector instructions per query, the vector pipe 60 % busy; per pair of candidates and four slots 16
    // v_mov + 16 v_bcnt + 6 v_add3 of 62 vector instructions were this count): there s_bcnt1_i32_b64 counts the mask where
    // v_cmp left it, on the scalar pipe, which the other waves' vector work hides.
    static constexpr bool SCALAR_COUNTS = S >= 2;
    struct Ranked {
        uint32_t shift[S]; // list entries: candidates that sort before my key
        uint32_t rankv;    // candidate lanes: entries of the list below my key
        uint32_t below;    // candidate lanes: candidates below my key
    };

    // One candidate of the loop below (`it` is not empty). What a lone wavefront pays for (tools/ubench.hip): ~4 clocks
    // per instruction, scalar ones included; ~20 for a taken branch; and ~16 more whenever a SCALAR instruction consumes
    // what a VECTOR one produced (a v_cmp's mask, a v_readlane's value) -- the other direction is free. So nothing scalar
    // here reads a vector result: the scalar chain (which lane is next) runs ahead on its own, the rank is counted by
    // v_bcnt on the vector side from the compare's mask and lands in the candidate's lane through a select.
    template <bool TWINS>
    __device__ __forceinline__ void rank_one(uint64_t& it, uint64_t& passm, const uint32_t ck_lo, const uint32_t ck_hi,
                                             const uint64_t ck, Ranked& rk) const {
        const uint32_t j = (uint32_t)__builtin_ctzll(it);
        const uint64_t bit = 1ull << j;
        it &= it - 1;
        const uint32_t k_lo = readlane32(ck_lo, j), k_hi = readlane32(ck_hi, j);
        const uint64_t K = ((uint64_t)k_hi << 32) | k_lo;
        if constexpr (TWINS) { // the second lane of a row that names this node twice leaves with this one
            const uint64_t twins = wave_ballot(ck_lo == k_lo) & it;
            it &= ~twins;
            passm &= ~twins;
        }
        uint32_t above = 0; // entries above K, counted in every lane
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool g = L.key[s] > K;
            rk.shift[s] += g ? 1u : 0u;
            const uint64_t gm = wave_ballot(g);
            if constexpr (SCALAR_COUNTS) {
                above += (uint32_t)__popcll(gm); // s_bcnt1_i32_b64
            } else {
                uint32_t g_lo = (uint32_t)gm, g_hi = (uint32_t)(gm >> 32);
                asm("" : "+v"(g_lo));
                asm("" : "+v"(g_hi)); // (in vector registers: the counts are v_bcnt's, not s_bcnt1's)
                above += (uint32_t)__builtin_popcount(g_lo) + (uint32_t)__builtin_popcount(g_hi);
            }
        }
        rk.rankv = __builtin_amdgcn_inverse_ballot_w64(bit) ? above : rk.rankv;
        rk.below += (ck > K) ? 1u : 0u;
    }

    // Two candidates at once (`it` holds at least two): the same operations as two rank_one, written side by side so that
    // the wait states between a compare and the use of its mask, and between a v_readlane and the use of its value, are
    // filled by the other candidate's instructions.
    __device__ __forceinline__ void rank_two(uint64_t& it, const uint32_t ck_lo, const uint32_t ck_hi, const uint64_t ck, Ranked& rk) const {
        const uint32_t ja = (uint32_t)__builtin_ctzll(it);
        it &= it - 1;
        const uint32_t jb = (uint32_t)__builtin_ctzll(it);
        it &= it - 1;
        const uint64_t bit_a = 1ull << ja, bit_b = 1ull << jb;
        const uint32_t a_lo = readlane32(ck_lo, ja), a_hi = readlane32(ck_hi, ja);
        const uint32_t b_lo = readlane32(ck_lo, jb), b_hi = readlane32(ck_hi, jb);
        const uint64_t Ka = ((uint64_t)a_hi << 32) | a_lo, Kb = ((uint64_t)b_hi << 32) | b_lo;
        uint32_t above_a = 0, above_b = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool ga = L.key[s] > Ka, gb = L.key[s] > Kb;
            const uint64_t gma = wave_ballot(ga), gmb = wave_ballot(gb);
            rk.shift[s] += (ga ? 1u : 0u) + (gb ? 1u : 0u);
            if constexpr (SCALAR_COUNTS) {
                above_a += (uint32_t)__popcll(gma);
                above_b += (uint32_t)__popcll(gmb);
            } else {
                uint32_t a0 = (uint32_t)gma, a1 = (uint32_t)(gma >> 32), b0 = (uint32_t)gmb, b1 = (uint32_t)(gmb >> 32);
                asm("" : "+v"(a0));
                asm("" : "+v"(b0));
                asm("" : "+v"(a1));
                asm("" : "+v"(b1));
                above_a += (uint32_t)__builtin_popcount(a0) + (uint32_t)__builtin_popcount(a1);
                above_b += (uint32_t)__builtin_popcount(b0) + (uint32_t)__builtin_popcount(b1);
            }
        }
        rk.rankv = __builtin_amdgcn_inverse_ballot_w64(bit_a) ? above_a : rk.rankv;
        rk.rankv = __builtin_amdgcn_inverse_ballot_w64(bit_b) ? above_b : rk.rankv;
        rk.below += ((ck > Ka) ? 1u : 0u) + ((ck > Kb) ? 1u : 0u);
    }

    // passm: lanes (odd) whose candidate passed the filter; on return, the ones that enter the list, ranked.
    // The loop ranks every candidate as if the list held none of them. Whether it does is looked up afterwards, for all of
    // them at once: a node the list holds has the candidate's key up to the flag, so it stands right where the candidate's
    // rank points -- at entry `rank` when it is expanded (K|1 is the smallest key above K), at `rank - 1` when it is not
    // (K itself is the largest key not above K). Both are read from the list's image (ids do not depend on the flags, which
    // the image of a short list does not follow). A hit is rare (the cache of entered ids has taken the revisits out) and
    // costs a second pass over the remaining candidates.
    template <bool TWINS>
    __device__ __forceinline__ uint64_t rank_candidates(uint64_t passm, const uint32_t ck_lo, const uint32_t ck_hi, Ranked& rk) const {
        const uint64_t ck = ((uint64_t)ck_hi << 32) | ck_lo;
        const uint32_t* img_lo = reinterpret_cast<const uint32_t*>(mslot);
        for (;;) {
#pragma unroll
            for (int s = 0; s < S; ++s) rk.shift[s] = 0;
            rk.rankv = 0;
            rk.below = 0;
            if (passm == 0) return 0;
            uint64_t it = passm;
            if constexpr (TWINS) {
                while (it) rank_one<true>(it, passm, ck_lo, ck_hi, ck, rk);
            } else {
                while (it & (it - 1)) rank_two(it, ck_lo, ck_hi, ck, rk); // two candidates per trip
                if (it) rank_one<false>(it, passm, ck_lo, ck_hi, ck, rk);
            }
            rk.rankv = CAP - rk.rankv; // entries below K
            const uint32_t r0 = rk.rankv < CAP ? rk.rankv : CAP - 1u, r1 = rk.rankv ? rk.rankv - 1u : 0u;
            const uint32_t e0 = img_lo[2u * r0], e1 = img_lo[2u * r1];
            const uint32_t me = ck_lo | 1u;
            const uint64_t known = passm & ((wave_ballot((e0 | 1u) == me) & wave_ballot(rk.rankv < CAP)) | wave_ballot((e1 | 1u) == me));
            if (known == 0) return passm;
            passm &= ~known; // in the list already, expanded or not: not a candidate (VisitedNone, wave_prims.h)
        }
    }

    // scatter the list and the candidates in passm to their places in the image, read the list back
    __device__ __forceinline__ void merge_ranked(const uint64_t passm, const uint64_t ck, const Ranked& rk, const uint32_t ef) {
        uint64_t* img = mslot;
#pragma unroll
        for (int s = 0; s < S; ++s) img[(uint32_t)s * 64u + lane + rk.shift[s]] = L.key[s];
        if (__builtin_amdgcn_inverse_ballot_w64(passm)) img[rk.rankv + rk.below] = ck;
        asm volatile("" ::: "memory"); // one wave: LDS executes its accesses in program order
#pragma unroll
        for (int s = 0; s < S; ++s) L.key[s] = img[(uint32_t)s * 64u + lane];
        const uint32_t lost = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(img)[2u * CAP + 1u]);
        theta = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(img)[2u * (ef - 1u) + 1u]);
        asm volatile("" ::: "memory");
        // every place up to CAP + m - 1 was written (a permutation of CAP entries and m candidates), so place CAP holds
        // the smallest key that fell off the end, 0xFFFFFFFF.. when that is an unused place
        if (lost == theta && theta != 0xFFFFFFFFu) bail = true;
    }

    // mod.rs:1029 for the candidates of one expansion: the lanes of candm whose candidate enters `pq`
    __device__ __forceinline__ uint64_t filter_mask(const uint64_t candm, const uint32_t dbits, const uint32_t ef) const {
        // `res` does not change during an expansion, so neither do the two thresholds:
        //   theta = dist of entry max_search-1 (0xFFFFFFFF while the list is shorter: every distance is below it); a
        //           candidate beyond it has max_search entries strictly closer: dead;
        //   worst = res.peek().dist = dist of the max_search-th EXPANDED entry (mod.rs:1029), >= theta.
        // d <= theta < worst needs no second look; only a candidate that ties with theta can still fail
        // `d < worst`, and only then is the max_search-th expanded entry looked up.
        uint64_t passm = candm & wave_ballot(dbits <= theta);
        const uint64_t tiem = passm & wave_ballot(dbits == theta);
        if (tiem) {
            const uint32_t w = L.nth_expanded(ef); // res.peek(): the max_search-th popped node
            if (w != WPOS_NONE) passm &= ~(tiem & ~wave_ballot(dbits < wkey_hi(L.at(w))));
        }
        return passm;
    }

    // mod.rs:1029 for the candidates of one expansion (`cand` lanes hold a distance): which of them enter the list
    __device__ __forceinline__ bool filter(bool cand, float d, uint32_t ef) {
        return __builtin_amdgcn_inverse_ballot_w64(filter_mask(wave_ballot(cand), __float_as_uint(d), ef));
    }

    // The walk on the two-level list. One iteration = one expansion of x (flagged when it was chosen): adjacency row, row
    // loads, the fetch-ahead of the adjacency row of the entry that is first in line now, distances, filter, room in F (or
    // a flush), the candidates into F, theta, who is next (a candidate that sorts before everything unexpanded has its
    // adjacency row requested at once), break test, flag.
    __device__ __forceinline__ void search_layer_long(const LayerDev& Ly, uint32_t entrypoint, uint32_t ef, uint32_t slots,
                                                      bool d0_known, float d0_value) {
        static_assert(!LONG || (NOVIS && !WIDE && !TOUCH), "lists beyond 1024 keys: no visited set, 32-id layers");
        PT_RESET();
        L.init_list(mslot, lane);
        long_init(0u);
        __syncthreads();
        [[maybe_unused]] const uint8_t* const adjx = XT ? Ly.adjx : nullptr; // ids + the neighbors' tails (LayerDev::adjx), or null
        const gptr_u32 adjg = (XT && adjx) ? (gptr_u32)adjx : (gptr_u32)Ly.adj;
        const uint32_t W = (XT && adjx) ? Ly.adjx_stride / 4u : 32u; // u32 from one node's ids to the next's
        const bool twin_rows = (Ly.flags & LAYER_TWIN_ROWS) != 0u;
        uint32_t pre_id = entrypoint, pre_nb; // adjacency row fetched ahead (one id per pair) and whose it is
        pre_nb = adjg[(size_t)entrypoint * W + R]; // get_neighbors(entrypoint): needed right after
        vis.count = 1;
        st.n_dist += 1;
        uint64_t xkey;
        if (d0_known) {
            xkey = wkey(d0_value, entrypoint) | 1ull;
        } else {
            RowRegs r0;
            issue_rows(entrypoint, r0);
            const float d0 = finish_rows(r0);
            xkey = readlane64(wkey(d0, entrypoint), 1) | 1ull;
        }
        L.set_first(xkey, lane); // M[0]: popped at once, flagged
        vcache_reset(entrypoint);
        nM = 1;
        m_un = CAP;
        m_un_key = KEY_INF;
        theta = ef == 1u ? wkey_hi(xkey) : 0xFFFFFFFFu;
        uint32_t xid = entrypoint;

        RowRegs rr;
        PT_WAIT_VM();
        PT_MARK(7); // layer setup: entry point distance
        for (;;) {
            st.n_expand += 1;
            // layer.get_neighbors(x), mod.rs:1025 / 540-552: row prefix until UNUSED, one id per pair
            uint32_t nb;
            if (pre_id == xid) nb = pre_nb;
            else nb = adjg[(size_t)xid * W + R];
            PT_MARK(0);
            PT_WAIT_VM();
            PT_MARK(8); // wait for the adjacency row
            PT_COUNT();
            const uint64_t unused = wave_ballot(nb == ID_EMPTY);
            const uint32_t nvalid = unused ? ((uint32_t)__builtin_ctzll(unused) >> 1) : 32u;
            st.n_adj += nvalid;
            {   // pairs beyond the row re-read its first neighbor (an empty row: the node itself)
                uint32_t fill = readlane32(nb, 0);
                fill = fill != ID_EMPTY ? fill : xid;
                const uint8_t* tails = nullptr;
                if constexpr (XT) {
                    if (adjx) tails = adjx + (size_t)xid * Ly.adjx_stride + 128u + (nb != ID_EMPTY ? R : 0u) * XTAILB;
                }
                issue_rows(nb != ID_EMPTY ? nb : fill, rr, tails);
            }
            // fetch ahead the adjacency row of the entry that is first in line now; always one load: static wait counts
            uint64_t ypre;
            {
                const uint64_t fun = wave_ballot((((uint32_t)fkey) & 1u) == 0u); // (KEY_INF carries the flag)
                const uint64_t yF = fun ? readlane64(fkey, (uint32_t)__builtin_ctzll(fun)) : KEY_INF;
                ypre = yF < m_un_key ? yF : m_un_key;
            }
            pre_id = ypre != KEY_INF ? wkey_id(ypre) : xid;
            pre_nb = adjg[(size_t)pre_id * W + R];
            const uint32_t cached = vcache[vcache_slot(nb)]; // what the cache of entered ids holds in this neighbor's slot
            asm volatile("" ::: "memory");
            PT_MARK(1); // row loads and the fetch-ahead issued
            const uint64_t fm = 0x5555555555555555ull & ~unused; // every neighbor is evaluated (valid ids come first)
            PT_MARK(2);
            PT_WAIT_VM();
            PT_MARK(3); // what is left of the wait for the rows
            const float d = finish_rows(rr);
            st.n_dist += (uint32_t)__popcll(fm);
            PT_PIN(d);
            PT_MARK(4); // distances
            const uint32_t dbits = __float_as_uint(d);
            // odd lanes whose even partner holds an id; a node that entered the list before is visited (mod.rs:1026): the cache
            // knows the recent ones -- most of the revisits -- and the look-ups in M and F find the rest
            uint64_t passm = filter_mask_long((fm << 1) & ~wave_ballot(cached == nb), dbits, ef);
            PT_ADD(6, (uint32_t)__popcll(passm));
            if (nF + (uint32_t)__popcll(passm) > FCAP) {
                flush(); // F must take them all: BEFORE they are ranked against M
                PT_ADD(1, 1u); // (the diagnostics build counts flushes where the short lists count expansions without a candidate)
            }
            PT_MARK(10); // filter (and a flush)
            const uint64_t ck = ((uint64_t)dbits << 32) | (nb << 1);
            uint64_t cmin;
            passm = insert_fresh(passm, ck, twin_rows, cmin);           // pq.push, mod.rs:1029-1031
            if (__builtin_amdgcn_inverse_ballot_w64(passm)) vcache[vcache_slot(nb)] = nb;
            PT_MARK(11); // look-ups, ranks, the candidates into F
            PT_ADD(0, (uint32_t)__popcll(passm));
            if (passm) {
                if (cmin < ypre) { // it is expanded next: its adjacency row is wanted
                    PT_ADD(5, 1u);
                    pre_id = wkey_id(cmin);
                    uint32_t pv = pre_id;
                    asm("" : "+v"(pv));
                    pre_nb = adjg[(size_t)pv * W + R];
                }
                update_theta(ef);
            }
            if (lost_bits != 0xFFFFFFFFu) { // what a flush pushed off M's end: dead unless the closest of it ties with theta
                if (lost_bits == theta) bail = true;
                lost_bits = 0xFFFFFFFFu;
            }
            PT_MARK(5); // theta, the next node's adjacency request
            if (bail) return;
            // who is next: the smaller of the two first unexpanded entries (pq.pop(), mod.rs:1018)
            const uint64_t fun = wave_ballot((((uint32_t)fkey) & 1u) == 0u);
            const uint32_t pF = fun ? (uint32_t)__builtin_ctzll(fun) : 0u;
            const uint64_t yF = fun ? readlane64(fkey, pF) : KEY_INF;
            const bool fromF = yF < m_un_key;
            const uint64_t y = fromF ? yF : m_un_key;
            if (y == KEY_INF) break;                 // the queue is empty
            if (theta < wkey_hi(y)) break;           // mod.rs:1019-1021: max_search entries are strictly closer
            if (fromF) {                             // res.push((d, idx)), mod.rs:1023
                if (lane == pF) {
                    fkey |= 1ull;
                    fimg[pF] = fkey;
                }
            } else {
                if (lane == 0) mslot[m_un] = y | 1ull;
                asm volatile("" ::: "memory");
                m_scan_unexpanded(m_un + 1u);
            }
            xid = wkey_id(y);
            PT_MARK(6); // pop, break test, flag
            PT_MARK(9);
        }
        flush(); // the layer's result is read from M
    }

    // search_for_neighbors (mod.rs:999-1037) on one layer; the result is the list's expanded entries.
    //
    // One iteration = one expansion of x, whose key, position and adjacency row the iteration before left behind (x is
    // flagged = res.push, mod.rs:1023, when it is chosen): row loads from the ids alone -> the first unexpanded entry y and
    // the fetch-ahead of ITS adjacency row, under the loads -> distances -> filter -> ranks (rank_candidates) -> who is next:
    // the smallest candidate when it sorts before y (its adjacency row is requested right there), else y, for which the
    // break test (mod.rs:1019) is taken -- before the merge, which a finished walk does not need -> merge -> flag.
    // A candidate that wins needs no break test: it passed the filter, so fewer than max_search entries are strictly closer.
    // The row loads have one site in the loop: two sites would meet in a phi, and the register copies at the
    // join wait for the data right after issuing it.
    __device__ __forceinline__ void search_layer(const LayerDev& Ly, uint32_t entrypoint, uint32_t ef, uint32_t slots,
                                                 bool d0_known = false, float d0_value = 0.0f) {
        if constexpr (LONG) {
            search_layer_long(Ly, entrypoint, ef, slots, d0_known, d0_value);
            return;
        } else {
        PT_RESET();
        if constexpr (!NOVIS) vis.reset(vis_tab, slots, lane, ef > 1u ? p.front_eighths : 7u);
        L.init_list(mslot, lane);
        __syncthreads();
        [[maybe_unused]] const uint8_t* const adjx = XT ? Ly.adjx : nullptr; // ids + the neighbors' tails (LayerDev::adjx), or null
        const gptr_u32 adjg = (XT && adjx) ? (gptr_u32)adjx : (gptr_u32)Ly.adj;
        // u32 from one node's ids to the next's: 32, 64 in a WIDE launch, a record of LayerDev::adjx
        const uint32_t W = WIDE ? Ly.width : (XT && adjx) ? Ly.adjx_stride / 4u : 32u;
        const bool twin_rows = (Ly.flags & LAYER_TWIN_ROWS) != 0u; // some row of the layer names a neighbor twice (rare: found at upload)
        // distance to the entry point (mod.rs:1012-1016), the first pop. On every layer but the first the entry point is
        // the node the layer above returned as its closest, and its distance to the same query was evaluated there: the
        // same operations on the same inputs give the same bits, so the value is reused (the evaluation still counts)
        // and the layer starts one memory round trip earlier.
        uint32_t next_nb = adjg[(size_t)entrypoint * W + R]; // get_neighbors(entrypoint): the first row
        if constexpr (!NOVIS) vis.insert(entrypoint, lane == 0, p.ovf);
        vis.count = 1;
        st.n_dist += 1;
        uint64_t xkey;
        if (d0_known) {
            xkey = wkey(d0_value, entrypoint) | 1ull;
        } else {
            RowRegs r0;
            issue_rows(entrypoint, r0);
            const float d0 = finish_rows(r0);
            xkey = readlane64(wkey(d0, entrypoint), 1) | 1ull;
        }
        L.set_first(xkey, lane); // popped at once: entry 0, flagged
        vcache_reset(entrypoint);
        theta = ef == 1u ? wkey_hi(xkey) : 0xFFFFFFFFu;
        uint32_t xid = entrypoint;

        RowRegs rr;
        [[maybe_unused]] uint32_t touched[NT] = {};
        [[maybe_unused]] uint32_t touched_x = 0;
        PT_WAIT_VM();
        PT_MARK(7); // layer setup: tables, entry point distance
        for (;;) {
            st.n_expand += 1;
            // layer.get_neighbors(x), mod.rs:1025 / 540-552: row prefix until UNUSED, one id per pair. The row was requested
            // an expansion ago (y's fetch-ahead) or before the merge (a candidate that won): two requests, one register --
            // taken over HERE, by an instruction the compiler cannot move up to where the second request is issued (a copy
            // placed there waits for the data right after asking for it)
            uint32_t nb;
            asm volatile("v_mov_b32 %0, %1" : "=v"(nb) : "v"(next_nb));
            [[maybe_unused]] uint32_t nb_hi = ID_EMPTY; // WIDE: ids 32..63 of the row, wanted once the first pass is through
            if constexpr (WIDE) {
                if (W > 32u) nb_hi = adjg[(size_t)xid * W + 32u + R];
            }
            PT_MARK(0);
            PT_WAIT_VM();
            PT_MARK(8); // wait for the adjacency row
            PT_COUNT();
            // y: the entry that is first in line now (x is flagged), or the candidate that has beaten it
            uint64_t ykey = KEY_INF;
            uint32_t ypos = WPOS_NONE, yid = xid;
            for (uint32_t half = 0; half < (WIDE ? 2u : 1u); ++half) {
            if constexpr (WIDE) {
                if (half == 1u) nb = nb_hi; // the row went through all of its first 32 places: its second 32
            }
            const uint64_t unused = wave_ballot(nb == ID_EMPTY);
            const uint32_t nvalid = unused ? ((uint32_t)__builtin_ctzll(unused) >> 1) : 32u;
            st.n_adj += nvalid;
            [[maybe_unused]] uint64_t seenm = 0; // SEEN: the pairs whose id the walk has evaluated before (both lanes of a pair)
            if constexpr (SEEN) {
                const bool seen = vcache[vcache_slot(nb)] == nb; // (the entry point is in the cache; UNUSED never is)
                seenm = wave_ballot(seen);
                if constexpr (GEN) {
                    // the streamed walker loads a row's later chunks inside finish_rows, for every lane: a revisit's and an
                    // empty pair's lanes follow the first NEW neighbor's row instead (lines that pair fetches anyway)
                    const uint64_t newm = wave_ballot(!seen && nb != ID_EMPTY);
                    const uint32_t fill = newm ? readlane32(nb, (uint32_t)__builtin_ctzll(newm)) : xid;
                    issue_rows((!seen && nb != ID_EMPTY) ? nb : fill, rr);
                } else if (!seen && nb != ID_EMPTY) { // rows of new ids only: a revisit's -- and an empty pair's -- loads are not issued
                    const uint8_t* tails = nullptr;
                    if constexpr (XT) {
                        if (adjx) tails = adjx + (size_t)xid * Ly.adjx_stride + 128u + R * XTAILB;
                    }
                    issue_rows(nb, rr, tails);
                }
            } else {
                // pairs past the row's end re-read its first neighbor (the same lines as pair 0: no traffic of their own; an
                // empty row: the node itself) -- chosen per lane, nothing scalar waits for a vector result
                uint32_t fill = readlane32(nb, 0);
                asm("" : "+v"(fill));
                fill = fill != ID_EMPTY ? fill : xid;
                const uint8_t* tails = nullptr;
                if constexpr (XT) { // (a pair past the row's end: slot 0's tail, the lines pair 0 reads anyway)
                    if (adjx) tails = adjx + (size_t)xid * Ly.adjx_stride + 128u + (nb != ID_EMPTY ? R : 0u) * XTAILB;
                }
                issue_rows(nb != ID_EMPTY ? nb : fill, rr, tails);
            }
            if (half == 0u) {
                // fetch ahead the row of the node that is first in line now; always one load: static wait counts
                const bool has_y = L.first_unexpanded(ypos);
                if (has_y) {
                    ykey = L.at(ypos);
                    yid = wkey_id(ykey);
                } else {
                    ypos = WPOS_NONE;
                }
                next_nb = adjg[(size_t)yid * W + R];
            }
            [[maybe_unused]] uint32_t cached = ID_EMPTY; // what the cache holds in this neighbor's slot (arrives under the row loads)
            if constexpr (NOVIS && !SEEN) cached = vcache[vcache_slot(nb)];
            if constexpr (SEEN) { // every id this expansion evaluates is seen from now on (lanes of one slot: the last one wins)
                if (nb != ID_EMPTY && !__builtin_amdgcn_inverse_ballot_w64(seenm)) vcache[vcache_slot(nb)] = nb;
            }
            asm volatile("" ::: "memory"); // the fetch-ahead and the look-up are issued here, not where their results are used
            __builtin_amdgcn_sched_barrier(0);
            PT_MARK(1); // row loads and the fetch-ahead issued

            // visited set under the loads, then the distances (mod.rs:1026-1027)
            uint64_t fm; // even lanes whose id is evaluated for the first time (no set: every neighbor)
            if constexpr (SEEN) fm = 0x5555555555555555ull & ~unused & ~seenm;
            else if constexpr (NOVIS) fm = 0x5555555555555555ull & ~unused; // (valid ids come first, mod.rs:540-552)
            else fm = wave_ballot(vis.insert(nb, h == 0u && R < nvalid, p.ovf));
            PT_MARK(2); // visited set (under the loads)
            PT_WAIT_VM();
            PT_MARK(3); // what is left of the wait for the rows
            const float d = finish_rows(rr);
            if constexpr (TOUCH) {
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(touched[j])); // (arrived before the rows did)
                if constexpr (XT) asm volatile("" ::"v"(touched_x));
                // y's adjacency row came in right behind the rows; a pair past its end touches y's own row
                const uint32_t tid = next_nb != ID_EMPTY ? next_nb : yid;
                const uint8_t* tb = p.elements + (size_t)tid * RSTRIDE;
                // what an expansion of y reads of a neighbor's row: all of it, or (XT with the tails in y's record) its chunks
                const uint32_t lim = (XT && adjx) ? (uint32_t)NB * 128u : ROWB;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const uint32_t off = h * 128u + (uint32_t)j * 256u;
                    touched[j] = *reinterpret_cast<const uint32_t*>(tb + (off < lim ? off : lim - 4u));
                }
                if constexpr (XT) { // ... and the lines of y's record that hold its neighbors' tails (eight pairs share one)
                    if (adjx) touched_x = *(gptr_u32)(adjx + (size_t)yid * Ly.adjx_stride + 128u + R * XTAILB);
                }
                asm volatile("" ::: "memory");
            }
            const uint32_t mf = (uint32_t)__popcll(fm);
            vis.added(mf);
            st.n_dist += mf;
            PT_PIN(d);
            PT_MARK(4); // distances
            const uint32_t dbits = __float_as_uint(d);
            uint64_t candm = fm << 1; // odd lanes whose even partner holds a new id
            if constexpr (NOVIS) candm &= ~wave_ballot(cached == nb); // entered the list before: visited (a pair past the row's end is no candidate anyway)
            uint64_t passm = filter_mask(candm, dbits, ef);
            PT_ADD(6, (uint32_t)__popcll(passm));
            PT_MARK(10); // cache look-up, filter
            const uint32_t ck_lo = nb << 1;
            const uint64_t ck = ((uint64_t)dbits << 32) | ck_lo;
            Ranked rk;
            passm = twin_rows ? rank_candidates<true>(passm, ck_lo, dbits, rk) : rank_candidates<false>(passm, ck_lo, dbits, rk);
            if constexpr (NOVIS && !SEEN) {
                if (__builtin_amdgcn_inverse_ballot_w64(passm)) vcache[vcache_slot(nb)] = nb;
            }
            const uint32_t m = (uint32_t)__popcll(passm);
            PT_MARK(11); // ranks
            PT_ADD(0, m);
            PT_ADD(1, m == 0u ? 1u : 0u);
            PT_ADD(2, (m == 1u || m == 2u) ? 1u : 0u);
            PT_ADD(3, (m >= 3u && m <= 6u) ? 1u : 0u);
            PT_ADD(4, m > 6u ? 1u : 0u);
            // who is next?
            [[maybe_unused]] uint64_t win_bit = 0; // the lane of a candidate that is expanded next
            if (m) {
                // the smallest candidate (nobody below it) wins iff it sorts before y: #{entries < K} <= pos(y)  (no y: 0xFFFFFFFF)
                const uint64_t wm = passm & wave_ballot(rk.below == 0u) & wave_ballot(rk.rankv <= ypos);
                if (wm) {
                    const uint32_t jw = (uint32_t)__builtin_ctzll(wm);
                    win_bit = wm;
                    yid = readlane32(nb, jw);
                    uint32_t yv = yid;
                    asm("" : "+v"(yv)); // (the address on the vector side: a scalar shift would wait for the v_readlane)
                    next_nb = adjg[(size_t)yv * W + R]; // requested here, arrives under the merge
                    ykey = readlane64(ck, jw);
                    ypos = readlane32(rk.rankv, jw);
                    PT_ADD(5, 1u);
                }
            }
            bool last = true;
            if constexpr (WIDE) last = half == 1u || nvalid < 32u || wave_ballot(nb_hi != ID_EMPTY) == 0;
            if (last) {
                if (ypos == WPOS_NONE) goto layer_done; // pq.pop() on an empty queue, mod.rs:1018
                // mod.rs:1019-1021. Every entry before y is expanded and at most as far; #{closer} = ypos - #{ties
                // before y}, so the count is only taken when ypos alone does not already decide
                if (ypos >= ef && L.count_closer(wkey_hi(ykey)) >= ef) goto layer_done; // (before the merge: a finished walk does not need it)
            }
            PT_MARK(5); // next-node decision, its adjacency request
            if constexpr (!WIDE) {
                // res.push((d, idx)), mod.rs:1023: the node expanded next is flagged on its way through the merge -- a
                // candidate in its own lane, y where it stands (no candidate sorts before it: it stays there)
                if (m) {
                    const uint32_t ck_lo_in = ck_lo | (__builtin_amdgcn_inverse_ballot_w64(win_bit) ? 1u : 0u);
#pragma unroll
                    for (int s = 0; s < S; ++s) L.key[s] |= (win_bit == 0 && (uint32_t)s * 64u + lane == ypos) ? 1ull : 0ull;
                    merge_ranked(passm, ((uint64_t)dbits << 32) | ck_lo_in, rk, ef);
                } else {
                    L.mark_expanded(ypos, ykey, lane);
                }
            } else {
                if (m) merge_ranked(passm, ck, rk, ef);
            }
            PT_MARK(6); // merge
            if (!vis.make_room(p.ovf, lane)) bail = true;
            PT_MARK(9); // visited-set housekeeping
            if (bail) return;
            if (last) break;
            } // passes over the row
            if constexpr (WIDE) L.mark_expanded(ypos, ykey, lane); // (two passes: flagged where it stands after the last one)
            xid = yid;
            PT_MARK(12); // flag
        }
    layer_done:;
        }
    }
};

template <int DT, int DIM, int S, bool TRAIL, int V16, bool WIDE = false>
__device__ __forceinline__ void fast_walk_one(const SearchParams& p, const uint32_t qi, uint8_t* smem) {
    const uint32_t lane = threadIdx.x;
    if (p.force_slow) {
        hand_over(p, qi);
        return;
    }
    FastWalker<DT, DIM, S, V16, WIDE> w(p, smem);
    w.load_query(qi);

    if constexpr (TRAIL) { // find_entrypoint_trail (reorder.rs:180-208): every walk starts at node 0
        const uint32_t take = min(min(p.trail_layers, TRAIL_WIDTH), p.n_layers);
        uint32_t mine = 0;
        for (uint32_t l = 0; l < take; ++l) {
            w.search_layer(p.layers[l], 0u, 1u, p.upper_slots);
            if (w.bail) break;
            const uint32_t found = wkey_id(w.L.at(0));
            if (lane == l) mine = found;
        }
        w.vis.release(p.ovf, lane);
        if (w.bail) {
            hand_over(p, qi);
        } else if (lane < TRAIL_WIDTH) {
            p.trail_out[(size_t)qi * TRAIL_WIDTH + lane] = mine;
        }
        return;
    } else {
        uint32_t entrypoint = 0; // mod.rs:989
        float ep_dist = 0.0f;
        for (uint32_t l = 0; l < p.n_layers; ++l) {
            const LayerDev Ly = p.layers[l];
            const bool bottom = (l + 1 == p.n_layers);
#if GRANNE_HIP_PHASE_TIMERS
            w.pt_bottom = bottom;
            if (l == 0) w.pt_t0 = __builtin_amdgcn_s_memtime();
#endif
            w.search_layer(Ly, entrypoint, bottom ? p.ef : 1u, bottom ? p.visited_slots : p.upper_slots, l > 0, ep_dist);
            if (w.bail) break;
            if (!bottom) { // res[0], mod.rs:993: the smallest popped key
                const uint64_t best = w.L.at(0);
                entrypoint = wkey_id(best);
                ep_dist = wkey_dist(best);
            }
        }
        w.vis.release(p.ovf, lane);
        if (w.bail) { // hand the untouched query to the exact global-memory walker
            hand_over(p, qi);
            return;
        }
        // res = the first max_search expanded entries; .take(num_neighbors), mod.rs:974-977
        uint32_t count;
        const QueryIO io = query_io(p, qi);
        if constexpr (FastWalker<DT, DIM, S, V16, WIDE>::LONG) { // the list is M (flushed at the layer's end), in LDS
            const uint32_t want = min(p.ef, p.k);
            uint32_t total = 0;
            const uint32_t nm = p.n_layers > 0 ? w.nM : 0u;
            for (uint32_t w0 = 0; w0 < nm && total < want; w0 += 64u) {
                const uint32_t e = w0 + lane;
                const uint64_t v = w.mslot[e < w.CAP ? e : w.CAP - 1u];
                const bool f = e < nm && (((uint32_t)v) & 1u) != 0u;
                const uint64_t em = wave_ballot(f);
                const uint32_t r = total + mbcnt64(em);
                if (f && r < want) {
                    io.ids[r] = (uint64_t)wkey_id(v);
                    io.dists[r] = wkey_dist(v);
                }
                total += (uint32_t)__popcll(em);
            }
            count = min(total, want);
        } else {
        uint32_t total = 0;
        uint32_t rank[S];
        bool flag[S];
        if (p.n_layers > 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                flag[s] = (((uint32_t)w.L.key[s]) & 1u) && wkey_hi(w.L.key[s]) != 0xFFFFFFFFu;
                const uint64_t em = wave_ballot(flag[s]);
                rank[s] = total + mbcnt64(em);
                total += (uint32_t)__popcll(em);
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) { flag[s] = false; rank[s] = 0; }
        }
        count = min(min(total, p.ef), p.k);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (flag[s] && rank[s] < count) {
                io.ids[rank[s]] = (uint64_t)wkey_id(w.L.key[s]);
                io.dists[rank[s]] = wkey_dist(w.L.key[s]);
            }
        }
        }
        for (uint32_t e = count + lane; e < p.k; e += 64) {
            io.ids[e] = ~0ull;
            io.dists[e] = __builtin_inff();
        }
#if GRANNE_HIP_PHASE_TIMERS
        if (lane == 0 && qi < PHASE_QUERIES) {
            uint64_t* o = g_phase + (size_t)qi * PHASE_SLOTS;
            for (int i = 0; i < 16; ++i) { o[i] = w.pt_b[i]; o[16 + i] = w.pt_u[i]; }
            o[32] = w.pt_nb; o[33] = w.pt_nu;
            o[34] = __builtin_amdgcn_s_memtime() - w.pt_t0;
            for (int i = 0; i < 8; ++i) o[36 + i] = w.pt_cnt[i];
            o[44] = w.vis.pt_rounds;
        }
#endif
        if (lane == 0) {
            *io.count = count;
            if (io.stats) {
                io.stats[0] = w.st.n_dist;
                io.stats[1] = w.st.n_expand;
                io.stats[2] = w.st.n_adj;
            }
        }
    }
}

// Block b walks query b (one wavefront). LDS: [query][S >= 8: the list's mirror, CAP keys][visited front table, if any].
// waves per SIMD the register allocator is asked to keep possible (__launch_bounds__'s second argument is
// per SIMD on AMD; 5 waves = 96 VGPRs, 4 = 128, 3 = 168, 2 = 256). Chosen from the unconstrained
// allocation of each instantiation so that none spills (tools/isa_report.py prints both).
constexpr int fast_waves_per_simd(int DT, int DIM, int S, bool WIDE = false) {
#if GRANNE_HIP_PHASE_TIMERS
    return 1; // the phase clocks live in registers too: no cap, the diagnostics run is one wave per SIMD anyway
#endif
    // lists of 2112 / 4160 keys: 2 registers per 64 keys + the merge's bookkeeping; 17 / 33 KB of LDS mirror each. Two
    // walkers per SIMD for the 33-slot lists (256 registers: measured 137 k against 97 k queries/s at max_search 1600 when the
    // allocation crept to 259), one for the 65-slot ones
    if (walk_list_is_long(S, WIDE)) return S >= 33 ? 2 : 3; // the two-level lists (M in LDS only): 17 / 33 / 66 KB of LDS each bound the walkers per CU before the registers do (17 slots: 9 KB)
    if (DT == DT_I8 && DIM >= 256) return DIM == 256 ? 3 : 2; // 2 / 4 blocks of row data and of query per lane
    if (DT == DT_I8) return S == 1 ? 5 : S <= 4 ? 4 : S == 8 ? 3 : 2;
    if (DIM == 0) return 2; // the streamed walker keeps a group of chunks, the tail and the accumulators: ~210 VGPRs
    if (DIM > 128) return S == 1 ? 3 : 2;
    if (WIDE && S == 2) return 3; // (the second half of the row is one register too many for 128)
    return S == 1 ? (GRANNE_HIP_QUERY_IN_LDS ? 4 : 3) : S == 2 ? 4 : S <= 8 ? 3 : 2;
}

// Blocks nq.. are the tail (slow_kernel.h): they serve the hand-over list inside the same launch.
template <int DT, int DIM, int S, bool TRAIL = false, int V16 = 0, bool WIDE = false>
__global__ __launch_bounds__(64, fast_waves_per_simd(DT, DIM, S, WIDE)) void fast_kernel(const SlowParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    if (blockIdx.x < P.sp.nq) {
        fast_walk_one<DT, DIM, S, TRAIL, V16, WIDE>(P.sp, blockIdx.x, smem);
        walker_done(P);
    } else {
        tail_block<DT>(P, smem);
    }
}

__host__ __device__ inline uint32_t fast_lds_bytes(bool i8, bool gen, uint32_t dim, uint32_t row_bytes, uint32_t S, uint32_t visited_slots, bool seen = false, bool wide = false) {
    const bool lng = walk_list_is_long((int)S, wide);
    // [query][the list's image][lists of up to 17 slots: the cache of entered ids][visited]
    // (lists beyond 1024 keys: M's image, then F's of 128 keys)
    return fast_query_bytes(i8, gen, dim, row_bytes, S) + (64u * S + (lng ? 0u : 32u)) * 8u + (lng ? 128u * 8u : 0u) + (seen ? VCACHE_SLOTS_SEEN : VCACHE_SLOTS) * 4u + visited_slots * 4u;
}

} // namespace granne_hip
