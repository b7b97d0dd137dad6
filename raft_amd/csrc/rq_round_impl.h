// rq_round_impl.h -- the quorum-round kernels, written once over the element
// type T of the term/index/id planes (int32_t or int64_t) and instantiated by
// rq_kernels.hip (int32_t) and rq_kernels64.hip (int64_t).  Internal to
// librq.so; everything here has internal linkage.
//
// One thread owns V consecutive groups, so a wave reads 64*V contiguous elements
// of each slot plane.  The library is built with V = 1 (measured fastest on
// MI355X: 8192 short blocks overlap the read, gather and write phases of
// different blocks; V = 2 / 4 builds were 10-25 % slower); the bodies stay
// written over V.  All quorum arithmetic stays in registers:
//   * commit: compile-time sorting network over the n matchIndex values
//     (v_min/v_max/v_med3), one gathered log-term load, one compare;
//   * vote: the reference's RequestVoteRPC predicate per slot (raft.go:181-223)
//     folded into two bit masks, popcount majority, running max of terms.
// The kernels are HBM-bound integer work; there is no matrix-shaped step.
#pragma once

#include "rq_kernels.h"

#include <algorithm>
#include <cstring>
#include <limits>
#include <type_traits>

namespace {

constexpr int kBlock = 256;            // launch bound
constexpr int kDefaultBlockFixed = 128;   // measured best for the fixed-size round
constexpr int kDefaultBlockCommit = 64;   // commit only: 40 bytes per group, short waves; finer blocks drain sooner (10.4 vs 11.3 us at 1M x 5)
constexpr int kDefaultBlockJoint = 64;    // and for the member-mask round

template <typename T> constexpr T kMinOf = std::numeric_limits<T>::min();

// ---------------------------------------------------------------------------
// vector load/store helpers
// ---------------------------------------------------------------------------
// r[0..V) = p[0..V) (FULL: one aligned vector load; else first `valid` only).
// Every array these helpers touch is device (global) memory: saying so lets the compiler emit
// global_load / global_store instead of flat accesses.
#define RQ_GLOBAL __attribute__((address_space(1)))
template <typename P>
__device__ __forceinline__ RQ_GLOBAL P *gptr(P *p) { return (RQ_GLOBAL P *)p; }
template <int V, bool NT, bool FULL, typename T>
__device__ __forceinline__ void ld_i32(const T *pg, int valid, T (&r)[V]) {
    const RQ_GLOBAL T *p = (const RQ_GLOBAL T *)pg;
    if constexpr (!FULL) {
#pragma unroll
        for (int k = 0; k < V; k++) r[k] = k < valid ? p[k] : 0;
    } else if constexpr (V == 1) {
        r[0] = NT ? __builtin_nontemporal_load(p) : *p;
    } else {
        typedef T vt __attribute__((ext_vector_type(V)));
        const RQ_GLOBAL vt *vp = (const RQ_GLOBAL vt *)p;
        vt v = NT ? __builtin_nontemporal_load(vp) : *vp;
#pragma unroll
        for (int k = 0; k < V; k++) r[k] = v[k];
    }
}

template <int V, bool NT, bool FULL, typename T>
__device__ __forceinline__ void st_i32(T *pg, int valid, const T (&r)[V]) {
    RQ_GLOBAL T *p = (RQ_GLOBAL T *)pg;
    if constexpr (!FULL) {
#pragma unroll
        for (int k = 0; k < V; k++) if (k < valid) p[k] = r[k];
    } else if constexpr (V == 1) {
        if (NT) __builtin_nontemporal_store(r[0], p); else *p = r[0];
    } else {
        typedef T vt __attribute__((ext_vector_type(V)));
        vt v;
#pragma unroll
        for (int k = 0; k < V; k++) v[k] = r[k];
        RQ_GLOBAL vt *vp = (RQ_GLOBAL vt *)p;
        if (NT) __builtin_nontemporal_store(v, vp); else *vp = v;
    }
}

// V 16-bit values (low halves of r[k]) -> one 2*V-byte store.
template <int V, bool NT, bool FULL>
__device__ __forceinline__ void st_u16(uint16_t *pg, int valid, const uint32_t (&r)[V]) {
    RQ_GLOBAL uint16_t *p = (RQ_GLOBAL uint16_t *)pg;
    if constexpr (!FULL) {
#pragma unroll
        for (int k = 0; k < V; k++) if (k < valid) p[k] = (uint16_t)r[k];
    } else if constexpr (V == 1) {
        if (NT) __builtin_nontemporal_store((uint16_t)r[0], p); else *p = (uint16_t)r[0];
    } else if constexpr (V == 2) {
        uint32_t w = (r[0] & 0xFFFFu) | (r[1] << 16);
        RQ_GLOBAL uint32_t *wp = (RQ_GLOBAL uint32_t *)p;
        if (NT) __builtin_nontemporal_store(w, wp); else *wp = w;
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 w;
        w[0] = (r[0] & 0xFFFFu) | (r[1] << 16);
        w[1] = (r[2] & 0xFFFFu) | (r[3] << 16);
        RQ_GLOBAL u32x2 *wp = (RQ_GLOBAL u32x2 *)p;
        if (NT) __builtin_nontemporal_store(w, wp); else *wp = w;
    }
}

// V 8-bit values -> one V-byte store.
template <int V, bool NT, bool FULL>
__device__ __forceinline__ void st_u8(uint8_t *pg, int valid, const uint32_t (&r)[V]) {
    RQ_GLOBAL uint8_t *p = (RQ_GLOBAL uint8_t *)pg;
    if constexpr (!FULL) {
#pragma unroll
        for (int k = 0; k < V; k++) if (k < valid) p[k] = (uint8_t)r[k];
    } else if constexpr (V == 1) {
        if (NT) __builtin_nontemporal_store((uint8_t)r[0], p); else *p = (uint8_t)r[0];
    } else if constexpr (V == 2) {
        const uint16_t w = (uint16_t)((r[0] & 0xFFu) | ((r[1] & 0xFFu) << 8));
        RQ_GLOBAL uint16_t *wp = (RQ_GLOBAL uint16_t *)p;
        if (NT) __builtin_nontemporal_store(w, wp); else *wp = w;
    } else {
        uint32_t w = (r[0] & 0xFFu) | ((r[1] & 0xFFu) << 8) | ((r[2] & 0xFFu) << 16) | (r[3] << 24);
        RQ_GLOBAL uint32_t *wp = (RQ_GLOBAL uint32_t *)p;
        if (NT) __builtin_nontemporal_store(w, wp); else *wp = w;
    }
}

template <int V, bool FULL>
__device__ __forceinline__ void ld_u16(const uint16_t *pg, int valid, uint32_t (&r)[V]) {
    const RQ_GLOBAL uint16_t *p = (const RQ_GLOBAL uint16_t *)pg;
    if constexpr (!FULL) {
#pragma unroll
        for (int k = 0; k < V; k++) r[k] = k < valid ? p[k] : 0;
    } else if constexpr (V == 1) {
        r[0] = *p;
    } else if constexpr (V == 2) {
        uint32_t w = *(const RQ_GLOBAL uint32_t *)p;
        r[0] = w & 0xFFFFu; r[1] = w >> 16;
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 w = *(const RQ_GLOBAL u32x2 *)p;
        r[0] = w[0] & 0xFFFFu; r[1] = w[0] >> 16;
        r[2] = w[1] & 0xFFFFu; r[3] = w[1] >> 16;
    }
}

// ---------------------------------------------------------------------------
// quorum index: ascending_sorted(s)[(N-1)/2] = the (N/2+1)-th largest value,
// i.e. the largest index replicated on a strict majority of the N slots.
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T tmin(T a, T b) { return b < a ? b : a; }
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a < b ? b : a; }

template <typename T>
__device__ __forceinline__ void cex(T &a, T &b) {
    T lo = tmin(a, b), hi = tmax(a, b);
    a = lo; b = hi;
}
template <typename T>
__device__ __forceinline__ T med3(T a, T b, T c) {
    return tmax(tmin(a, b), tmin(tmax(a, b), c));  // int32: folds to v_med3_i32
}

template <int N, typename T>
__device__ __forceinline__ T quorum_index(T (&s)[N]) {
    if constexpr (N == 1) {
        return s[0];
    } else if constexpr (N == 2) {
        return tmin(s[0], s[1]);
    } else if constexpr (N == 3) {
        return med3(s[0], s[1], s[2]);
    } else if constexpr (N == 4) {
        return tmin(tmax(tmin(s[0], s[1]), tmin(s[2], s[3])),
                    tmin(tmax(s[0], s[1]), tmax(s[2], s[3])));
    } else if constexpr (N == 5) {
        return med3(s[4], tmax(tmin(s[0], s[1]), tmin(s[2], s[3])),
                          tmin(tmax(s[0], s[1]), tmax(s[2], s[3])));
    } else if constexpr (N == 6) {
        cex(s[0], s[5]); cex(s[1], s[3]); cex(s[2], s[4]);
        cex(s[1], s[2]); cex(s[3], s[4]);
        cex(s[0], s[3]); cex(s[2], s[5]);
        cex(s[0], s[1]); cex(s[2], s[3]); cex(s[4], s[5]);
        cex(s[1], s[2]); cex(s[3], s[4]);
        return s[2];
    } else if constexpr (N == 7) {
        cex(s[0], s[6]); cex(s[2], s[3]); cex(s[4], s[5]);
        cex(s[0], s[2]); cex(s[1], s[4]); cex(s[3], s[6]);
        cex(s[0], s[1]); cex(s[2], s[5]); cex(s[3], s[4]);
        cex(s[1], s[2]); cex(s[4], s[6]);
        cex(s[2], s[3]); cex(s[4], s[5]);
        cex(s[1], s[2]); cex(s[3], s[4]); cex(s[5], s[6]);
        return s[3];
    } else if constexpr (N == 8) {
        cex(s[0], s[2]); cex(s[1], s[3]); cex(s[4], s[6]); cex(s[5], s[7]);
        cex(s[0], s[4]); cex(s[1], s[5]); cex(s[2], s[6]); cex(s[3], s[7]);
        cex(s[0], s[1]); cex(s[2], s[3]); cex(s[4], s[5]); cex(s[6], s[7]);
        cex(s[2], s[4]); cex(s[3], s[5]);
        cex(s[1], s[4]); cex(s[3], s[6]);
        cex(s[1], s[2]); cex(s[3], s[4]); cex(s[5], s[6]);
        return s[3];
    } else {
        static_assert(N == 9, "fixed-size networks cover 1..9 slots");
        cex(s[0], s[3]); cex(s[1], s[7]); cex(s[2], s[5]); cex(s[4], s[8]);
        cex(s[0], s[7]); cex(s[2], s[4]); cex(s[3], s[8]); cex(s[5], s[6]);
        cex(s[0], s[2]); cex(s[1], s[3]); cex(s[4], s[5]); cex(s[7], s[8]);
        cex(s[1], s[4]); cex(s[3], s[6]); cex(s[5], s[7]);
        cex(s[0], s[1]); cex(s[2], s[4]); cex(s[3], s[5]); cex(s[6], s[8]);
        cex(s[2], s[3]); cex(s[4], s[5]); cex(s[6], s[7]);
        cex(s[1], s[2]); cex(s[3], s[4]); cex(s[5], s[6]);
        return s[4];
    }
}

// ---------------------------------------------------------------------------
// One RequestVoteRPC evaluation (raft.go:181-210 with :212-223 and :234-236).
//   stepdown : term > cur                         (:219, role := Follower :221)
//   cur2     : max(cur, term)                     (:220)
//   reach    : !(term < cur2) == term >= cur      (:190)
//   grant    : reach && (votedFor == -1 || votedFor == candId)   (:202-203)
//                    && lastEntry(log).TermNum <= term           (:204)
//   panic    : reach && log empty                 (:204 -> :235 index panic)
// An empty voter log is encoded as last_log_term == the minimum of T.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void vote_slot(T cur, T vf, T llt, T term, T cid, T &cur2,
                                          uint32_t &stepdown, uint32_t &grant,
                                          uint32_t &panic) {
    // bitwise on purpose: && / || compile to a chain of divergent branches
    stepdown = (uint32_t)(term > cur);
    cur2 = tmax(cur, term);
    const uint32_t reach = (uint32_t)(term >= cur);
    const uint32_t empty = (uint32_t)(llt == kMinOf<T>);
    const uint32_t free_vote = (uint32_t)(vf == (T)-1) | (uint32_t)(vf == cid);
    panic = reach & empty;
    grant = reach & free_vote & (empty ^ 1u) & (uint32_t)(llt <= term);
}

// Pack each thread's V elected bits (groups g0..g0+V-1) into the 1-bit-per-
// group summary.  Must be reached by every lane of the wave.
template <int V>
__device__ __forceinline__ void pack_elected(uint32_t field, uint64_t g0, uint32_t G,
                                             uint64_t *bits) {
    if constexpr (V == 1) {
        unsigned long long w = __ballot(field & 1u);
        if ((threadIdx.x & 63) == 0 && g0 < G) ((RQ_GLOBAL uint64_t *)bits)[g0 >> 6] = w;
    } else {
        constexpr int lanes_per_word = 64 / V;
        unsigned long long w = (unsigned long long)field << (g0 & 63);
#pragma unroll
        for (int o = lanes_per_word / 2; o > 0; o >>= 1) w |= __shfl_xor(w, o, 64);
        if ((threadIdx.x & (lanes_per_word - 1)) == 0 && g0 < G) ((RQ_GLOBAL uint64_t *)bits)[g0 >> 6] = w;
    }
}

constexpr uint8_t kStLogOob = 0x01, kStEmptyLog = 0x02, kStBadCsr = 0x04;

// The one dependent, data-indexed load of a round: log[N].TermNum.  Arena
// entries are `bits` wide -- 8 or 16 (unsigned), 32 (signed) or the full element,
// a wave-uniform choice.  The cache policy (a.nt_gather) is wave-uniform as well.
template <typename P>
__device__ __forceinline__ P load_one(const P *pg, bool nt) {
    const RQ_GLOBAL P *p = (const RQ_GLOBAL P *)pg;
    // two instructions behind a scalar branch: the hint is part of the encoding
    if (nt) return __builtin_nontemporal_load(p);
    return *p;
}
// A scalar branch around loads of the entry's own width: for callers that have
// nothing else in flight when they look a term up (the AppendEntries batch).
template <typename T>
__device__ __forceinline__ T load_term(const void *arena, uint32_t bits, size_t idx, bool nt) {
    if (bits == 8) return (T)load_one(static_cast<const uint8_t *>(arena) + idx, nt);
    if (bits == 16) return (T)load_one(static_cast<const uint16_t *>(arena) + idx, nt);
    if constexpr (sizeof(T) == 8) {
        if (bits == 32) return (T)load_one(static_cast<const int32_t *>(arena) + idx, nt);
    }
    return load_one(static_cast<const T *>(arena) + idx, nt);
}
// The rounds look the term up while their streamed planes are still arriving, and
// there the branch above costs the overlap: before the narrow loads the compiler
// waits for EVERY outstanding load (their address registers might be the target of
// the other width's load, for all it can tell).  So entries of every width are
// fetched by one load instruction whatever their width -- the aligned word (as wide
// as the layout's element) that holds the entry -- and term_of() cuts the entry out
// where the term is consumed, late.  The word never leaves the page the entry is
// in; entries are naturally aligned (checked at launch).
template <typename T>
struct term_word {
    typename std::make_unsigned<T>::type raw;
    uint32_t shift;
};
template <typename T>
__device__ __forceinline__ term_word<T> load_term_word(const void *arena, uint32_t bits, size_t idx, bool nt) {
    typedef typename std::make_unsigned<T>::type U;          // the word: as wide as the layout's element
    const uint32_t lg = bits == 8 ? 0u : bits == 16 ? 1u : bits == 32 ? 2u : 3u;
    const uintptr_t at = reinterpret_cast<uintptr_t>(arena) + (idx << lg);
    const U w = load_one(reinterpret_cast<const U *>(at & ~(uintptr_t)(sizeof(U) - 1)), nt);
    return term_word<T>{w, (uint32_t)(at & (sizeof(U) - 1)) * 8u};
}
template <typename T>
__device__ __forceinline__ T term_of(term_word<T> w, uint32_t bits) {
    typedef typename std::make_unsigned<T>::type U;
    const U v = w.raw >> w.shift;
    if (bits == 8 * sizeof(T)) return (T)v;
    if (bits == 32) return (T)(int32_t)(uint32_t)v;          // int32 entries of the wide layout
    return (T)(v & (U)(bits == 8 ? 0xFFu : 0xFFFFu));
}

// A group's slice [beg, end) of a CSR arena of `total` entries is usable when the
// offsets are ordered and stay inside the arena; then beg + i < total for every
// i < end - beg, so a lookup guarded by it cannot leave the arena whatever the
// offsets hold.
__device__ __forceinline__ bool slice_ok(uint32_t beg, uint32_t end, uint64_t total) {
    return beg <= end && (uint64_t)end <= total;
}
// With a length plane group g's log is the `len` entries from `beg` on, wherever
// that lies in the arena (a resident arena whose slices have room to grow and
// move when they run out of it): usable when it ends inside the arena.
__device__ __forceinline__ bool span_ok(uint32_t beg, uint32_t len, uint64_t total) {
    return (uint64_t)beg + (uint64_t)len <= total;
}

// Narrow commit summary: commit' as an offset from a per-group base in `bits`
// (16, 8 or 4) bits, the all-ones value meaning "does not fit, fetch the full
// vector" (counted in summary_escapes).  When the base is the round's own input
// commit plane the offset is the round's advance and the value already in
// registers is used.  16- and 8-bit values are stored here; 4-bit values are
// returned (4 bits per group from bit 0) and packed by pack_nibbles() where the
// whole wave is convergent.
template <int V, bool NTS, bool FULL, typename T>
__device__ __forceinline__ uint32_t store_commit_summary(const rq_round_args_t<T> &a, uint32_t g0, int valid,
                                                         const T (&co)[V], const T (&cm)[V],
                                                         uint32_t &escapes_out) {
    const uint32_t bits = a.summary_bits;
    const uint32_t escape = (1u << bits) - 1u;
    const bool base_is_input = a.summary_base == a.commit;
    uint32_t lo[V];
    uint32_t escapes = 0, nibbles = 0;
#pragma unroll
    for (int k = 0; k < V; k++) {
        T base = 0;
        if (base_is_input) base = cm[k];
        else if (a.summary_base && k < valid) base = ((const RQ_GLOBAL T *)a.summary_base)[g0 + k];
        // unsigned difference: exact whenever co >= base, whatever the magnitudes
        const uint64_t d = (uint64_t)(int64_t)co[k] - (uint64_t)(int64_t)base;
        const bool fit = co[k] >= base && d < escape;
        lo[k] = fit ? (uint32_t)d : escape;
        escapes += (!fit && k < valid) ? 1u : 0u;
        nibbles |= (k < valid ? lo[k] : 0u) << (4 * k);
    }
    if (bits == 16) st_u16<V, NTS, FULL>(a.commit_lo16 + g0, valid, lo);
    else if (bits == 8) st_u8<V, NTS, FULL>(reinterpret_cast<uint8_t *>(a.commit_lo16) + g0, valid, lo);
    escapes_out = escapes;      // summed over the wave by count_escapes(): one atomic per wave
    return nibbles;
}

// Adds the wave's groups that did not fit the summary window to the caller's
// counter with one atomic per wave (a counter hit by every escaping thread
// serialises in L2: 19k escapes in a 1M-group launch cost 100 us that way).
// The count is exact up to kEscapeCountCap and stops growing soon after: a width
// that most groups do not fit is a configuration error the host needs to see,
// not to measure, and must not slow the round down.
// Must be reached by every lane of the wave.
constexpr uint32_t kEscapeCountCap = 1u << 16;
__device__ __forceinline__ void count_escapes(uint32_t escapes, uint32_t *counter) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) escapes += __shfl_xor(escapes, o, 64);
    if ((threadIdx.x & 63) == 0 && escapes) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kEscapeCountCap)
            atomicAdd(counter, escapes);
    }
}

// 4-bit summary: 16 groups per 64-bit word (group g in bits 4*(g%16) of word
// g/16).  `field` holds the thread's V nibbles, zero for groups outside the
// batch.  Must be reached by every lane of the wave.
template <int V>
__device__ __forceinline__ void pack_nibbles(uint32_t field, uint64_t g0, uint32_t G, uint64_t *words) {
    constexpr int lanes_per_word = 16 / V;
    unsigned long long w = (unsigned long long)field << (4 * (g0 & 15));
#pragma unroll
    for (int o = lanes_per_word / 2; o > 0; o >>= 1) w |= __shfl_xor(w, o, 64);
    if ((threadIdx.x & (lanes_per_word - 1)) == 0 && g0 < G) ((RQ_GLOBAL uint64_t *)words)[g0 >> 4] = w;
}

// What a round body hands back to the kernel for the wave-level packing.
struct round_fields {
    uint32_t elected;   // bit k: group g0+k elected
    uint32_t nibbles;   // 4 bits per group: 4-bit commit summary
    uint32_t escapes;   // groups of this thread whose summary offset did not fit
    uint32_t changed;   // bit k: the round moved commitIndex, stepped a voter down, elected or flagged group g0+k
};

// Slot of log[N] inside the group's arena slice, as an unsigned number: with a
// window, arena slot i holds log[base + i]; N below the base wraps to a huge
// value and is reported like any other N outside the arena.
template <typename T>
__device__ __forceinline__ typename std::make_unsigned<T>::type window_slot(T nq, T base) {
    typedef typename std::make_unsigned<T>::type U;
    return (U)nq - (U)base;
}

// ---------------------------------------------------------------------------
// Where group g's element of a field lives.  Plane layout: field[p * ld + g].
// Tiled layout (a.tile_bytes != 0): the groups are cut into tiles of 64, a tile
// holds its 64 groups' elements of every input field back to back (tile_bytes
// per tile; the outputs likewise in a block of their own, out_tile_bytes per
// tile), a field pointer names the field's place inside tile 0, slot p of a
// field follows slot p - 1 after 64 elements -- so a wave reads one contiguous
// tile and writes one instead of touching 26 planes 4 MiB apart.  Reads and
// writes in separate address streams: one mixed tile gained nothing, two
// blocks 5 % at 1M x 5.
// ---------------------------------------------------------------------------
struct group_addr {
    size_t tile_off;     // bytes
    size_t slot_stride;  // elements
    size_t lane;         // elements
};
template <typename T>
__device__ __forceinline__ group_addr addr_of(const rq_round_args_t<T> &a, uint32_t g) {
    if (a.tile_bytes) return group_addr{(size_t)(g >> 6) * a.tile_bytes, 64, g & 63u};
    return group_addr{0, a.ld, g};
}
template <typename P>
__device__ __forceinline__ P *at(P *base, const group_addr &ga, uint32_t p = 0) {
    return reinterpret_cast<P *>(reinterpret_cast<uintptr_t>(base) + ga.tile_off) + (p * ga.slot_stride + ga.lane);
}

// ---------------------------------------------------------------------------
// Start signal.  A stream that consumes the outputs of the rounds enqueued so
// far (the summary all-gather of a sharded deployment) normally waits for an
// event recorded behind the last of them; that marker keeps the next round from
// starting while the previous one drains and costs 8-13 us per dependency when
// a round is 26 us long.  Instead the NEXT launch on the compute stream reports
// that it has started -- which, a stream being in order, means every earlier
// launch on it has completed and its outputs are visible -- by adding one to
// start_signal, a word the consumer's stream polls (rq_stream_wait_signal).
// One atomic by one thread: nothing is added to the stream but kernels.
// (Reporting the END of a launch from inside it was built first and rejected:
// a release fence per block wrote the L2 back 8192 times, 550 us per launch;
// counting blocks with write-through outputs instead still cost 4.5 us.)
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void signal_started(const rq_round_args_t<T> &a) {
    if (a.start_signal && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_fetch_add(a.start_signal, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// fixed cluster size n = N for every group
// ---------------------------------------------------------------------------
template <int N, int V, int WHAT, bool NT, bool FULL, typename T>
__device__ __forceinline__ round_fields round_fixed_body(const rq_round_args_t<T> &a,
                                                     uint32_t g0, int valid) {
    typedef typename std::make_unsigned<T>::type U;
    constexpr bool NTL = NT, NTS = NT;
    constexpr bool DO_COMMIT = (WHAT & RQ_WHAT_COMMIT) != 0;
    constexpr bool DO_VOTE = (WHAT & RQ_WHAT_VOTE) != 0;
    const group_addr ga = addr_of(a, g0);

    T mt[N][V], cm[V], lt[V], lb[V], lbs[V], tb[V];
    int32_t off[V + 1], ln[V], lns[V];
#pragma unroll
    for (int k = 0; k < V; k++) { lb[k] = 0; lbs[k] = 0; tb[k] = 0; ln[k] = 0; lns[k] = 0; lt[k] = 0; }
    T vt[N][V], vf[N][V], ll[N][V], ct[V], ci[V];
    uint32_t st[V];
#pragma unroll
    for (int k = 0; k < V; k++) st[k] = 0;

    // ---- issue every streamed load up front.  First what the dependent log-term gather needs
    // (matchIndex, commitIndex, the log's place in the arena): loads complete in the order they
    // were issued, so the gather can go out while the vote planes are still on their way.  The
    // first five arrays are also the ones whose pointers arrive with the wave (the kernel's
    // leading arguments); the rest of the arguments names the others ----
    const bool lookup = a.log_off != nullptr;       // uniform; the launcher clears log_off when log_term_start replaces the arena
    if constexpr (DO_COMMIT) {
#pragma unroll
        for (int p = 0; p < N; p++) ld_i32<V, NTL, FULL>(at(a.match, ga, p), valid, mt[p]);
        ld_i32<V, NTL, FULL>(at(a.commit, ga), valid, cm);
#pragma unroll
        for (int k = 0; k <= V; k++) off[k] = 0;
        if (lookup) {
            const int32_t *offp = at(reinterpret_cast<const int32_t *>(a.log_off), ga);
            int32_t o[V];
            ld_i32<V, NTL, FULL>(offp, valid, o);
            // plain on purpose: it re-reads the line the vector load above brought in
            // (non-temporal here bypasses L1 and was measured 1 us slower per round)
            const int32_t oend = *(const RQ_GLOBAL int32_t *)at(reinterpret_cast<const int32_t *>(a.log_off), addr_of(a, g0 + valid));  // log_off has G+1 entries
#pragma unroll
            for (int k = 0; k < V; k++) off[k] = k < valid ? o[k] : oend;
            off[V] = oend;
            // what else the gather's address needs; these pointers come with the rest of the arguments
            if (a.log_base) ld_i32<V, NTL, FULL>(at(a.log_base, ga), valid, lb);
            if (a.log_len) ld_i32<V, NTL, FULL>(at(reinterpret_cast<const int32_t *>(a.log_len), ga), valid, ln);
        }
    }
    if constexpr (DO_VOTE) {
#pragma unroll
        for (int p = 0; p < N; p++) {
            ld_i32<V, NTL, FULL>(at(a.voter_term, ga, p), valid, vt[p]);
            ld_i32<V, NTL, FULL>(at(a.voted_for, ga, p), valid, vf[p]);
        }
    }
    if constexpr (DO_VOTE) {
#pragma unroll
        for (int p = 0; p < N; p++) ld_i32<V, NTL, FULL>(at(a.last_log_term, ga, p), valid, ll[p]);
        ld_i32<V, NTL, FULL>(at(a.cand_term, ga), valid, ct);
        ld_i32<V, NTL, FULL>(at(a.cand_id, ga), valid, ci);
    }
    const bool by_start = !lookup;
    if constexpr (DO_COMMIT) {
        if (lookup) {
            ld_i32<V, NTL, FULL>(at(a.leader_term, ga), valid, lt);
            if (a.log_term_base) ld_i32<V, NTL, FULL>(at(a.log_term_base, ga), valid, tb);
        } else {
            // no gather to hurry, so nothing here was worth waiting for the arguments.  Registers of
            // their own: a second load into lb / ln down here would make the gather above wait for it
            if (a.log_base) ld_i32<V, NTL, FULL>(at(a.log_base, ga), valid, lbs);
            ld_i32<V, NTL, FULL>(at(reinterpret_cast<const int32_t *>(a.log_len), ga), valid, lns);
            // log_term_start itself is read by the load that is the gather in the other mode (below)
        }
    }

    group_addr go = ga;                       // outputs: their own tile block ...
    if (a.tile_bytes) go.tile_off = (size_t)(g0 >> 6) * a.out_tile_bytes;
    // ... unless they are the inputs themselves (state updated in place)
    const group_addr gco = (const void *)a.commit_out == (const void *)a.commit ? ga : go;
    const group_addr gvo = (const void *)a.voter_term_out == (const void *)a.voter_term ? ga : go;

    // ---- commit: quorum index, then the dependent log-term gather ----
    T nq[V];
    term_word<T> gw[V];
    size_t ask[V];
    bool adv[V], inb[V];
    if constexpr (DO_COMMIT) {
#pragma unroll
        for (int k = 0; k < V; k++) {
            T s[N];
#pragma unroll
            for (int p = 0; p < N; p++) s[p] = mt[p][k];
            nq[k] = quorum_index<N>(s);
            adv[k] = (k < valid) && nq[k] > cm[k];
            if (lookup) {
                const uint32_t beg = (uint32_t)off[k], end = (uint32_t)off[k + 1];
                const bool csr = a.log_len ? span_ok(beg, (uint32_t)ln[k], a.log_terms) : slice_ok(beg, end, a.log_terms);
                const uint32_t len = a.log_len ? (uint32_t)ln[k] : end - beg;
                const U wi = window_slot(nq[k], lb[k]);
                inb[k] = csr && wi < (U)len;
                st[k] |= (k < valid && !csr) ? kStBadCsr : 0;
                ask[k] = (adv[k] && inb[k]) ? (size_t)beg + (size_t)wi : (size_t)0;
            } else {
                inb[k] = window_slot(nq[k], lbs[k]) < (U)(uint32_t)lns[k];
                ask[k] = 0;
            }
            // Every lane of every launch issues this last load: the looked-up term; without an arena
            // the group's log_term_start; word 0 of the commit plane when a lane has nothing to ask
            // (result unused).  A branch around it, scalar or per lane, would make the vote below
            // wait for all outstanding loads, this one included, where it now waits for all but it.
            const bool can = lookup && a.log_terms != 0;
            const void *src = can ? a.log_term : (const void *)a.commit;
            if (by_start && k < valid) src = at(a.log_term_start, ga) + k;
            gw[k] = load_term_word<T>(src, can ? a.log_term_bits : by_start ? 8u * (uint32_t)sizeof(T) : 32u, ask[k],
                                      by_start || (can && a.nt_gather != 0));
        }
    }

    // ---- vote: per-slot predicate, tally ----
    uint32_t el_field = 0, nib_field = 0, esc_field = 0, ch_field = 0;
    if constexpr (DO_VOTE) {
        uint32_t gm[V], sm[V], el[V];
        T mx[V];
#pragma unroll
        for (int k = 0; k < V; k++) { gm[k] = 0; sm[k] = 0; mx[k] = kMinOf<T>; }
#pragma unroll
        for (int p = 0; p < N; p++) {
            T nt[V];
#pragma unroll
            for (int k = 0; k < V; k++) {
                uint32_t sd, gr, pn;
                vote_slot(vt[p][k], vf[p][k], ll[p][k], ct[k], ci[k], nt[k], sd, gr, pn);
                gm[k] |= gr << p;
                sm[k] |= sd << p;
                mx[k] = tmax(mx[k], nt[k]);
                st[k] |= pn ? kStEmptyLog : 0;
            }
            st_i32<V, NTS, FULL>(at(a.voter_term_out, gvo, p), valid, nt);
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            el[k] = (2 * __popc(gm[k]) > N) && (mx[k] == ct[k]);
            if (k < valid) el_field |= el[k] << k;
            if (k < valid) ch_field |= (uint32_t)(el[k] != 0 || sm[k] != 0) << k;
        }
        st_u16<V, NTS, FULL>(at(a.grant_mask, go), valid, gm);
        st_u16<V, NTS, FULL>(at(a.stepdown_mask, go), valid, sm);
        st_u8<V, NTS, FULL>(at(a.elected, go), valid, el);
        st_i32<V, NTS, FULL>(at(a.max_term, go), valid, mx);
    }

    // ---- commit: finish ----
    if constexpr (DO_COMMIT) {
        T co[V];
#pragma unroll
        for (int k = 0; k < V; k++) {
            // the looked-up term (or log_term_start) is consumed here, late
            const T got = term_of<T>(gw[k], by_start ? 8u * (uint32_t)sizeof(T) : a.log_term_bits);
            const bool hit = by_start ? nq[k] >= got : (T)((U)got + (U)tb[k]) == lt[k];
            co[k] = (adv[k] && inb[k] && hit) ? nq[k] : cm[k];
            st[k] |= (adv[k] && !inb[k]) ? kStLogOob : 0;
            if (k < valid) ch_field |= (uint32_t)(co[k] != cm[k]) << k;
        }
        st_i32<V, NTS, FULL>(at(a.commit_out, gco), valid, co);
        if (a.commit_lo16) nib_field = store_commit_summary<V, NTS, FULL>(a, g0, valid, co, cm, esc_field);
    }
    if (a.status_out) st_u8<V, NTS, FULL>(at(a.status_out, go), valid, st);
#pragma unroll
    for (int k = 0; k < V; k++)
        if (k < valid) ch_field |= (uint32_t)(st[k] != 0) << k;
    return round_fields{el_field, nib_field, esc_field, ch_field};
}

// The arguments a wave needs before it can issue its first loads -- the group count, the block
// size, the layout's stride, the three arrays the log-term lookup depends on and the first two
// vote planes -- travel as leading scalar kernel arguments, 14 dwords: the build preloads those
// into SGPRs at wave launch (kernarg preload), so the streamed loads go out at once while the
// rest of the argument structure is still on its way from memory.  stride: bit 63 set = 64-group tiles of
// (stride & ~bit 63) bytes, else the plane stride ld in elements.
constexpr uint64_t kStrideTiled = 1ull << 63;
template <int N, int V, int WHAT, bool NT, typename T>
__global__ __launch_bounds__(kBlock) void rq_round_fixed(uint32_t G, uint32_t block, uint64_t stride,
                                                         const T *match, const T *commit, const uint32_t *log_off,
                                                         const T *voter_term, const T *voted_for,
                                                         rq_round_args_t<T> rest) {
    rq_round_args_t<T> a = rest;
    a.G = G;
    a.tile_bytes = (stride & kStrideTiled) ? (stride & ~kStrideTiled) : 0;
    a.ld = (stride & kStrideTiled) ? 0u : (uint32_t)stride;
    a.match = match; a.commit = commit; a.log_off = log_off; a.voter_term = voter_term; a.voted_for = voted_for;
    const uint64_t g0 = ((uint64_t)blockIdx.x * block + threadIdx.x) * V;
    round_fields f{0, 0, 0, 0};
    if (g0 + V <= a.G)
        f = round_fixed_body<N, V, WHAT, NT, true>(a, (uint32_t)g0, V);
    else if (g0 < a.G)
        f = round_fixed_body<N, V, WHAT, NT, false>(a, (uint32_t)g0, (int)(a.G - g0));
    signal_started(a);          // any time while the launch runs: it only has to have started
    if constexpr ((WHAT & RQ_WHAT_VOTE) != 0) {
        if (a.elected_bits) pack_elected<V>(f.elected, g0, a.G, a.elected_bits);
    }
    if (a.changed_bits) pack_elected<V>(f.changed, g0, a.G, a.changed_bits);
    if constexpr ((WHAT & RQ_WHAT_COMMIT) != 0) {
        if (a.commit_lo16 && a.summary_bits == 4)
            pack_nibbles<V>(f.nibbles, g0, a.G, reinterpret_cast<uint64_t *>(a.commit_lo16));
        if (a.commit_lo16 && a.summary_escapes) count_escapes(f.escapes, a.summary_escapes);
    }
}

// ---------------------------------------------------------------------------
// Quorum index over a member subset: non-members are replaced by the minimum
// of T, the vector is sorted in DESCENDING order by a Batcher odd-even merge
// network (19 compare-exchanges for 8 inputs, 63 for 16) and element |cfg|/2
// -- the (|cfg|/2+1)-th largest member value -- is selected.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void cexd(T &a, T &b) {
    T hi = tmax(a, b), lo = tmin(a, b);
    a = hi; b = lo;
}
#define D(i, j) cexd(s[i], s[j]);
template <typename T>
__device__ __forceinline__ void sort_desc8(T (&s)[8]) {
    D(0,1) D(2,3) D(4,5) D(6,7) D(0,2) D(1,3) D(4,6) D(5,7) D(1,2) D(5,6) D(0,4) D(1,5) D(2,6)
    D(3,7) D(2,4) D(3,5) D(1,2) D(3,4) D(5,6)
}
template <typename T>
__device__ __forceinline__ void sort_desc16(T (&s)[16]) {
    D(0,1) D(2,3) D(4,5) D(6,7) D(8,9) D(10,11) D(12,13) D(14,15) D(0,2) D(1,3) D(4,6) D(5,7)
    D(8,10) D(9,11) D(12,14) D(13,15) D(1,2) D(5,6) D(9,10) D(13,14) D(0,4) D(1,5) D(2,6)
    D(3,7) D(8,12) D(9,13) D(10,14) D(11,15) D(2,4) D(3,5) D(10,12) D(11,13) D(1,2) D(3,4)
    D(5,6) D(9,10) D(11,12) D(13,14) D(0,8) D(1,9) D(2,10) D(3,11) D(4,12) D(5,13) D(6,14)
    D(7,15) D(4,8) D(5,9) D(6,10) D(7,11) D(2,4) D(3,5) D(6,8) D(7,9) D(10,12) D(11,13) D(1,2)
    D(3,4) D(5,6) D(7,8) D(9,10) D(11,12) D(13,14)
}
#undef D

// W = 8 or 16 slots considered; mask has no bit at or above W.
template <int W, int MMAX, int V, typename T>
__device__ __forceinline__ T masked_quorum(const T (&mt)[MMAX][V], int k, uint32_t mask) {
    T s[W];
#pragma unroll
    for (int p = 0; p < W; p++) s[p] = ((mask >> p) & 1u) ? mt[p][k] : kMinOf<T>;
    if constexpr (W == 8) sort_desc8(s); else sort_desc16(s);
    const int idx = __popc(mask) >> 1;
    T r = s[0];
#pragma unroll
    for (int i = 1; i <= W / 2; i++) r = (idx == i) ? s[i] : r;
    return r;
}

// ---------------------------------------------------------------------------
// per-group membership: m slot planes, old/new member masks (joint consensus,
// mixed cluster sizes).  Null masks mean "all m slots, one configuration".
// ---------------------------------------------------------------------------
// LATE_VOTE: the vote side's first chunk of loads is issued after the commit side's
// sort instead of before it -- fewer loads in flight, a third fewer registers.
template <int MMAX, int V, int WHAT, bool NT, bool FULL, typename T, bool LATE_VOTE = false>
__device__ __forceinline__ round_fields round_joint_body(const rq_round_args_t<T> &a,
                                                     uint32_t g0, int valid,
                                                     bool in_range) {
    typedef typename std::make_unsigned<T>::type U;
    constexpr bool NTL = NT, NTS = NT;
    constexpr bool DO_COMMIT = (WHAT & RQ_WHAT_COMMIT) != 0;
    constexpr bool DO_VOTE = (WHAT & RQ_WHAT_VOTE) != 0;
    const group_addr ga = addr_of(a, g0);
    group_addr go = ga;                       // outputs: their own tile block ...
    if (a.tile_bytes) go.tile_off = (size_t)(g0 >> 6) * a.out_tile_bytes;
    // ... unless they are the inputs themselves (state updated in place)
    const group_addr gco = (const void *)a.commit_out == (const void *)a.commit ? ga : go;
    const group_addr gvo = (const void *)a.voter_term_out == (const void *)a.voter_term ? ga : go;
    const uint32_t m = a.m;
    const uint32_t lim = m >= 16 ? 0xFFFFu : ((1u << m) - 1u);

    uint32_t om[V], nm[V], mem[V];
    uint32_t anymem = 0;
    if (in_range) {
        if (a.old_mask) {
            ld_u16<V, FULL>(at(a.old_mask, ga), valid, om);
            ld_u16<V, FULL>(at(a.new_mask, ga), valid, nm);
        } else {
#pragma unroll
            for (int k = 0; k < V; k++) om[k] = nm[k] = (k < valid) ? lim : 0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < V; k++) om[k] = nm[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < V; k++) {
        om[k] &= lim; nm[k] &= lim;
        mem[k] = om[k] | nm[k];
        anymem |= mem[k];
    }
    // slots no lane of this wave uses are skipped with scalar branches
    uint32_t wm = 0;
#pragma unroll
    for (int p = 0; p < MMAX; p++)
        if (__ballot((anymem >> p) & 1u) != 0ull) wm |= 1u << p;
    wm = __builtin_amdgcn_readfirstlane(wm);

    uint32_t st[V];
#pragma unroll
    for (int k = 0; k < V; k++) st[k] = 0;

    // ---- vote: slots are processed in chunks of kChunk; a chunk's 3*kChunk plane
    // loads are all issued before any of them is consumed, and chunk 0 is issued
    // before the commit side's sort so that it is in flight behind it.
    constexpr int kChunk = 8;
    uint32_t el_field = 0, nib_field = 0, esc_field = 0, ch_field = 0;
    T ct[V], ci[V], mx[V];
    uint32_t gm[V], sm[V];
    T cvt[kChunk][V], cvf[kChunk][V], cll[kChunk][V];
    auto issue_chunk = [&](int c0) {
#pragma unroll
        for (int q = 0; q < kChunk; q++) {
            const int p = c0 + q;
#pragma unroll
            for (int k = 0; k < V; k++) { cvt[q][k] = 0; cvf[q][k] = 0; cll[q][k] = 0; }
            if (p < MMAX && ((wm >> p) & 1u)) {
                if ((anymem >> p) & 1u) {
                    ld_i32<V, NTL, FULL>(at(a.voter_term, ga, p), valid, cvt[q]);
                    ld_i32<V, NTL, FULL>(at(a.voted_for, ga, p), valid, cvf[q]);
                    ld_i32<V, NTL, FULL>(at(a.last_log_term, ga, p), valid, cll[q]);
                }
            }
        }
    };
    auto compute_chunk = [&](int c0) {
#pragma unroll
        for (int q = 0; q < kChunk; q++) {
            const int p = c0 + q;
            if (!(p < MMAX && ((wm >> p) & 1u))) continue;
            if (!((anymem >> p) & 1u)) continue;
            T nt[V];
            bool all_members = true;
#pragma unroll
            for (int k = 0; k < V; k++) {
                const bool member = (mem[k] >> p) & 1u;
                uint32_t sd, gr, pn;
                vote_slot(cvt[q][k], cvf[q][k], cll[q][k], ct[k], ci[k], nt[k], sd, gr, pn);
                if (member) {
                    gm[k] |= gr << p;
                    sm[k] |= sd << p;
                    mx[k] = tmax(mx[k], nt[k]);
                    st[k] |= pn ? kStEmptyLog : 0;
                } else {
                    nt[k] = cvt[q][k];
                    if (k < valid) all_members = false;
                }
            }
            if (all_members) {
                st_i32<V, NTS, FULL>(at(a.voter_term_out, gvo, p), valid, nt);
            } else {
#pragma unroll
                for (int k = 0; k < V; k++)
                    if (k < valid && ((mem[k] >> p) & 1u)) at(a.voter_term_out, gvo, p)[k] = nt[k];
            }
        }
    };
    if constexpr (DO_VOTE) {
#pragma unroll
        for (int k = 0; k < V; k++) { ct[k] = 0; ci[k] = 0; gm[k] = 0; sm[k] = 0; mx[k] = kMinOf<T>; }
        if (in_range) {
            ld_i32<V, NTL, FULL>(at(a.cand_term, ga), valid, ct);
            ld_i32<V, NTL, FULL>(at(a.cand_id, ga), valid, ci);
        }
        if constexpr (!LATE_VOTE) issue_chunk(0);
    }

    T nq[V], gt[V], cm[V], lt[V], ts[V];
    bool adv[V], inb[V];
    const bool by_start = a.log_term_start != nullptr;         // uniform: no arena lookup
    if constexpr (DO_COMMIT) {
        T mt[MMAX][V];
        int32_t off[V + 1];
#pragma unroll
        for (int p = 0; p < MMAX; p++) {
#pragma unroll
            for (int k = 0; k < V; k++) mt[p][k] = 0;
            if ((wm >> p) & 1u) {
                if ((anymem >> p) & 1u) ld_i32<V, NTL, FULL>(at(a.match, ga, p), valid, mt[p]);
            }
        }
        T lb[V], tb[V];
        int32_t ln[V];
#pragma unroll
        for (int k = 0; k < V; k++) { cm[k] = 0; lt[k] = 0; off[k] = 0; lb[k] = 0; tb[k] = 0; ln[k] = 0; ts[k] = 0; }
        off[V] = 0;
        if (in_range) {
            ld_i32<V, NTL, FULL>(at(a.commit, ga), valid, cm);
            if (a.log_term_start) {
                if (a.log_base) ld_i32<V, NTL, FULL>(at(a.log_base, ga), valid, lb);
                ld_i32<V, NTL, FULL>(at(reinterpret_cast<const int32_t *>(a.log_len), ga), valid, ln);
                ld_i32<V, NTL, FULL>(at(a.log_term_start, ga), valid, ts);
            } else {
                ld_i32<V, NTL, FULL>(at(a.leader_term, ga), valid, lt);
                if (a.log_base) ld_i32<V, NTL, FULL>(at(a.log_base, ga), valid, lb);
                if (a.log_term_base) ld_i32<V, NTL, FULL>(at(a.log_term_base, ga), valid, tb);
                if (a.log_len) ld_i32<V, NTL, FULL>(at(reinterpret_cast<const int32_t *>(a.log_len), ga), valid, ln);
                const int32_t *offp = at(reinterpret_cast<const int32_t *>(a.log_off), ga);
                int32_t o[V];
                ld_i32<V, NTL, FULL>(offp, valid, o);
                const int32_t oend = *(const RQ_GLOBAL int32_t *)at(reinterpret_cast<const int32_t *>(a.log_off), addr_of(a, g0 + valid));
#pragma unroll
                for (int k = 0; k < V; k++) off[k] = k < valid ? o[k] : oend;
                off[V] = oend;
            }
        }
        // one configuration for the whole wave (the common case) needs one sort
        bool two_cfgs = false;
#pragma unroll
        for (int k = 0; k < V; k++) two_cfgs = two_cfgs || (om[k] != nm[k]);
        const bool wave_two_cfgs = __ballot(two_cfgs) != 0ull;
        const bool wave_small = MMAX == 8 || wm < 256u;
#pragma unroll
        for (int k = 0; k < V; k++) {
            T no, nn;
            if (wave_small) {
                no = masked_quorum<8, MMAX, V>(mt, k, om[k]);
                nn = wave_two_cfgs ? masked_quorum<8, MMAX, V>(mt, k, nm[k]) : no;
            } else if constexpr (MMAX == 16) {
                no = masked_quorum<16, MMAX, V>(mt, k, om[k]);
                nn = wave_two_cfgs ? masked_quorum<16, MMAX, V>(mt, k, nm[k]) : no;
            }
            nq[k] = tmin(no, nn);
            const uint32_t beg = (uint32_t)off[k], end = (uint32_t)off[k + 1];
            const bool csr = by_start || (a.log_len ? span_ok(beg, (uint32_t)ln[k], a.log_terms) : slice_ok(beg, end, a.log_terms));
            const uint32_t len = a.log_len ? (uint32_t)ln[k] : end - beg;
            const U wi = window_slot(nq[k], lb[k]);
            adv[k] = (k < valid) && mem[k] != 0 && nq[k] > cm[k];
            inb[k] = csr && wi < (U)len;
            st[k] |= (k < valid && !csr) ? kStBadCsr : 0;
            gt[k] = 0;
            if (!by_start && adv[k] && inb[k])
                gt[k] = (T)((U)load_term<T>(a.log_term, a.log_term_bits, (size_t)beg + (size_t)wi, a.nt_gather != 0) + (U)tb[k]);
        }
    }

    if constexpr (DO_VOTE) {
        if constexpr (LATE_VOTE) issue_chunk(0);
        compute_chunk(0);
        if constexpr (MMAX > kChunk) {
            if ((wm >> kChunk) != 0u) {          // wave-uniform: some lane has a slot >= kChunk
                issue_chunk(kChunk);
                compute_chunk(kChunk);
            }
        }
        uint32_t el[V];
#pragma unroll
        for (int k = 0; k < V; k++) {
            const bool maj_old = 2 * __popc(gm[k] & om[k]) > __popc(om[k]);
            const bool maj_new = 2 * __popc(gm[k] & nm[k]) > __popc(nm[k]);
            el[k] = maj_old && maj_new && (mx[k] == ct[k]) && mem[k] != 0;
            if (k < valid) el_field |= el[k] << k;
            if (k < valid) ch_field |= (uint32_t)(el[k] != 0 || sm[k] != 0) << k;
        }
        if (in_range) {
            st_u16<V, NTS, FULL>(at(a.grant_mask, go), valid, gm);
            st_u16<V, NTS, FULL>(at(a.stepdown_mask, go), valid, sm);
            st_u8<V, NTS, FULL>(at(a.elected, go), valid, el);
            st_i32<V, NTS, FULL>(at(a.max_term, go), valid, mx);
        }
    }

    if constexpr (DO_COMMIT) {
        T co[V];
#pragma unroll
        for (int k = 0; k < V; k++) {
            const bool hit = by_start ? nq[k] >= ts[k] : gt[k] == lt[k];     // the looked-up term is consumed here, late
            co[k] = (adv[k] && inb[k] && hit) ? nq[k] : cm[k];
            st[k] |= (adv[k] && !inb[k]) ? kStLogOob : 0;
            if (k < valid) ch_field |= (uint32_t)(co[k] != cm[k]) << k;
        }
        if (in_range) {
            st_i32<V, NTS, FULL>(at(a.commit_out, gco), valid, co);
            if (a.commit_lo16) nib_field = store_commit_summary<V, NTS, FULL>(a, g0, valid, co, cm, esc_field);
        }
    }
    if (in_range && a.status_out) st_u8<V, NTS, FULL>(at(a.status_out, go), valid, st);
#pragma unroll
    for (int k = 0; k < V; k++)
        if (k < valid) ch_field |= (uint32_t)(st[k] != 0) << k;
    return round_fields{el_field, nib_field, esc_field, ch_field};
}

template <int MMAX, int V, int WHAT, bool NT, typename T>
__global__ __launch_bounds__(kBlock) void rq_round_joint(rq_round_args_t<T> a) {
    signal_started(a);
    const uint64_t g0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    round_fields f;
    // Every lane runs a body: the wave-wide ballots inside need all of them.
    if (g0 + V <= a.G)
        f = round_joint_body<MMAX, V, WHAT, NT, true>(a, (uint32_t)g0, V, true);
    else {
        const bool in_range = g0 < a.G;
        f = round_joint_body<MMAX, V, WHAT, NT, false>(
            a, in_range ? (uint32_t)g0 : 0u, in_range ? (int)(a.G - g0) : 0, in_range);
    }
    if constexpr ((WHAT & RQ_WHAT_VOTE) != 0) {
        if (a.elected_bits) pack_elected<V>(f.elected, g0, a.G, a.elected_bits);
    }
    if (a.changed_bits) pack_elected<V>(f.changed, g0, a.G, a.changed_bits);
    if constexpr ((WHAT & RQ_WHAT_COMMIT) != 0) {
        if (a.commit_lo16 && a.summary_bits == 4)
            pack_nibbles<V>(f.nibbles, g0, a.G, reinterpret_cast<uint64_t *>(a.commit_lo16));
        if (a.commit_lo16 && a.summary_escapes) count_escapes(f.escapes, a.summary_escapes);
    }
}

// ---------------------------------------------------------------------------
// dispatch.  One build per kernel family, element type and decision set (every
// streamed access non-temporal; RQ_BUILD_PLAIN adds the default-policy builds
// behind RQ_VAR_PLAIN); the log-term
// gather is non-temporal only while neighbouring groups
// share arena lines (short logs: the wave's gathers sweep the arena like one
// more stream, measured 4 % faster at 8-34 bytes of arena per group and 2 %
// faster at 68 bytes, the int64 arena of the same short logs), and plain
// when every gather lands in a line of its own (long logs: non-temporal was
// measured 8-13 % slower at 200 bytes - 3 KB per group).
// ---------------------------------------------------------------------------

template <typename K, typename T>
hipError_t launch_grid(K kernel, const rq_round_args_t<T> &a, uint32_t block, hipStream_t s) {
    const uint32_t grid = (uint32_t)(((uint64_t)a.G + block - 1) / block);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, a);
    return hipGetLastError();
}
template <typename K, typename T>
hipError_t launch_grid_lead(K kernel, const rq_round_args_t<T> &a, uint32_t block, hipStream_t s) {
    const uint32_t grid = (uint32_t)(((uint64_t)a.G + block - 1) / block);
    const uint64_t stride = a.tile_bytes ? (a.tile_bytes | kStrideTiled) : (uint64_t)a.ld;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, a.G, block, stride,
                       a.match, a.commit, a.log_off, a.voter_term, a.voted_for, a);
    return hipGetLastError();
}

// plain: default cache policy for every access (only in builds with RQ_BUILD_PLAIN; a build
// without them runs the streaming kernels instead -- the policy never changes a result).
template <int N, int WHAT, typename T>
hipError_t launch_fixed_policy(const rq_round_args_t<T> &a, bool plain, uint32_t block, hipStream_t s) {
#ifdef RQ_BUILD_PLAIN
    if (plain) return launch_grid_lead(rq_round_fixed<N, 1, WHAT, false, T>, a, block, s);
#endif
    (void)plain;
    return launch_grid_lead(rq_round_fixed<N, 1, WHAT, true, T>, a, block, s);
}

template <int N, typename T>
hipError_t launch_fixed_what(const rq_round_args_t<T> &a, int what, bool plain, uint32_t block, hipStream_t s) {
    switch (what) {
    case RQ_WHAT_COMMIT: return launch_fixed_policy<N, RQ_WHAT_COMMIT>(a, plain, block, s);
    case RQ_WHAT_VOTE: return launch_fixed_policy<N, RQ_WHAT_VOTE>(a, plain, block, s);
    default: return launch_fixed_policy<N, RQ_WHAT_COMMIT | RQ_WHAT_VOTE>(a, plain, block, s);
    }
}

template <int MMAX, int WHAT, typename T>
hipError_t launch_joint_policy(const rq_round_args_t<T> &a, bool plain, uint32_t block, hipStream_t s) {
#ifdef RQ_BUILD_PLAIN
    if (plain) return launch_grid(rq_round_joint<MMAX, 1, WHAT, false, T>, a, block, s);
#endif
    (void)plain;
    return launch_grid(rq_round_joint<MMAX, 1, WHAT, true, T>, a, block, s);
}

template <int MMAX, typename T>
hipError_t launch_joint_what(const rq_round_args_t<T> &a, int what, bool plain, uint32_t block, hipStream_t s) {
    switch (what) {
    case RQ_WHAT_COMMIT: return launch_joint_policy<MMAX, RQ_WHAT_COMMIT>(a, plain, block, s);
    case RQ_WHAT_VOTE: return launch_joint_policy<MMAX, RQ_WHAT_VOTE>(a, plain, block, s);
    default: return launch_joint_policy<MMAX, RQ_WHAT_COMMIT | RQ_WHAT_VOTE>(a, plain, block, s);
    }
}

// ---------------------------------------------------------------------------
// Several sub-rounds in one launch.  Sub-round s decides the groups
// [first_group[s], end_group[s]) of an image with the fixed-size body of kind[s]
// (1..9: the same code as rq_round_fixed<n>) or, kind 0, with the member-mask
// body; blocks are dealt to the sub-rounds in order, ceil(len / blockDim) each,
// and every first_group is a multiple of 64, so a wave never straddles two.
// Two uses: ONE image whose groups are ordered by cluster size (the segment
// table of a round; every sub-round reads the launch's own arguments), and
// SEVERAL images -- each with its own arrays, tiling and size -- decided by one
// grid (`table` in device memory, sub-round s reads table[image[s]]), so that
// short rounds share one ramp-up and one drain.  Finding the sub-round and
// fetching its arguments is scalar work.
// ---------------------------------------------------------------------------
struct multi_subs {
    uint32_t count;
    uint32_t first_block[RQ_MAX_SEGS + 1];
    uint32_t first_group[RQ_MAX_SEGS];
    uint32_t end_group[RQ_MAX_SEGS];
    uint8_t kind[RQ_MAX_SEGS];
    uint8_t image[RQ_MAX_SEGS];
};

template <int WHAT, bool NT, typename T>
__global__ __launch_bounds__(kBlock) void rq_round_multi(rq_round_args_t<T> a0,
                                                         const rq_round_args_t<T> *__restrict__ table,
                                                         multi_subs sb) {
    signal_started(a0);
    uint32_t fb = 0, fg = sb.first_group[0], end = sb.end_group[0], kind = sb.kind[0], img = sb.image[0];
#pragma unroll
    for (int i = 1; i < RQ_MAX_SEGS; i++) {
        if (i < (int)sb.count && blockIdx.x >= sb.first_block[i]) {
            fb = sb.first_block[i]; fg = sb.first_group[i]; end = sb.end_group[i]; kind = sb.kind[i]; img = sb.image[i];
        }
    }
    rq_round_args_t<T> a = a0;
    if (table) a = table[img];
    const uint32_t g = fg + (blockIdx.x - fb) * blockDim.x + threadIdx.x;
    const bool in = g < end;
    round_fields f{0, 0, 0, 0};
    switch (kind) {
#define RQ_SUB_CASE(N) case N: if (in) f = round_fixed_body<N, 1, WHAT, NT, true>(a, g, 1); break;
    RQ_SUB_CASE(1) RQ_SUB_CASE(2) RQ_SUB_CASE(3) RQ_SUB_CASE(4) RQ_SUB_CASE(5)
    RQ_SUB_CASE(6) RQ_SUB_CASE(7) RQ_SUB_CASE(8) RQ_SUB_CASE(9)
#undef RQ_SUB_CASE
    default:   // every lane runs the member-mask body: its wave-wide ballots need all of them
        f = round_joint_body<16, 1, WHAT, NT, true, T, true>(a, in ? g : 0u, in ? 1 : 0, in);
    }
    if constexpr ((WHAT & RQ_WHAT_VOTE) != 0) {
        if (a.elected_bits) pack_elected<1>(f.elected, g, end, a.elected_bits);
    }
    if (a.changed_bits) pack_elected<1>(f.changed, g, end, a.changed_bits);
    if constexpr ((WHAT & RQ_WHAT_COMMIT) != 0) {
        if (a.commit_lo16 && a.summary_bits == 4)
            pack_nibbles<1>(f.nibbles, g, end, reinterpret_cast<uint64_t *>(a.commit_lo16));
        if (a.commit_lo16 && a.summary_escapes) count_escapes(f.escapes, a.summary_escapes);
    }
}

// Deals blocks to the sub-rounds and enqueues the grid.  The sub-rounds whose waves live
// longest get the first blocks -- member masks (one more dependent load), then the fixed
// sizes from 9 down -- so that the grid drains through its shortest waves.
template <int WHAT, typename T>
hipError_t launch_multi(const rq_round_args_t<T> &a0, const rq_round_args_t<T> *table, const multi_subs &in,
                        uint32_t block, hipStream_t s) {
    uint32_t order[RQ_MAX_SEGS];
    for (uint32_t i = 0; i < in.count; i++) order[i] = i;
    auto weight = [&](uint32_t i) { return in.kind[i] ? (uint32_t)in.kind[i] : 100u; };
    std::stable_sort(order, order + in.count, [&](uint32_t x, uint32_t y) { return weight(x) > weight(y); });
    multi_subs sb;
    std::memset(&sb, 0, sizeof sb);
    sb.count = in.count;
    uint64_t nb = 0;
    for (uint32_t i = 0; i < in.count; i++) {
        const uint32_t j = order[i];
        sb.first_group[i] = in.first_group[j]; sb.end_group[i] = in.end_group[j];
        sb.kind[i] = in.kind[j]; sb.image[i] = in.image[j];
        sb.first_block[i] = (uint32_t)nb;
        nb += ((uint64_t)(sb.end_group[i] - sb.first_group[i]) + block - 1) / block;
    }
    for (uint32_t i = sb.count; i <= RQ_MAX_SEGS; i++) sb.first_block[i] = (uint32_t)nb;
    if (nb == 0) return hipSuccess;
    if (nb > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((rq_round_multi<WHAT, true, T>), dim3((uint32_t)nb), dim3(block), 0, s, a0, table, sb);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_multi_what(const rq_round_args_t<T> &a0, const rq_round_args_t<T> *table, const multi_subs &sb,
                             int what, uint32_t block, hipStream_t s) {
    switch (what) {
    case RQ_WHAT_COMMIT: return launch_multi<RQ_WHAT_COMMIT>(a0, table, sb, block, s);
    case RQ_WHAT_VOTE: return launch_multi<RQ_WHAT_VOTE>(a0, table, sb, block, s);
    default: return launch_multi<RQ_WHAT_COMMIT | RQ_WHAT_VOTE>(a0, table, sb, block, s);
    }
}

// The table is usable when it covers [0, G) in order with 64-aligned inner
// boundaries and every fixed size fits the image's planes.
inline bool segs_ok(const rq_round_segs &t, uint32_t G, uint32_t m) {
    if (t.count < 1 || t.count > RQ_MAX_SEGS || t.first_group[0] != 0 || t.first_group[t.count] != G) return false;
    for (uint32_t i = 0; i < t.count; i++) {
        if (t.first_group[i + 1] < t.first_group[i]) return false;
        if (i + 1 < t.count && (t.first_group[i + 1] & 63u)) return false;
        if (t.n[i] > 9 || t.n[i] > m) return false;
    }
    return true;
}

inline bool aligned_to(const void *p, size_t a) {
    return p == nullptr || (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0;
}

template <typename T>
hipError_t launch_round_impl(const rq_round_args_t<T> &args, uint32_t n, bool joint, int what,
                             int variant, hipStream_t stream, const rq_round_segs *segs = nullptr) {
    if (args.G == 0) return hipSuccess;
    if (args.G > 0x7FFFFFFFu || (args.ld < args.G && !args.tile_bytes)) return hipErrorInvalidValue;
    if (!(what & (RQ_WHAT_COMMIT | RQ_WHAT_VOTE))) return hipErrorInvalidValue;
    if (args.commit_lo16 && args.summary_bits != 16 && args.summary_bits != 8 && args.summary_bits != 4)
        return hipErrorInvalidValue;
    if (args.commit_lo16 && args.summary_bits == 4 && !aligned_to(args.commit_lo16, 8)) return hipErrorInvalidValue;
    const uint32_t tb = args.log_term_bits;
    if ((what & RQ_WHAT_COMMIT) && tb != 8 && tb != 16 && tb != 32 && tb != 8 * sizeof(T)) return hipErrorInvalidValue;
    if (!aligned_to(args.start_signal, 8)) return hipErrorInvalidValue;
    if ((what & RQ_WHAT_COMMIT) && !aligned_to(args.log_term, tb / 8)) return hipErrorInvalidValue;   // load_term reads aligned entries
    if ((what & RQ_WHAT_COMMIT) && args.log_term_start && !args.log_len) return hipErrorInvalidValue;
    if (variant & ~(RQ_VAR_PLAIN | 0xF00)) return hipErrorInvalidValue;
    const int bsel = (variant >> 8) & 0xF;
    if (bsel > 4) return hipErrorInvalidValue;
    const bool fixed = !joint && n >= 1 && n <= 9;
    const uint32_t block = bsel ? 64u * (uint32_t)bsel
                                : (uint32_t)(!fixed ? kDefaultBlockJoint : what == RQ_WHAT_COMMIT ? kDefaultBlockCommit : kDefaultBlockFixed);
    bool plain = false;
#ifdef RQ_BUILD_PLAIN
    plain = (variant & RQ_VAR_PLAIN) != 0;
#endif
    rq_round_args_t<T> a = args;
    a.nt_gather = rq_nt_gather(args);
    if (a.log_term_start) a.log_off = nullptr;      // the kernels take "no log_off" for "no arena lookup"
    if (segs) {
        // Groups ordered by size in an image of n planes; member masks are needed only by
        // segments of kind 0 (without masks those take all n slots as their one configuration).
        rq_round_args_t<T> b = a;
        b.m = n;
        if (!joint) { b.old_mask = nullptr; b.new_mask = nullptr; }
        if (n < 1 || n > 16 || !segs_ok(*segs, b.G, n)) return hipErrorInvalidValue;
        if ((b.old_mask == nullptr) != (b.new_mask == nullptr)) return hipErrorInvalidValue;
        // The table never changes a result, only which code decides a group: the plain
        // cache policy has no segmented build and decides every group through its masks.
        if (!plain) {
            multi_subs sb;
            std::memset(&sb, 0, sizeof sb);
            sb.count = segs->count;
            for (uint32_t i = 0; i < segs->count; i++) {
                sb.first_group[i] = segs->first_group[i];
                sb.end_group[i] = segs->first_group[i + 1];
                sb.kind[i] = segs->n[i];
            }
            return launch_multi_what<T>(b, nullptr, sb, what, bsel ? block : (uint32_t)kDefaultBlockFixed, stream);
        }
        if (!joint) return hipErrorInvalidValue;
    }

    if (fixed && !segs) {
        switch (n) {
        case 1: return launch_fixed_what<1>(a, what, plain, block, stream);
        case 2: return launch_fixed_what<2>(a, what, plain, block, stream);
        case 3: return launch_fixed_what<3>(a, what, plain, block, stream);
        case 4: return launch_fixed_what<4>(a, what, plain, block, stream);
        case 5: return launch_fixed_what<5>(a, what, plain, block, stream);
        case 6: return launch_fixed_what<6>(a, what, plain, block, stream);
        case 7: return launch_fixed_what<7>(a, what, plain, block, stream);
        case 8: return launch_fixed_what<8>(a, what, plain, block, stream);
        default: return launch_fixed_what<9>(a, what, plain, block, stream);
        }
    }
    // per-group membership, or a fixed size beyond the register networks
    rq_round_args_t<T> b = a;
    if (!joint) { b.m = n; b.old_mask = nullptr; b.new_mask = nullptr; }
    if (b.m < 1 || b.m > 16) return hipErrorInvalidValue;
    if ((b.old_mask == nullptr) != (b.new_mask == nullptr)) return hipErrorInvalidValue;
    if (b.m <= 8) return launch_joint_what<8>(b, what, plain, block, stream);
    return launch_joint_what<16>(b, what, plain, block, stream);
}

// Several images in one grid.  args[i] / n[i] / joint[i]: image i as rq_launch_round takes it;
// table: the same `count` argument structs in device memory (the caller keeps them there).
template <typename T>
hipError_t launch_rounds_impl(const rq_round_args_t<T> *args, const uint32_t *n, const bool *joint, uint32_t count,
                              const rq_round_args_t<T> *table, int what, int variant, hipStream_t stream) {
    if (count < 1 || count > RQ_MAX_SEGS || !table) return hipErrorInvalidValue;
    if (!(what & (RQ_WHAT_COMMIT | RQ_WHAT_VOTE)) || (variant & ~0xF00)) return hipErrorInvalidValue;
    const int bsel = (variant >> 8) & 0xF;
    if (bsel > 4) return hipErrorInvalidValue;
    multi_subs sb;
    std::memset(&sb, 0, sizeof sb);
    sb.count = count;
    for (uint32_t i = 0; i < count; i++) {
        const rq_round_args_t<T> &a = args[i];
        if (a.G > 0x7FFFFFFFu || (a.ld < a.G && !a.tile_bytes)) return hipErrorInvalidValue;
        if ((what & RQ_WHAT_COMMIT) && !aligned_to(a.log_term, a.log_term_bits / 8)) return hipErrorInvalidValue;
        const bool fixed = !joint[i] && n[i] >= 1 && n[i] <= 9;
        if (!fixed && (a.m < 1 || a.m > 16 || (a.old_mask == nullptr) != (a.new_mask == nullptr))) return hipErrorInvalidValue;
        sb.first_group[i] = 0; sb.end_group[i] = a.G;
        sb.kind[i] = fixed ? (uint8_t)n[i] : 0;
        sb.image[i] = (uint8_t)i;
    }
    return launch_multi_what<T>(args[0], table, sb, what, bsel ? 64u * (uint32_t)bsel : (uint32_t)kDefaultBlockFixed, stream);
}

}  // namespace
