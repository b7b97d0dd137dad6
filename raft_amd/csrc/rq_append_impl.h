// rq_append_impl.h -- the AppendEntries consistency batch and the follower commit
// clamp, written once over the element type T (int32_t or int64_t) like the
// round kernels.  Internal to librq.so.
#pragma once

#include "rq_round_impl.h"

namespace {

// ---------------------------------------------------------------------------
// AppendEntries consistency batch: one thread per group; all plane loads of
// the group's slots are issued before the dependent log gathers.
//   stepdown : term > cur, cur' = max(cur, term)                 (raft.go:142, :219-221)
//   reach    : !(term < cur')                                    (:145-147)
//   success  : reach && log[prevLogIndex].TermNum == prevLogTerm (:151-153)
//   prevLogIndex outside the follower's log is an index panic in Go: status
//   bit, no success.  Follower logs may be windows (flog_base) of narrow entries
//   (flog_term_bits, flog_term_base), like the leader arena of the round.
// Leader bookkeeping for the reply (Raft Fig. 2; the reference only declares
// nextIndex/matchIndex, raft.go:60-68): success -> match = prev + n_new,
// next = match + 1; rejected at the log check -> next = max(next - 1, 1);
// rejected for a stale term -> unchanged.
// Follower commit (optional, raft.go:172-176): a slot whose RPC succeeds and whose
// commitIndex is below leaderCommit takes min(leaderCommit, index of the last new
// entry); with no new entries (n_new == 0) the reference's lastEntry(newEntries)
// panics (:175, :235): status bit, commitIndex unchanged.
// ---------------------------------------------------------------------------
// NMAX = compile-time slot bound (the loop over slots is fully unrolled so that
// every plane load of a group is in flight before the first gather).
// Index of element (p, g) of an array of P: planes field[p * ld + g], or (TILED) the round
// kernels' 64-group tiles, whose stride is a multiple of 8 bytes and so of every element size.
template <bool TILED, typename P>
__device__ __forceinline__ size_t append_index(uint64_t tile_bytes, size_t ld, uint32_t g, uint32_t p) {
    if constexpr (TILED) return (size_t)(g >> 6) * (size_t)(tile_bytes / sizeof(P)) + (size_t)(p * 64u + (g & 63u));
    else return (size_t)p * ld + g;
}

// The leader's commit rule in the same launch (rq_append_desc.commit_round): what a commit
// round over the batch's matchIndex array would read back from memory right after this batch
// is still in registers here.  r describes that round (its `match` IS match_index_io; its
// slots are the batch's follower slots, optionally followed by the leader's own); the
// decision is the round kernels' -- quorum index of the r.m values, N > commitIndex, log[N]
// inside the log and of the leader's term (or N >= log_term_start) -- with the sort done by
// the member-mask round's network over r.m live values.
template <typename T>
struct commit_inputs {
    T cm, lt, lb, tb, ts;
    uint32_t beg, end, ln;
};
template <typename T>
__device__ __forceinline__ commit_inputs<T> load_commit_inputs(const rq_round_args_t<T> &r, uint32_t g) {
    commit_inputs<T> in;
    const group_addr ga = addr_of(r, g);
    in.lt = 0; in.lb = 0; in.tb = 0; in.ts = 0; in.beg = 0; in.end = 0; in.ln = 0;
    in.cm = __builtin_nontemporal_load(gptr(at(r.commit, ga)));
    if (r.log_base) in.lb = __builtin_nontemporal_load(gptr(at(r.log_base, ga)));
    if (r.log_len) in.ln = __builtin_nontemporal_load(gptr(at(r.log_len, ga)));
    if (r.log_term_start) {
        in.ts = __builtin_nontemporal_load(gptr(at(r.log_term_start, ga)));
    } else {
        in.lt = __builtin_nontemporal_load(gptr(at(r.leader_term, ga)));
        if (r.log_term_base) in.tb = __builtin_nontemporal_load(gptr(at(r.log_term_base, ga)));
        in.beg = __builtin_nontemporal_load(gptr(at(r.log_off, ga)));
        in.end = *gptr(at(r.log_off, addr_of(r, g + 1)));
    }
    return in;
}
// Quorum index of the first n of the L values in v (n = L - 1 or L): the member-mask round's
// descending network over 8 or 16 inputs, the inputs beyond L being compile-time fillers.
template <int L, typename T>
__device__ __forceinline__ T live_quorum(const T (&v)[L], uint32_t n) {
    constexpr int W = L > 8 ? 16 : 8;
    T s[W];
#pragma unroll
    for (int p = 0; p < W; p++) s[p] = kMinOf<T>;
#pragma unroll
    for (int p = 0; p < L; p++) s[p] = (uint32_t)p < n ? v[p] : kMinOf<T>;
    if constexpr (W == 8) sort_desc8(s); else sort_desc16(s);
    const int idx = (int)(n >> 1);
    T q = s[0];
#pragma unroll
    for (int i = 1; i <= W / 2; i++) q = (idx == i) ? s[i] : q;
    return q;
}
template <typename T>
__device__ __forceinline__ T commit_decision(const rq_round_args_t<T> &r, const commit_inputs<T> &in, T nq,
                                             uint32_t &st) {
    typedef typename std::make_unsigned<T>::type U;
    const bool by_start = r.log_term_start != nullptr;
    const bool csr = by_start || (r.log_len ? span_ok(in.beg, in.ln, r.log_terms) : slice_ok(in.beg, in.end, r.log_terms));
    const uint32_t len = r.log_len ? in.ln : in.end - in.beg;
    const U wi = window_slot(nq, in.lb);
    const bool adv = nq > in.cm, inb = csr && wi < (U)len;
    st |= csr ? 0u : kStBadCsr;
    st |= (adv && !inb) ? kStLogOob : 0u;
    bool hit = by_start ? nq >= in.ts : false;
    if (!by_start && adv && inb)
        hit = (T)((U)load_term<T>(r.log_term, r.log_term_bits, (size_t)in.beg + (size_t)wi, true) + (U)in.tb) == in.lt;
    return (adv && inb && hit) ? nq : in.cm;
}

// CLAMP: the follower commit rule is part of the launch (a build of its own: its two extra
// planes per slot would otherwise cost every launch 12 registers per slot).  COMMIT: the
// leader's commit rule as well (r is only read then).
template <int NMAX, bool TILED, bool CLAMP, bool COMMIT, typename T>
__global__ __launch_bounds__(kBlock) void rq_append_check_kernel(rq_append_args_t<T> a, rq_round_args_t<T> r) {
    typedef typename std::make_unsigned<T>::type U;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    const size_t ld = a.ld;
    const uint32_t n = a.n;
    const size_t gT = append_index<TILED, T>(a.tile_bytes, ld, g, 0);
    const T lt = __builtin_nontemporal_load(gptr(a.leader_term + gT));
    const T tbase = a.flog_term_base ? __builtin_nontemporal_load(gptr(a.flog_term_base + gT)) : (T)0;
    // without a mask nothing below waits for a load before the slots' own loads go out
    uint32_t act = n >= 16 ? 0xFFFFu : ((1u << n) - 1u);
    // wact: the slots SOME lane of the wave sends an RPC to.  Those are loaded and stored by every
    // lane (a lane without an RPC for the slot writes back what it read): rows with holes -- every
    // plane has them when each group leaves out its own leader's slot -- would be partial writes,
    // which the memory turns into a read and a write each.  Slots no lane uses are skipped whole.
    uint32_t wact = act;
    if (a.active_mask) {
        act &= (uint32_t)gptr(a.active_mask)[append_index<TILED, uint16_t>(a.tile_bytes, ld, g, 0)];
        wact = 0;
#pragma unroll
        for (int p = 0; p < NMAX; p++)
            if (__ballot((act >> p) & 1u) != 0ull) wact |= 1u << p;
        wact = __builtin_amdgcn_readfirstlane(wact);
    }
    const bool upd = a.match_index_io != nullptr || a.next_index_io != nullptr;
    constexpr bool clamp = CLAMP;
    const T lc = clamp ? __builtin_nontemporal_load(gptr(a.leader_commit + gT)) : (T)0;
    T ft[NMAX], pi[NMAX], pt[NMAX], nn[NMAX], mi[NMAX], ni[NMAX], tg[NMAX], fb[NMAX], fc[NMAX], li[NMAX];
    uint32_t o0[NMAX], o1[NMAX];
#pragma unroll
    for (int p = 0; p < NMAX; p++) {
        ft[p] = pi[p] = pt[p] = nn[p] = mi[p] = fb[p] = fc[p] = li[p] = 0; ni[p] = 1; o0[p] = o1[p] = 0;
        if ((wact >> p) & 1u) {
            const size_t s = append_index<TILED, T>(a.tile_bytes, ld, g, p);
            const size_t so = append_index<TILED, uint32_t>(a.tile_bytes, ld, g, p);
            ft[p] = __builtin_nontemporal_load(gptr(a.follower_term + s));
            pi[p] = __builtin_nontemporal_load(gptr(a.prev_log_index + s));
            pt[p] = __builtin_nontemporal_load(gptr(a.prev_log_term + s));
            o0[p] = __builtin_nontemporal_load(gptr(a.flog_off + so));
            // the end of the log = the next entry: plain load, the neighbour's line is already on its way
            o1[p] = TILED ? gptr(a.flog_off)[append_index<TILED, uint32_t>(a.tile_bytes, ld, g + 1, p)] : gptr(a.flog_off)[so + 1];
            if (a.flog_base) fb[p] = __builtin_nontemporal_load(gptr(a.flog_base + s));
            if (a.n_new) nn[p] = __builtin_nontemporal_load(gptr(a.n_new + s));
            if (a.match_index_io) mi[p] = __builtin_nontemporal_load(gptr(a.match_index_io + s));
            if (a.next_index_io) ni[p] = __builtin_nontemporal_load(gptr(a.next_index_io + s));
            if constexpr (clamp) {
                fc[p] = __builtin_nontemporal_load(gptr(a.follower_commit_io + s));
                li[p] = __builtin_nontemporal_load(gptr(a.last_new_index + s));
            }
        } else if constexpr (COMMIT) {
            // a slot without an RPC still counts in the commit round: its stored matchIndex
            if ((uint32_t)p < r.m) mi[p] = __builtin_nontemporal_load(gptr(a.match_index_io + append_index<TILED, T>(a.tile_bytes, ld, g, p)));
        }
    }
    // COMMIT: matchIndex of every slot the commit round counts -- what this batch produces stays
    // in registers; slots without an RPC were loaded above with the batch's own planes, the
    // leader's own slot beyond the batch's planes is read at the end
    constexpr int L = NMAX < 16 ? NMAX + 1 : 16;
    T mv[L];
    if constexpr (COMMIT) {
#pragma unroll
        for (int p = 0; p < L; p++) mv[p] = p < NMAX ? mi[p] : (T)0;
    }
    // The followers' log[prevLogIndex].TermNum (raft.go:151), one lookup per slot.  Every slot of
    // every lane issues one -- the same load instruction whatever the entry width, entry 0 (or,
    // without an arena, word 0 of the leader terms) for a slot with nothing to ask, result unused --
    // and the words are cut where the terms are compared: branches around the loads, per width or
    // per lane, made each lookup wait for the one before it, four memory latencies in a row.
    uint32_t reach = 0, inb = 0, bad = 0;
    const bool arena = a.flog_terms != 0;
    const void *tsrc = arena ? a.flog_term : (const void *)a.leader_term;
    const uint32_t tbits = arena ? a.flog_term_bits : 32u;
    term_word<T> tw[NMAX];
#pragma unroll
    for (int p = 0; p < NMAX; p++) {
        const bool on = (act >> p) & 1u;
        const bool rch = on && lt >= ft[p];
        const bool csr = slice_ok(o0[p], o1[p], a.flog_terms);
        const U wi = window_slot(pi[p], fb[p]);
        const bool in = csr && wi < (U)(o1[p] - o0[p]);
        reach |= (uint32_t)rch << p;
        inb |= (uint32_t)in << p;
        bad |= (uint32_t)(on && !csr) << p;
        tw[p] = load_term_word<T>(tsrc, tbits, (rch && in) ? (size_t)o0[p] + (size_t)wi : (size_t)0, arena);
    }
#pragma unroll
    for (int p = 0; p < NMAX; p++) tg[p] = (T)((U)term_of<T>(tw[p], tbits) + (U)tbase);
    // pure outputs live in their own tile block; follower terms updated in place stay with the inputs
    const uint64_t out_tile = a.out_tile_bytes;
    const uint64_t fo_tile = (const void *)a.follower_term_out == (const void *)a.follower_term ? a.tile_bytes : out_tile;
    uint32_t ok = 0, sd = 0, st = bad ? kStBadCsr : 0;
#pragma unroll
    for (int p = 0; p < NMAX; p++) {
        if (!((wact >> p) & 1u)) continue;
        const size_t s = append_index<TILED, T>(a.tile_bytes, ld, g, p);
        const bool on = (act >> p) & 1u;
        const bool rch = (reach >> p) & 1u, in = (inb >> p) & 1u;          // rch implies on
        const bool success = rch && in && tg[p] == pt[p];
        sd |= (uint32_t)(on && lt > ft[p]) << p;
        ok |= (uint32_t)success << p;
        st |= (rch && !in) ? kStLogOob : 0;
        __builtin_nontemporal_store(on ? tmax(ft[p], lt) : ft[p], gptr(a.follower_term_out + append_index<TILED, T>(fo_tile, ld, g, p)));
        if (upd) {
            T m2 = mi[p], n2 = ni[p];
            if (success) { m2 = (T)((U)pi[p] + (U)nn[p]); n2 = (T)((U)m2 + (U)1); }
            else if (rch) { n2 = n2 > 1 ? n2 - 1 : 1; }
            if (a.match_index_io) __builtin_nontemporal_store(m2, gptr(a.match_index_io + s));
            if (a.next_index_io) __builtin_nontemporal_store(n2, gptr(a.next_index_io + s));
            if constexpr (COMMIT) { if (a.match_index_io) mv[p] = m2; }
        }
        if constexpr (clamp) {
            T c2 = fc[p];
            if (success && lc > fc[p]) {                                  // raft.go:174
                if (a.n_new && nn[p] == 0) st |= kStEmptyLog;             // lastEntry of no entries, :175 / :235
                else c2 = tmin(lc, li[p]);
            }
            __builtin_nontemporal_store(c2, gptr(a.follower_commit_io + s));
        }
    }
    const size_t g16 = append_index<TILED, uint16_t>(out_tile, ld, g, 0);
    __builtin_nontemporal_store((uint16_t)ok, gptr(a.success_mask + g16));
    __builtin_nontemporal_store((uint16_t)sd, gptr(a.stepdown_mask + g16));
    if (a.status_out) gptr(a.status_out)[append_index<TILED, uint8_t>(out_tile, ld, g, 0)] = (uint8_t)st;
    if constexpr (COMMIT) {
        // the round's own inputs are read here, after the batch's stores: issued with the batch's
        // planes they cost 11 registers and two of the eight resident waves (measured 3 % slower)
        const group_addr rga = addr_of(r, g);
        group_addr rgo = rga;                     // the round's outputs: their own tile block unless in place
        if (r.tile_bytes) rgo.tile_off = (size_t)(g >> 6) * r.out_tile_bytes;
        const group_addr gco = (const void *)r.commit_out == (const void *)r.commit ? rga : rgo;
        if constexpr (L > NMAX) {
            if (r.m > (uint32_t)NMAX) mv[NMAX] = __builtin_nontemporal_load(gptr(at(r.match, rga, NMAX)));
        }
        const commit_inputs<T> cin = load_commit_inputs(r, g);
        uint32_t rst = 0;
        const T co = commit_decision(r, cin, live_quorum<L>(mv, r.m), rst);
        __builtin_nontemporal_store(co, gptr(at(r.commit_out, gco)));
        if (r.status_out) *gptr(at(r.status_out, rgo)) = (uint8_t)rst;
    }
}

// raft.go:172-176 elementwise
template <typename T>
__global__ __launch_bounds__(kBlock) void rq_follower_clamp_kernel(
        uint64_t count, const T *lc, const T *ci, const T *ln, T *out, int vec_ok) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i0 >= count) return;
    T a[4], b[4], c[4], r[4];
    const bool full = vec_ok && i0 + 4 <= count;
    const int valid = full ? 4 : (int)min((uint64_t)4, count - i0);
    if (full) {
        ld_i32<4, false, true>(lc + i0, 4, a);
        ld_i32<4, false, true>(ci + i0, 4, b);
        ld_i32<4, false, true>(ln + i0, 4, c);
    } else {
        ld_i32<4, false, false>(lc + i0, valid, a);
        ld_i32<4, false, false>(ci + i0, valid, b);
        ld_i32<4, false, false>(ln + i0, valid, c);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = a[k] > b[k] ? tmin(a[k], c[k]) : b[k];
    if (full) st_i32<4, false, true>(out + i0, 4, r);
    else st_i32<4, false, false>(out + i0, valid, r);
}


template <typename T>
hipError_t launch_append_check_impl(const rq_append_args_t<T> &a, const rq_round_args_t<T> *commit_round,
                                    hipStream_t stream) {
    if (a.G == 0) return hipSuccess;
    if (a.G > 0x7FFFFFFFu || (a.ld < a.G && !a.tile_bytes) || a.n < 1 || a.n > 16) return hipErrorInvalidValue;
    if ((a.tile_bytes & 7u) || (a.out_tile_bytes & 7u) || ((a.tile_bytes != 0) != (a.out_tile_bytes != 0))) return hipErrorInvalidValue;
    const uint32_t tb = a.flog_term_bits;
    if (tb != 8 && tb != 16 && tb != 32 && tb != 8 * sizeof(T)) return hipErrorInvalidValue;
    if (reinterpret_cast<uintptr_t>(a.flog_term) & (tb / 8 - 1)) return hipErrorInvalidValue;   // load_term reads aligned entries
    const uint32_t block = 128;
    const dim3 grid((a.G + block - 1) / block), blk(block);
    const int clamp_ptrs = (a.leader_commit != nullptr) + (a.last_new_index != nullptr) + (a.follower_commit_io != nullptr);
    if (clamp_ptrs != 0 && clamp_ptrs != 3) return hipErrorInvalidValue;
    const bool clamp = clamp_ptrs == 3, tiled = a.tile_bytes != 0;
    rq_round_args_t<T> r = rq_round_args_t<T>();      // zeroed
    if (commit_round) {
        r = *commit_round;
        // the round reads the array this batch updates, in the same arrangement, and nothing else of it is fused
        if (r.G != a.G || r.m < a.n || r.m > a.n + 1 || r.m > 16 || r.tile_bytes != a.tile_bytes ||
            (!r.tile_bytes && r.ld != a.ld) || (const void *)r.match != (const void *)a.match_index_io ||
            !a.match_index_io || r.old_mask || r.commit_lo16 || r.changed_bits || r.elected_bits || r.start_signal ||
            !r.commit || !r.commit_out)
            return hipErrorInvalidValue;
        if (r.log_term_start ? !r.log_len : (!r.leader_term || !r.log_off || (!r.log_term && r.log_terms)))
            return hipErrorInvalidValue;
        const uint32_t rb = r.log_term_bits;
        if (rb != 8 && rb != 16 && rb != 32 && rb != 8 * sizeof(T)) return hipErrorInvalidValue;
        if (reinterpret_cast<uintptr_t>(r.log_term) & (rb / 8 - 1)) return hipErrorInvalidValue;
    }
    const bool commit = commit_round != nullptr;
#define RQ_APPEND_LAUNCH(N, TILED, CLAMP)                                                                        \
    do {                                                                                                         \
        if (commit) hipLaunchKernelGGL((rq_append_check_kernel<N, TILED, CLAMP, true, T>), grid, blk, 0, stream, a, r); \
        else hipLaunchKernelGGL((rq_append_check_kernel<N, TILED, CLAMP, false, T>), grid, blk, 0, stream, a, r);      \
    } while (0)
#define RQ_APPEND_CASE(N)                                             \
    case N:                                                           \
        if (tiled && clamp) RQ_APPEND_LAUNCH(N, true, true);          \
        else if (tiled) RQ_APPEND_LAUNCH(N, true, false);             \
        else if (clamp) RQ_APPEND_LAUNCH(N, false, true);             \
        else RQ_APPEND_LAUNCH(N, false, false);                       \
        break;
    // Builds exist for a subset of slot counts; a batch runs the next larger one (slots beyond
    // a.n are masked out wave-wide, so they cost a skipped branch, not traffic).  The compact
    // layout has its own build for the follower counts of 3- to 10-peer clusters that matter
    // most (1-5, 7, 9); the wide layout, a fallback for values beyond int32, fewer.
    uint32_t build = 16;
    if constexpr (sizeof(T) == 4) {
        for (uint32_t c : {9u, 7u, 5u, 4u, 3u, 2u, 1u}) if (a.n <= c) build = c;
    } else {
        for (uint32_t c : {9u, 5u, 4u, 2u}) if (a.n <= c) build = c;
    }
    switch (build) {
    RQ_APPEND_CASE(2) RQ_APPEND_CASE(4) RQ_APPEND_CASE(5) RQ_APPEND_CASE(9) RQ_APPEND_CASE(16)
    default:
        if constexpr (sizeof(T) == 4) {
            switch (build) {
            RQ_APPEND_CASE(1) RQ_APPEND_CASE(3) RQ_APPEND_CASE(7)
            default: break;
            }
        }
        break;
    }
#undef RQ_APPEND_CASE
#undef RQ_APPEND_LAUNCH
    return hipGetLastError();
}

template <typename T>
hipError_t launch_follower_clamp_impl(uint64_t count, const T *leader_commit, const T *commit,
                                      const T *last_new_index, T *commit_out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    constexpr size_t kVec = 4 * sizeof(T);
    const int vec_ok = aligned_to(leader_commit, kVec) && aligned_to(commit, kVec) &&
                       aligned_to(last_new_index, kVec) && aligned_to(commit_out, kVec);
    const uint64_t threads = (count + 3) / 4;
    const uint64_t grid = (threads + kBlock - 1) / kBlock;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((rq_follower_clamp_kernel<T>), dim3((uint32_t)grid), dim3(kBlock), 0, stream,
                       count, leader_commit, commit, last_new_index, commit_out, vec_ok);
    return hipGetLastError();
}

}  // namespace
