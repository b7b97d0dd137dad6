// rq_kernels.h -- launch interface between the C-ABI (rq_engine.hip) and the
// gfx950 kernels (rq_kernels.hip).  Internal to librq.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// One round's device pointers.  Planes are field[p * ld + g].  T is the
// element type of every term, index and id: int32_t (the compact layout) or
// int64_t (Go's int on amd64, raft.go:35-68).
template <typename T>
struct rq_round_args_t {
    uint32_t G;
    uint32_t ld;
    uint32_t m;  // joint kernels: number of slot planes
    uint64_t tile_bytes;      // 0 = planes field[p * ld + g]; else 64-group tiles of this many bytes (rq_round_impl.h)
    uint64_t out_tile_bytes;  // stride of the tiles that hold the output arrays
    // commit side (Node.matchIndex / commitIndex / currentTerm / log, raft.go:35-68)
    const T *match;
    const T *commit;
    const T *leader_term;
    const uint32_t *log_off;
    const void *log_term;     // arena of log_term_bits-wide entries
    const T *log_base;        // may be null (0): arena slot 0 of group g holds log[log_base[g]]
    const T *log_term_base;   // may be null (0): added to every arena entry of group g
    const uint32_t *log_len;  // may be null: group g's log is the first log_len[g] entries of its slice
    // may be null.  Else no arena lookup: log[N].TermNum == leader_term is taken to be
    // N >= log_term_start[g], the index of the first entry of the leader's current term
    // (valid for logs whose terms never decrease and never exceed the leader's term, as
    // Raft's own logs); log_len is then required, log_off / log_term / leader_term unused.
    const T *log_term_start;
    uint64_t log_terms;       // entries in the arena: no lookup goes at or beyond it
    uint32_t log_term_bits;   // 8, 16 (unsigned entries), 32 (signed) or 8 * sizeof(T)
    uint32_t nt_gather;       // set by the launcher: the looked-up term is loaded with the non-temporal hint
    // vote side (RequestVoteRPC inputs, raft.go:181-210)
    const T *voter_term;
    const T *voted_for;
    const T *last_log_term;
    const T *cand_term;
    const T *cand_id;
    // joint consensus member masks
    const uint16_t *old_mask;
    const uint16_t *new_mask;
    // outputs
    T *commit_out;
    T *voter_term_out;
    uint16_t *grant_mask;
    uint16_t *stepdown_mask;
    uint8_t  *elected;
    T *max_term;
    uint8_t  *status_out;    // may be null
    uint64_t *elected_bits;  // may be null
    uint64_t *changed_bits;  // may be null: 1 bit per group, set when the round changed or flagged anything
    // narrow summary for the sharded exchange (all may be null): per group
    // commit'[g] - summary_base[g] in summary_bits (16, 8 or 4) bits, all ones = "does
    // not fit".  16: uint16 per group; 8: uint8 per group; 4: 16 groups per 64-bit
    // word (the buffer then must be 8-byte aligned and hold ceil(G/16) words).
    uint16_t *commit_lo16;
    const T *summary_base;         // null = 0; may be the input commit plane (offset = the round's advance)
    uint32_t *summary_escapes;     // incremented once per group that did not fit
    uint32_t summary_bits;
    // may be null: incremented by one when the launch starts, i.e. when everything
    // enqueued before it on its stream has completed
    uint64_t *start_signal;
};
// Cache policy of the log-term gather: non-temporal only while neighbouring groups share
// arena lines (short logs: the wave's gathers sweep the arena like one more stream, measured
// 4 % faster at 8-34 bytes of arena per group and 2 % faster at 68 bytes), plain when every
// gather lands in a line of its own (long logs: non-temporal measured 8-13 % slower at
// 200 bytes - 3 KB per group).
template <typename T>
inline uint32_t rq_nt_gather(const rq_round_args_t<T> &a) {
    return a.log_terms * (uint64_t)(a.log_term_bits / 8) <= 128ull * (uint64_t)a.G ? 1u : 0u;
}
typedef rq_round_args_t<int32_t> rq_round_args;
typedef rq_round_args_t<int64_t> rq_round64_args;

// Kernel build selector ("variant" in the C-ABI).  Never changes results.
//   0          library default
//   bit  0     RQ_VAR_PLAIN: default cache policy for every load and store
//              (the default build streams everything with non-temporal accesses:
//              every byte is touched once per round, so nothing is worth keeping
//              in L2 or the Infinity Cache)
//   bits 8-11  threads per block / 64: 1..4 (0 = the family's default)
// Measured on MI355X at 1M groups x 5 slots: one group per thread in
// 128-thread blocks with non-temporal loads and stores is the fastest fused
// round (many short blocks keep reads, gathers and writes of different blocks
// overlapped in time).
#define RQ_VAR_PLAIN 0x1

#define RQ_WHAT_COMMIT 1
#define RQ_WHAT_VOTE   2

// Segment table of one round (may be absent: count == 0).  The groups
// [first_group[s], first_group[s + 1]) all have the SAME fixed cluster size n[s]
// (1..9: their member masks, if the round has any, are old == new == the n[s] lowest
// slots -- the caller vouches for it) or, with n[s] == 0, are decided through their
// member masks.  The segments cover [0, G) in order; every boundary but the last
// is a multiple of 64.  A host that keeps groups of different sizes in one image
// orders them by size; the fixed-size segments then run the register networks of
// the fixed-size round inside the same launch and only the rest pays for masks.
#define RQ_MAX_SEGS 16
struct rq_round_segs {
    uint32_t count;
    uint32_t first_group[RQ_MAX_SEGS + 1];
    uint8_t n[RQ_MAX_SEGS];
};

// Enqueue one round on `stream`.  `what` = RQ_WHAT_COMMIT | RQ_WHAT_VOTE.
// n is the fixed cluster size (1..16) or, with joint = true, ignored in favour
// of a.m and the masks.  Returns hipErrorInvalidValue for shapes it rejects.
// segs: null or a table with count >= 1 (then joint must be true: a.m planes).
hipError_t rq_launch_round(const rq_round_args &a, uint32_t n, bool joint,
                           int what, int variant, hipStream_t stream, const rq_round_segs *segs = nullptr);
// The same round over 64-bit planes.
hipError_t rq_launch_round(const rq_round64_args &a, uint32_t n, bool joint,
                           int what, int variant, hipStream_t stream, const rq_round_segs *segs = nullptr);

// Several independent images decided by ONE grid (short rounds share one ramp-up and
// one drain; images of different cluster sizes each keep their own compact tiles).
// args / n / joint: image i as rq_launch_round takes it (host memory); table: the same
// `count` (1..RQ_MAX_SEGS) argument structs in device memory.  variant: block size bits only.
hipError_t rq_launch_rounds(const rq_round_args *args, const uint32_t *n, const bool *joint, uint32_t count,
                            const rq_round_args *table, int what, int variant, hipStream_t stream);
hipError_t rq_launch_rounds(const rq_round64_args *args, const uint32_t *n, const bool *joint, uint32_t count,
                            const rq_round64_args *table, int what, int variant, hipStream_t stream);

// AppendEntries receiver-side consistency batch (raft.go:142-153) plus the
// leader's nextIndex/matchIndex bookkeeping for the replies.  Slot id
// s = p * ld + g indexes both the planes and the follower-log CSR offsets.
template <typename T>
struct rq_append_args_t {
    uint32_t G, ld, n;
    const uint16_t *active_mask;    // may be null: all n slots receive the RPC
    const T *follower_term;         // [n][ld]
    const T *leader_term;           // [G]
    const T *prev_log_index;        // [n][ld]
    const T *prev_log_term;         // [n][ld]
    const T *n_new;                 // [n][ld], may be null (0)
    const uint32_t *flog_off;       // [n*ld + 1]
    const void *flog_term;          // arena of flog_term_bits-wide entries
    const T *flog_base;             // [n][ld], may be null (0): arena slot 0 of slot s holds log[flog_base[s]]
    const T *flog_term_base;        // [G], may be null (0): added to every arena entry of the group's followers
    uint64_t flog_terms;            // entries in the arena
    uint32_t flog_term_bits;        // 8, 16, 32 or 8 * sizeof(T)
    T *follower_term_out;           // [n][ld]
    uint16_t *success_mask;         // [G]
    uint16_t *stepdown_mask;        // [G]
    T *match_index_io;              // [n][ld], may be null
    T *next_index_io;               // [n][ld], may be null
    uint8_t  *status_out;           // [G], may be null
    // The follower's commit clamp (raft.go:172-176) for the slots whose RPC succeeds; all three
    // null, or all three given (updated in place like match_index_io):
    const T *leader_commit;         // [G]
    const T *last_new_index;        // [n][ld]: lastEntry(newEntries).Index
    T *follower_commit_io;          // [n][ld]
    // 0 = slot-major planes and a CSR over slot ids s = p * ld + g.  Else the 64-group tiles of
    // the round kernels (rq_round_args_t): tile_bytes is the stride of the block that holds the
    // inputs and the arrays updated in place, out_tile_bytes that of the pure outputs
    // (success_mask, stepdown_mask, status_out, and follower_term_out unless it IS follower_term);
    // flog_off then is per slot: element (p, g) is the start of follower (p, g)'s log and
    // element (p, g + 1) its end, g = 0 .. G.
    uint64_t tile_bytes, out_tile_bytes;
};
typedef rq_append_args_t<int32_t> rq_append_args;
typedef rq_append_args_t<int64_t> rq_append64_args;
// commit_round: null, or the commit round to decide in the same launch (its match array is
// the batch's match_index_io; only its commit-side fields are used)
hipError_t rq_launch_append_check(const rq_append_args &a, const rq_round_args *commit_round, hipStream_t stream);
hipError_t rq_launch_append_check(const rq_append64_args &a, const rq_round64_args *commit_round, hipStream_t stream);

// commit'[i] = lc[i] > ci[i] ? min(lc[i], last_new[i]) : ci[i]   (raft.go:172-176)
hipError_t rq_launch_follower_clamp(uint64_t count, const int32_t *leader_commit,
                                    const int32_t *commit,
                                    const int32_t *last_new_index,
                                    int32_t *commit_out, hipStream_t stream);
hipError_t rq_launch_follower_clamp(uint64_t count, const int64_t *leader_commit,
                                    const int64_t *commit,
                                    const int64_t *last_new_index,
                                    int64_t *commit_out, hipStream_t stream);

// Incremental host <-> device exchange of single groups for a host that keeps the
// state of a round_args set resident.  Scatter: row r overwrites group gids[r]:
// rows[r] = { match[n], voter_term[n], voted_for[n], last_log_term[n], commit,
// leader_term, cand_term, cand_id, log_base } as T words; the group's log terms are
// row_terms[row_off[r] .. row_off[r+1]) (T words), written into its arena slice in
// the arena's entry width after subtracting log_term_base[g]; log_len[g] is set.
// row_status[r] (may be null): 1 = the terms do not fit the slice's capacity
// (nothing of the log written), 2 = a term does not fit the entry width, 4 = gid
// out of range (row ignored).  Gather: rows_out[r] = { commit, max_term,
// voter_term[n] } as T words, flags_out[r] = grant | stepdown << 16 | elected << 32
// | status << 40.
hipError_t rq_launch_groups_scatter(const rq_round_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                    const int32_t *rows, const uint32_t *row_off, const int32_t *row_terms,
                                    uint8_t *row_status, hipStream_t stream);
hipError_t rq_launch_groups_scatter(const rq_round64_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                    const int64_t *rows, const uint32_t *row_off, const int64_t *row_terms,
                                    uint8_t *row_status, hipStream_t stream);
hipError_t rq_launch_groups_gather(const rq_round_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                   int32_t *rows_out, uint64_t *flags_out, hipStream_t stream);
hipError_t rq_launch_groups_gather(const rq_round64_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                   int64_t *rows_out, uint64_t *flags_out, hipStream_t stream);

// Summary consumer of a sharded deployment: adds the gathered narrow commit
// summaries (advance coding) of `ranks` ranks to this rank's replica of every
// rank's commitIndex.  msg: ranks messages of msg_stride bytes, each holding
// `rounds` rounds round_stride bytes apart, the commit section of a round --
// `groups` offsets of `bits` (16, 8, 4) bits -- at byte offset section_off;
// replica: [ranks][groups] int32.  Escape values add nothing and are counted in
// *escapes.
hipError_t rq_launch_summary_apply(const uint8_t *msg, uint64_t msg_stride, uint64_t section_off,
                                   uint64_t round_stride, uint32_t rounds, uint32_t ranks, uint32_t groups, uint32_t bits,
                                   int32_t *replica, uint32_t *escapes, hipStream_t stream);
