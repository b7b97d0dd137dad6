/*
 * rq.h -- C ABI of librq.so, the MI355X (gfx950) batched Raft quorum engine.
 *
 * This is the drop-in boundary a Go host (package raft, cgo) binds for the
 * per-group quorum arithmetic of tawawhite/raft's raft.go, evaluated for very
 * many independent Raft groups per call.  The reference has no FFI of its
 * own; each entry point below cites the reference code it replaces.
 *
 * Layout (all entry points): structure-of-arrays, slot-major planes
 *     field[p * ld + g]      p = slot (peer) 0..n-1,  g = group 0..G-1
 * with ld >= G the plane stride in elements.  n counts every member of the
 * cluster INCLUDING the node itself (raft.go:94-97, :114).  Terms, indices
 * and ids are int32 in the compact layout (Go int is 64-bit: the host narrows
 * and rejects values outside int32) and int64 in the wide layout of the
 * rq_round64* entry points.  Leader logs are a CSR arena: group g's log is
 *     log_term[log_off[g] .. log_off[g+1])      slot i holds log[i].TermNum
 * indexed directly by slot (the convention of raft.go:151).
 *
 * Conventions: every function returns 0 (RQ_OK) or a negative RQ_E* code and
 * never throws or aborts.  Per-group data errors that are Go panics in the
 * reference go to status_out bits, not to the return code.  The library
 * copies caller buffers before returning and never retains a caller pointer.
 * Output planes may alias the matching input plane (in-place update).  One
 * engine owns one device, one stream and one mutex; distinct engines may be
 * used concurrently.  Every entry point makes its device current first, so a
 * caller (goroutine) may arrive on any OS thread.
 *
 * No CPU fallback exists: without a usable GPU rq_engine_create fails.
 *
 * Threading.  The host-buffer entry points (rq_fused_round, rq_commit_advance,
 * rq_vote_tally, their _joint forms, rq_round64, rq_append_check(64),
 * rq_follower_clamp(64), rq_upload / rq_step / rq_download) take the engine's
 * mutex, use its stream and return when the results are in the caller's
 * buffers: any number of threads may share an engine, the calls are serialised.
 * The device-pointer calls take NO lock and touch no engine state beyond the
 * device number and the last-error word: rq_round_dev, rq_append_check_dev,
 * rq_groups_scatter_dev, rq_groups_gather_dev, rq_summary_apply_dev,
 * rq_plan_round(s), rq_plan_launch, rq_stream_wait_signal, rq_stream_sync,
 * rq_device_alloc / _free, rq_host_alloc / _free, rq_copy_to / _from_device.
 * They only enqueue on the stream they are given (the copies and rq_stream_sync
 * block the calling thread), so they may run concurrently with each other and
 * with the host-buffer calls, from any threads, on the same engine.  What the
 * caller must not do: destroy a plan, an engine or a communicator while another
 * thread is inside a call that uses it; free or overwrite device buffers that a
 * launch still in flight reads or writes (ordering between launches is the
 * stream's: two launches on one stream run in order, on two streams they may
 * overlap); launch two rounds that update the same image in place on different
 * streams.  A plan is immutable after rq_plan_round(s) and may be launched from
 * several threads at once.  rq_last_hip_error is the last failure of any thread.
 *
 * Stability.  STABLE (kept source- and binary-compatible within a major
 * version; descriptors grow only at their end and older struct_size values stay
 * accepted): the engine lifetime calls, every host-buffer entry point,
 * rq_round_dev / rq_plan_round / rq_plan_launch / rq_plan_destroy with
 * rq_round_desc, rq_append_check_dev with rq_append_desc, rq_groups_scatter_dev /
 * _gather_dev, rq_stream_sync, the memory calls with RQ_MEM_DEFAULT,
 * rq_fused_round_noalloc, the status bits and return codes.  EXPERIMENTAL (may
 * change shape between minor versions as the multi-GPU path is measured on real
 * nodes): start signals (rq_round_desc.start_signal, rq_stream_wait_signal,
 * RQ_MEM_SIGNAL), the rq_comm_* communicator, rq_summary_apply_dev and the narrow
 * commit summary fields, rq_plan_rounds, the segment table, the RQ_MEM_FINEGRAINED
 * / RQ_MEM_UNCACHED kinds and the `variant` bits.
 */
#ifndef RQ_H
#define RQ_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RQ_VERSION 0x000300 /* 0.3.0: rq_round_desc gained the segment table (n_segments, seg_first_group, seg_n); 0.2.x descriptors are still accepted */

/* return codes */
#define RQ_OK            0
#define RQ_E_ARG        -1  /* null pointer, n out of range, ld < G, ...        */
#define RQ_E_RANGE      -2  /* size exceeds what the engine was created for    */
#define RQ_E_NOMEM      -3  /* host or device allocation failed                */
#define RQ_E_HIP        -4  /* HIP runtime failure, see rq_last_hip_error      */
#define RQ_E_NODEVICE   -5  /* no such GPU / not a gfx950 device               */
#define RQ_E_STATE      -6  /* rq_step/rq_download before rq_upload            */
#define RQ_E_MASK       -7  /* joint: an empty configuration mask              */
#define RQ_E_COMM       -8  /* RCCL unavailable or a collective call failed    */

/* status_out bits (one byte per group) */
#define RQ_ST_LOG_OOB   0x01u /* commit: N > commitIndex but N is outside the
                                 leader's log; reference would panic indexing
                                 log[N] (convention raft.go:151); no advance   */
#define RQ_ST_EMPTY_LOG 0x02u /* vote: a voter reached lastEntry(log)
                                 (raft.go:204, :235) with an empty log; the
                                 reference panics; that voter does not grant  */
#define RQ_ST_BAD_CSR   0x04u /* device launches: the group's log offsets are
                                 out of order or point past the arena; nothing
                                 was read through them, no advance / success   */

/* voter_last_log_term value meaning "this voter's log is empty" */
#define RQ_LOG_EMPTY INT32_MIN

#define RQ_MAX_SLOTS 16

typedef struct rq_engine rq_engine;

int         rq_version(void);
const char *rq_strerror(int code);

/* Engine lifetime.  Device buffers for max_groups x max_slots planes and a
 * log arena of max_log_terms entries are allocated once here.  Replaces the
 * Go heap ownership of Node state (raft.go:15-69) for the batched path. */
int rq_engine_create(int device, uint32_t max_groups, uint32_t max_slots,
                     uint64_t max_log_terms, rq_engine **out);
int rq_engine_destroy(rq_engine *e);
int rq_last_hip_error(const rq_engine *e);      /* last hipError_t seen       */
int rq_engine_device(const rq_engine *e);

/* ---- synchronous entry points on caller (host) buffers ------------------ */

/* Leader commitIndex advance for G groups of n slots.  The reference declares
 * the state (Node.matchIndex raft.go:65-68 zero-initialised at :114-117,
 * commitIndex :48-51, currentTerm :33-35) and never computes the rule; the
 * rule is Raft Fig. 2 "Leaders":
 *     N = ascending_sorted(match[0..n))[(n-1)/2]
 *     commit' = (N > commit && log[N].TermNum == leader_term) ? N : commit
 * match: [n][ld]; commit, leader_term, commit_out: [G]; log_off: [G+1];
 * status_out: [G] or NULL. */
int rq_commit_advance(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                      const int32_t *match, const int32_t *commit,
                      const int32_t *leader_term, const uint32_t *log_off,
                      const int32_t *log_term, int32_t *commit_out,
                      uint8_t *status_out);

/* RequestVote for all n slots of each group plus the candidate-side tally.
 * Per slot this is exactly Node.RequestVoteRPC (raft.go:181-210) with
 * testToAbdicateLeadership (raft.go:212-223) and lastEntry (raft.go:234-236):
 *     cur'   = max(cur, term);  stepdown bit p = term > cur      (:219-221)
 *     grant  = term >= cur && (votedFor == -1 || votedFor == candId)
 *              && voterLastLogTerm <= term                       (:190-205)
 * (lastLogIndex/lastLogTerm of the RPC are never read by the reference and
 * are not part of this ABI; votedFor is never written.)  The tally is the
 * caller BecomeLeader (raft.go:101) never had:
 *     max_term = max_p cur'_p ;  elected = 2*popcount(grant) > n
 *                                          && max_term == cand_term
 * voter_term, voted_for, voter_last_log_term, voter_term_out: [n][ld];
 * cand_term, cand_id, max_term: [G]; grant_mask, stepdown_mask: [G] bit p =
 * slot p; elected: [G] 0/1; status_out: [G] or NULL. */
int rq_vote_tally(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                  const int32_t *voter_term, const int32_t *voted_for,
                  const int32_t *voter_last_log_term, const int32_t *cand_term,
                  const int32_t *cand_id, int32_t *voter_term_out,
                  uint16_t *grant_mask, uint16_t *stepdown_mask,
                  uint8_t *elected, int32_t *max_term, uint8_t *status_out);

/* Both decisions for every group in ONE kernel launch (one "group-round").
 * status_out carries the OR of both status bit sets. */
int rq_fused_round(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                   const int32_t *match, const int32_t *commit,
                   const int32_t *leader_term, const uint32_t *log_off,
                   const int32_t *log_term,
                   const int32_t *voter_term, const int32_t *voted_for,
                   const int32_t *voter_last_log_term, const int32_t *cand_term,
                   const int32_t *cand_id,
                   int32_t *commit_out, int32_t *voter_term_out,
                   uint16_t *grant_mask, uint16_t *stepdown_mask,
                   uint8_t *elected, int32_t *max_term, uint8_t *status_out);

/* Joint-consensus variants (no counterpart in raft.go; membership there is
 * the fixed peers slice of raft.go:26).  m slot planes, two member masks per
 * group (bit p = slot p).  Commit: N = min(quorum index of old members,
 * quorum index of new members), each over its own member count.  Election:
 * strict majority of grants in BOTH configurations.  Slots outside old|new
 * are not evaluated; their voter_term_out keeps the input value on the host
 * entry points and in-place device use (out-of-place device launches leave it
 * unwritten or copy the input).  Mixed cluster sizes are expressed as
 * old_mask == new_mask == (1<<n_g)-1. */
int rq_commit_advance_joint(rq_engine *e, uint32_t G, uint32_t m, uint32_t ld,
                            const uint16_t *old_mask, const uint16_t *new_mask,
                            const int32_t *match, const int32_t *commit,
                            const int32_t *leader_term, const uint32_t *log_off,
                            const int32_t *log_term, int32_t *commit_out,
                            uint8_t *status_out);
int rq_vote_tally_joint(rq_engine *e, uint32_t G, uint32_t m, uint32_t ld,
                        const uint16_t *old_mask, const uint16_t *new_mask,
                        const int32_t *voter_term, const int32_t *voted_for,
                        const int32_t *voter_last_log_term,
                        const int32_t *cand_term, const int32_t *cand_id,
                        int32_t *voter_term_out, uint16_t *grant_mask,
                        uint16_t *stepdown_mask, uint8_t *elected,
                        int32_t *max_term, uint8_t *status_out);
int rq_fused_round_joint(rq_engine *e, uint32_t G, uint32_t m, uint32_t ld,
                         const uint16_t *old_mask, const uint16_t *new_mask,
                         const int32_t *match, const int32_t *commit,
                         const int32_t *leader_term, const uint32_t *log_off,
                         const int32_t *log_term,
                         const int32_t *voter_term, const int32_t *voted_for,
                         const int32_t *voter_last_log_term,
                         const int32_t *cand_term, const int32_t *cand_id,
                         int32_t *commit_out, int32_t *voter_term_out,
                         uint16_t *grant_mask, uint16_t *stepdown_mask,
                         uint8_t *elected, int32_t *max_term,
                         uint8_t *status_out);

/* Follower commit clamp, AppendEntriesRPC step 5 (raft.go:172-176, minInt
 * raft.go:226-231):  commit' = lc > ci ? min(lc, last_new_index) : ci.
 * May lower commitIndex, as the reference does.  All arrays [count]. */
int rq_follower_clamp(rq_engine *e, uint64_t count, const int32_t *leader_commit,
                      const int32_t *commit, const int32_t *last_new_index,
                      int32_t *commit_out);

/* AppendEntries receiver-side consistency batch for every (group, follower)
 * slot s = p*ld + g, and the leader's bookkeeping for the replies.
 * Receiver (AppendEntriesRPC raft.go:142-153, testToAbdicateLeadership :212-223):
 *     cur' = max(cur, term); stepdown bit p = term > cur
 *     success bit p = term >= cur && log[prevLogIndex].TermNum == prevLogTerm
 * with log[] indexed directly by slot (:151); prevLogIndex outside the
 * follower's log is an index panic in the reference: RQ_ST_LOG_OOB, no success.
 * Steps 3-5 of the RPC (:155-176: log mutation, string payloads, commit clamp)
 * are not part of this call; the clamp is rq_follower_clamp.
 * Leader (the reference declares nextIndex :60-63 / matchIndex :65-68 and
 * never updates them; rule: Raft Fig. 2): success -> match = prevLogIndex +
 * n_new, next = match + 1; rejected at the log check -> next = max(next-1, 1);
 * rejected for a stale term -> unchanged.
 * active_mask ([G], bit p; NULL = all n slots) selects the slots that receive
 * an RPC this round; other slots keep their values (arrays updated in place may
 * be rewritten with what they held, and a follower_term_out that is not
 * follower_term itself either keeps its old content there or receives a copy of
 * follower_term: whole rows are written wherever some group of a wave uses the slot).  Follower logs are one
 * CSR arena indexed by slot id: flog_term[flog_off[s] .. flog_off[s+1]),
 * flog_off has n*ld + 1 entries.  n_new, match_index_io, next_index_io and
 * status_out may be NULL. */
int rq_append_check(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                    const uint16_t *active_mask,
                    const int32_t *follower_term, const int32_t *leader_term,
                    const int32_t *prev_log_index, const int32_t *prev_log_term,
                    const int32_t *n_new,
                    const uint32_t *flog_off, const int32_t *flog_term,
                    int32_t *follower_term_out, uint16_t *success_mask,
                    uint16_t *stepdown_mask, int32_t *match_index_io,
                    int32_t *next_index_io, uint8_t *status_out);

/* ---- device-resident round: state stays in HBM between rounds ----------- */

/* rq_upload copies one round's inputs (same arguments as rq_fused_round's
 * inputs) into the engine's device buffers; rq_step runs `rounds` fused
 * rounds on them with no host transfer (voter terms and commitIndex are
 * updated in place, as the Go host would apply them; 8 or more rounds are
 * captured once into a HIP graph and replayed); rq_download copies the
 * current outputs back (voter_term_out dense, [n][G]).  Any output pointer may
 * be NULL to skip it. */
int rq_upload(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
              const int32_t *match, const int32_t *commit,
              const int32_t *leader_term, const uint32_t *log_off,
              const int32_t *log_term,
              const int32_t *voter_term, const int32_t *voted_for,
              const int32_t *voter_last_log_term, const int32_t *cand_term,
              const int32_t *cand_id);
int rq_step(rq_engine *e, uint32_t rounds);
int rq_download(rq_engine *e, int32_t *commit_out, int32_t *voter_term_out,
                uint16_t *grant_mask, uint16_t *stepdown_mask,
                uint8_t *elected, int32_t *max_term, uint8_t *status_out);

/* ---- asynchronous launches on caller-owned DEVICE buffers --------------- *
 * For hosts that keep group state resident in HBM (and for benchmarking
 * without PCIe in the loop).  One descriptor names a round's device buffers
 * and options; rq_round_dev enqueues it on `stream` (a hipStream_t used as
 * given; NULL = the device's default stream, which is also what torch calls its
 * default stream) and returns without synchronising.  The element type of every
 * term, index and id plane is int32_t (RQ_LAYOUT_I32, the compact layout) or
 * int64_t (RQ_LAYOUT_I64, the wide layout: Go's int is 64-bit on amd64,
 * raft.go:35-68; an empty voter log is then RQ_LOG_EMPTY64).  Same rules, same
 * status bits, same masks and byte outputs in both.
 *
 *   what        RQ_ROUND_COMMIT, RQ_ROUND_VOTE or both (one launch either way);
 *               pointers of the side that is not selected may be NULL.
 *   old_mask / new_mask   both NULL: every group has the n slots as its one
 *               configuration (the fixed-size round); both given: the
 *               joint-consensus round described above, n = number of planes.
 *   log_off / log_term / log_terms   the leader-log CSR arena and the number of
 *               entries in it.  No lookup goes at or beyond log_terms, whatever
 *               log_off holds: a group whose offsets are out of order or point
 *               past log_terms gets RQ_ST_BAD_CSR and does not advance.
 *   log_term_bits   width of one arena entry: 0 = the layout's element; 8 or 16
 *               = unsigned narrow entries, 32 = int32 entries in the wide
 *               layout.  A round reads ONE term per group but HBM hands out
 *               128-byte lines, so a narrow arena cuts the gather's traffic by
 *               the same factor.  log_term_base ([G], may be NULL = 0) is added
 *               to group g's entries before the comparison with leader_term
 *               (the host keeps base[g] <= every resident term of the group
 *               and picks the narrowest width the differences fit).  log_term
 *               is aligned to its entry width (else RQ_E_ARG); the lookup may read
 *               the other bytes of the aligned word (of the layout's element size) an entry lies in.
 *   log_base    [G], may be NULL: the arena is a window, slot i of group g's
 *               slice holds log[log_base[g] + i], so a host may keep only the
 *               uncommitted suffix of every log on the device (entries at or
 *               below commitIndex can never be looked up: the rule needs N >
 *               commitIndex); an N below the base is RQ_ST_LOG_OOB like an N
 *               past the end.
 *   log_len     [G], may be NULL: group g's log is then the log_len[g] entries
 *               from log_off[g] on, wherever that lies in the arena (log_off[g+1]
 *               is not read and the offsets need not be ordered), so a host can
 *               leave room behind each log to grow in place and move a log that
 *               ran out of room to free arena space (rq_groups_scatter_dev); a
 *               log that would end past log_terms is RQ_ST_BAD_CSR.
 *   changed_bits   may be NULL; else 1 bit per group, packed like elected_bits,
 *               set when the round moved the group's commitIndex, stepped one of
 *               its voters down, elected its candidate or set a status bit: the
 *               groups a host has to look at after the round.
 *   elected_bits   may be NULL; else the elected decisions packed 1 bit per
 *               group ([ceil(G/64)] uint64, bit g%64 of word g/64): the summary
 *               vector ranks exchange in a sharded deployment.
 *   commit_summary / summary_bits / summary_base / summary_escapes   may be
 *               NULL; else the other half of that summary, commit'[g] -
 *               summary_base[g] (NULL = 0) in summary_bits bits per group:
 *               16 (uint16 per group; 0 means 16), 8 (one byte per group) or 4
 *               (sixteen groups per 64-bit word, group g in bits 4*(g%16) of
 *               word g/16; the buffer is ceil(G/16) words, 8-byte aligned).
 *               A difference that does not fit below the all-ones value is
 *               stored as all ones (the escape value) and counted in
 *               *summary_escapes (exact up to 65536, it stops growing soon
 *               after: the count tells a host that its window is wrong, the
 *               escape values tell the receivers which groups); the receiver
 *               then fetches the full commit vector.  summary_base may be the
 *               round's own input commit plane: the summary is then the round's
 *               advance, which a receiver adds to the commitIndex it already
 *               holds (xGMI links are point-to-point, so the bytes per group on
 *               the wire decide how far the exchange of a 30 us round scales).
 *   log_term_start   may be NULL.  Else the round makes NO arena lookup: the test
 *               log[N].TermNum == currentTerm is evaluated as N >= log_term_start[g],
 *               the index (in matchIndex's index space) of the first entry of the
 *               leader's current term, or the index one past its last entry when
 *               it has none yet.  The two are the same whenever the terms of a log
 *               never decrease and never exceed the leader's term -- every log Raft
 *               itself produces; the reference does not enforce it (raft.go:158-170
 *               appends whatever it is sent), so the lookup stays the default.  A
 *               host knows the value when it appends.  log_len is then required
 *               (N outside [log_base, log_base + log_len) is RQ_ST_LOG_OOB as before);
 *               log_off, log_term and leader_term are not read.  One gathered term
 *               costs a 128-byte HBM line of its own once logs are long: at 1M
 *               groups with Zipf log lengths this path moves 40 MB less per round.
 *   tile_bytes / out_tile_bytes   0: every array is a slot-major plane,
 *               field[p * ld + g].  Else the tiled layout: groups are cut into
 *               tiles of 64; tile t of the INPUT block holds the elements of its
 *               64 groups for every input array back to back in tile_bytes bytes,
 *               tile t of the OUTPUT block the same for the output arrays in
 *               out_tile_bytes bytes (multiples of 8, both zero or both not), and
 *               each array pointer names
 *               that array's place inside tile 0 of its block:
 *                   element (p, g) of array A = (char *)A + (g / 64) * stride
 *                                               + (p * 64 + g % 64) * sizeof(element)
 *               commit_out / voter_term_out equal to commit / voter_term (state
 *               updated in place) are addressed as inputs.  ld is ignored;
 *               log_off[G] is element G of that array (tile G/64 must exist).  A
 *               wave then streams one contiguous input tile and one contiguous
 *               output tile instead of 26 planes 4 MiB apart: 5 % faster at
 *               1M x 5 (one mixed tile for reads and writes gains nothing).  The
 *               log arena, elected_bits, changed_bits, the commit summary and an
 *               explicit summary_base stay plain arrays (rq_groups_scatter_dev /
 *               _gather_dev address either layout).
 *   variant     kernel build: 0 = library default; RQ_VARIANT_PLAIN = default
 *               cache policy instead of non-temporal streaming (a diagnostic: only
 *               libraries built with `make PLAIN=1` contain those kernels, any other
 *               runs the default build for it); bits 8-11 = threads per block / 64
 *               (1..4).  Never changes results.
 *   n_segments / seg_first_group / seg_n   a host that keeps groups of DIFFERENT
 *               cluster sizes in one image of n planes orders them by size and
 *               says so here: the groups seg_first_group[s] .. seg_first_group[s+1]
 *               - 1 all have the fixed cluster size seg_n[s] (1..9, at most n: their
 *               members are the seg_n[s] lowest slots and, if the round has member
 *               masks, old == new == (1 << seg_n[s]) - 1 -- the caller vouches for
 *               it, the masks of these groups are not read), or, with seg_n[s] == 0,
 *               are decided through their member masks (joint consensus, tiles in
 *               which sizes meet; with NULL masks: all n slots).  1 .. RQ_MAX_SEGMENTS
 *               segments covering [0, G) in order, seg_first_group[0] = 0,
 *               seg_first_group[n_segments] = G, every boundary between two
 *               segments a multiple of 64.  Still ONE launch: each block runs the
 *               fixed-size code of its segment, so the nine groups in ten of a
 *               mixed deployment that are plain clusters are decided at the speed
 *               of the fixed-size round and only the rest pays for masks.  The table
 *               never changes a result for groups it describes truthfully (a group
 *               listed under the wrong size is decided as a cluster of that size;
 *               nothing is read out of bounds).  old_mask / new_mask may be NULL
 *               when no group is in joint consensus.
 * Output planes may alias the matching input plane (commit_out = commit,
 * voter_term_out = voter_term: state updated in place between rounds).
 * A struct_size of RQ_ROUND_DESC_SIZE_V2 (the descriptor up to log_term_start, as
 * built against 0.2.x headers) is accepted and means "no segment table". */
#define RQ_LAYOUT_I32 4
#define RQ_LAYOUT_I64 8
#define RQ_MAX_SEGMENTS 16
#define RQ_ROUND_DESC_SIZE_V2 ((uint32_t)offsetof(struct rq_round_desc, n_segments))
#define RQ_LOG_EMPTY64  INT64_MIN
#define RQ_ROUND_COMMIT 1
#define RQ_ROUND_VOTE   2
#define RQ_VARIANT_PLAIN 0x1

typedef struct rq_round_desc {
    uint32_t struct_size;       /* sizeof(rq_round_desc) */
    uint32_t layout;            /* RQ_LAYOUT_I32 or RQ_LAYOUT_I64 */
    uint32_t what;              /* RQ_ROUND_COMMIT | RQ_ROUND_VOTE */
    int32_t  variant;
    uint32_t G, n, ld;
    uint32_t log_term_bits;
    uint32_t summary_bits;
    uint32_t reserved;          /* 0 */
    uint64_t log_terms;
    uint64_t tile_bytes;        /* 0 = slot-major planes */
    uint64_t out_tile_bytes;    /* tiled: stride of the output arrays' tiles */
    const uint16_t *old_mask, *new_mask;
    /* commit side */
    const void *match, *commit, *leader_term;
    const uint32_t *log_off;
    const void *log_term, *log_base, *log_term_base;
    const uint32_t *log_len;    /* may be NULL */
    /* vote side */
    const void *voter_term, *voted_for, *voter_last_log_term, *cand_term, *cand_id;
    /* outputs */
    void *commit_out, *voter_term_out;
    uint16_t *grant_mask, *stepdown_mask;
    uint8_t *elected;
    void *max_term;
    uint8_t *status_out;        /* may be NULL */
    uint64_t *elected_bits;     /* may be NULL */
    uint64_t *changed_bits;     /* may be NULL */
    void *commit_summary;       /* may be NULL */
    const void *summary_base;
    uint32_t *summary_escapes;
    uint64_t *start_signal;     /* may be NULL (see rq_stream_wait_signal) */
    const void *log_term_start; /* may be NULL (see above) */
    /* segment table, see above; n_segments = 0: none */
    uint32_t n_segments;
    uint32_t seg_first_group[RQ_MAX_SEGMENTS + 1];
    uint8_t  seg_n[RQ_MAX_SEGMENTS];
} rq_round_desc;

int rq_round_dev(rq_engine *e, void *stream, const rq_round_desc *d);

/* Prepared launches.  rq_plan_round validates one descriptor once and keeps a
 * copy; rq_plan_launch enqueues it on `stream` with nothing to re-marshal,
 * which matters when a round is only a few microseconds long (10k groups:
 * 1.7 us on the device).  The plan refers to the caller's device buffers, it
 * does not own them; it may be launched any number of times, also under stream
 * capture. */
typedef struct rq_plan rq_plan;
int rq_plan_round(rq_engine *e, const rq_round_desc *d, rq_plan **out);
/* Several rounds in ONE launch.  descs[0 .. count) (1 .. RQ_MAX_SEGMENTS descriptors of
 * one struct_size, layout, `what` and block size; no RQ_VARIANT_PLAIN, no segment
 * tables; start_signal only in descs[0]) name independent images, each with its own
 * arrays, cluster size or member masks, tiling and log arena; rq_plan_launch decides all
 * of them with one grid, every block running the fixed-size or the member-mask code of its
 * image.  Two uses.  A deployment whose groups have DIFFERENT cluster sizes keeps one
 * image per size, each in its own compact 64-group tiles (a tile of a 3-peer image holds
 * 3 slots, not the 9 or 16 of the widest group), plus one image with member masks for the
 * groups in joint consensus: measured at 1M groups of 3-9 peers this is the fastest way
 * to decide a mixed population.  And a host with several short batches amortises the
 * ramp-up and drain of a launch over all of them.  The argument structs are copied
 * to device memory here (the plan owns 300 bytes per image); results are those of
 * `count` separate rq_plan_launch calls.  No counterpart in the reference. */
int rq_plan_rounds(rq_engine *e, const rq_round_desc *descs, uint32_t count, rq_plan **out);
/* Start signal (rq_round_desc.start_signal).  A second stream that consumes the
 * outputs of the rounds enqueued so far -- the summary all-gather of a sharded
 * deployment -- would normally wait for an event recorded behind the last of
 * them; on MI355X that marker keeps the next round from starting while the
 * previous one drains and costs 8-13 us per dependency, a third of a 1M-group
 * round.  With start_signal set a launch adds one to *start_signal when it
 * STARTS; a stream is in order, so by then every launch enqueued before it on
 * its stream has completed and all of their outputs are visible.  The word is
 * 64 bits from rq_device_alloc(e, 8, RQ_MEM_SIGNAL, ..) and only ever grows;
 * rq_stream_wait_signal makes `stream` wait until *signal >= value without
 * touching the rounds' stream.  Typical use: the first round of the next batch
 * carries the signal that releases the exchange of the previous batch; the host
 * counts the launches it has enqueued with a signal and waits for that count.
 * Enqueue the signalling launch BEFORE the wait (streams may share a hardware
 * queue).  The last batch of a run, with no launch after it, is fenced with an
 * event as usual.  No counterpart in the reference (its peers are in-process
 * pointers, raft.go:26). */
int rq_stream_wait_signal(rq_engine *e, void *stream, const uint64_t *signal, uint64_t value);
int rq_plan_launch(rq_plan *p, void *stream);
/* A plan made by rq_plan_rounds owns device memory: destroying it waits for the device. */
int rq_plan_destroy(rq_plan *p);

/* AppendEntries consistency batch on device buffers: the fields of
 * rq_append_check, plus
 *   flog_terms      entries in the follower-log arena (same guarantee as
 *                   log_terms above: RQ_ST_BAD_CSR instead of a stray read);
 *   flog_term_bits  width of one arena entry (0, 8, 16, 32 as above; flog_term aligned to it);
 *   flog_term_base  [G] or NULL: added to the entries of all followers of group g;
 *   flog_base       [n][ld] or NULL: follower s keeps only log[flog_base[s] ...]
 *                   on the device (the suffix around nextIndex is all a
 *                   consistency check can ask for);
 *   tile_bytes / out_tile_bytes   0: slot-major planes, and flog_off is a CSR over
 *                   slot ids s = p * ld + g as in rq_append_check.  Else the 64-group
 *                   tiles of rq_round_desc, so that one resident image serves rounds
 *                   and replication (follower_term IS the round's voter_term array,
 *                   match_index_io its match array): the inputs and the arrays updated
 *                   in place are addressed with tile_bytes, the pure outputs
 *                   (success_mask, stepdown_mask, status_out, and follower_term_out
 *                   unless it is follower_term itself) with out_tile_bytes; ld is
 *                   ignored; flog_off is per slot, element (p, g) the start of follower
 *                   (p, g)'s log and element (p, g + 1) its end, g = 0 .. G (tile G / 64
 *                   must exist);
 *   leader_commit [G] / last_new_index [n][ld] / follower_commit_io [n][ld]
 *                   all NULL, or all given: the same launch then also applies the
 *                   follower's commit rule of raft.go:172-176 to every slot whose RPC
 *                   succeeds, in place: commitIndex below leaderCommit becomes
 *                   min(leaderCommit, last_new_index), last_new_index being
 *                   lastEntry(newEntries).Index as the host read it (it may lower
 *                   commitIndex, as the reference does).  A succeeding slot with
 *                   leaderCommit above its commitIndex and n_new == 0 (n_new given) is
 *                   the reference's lastEntry(newEntries) panic (:175, :235):
 *                   RQ_ST_EMPTY_LOG, commitIndex unchanged.  Other slots keep their
 *                   commitIndex.  The steps between (:158-170, the log mutation) stay
 *                   with the host, which also vouches that they do not panic.
 *   commit_round    NULL, or the descriptor of the COMMIT round (what = RQ_ROUND_COMMIT,
 *                   same layout, G and tiling / ld) that would follow this batch on the
 *                   same image: its `match` must BE match_index_io, its n the batch's n
 *                   (the followers' slots) or n + 1 (the leader's own slot after them).
 *                   The leader's commit rule is then decided in this launch from the
 *                   matchIndex values still in registers -- same results as
 *                   rq_append_check_dev followed by rq_round_dev(commit_round), one
 *                   launch and 20 bytes per group less.  The round's log arena options
 *                   (log_term_bits, log_term_base, log_base, log_len, log_term_start),
 *                   commit_out (may be commit) and status_out apply; member masks,
 *                   summaries, bit outputs and start_signal must be NULL there. */
typedef struct rq_append_desc {
    uint32_t struct_size;       /* sizeof(rq_append_desc) */
    uint32_t layout;            /* RQ_LAYOUT_I32 or RQ_LAYOUT_I64 */
    uint32_t G, n, ld;
    uint32_t flog_term_bits;
    uint64_t flog_terms;
    const uint16_t *active_mask;
    const void *follower_term, *leader_term, *prev_log_index, *prev_log_term, *n_new;
    const uint32_t *flog_off;
    const void *flog_term, *flog_base, *flog_term_base;
    void *follower_term_out;
    uint16_t *success_mask, *stepdown_mask;
    void *match_index_io, *next_index_io;
    uint8_t *status_out;
    uint64_t tile_bytes, out_tile_bytes;
    const void *leader_commit, *last_new_index;
    void *follower_commit_io;
    const rq_round_desc *commit_round;
} rq_append_desc;
int rq_append_check_dev(rq_engine *e, void *stream, const rq_append_desc *d);

/* Single groups in and out of a resident round (no counterpart in the
 * reference: there the state IS the Go objects).  A host that keeps the planes of
 * descriptor `d` resident and applies RPCs to a few of its Node objects between
 * two rounds moves only those groups:
 * rq_groups_scatter_dev overwrites group gids[r] with row r,
 *     rows[r] = { match[n], voter_term[n], voted_for[n], voter_last_log_term[n],
 *                 commit, leader_term, cand_term, cand_id, log_base,
 *                 old_mask | new_mask << 16, log_off }     (4n + 7 elements)
 * in the layout's element type, and replaces the group's log by the terms
 * row_log_terms[row_log_off[r] .. row_log_off[r+1]) (elements; stored in the
 * arena's entry width after subtracting log_term_base[g]).  With a log_len plane
 * the log is written at the row's log_off (the host's choice: the old place if
 * there is room, else free arena space) and log_off[g], log_len[g] are set;
 * without one the row's log_off is ignored and the length must equal the group's
 * slice.  With log_term_start in the descriptor there is no arena: the row's last
 * element is the index where the leader's term starts, row_log_off[r+1] -
 * row_log_off[r] is the log's length (it goes to log_len[g]) and row_log_terms is
 * not read (may be NULL).  Fields whose
 * descriptor pointer is NULL are skipped.  A row is applied completely or not at
 * all; row_status[r] (may be NULL) = 0, or 1 the log does not fit the group's
 * slice, 2 a term does not fit the arena's entry width, 4 no such group.
 * rq_groups_gather_dev packs the outputs of the listed groups:
 *     rows_out[r] = { commit_out, max_term, voter_term_out[n] }   (n + 2 elements)
 *     flags_out[r] = grant_mask | stepdown_mask << 16 | elected << 32 | status << 40
 * All pointers are device pointers; both calls are enqueued on `stream`. */
int rq_groups_scatter_dev(rq_engine *e, void *stream, const rq_round_desc *d, uint32_t count,
                          const uint32_t *gids, const void *rows, const uint32_t *row_log_off,
                          const void *row_log_terms, uint8_t *row_status);
int rq_groups_gather_dev(rq_engine *e, void *stream, const rq_round_desc *d, uint32_t count,
                         const uint32_t *gids, void *rows_out, uint64_t *flags_out);

/* Blocks the calling thread until everything enqueued on `stream` has finished
 * (the device-pointer calls above only enqueue).  Polls the stream for the first
 * two milliseconds before it blocks, so that a host loop over short rounds does
 * not pay a thread wake-up per round.  No counterpart in the reference, whose
 * calls are synchronous (raft.go:132, 181). */
int rq_stream_sync(rq_engine *e, void *stream);

/* rq_fused_round for callers that cannot allocate.  It has no counterpart in the
 * reference: it exists for callers that may not call an allocator (tight
 * serving loops, harnesses that forbid hipMalloc) and therefore cannot create
 * an rq_engine, whose staging memory comes from hipMalloc.  Same arguments,
 * rules and outputs as rq_fused_round, but no engine: the round is staged
 * through a fixed RQ_NOALLOC_SCRATCH_BYTES array inside the library's device
 * image (every process holds one copy per GPU it uses until it ends) and runs on
 * `stream` (NULL = the default stream) of `device`; the call returns when the
 * outputs are in the caller's buffers.  It calls no hipMalloc / hipFree or any
 * other allocating routine.  A shape whose staging image (16n + 30 bytes per
 * group plus 4 per log entry) exceeds the array is refused with RQ_E_RANGE:
 * with n = 5 and logs of 8 entries about 110 000 groups.  Calls on one GPU share the
 * array: within a process they are serialised by the library; the array keeps
 * the previous call's contents, every byte a call reads it has written first. */
#define RQ_NOALLOC_SCRATCH_BYTES (16u << 20)
int rq_fused_round_noalloc(int device, void *stream, uint32_t G, uint32_t n, uint32_t ld,
                           const int32_t *match, const int32_t *commit,
                           const int32_t *leader_term, const uint32_t *log_off,
                           const int32_t *log_term,
                           const int32_t *voter_term, const int32_t *voted_for,
                           const int32_t *voter_last_log_term, const int32_t *cand_term,
                           const int32_t *cand_id,
                           int32_t *commit_out, int32_t *voter_term_out,
                           uint16_t *grant_mask, uint16_t *stepdown_mask,
                           uint8_t *elected, int32_t *max_term, uint8_t *status_out);

/* ---- wide layout on host buffers ----------------------------------------- *
 * rq_fused_round / rq_follower_clamp / rq_append_check over int64_t planes
 * (the 64-bit staging buffers are allocated by the first call and kept).
 * `what` and the optional masks as in rq_round_desc; log_base may be NULL;
 * log_off stays 32-bit (it indexes the arena, it is not a Raft value). */
int rq_round64(rq_engine *e, int what, uint32_t G, uint32_t m, uint32_t ld,
               const uint16_t *old_mask, const uint16_t *new_mask,
               const int64_t *match, const int64_t *commit,
               const int64_t *leader_term, const uint32_t *log_off,
               const int64_t *log_term, const int64_t *log_base,
               const int64_t *voter_term, const int64_t *voted_for,
               const int64_t *voter_last_log_term,
               const int64_t *cand_term, const int64_t *cand_id,
               int64_t *commit_out, int64_t *voter_term_out,
               uint16_t *grant_mask, uint16_t *stepdown_mask,
               uint8_t *elected, int64_t *max_term, uint8_t *status_out);
int rq_follower_clamp64(rq_engine *e, uint64_t count, const int64_t *leader_commit,
                        const int64_t *commit, const int64_t *last_new_index,
                        int64_t *commit_out);
int rq_append_check64(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                      const uint16_t *active_mask,
                      const int64_t *follower_term, const int64_t *leader_term,
                      const int64_t *prev_log_index, const int64_t *prev_log_term,
                      const int64_t *n_new,
                      const uint32_t *flog_off, const int64_t *flog_term,
                      int64_t *follower_term_out, uint16_t *success_mask,
                      uint16_t *stepdown_mask, int64_t *match_index_io,
                      int64_t *next_index_io, uint8_t *status_out);

/* Device memory for hosts that keep group state resident without another GPU
 * runtime binding (a Go host has none), to be used with the *_dev launches.
 * kind selects the hipExtMallocWithFlags flavour: 0 = ordinary device memory,
 * 1 = fine-grained, 2 = uncached.  (On MI355X the flavour did not change how
 * the log-term gather is fetched: every HBM read stayed a 128-byte request.)
 * The copies are synchronous. */
#define RQ_MEM_DEFAULT     0
#define RQ_MEM_FINEGRAINED 1
#define RQ_MEM_UNCACHED    2
#define RQ_MEM_SIGNAL      3   /* bytes must be 8: one zeroed 64-bit word a stream can wait on */
int rq_device_alloc(rq_engine *e, uint64_t bytes, int kind, void **out);
/* Page-locked host memory for the buffers handed to the host-buffer entry
 * points: copies from it run at the PCIe rate instead of being staged through
 * the runtime's bounce buffers (a Go host keeps its flat slices here, outside
 * the Go heap, so nothing is pinned by cgo either). */
int rq_host_alloc(rq_engine *e, uint64_t bytes, void **out);
int rq_host_free(rq_engine *e, void *ptr);
int rq_device_free(rq_engine *e, void *ptr);
int rq_copy_to_device(rq_engine *e, void *dst, const void *src, uint64_t bytes);
int rq_copy_from_device(rq_engine *e, void *dst, const void *src, uint64_t bytes);

/* ---- sharded deployment: the per-round summary exchange ------------------ *
 * Groups are independent, so ranks (one process per GPU) share nothing in the
 * decision kernels; the only exchange step of a round is an all-gather of each
 * rank's summary message (commit summary + elected bits, laid out by the
 * host) over RCCL/xGMI.  No counterpart in the reference (its "RPC" is a
 * method call on a pointer in the same address space, raft.go:26).
 * rq_comm_unique_id: rank 0 creates the 128-byte id and hands it to the other
 * ranks by any means (the Python host uses torch.distributed's store).
 * rq_comm_allgather: posts an all-gather of `bytes` bytes per rank (send ->
 * recv[rank*bytes ...]) ordered after everything already enqueued on
 * compute_stream; a worker thread owned by the communicator enqueues it on the
 * communicator's side stream, so the call returns in a few microseconds.
 * rq_comm_wait: orders compute_stream after slot's last all-gather (call
 * before a round overwrites that slot's buffers; required between two uses of
 * one slot).  rq_comm_sync: host waits for every posted all-gather.  `slots`
 * independent slots let the all-gather of round i overlap the kernels of
 * rounds i+1..; a rq_comm is driven by one caller thread at a time. */
typedef struct rq_comm rq_comm;
int rq_comm_unique_id(uint8_t *out128);
int rq_comm_create(int device, const uint8_t *id128, int rank, int world, uint32_t slots,
                   rq_comm **out);
int rq_comm_destroy(rq_comm *c);
int rq_comm_allgather(rq_comm *c, void *compute_stream, uint32_t slot, const void *send,
                      void *recv, uint64_t bytes);
int rq_comm_wait(rq_comm *c, void *compute_stream, uint32_t slot);
int rq_comm_sync(rq_comm *c);
int rq_comm_last_error(const rq_comm *c, int *nccl_result, int *hip_error);

/* Consumer side of the exchange, on the device: every rank keeps a replica of
 * every rank's commitIndex ([ranks][groups] int32) and adds the gathered
 * advance-coded summaries to it, without a host synchronisation.  gathered:
 * `ranks` messages of msg_stride bytes (the all-gather's receive buffer); each
 * message carries `rounds` rounds round_stride bytes apart, the commit section
 * of a round at byte section_off inside it, `bits` (16, 8 or 4) bits per group
 * as written by rq_round_dev with summary_base = the round's input commit
 * plane.  All rounds of the message are summed in one launch.  Escape values add
 * nothing and are counted in *escapes (may be NULL; same saturating count as
 * summary_escapes): the host then fetches the full vectors of that period.
 * Enqueued on `stream`; order it after the all-gather with rq_comm_wait. */
int rq_summary_apply_dev(rq_engine *e, void *stream, const void *gathered, uint64_t msg_stride,
                         uint64_t section_off, uint64_t round_stride, uint32_t rounds,
                         uint32_t ranks, uint32_t groups, uint32_t bits,
                         int32_t *replica, uint32_t *escapes);

#ifdef __cplusplus
}
#endif
#endif /* RQ_H */
