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
 */
#ifndef RQ_H
#define RQ_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RQ_VERSION 0x000100 /* 0.1.0 */

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
int rq_engine_set_summary_width(rq_engine *e, int bits); /* 16 (default), 8 or 4; see
                                                   the device-pointer section  */
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
 * an RPC this round; other slots are left untouched.  Follower logs are one
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
 * without PCIe in the loop).  Every pointer is a device pointer on the
 * engine's device; `stream` is a hipStream_t used as given (NULL = the device's
 * default stream, which is also what torch calls its default stream).
 * The launch is enqueued and the call returns without synchronising.
 * elected_bits, if not NULL, receives the elected decisions packed 1 bit per
 * group ([ceil(G/64)] uint64, bit g%64 of word g/64): the summary vector
 * ranks exchange in a sharded deployment.  commit_lo16, if not NULL, receives
 * the other half of that summary in 16 bits per group: commit'[g] -
 * summary_base[g] (summary_base NULL = 0) when that lies in [0, 65535), else
 * the escape value 65535, in which case *summary_escapes (if not NULL) is
 * incremented (exact up to 65536, it stops growing soon after: the count tells
 * a host that its window is wrong, the escape values themselves tell the
 * receivers which groups) and the receiver must fetch the full commit vector; the host
 * moves summary_base forward before offsets reach the window's end.
 * rq_engine_set_summary_width narrows the summary further for launches and
 * plans made afterwards on that engine: 8 = one byte per group (escape 255),
 * 4 = sixteen groups per 64-bit word, group g in bits 4*(g%16) of word g/16
 * (escape 15; the buffer is then ceil(G/16) words, 8-byte aligned); the buffer
 * is still passed as commit_lo16.  summary_base may be the round's own input
 * commit plane: the summary is then the round's advance, which a receiver adds
 * to the commitIndex it already holds (xGMI links are point-to-point, so the
 * bytes per group on the wire decide how far the exchange of a 30 us round
 * scales).
 * log_base, if not NULL, makes the log arena a window: slot i of group g's
 * arena slice holds log[log_base[g] + i], so a host may keep only the
 * uncommitted suffix of every log on the device (entries at or below
 * commitIndex can never be looked up: the rule needs N > commitIndex); an N
 * below the base is reported as RQ_ST_LOG_OOB like an N past the end.
 * variant selects a kernel build:
 * 0 = library default; other values are tuning knobs listed in rq_kernels.h
 * and never change results. */
int rq_fused_round_dev(rq_engine *e, void *stream, int variant,
                       uint32_t G, uint32_t n, uint32_t ld,
                       const int32_t *match, const int32_t *commit,
                       const int32_t *leader_term, const uint32_t *log_off,
                       const int32_t *log_term,
                       const int32_t *voter_term, const int32_t *voted_for,
                       const int32_t *voter_last_log_term,
                       const int32_t *cand_term, const int32_t *cand_id,
                       int32_t *commit_out, int32_t *voter_term_out,
                       uint16_t *grant_mask, uint16_t *stepdown_mask,
                       uint8_t *elected, int32_t *max_term,
                       uint8_t *status_out, uint64_t *elected_bits,
                       uint16_t *commit_lo16, const int32_t *summary_base,
                       uint32_t *summary_escapes, const int32_t *log_base);
int rq_commit_advance_dev(rq_engine *e, void *stream, int variant,
                          uint32_t G, uint32_t n, uint32_t ld,
                          const int32_t *match, const int32_t *commit,
                          const int32_t *leader_term, const uint32_t *log_off,
                          const int32_t *log_term, int32_t *commit_out,
                          uint8_t *status_out, const int32_t *log_base);
int rq_vote_tally_dev(rq_engine *e, void *stream, int variant,
                      uint32_t G, uint32_t n, uint32_t ld,
                      const int32_t *voter_term, const int32_t *voted_for,
                      const int32_t *voter_last_log_term,
                      const int32_t *cand_term, const int32_t *cand_id,
                      int32_t *voter_term_out, uint16_t *grant_mask,
                      uint16_t *stepdown_mask, uint8_t *elected,
                      int32_t *max_term, uint8_t *status_out,
                      uint64_t *elected_bits);
int rq_fused_round_joint_dev(rq_engine *e, void *stream, int variant,
                             uint32_t G, uint32_t m, uint32_t ld,
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
                             uint8_t *status_out, uint64_t *elected_bits,
                             uint16_t *commit_lo16, const int32_t *summary_base,
                             uint32_t *summary_escapes, const int32_t *log_base);
/* Prepared launches.  rq_plan_fused_round validates one rq_fused_round_dev
 * argument set once (joint = 0) or one rq_fused_round_joint_dev set (joint = 1,
 * old_mask/new_mask then required, n = number of slot planes) and keeps it;
 * rq_plan_launch enqueues it on `stream` with nothing to re-marshal, which
 * matters when a round is only a few microseconds long (10k groups: 1.7 us on
 * the device).  The plan refers to the caller's device buffers, it does not own
 * them; it may be launched any number of times, also under stream capture. */
typedef struct rq_plan rq_plan;
int rq_plan_fused_round(rq_engine *e, int variant, int joint,
                        uint32_t G, uint32_t n, uint32_t ld,
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
                        uint8_t *status_out, uint64_t *elected_bits,
                        uint16_t *commit_lo16, const int32_t *summary_base,
                        uint32_t *summary_escapes, const int32_t *log_base, rq_plan **out);
/* The same for one rq_round64_dev argument set (any `what`, masks optional). */
int rq_plan_round64(rq_engine *e, int variant, int what,
                    uint32_t G, uint32_t m, uint32_t ld,
                    const uint16_t *old_mask, const uint16_t *new_mask,
                    const int64_t *match, const int64_t *commit,
                    const int64_t *leader_term, const uint32_t *log_off,
                    const int64_t *log_term, const int64_t *log_base,
                    const int64_t *voter_term, const int64_t *voted_for,
                    const int64_t *voter_last_log_term,
                    const int64_t *cand_term, const int64_t *cand_id,
                    int64_t *commit_out, int64_t *voter_term_out,
                    uint16_t *grant_mask, uint16_t *stepdown_mask,
                    uint8_t *elected, int64_t *max_term,
                    uint8_t *status_out, uint64_t *elected_bits,
                    uint16_t *commit_lo16, const int64_t *summary_base,
                    uint32_t *summary_escapes, rq_plan **out);
int rq_plan_launch(rq_plan *p, void *stream);
int rq_plan_destroy(rq_plan *p);

int rq_append_check_dev(rq_engine *e, void *stream, uint32_t G, uint32_t n, uint32_t ld,
                        const uint16_t *active_mask,
                        const int32_t *follower_term, const int32_t *leader_term,
                        const int32_t *prev_log_index, const int32_t *prev_log_term,
                        const int32_t *n_new,
                        const uint32_t *flog_off, const int32_t *flog_term,
                        int32_t *follower_term_out, uint16_t *success_mask,
                        uint16_t *stepdown_mask, int32_t *match_index_io,
                        int32_t *next_index_io, uint8_t *status_out);
int rq_stream_sync(rq_engine *e, void *stream);

/* ---- wide layout: 64-bit terms, indices and ids -------------------------- *
 * The same round over int64_t planes, for hosts whose values do not fit the
 * compact layout (Go's int is 64-bit on amd64, raft.go:35-68).  Same rules,
 * same status bits, same masks and byte outputs; an empty voter log is
 * RQ_LOG_EMPTY64.  `what` selects the decisions: RQ_ROUND_COMMIT, RQ_ROUND_VOTE
 * or both (one launch either way); the pointers of the side that is not
 * selected may be NULL.  m slot planes; old_mask/new_mask are either both NULL
 * (every group has the m slots as its one configuration: the fixed-size round)
 * or both given (the joint-consensus round described above).  log_base may be
 * NULL (see the device-pointer section for its meaning); log_off stays 32-bit
 * (it indexes the arena, it is not a Raft value).
 * rq_round64 works on host buffers like rq_fused_round (its 64-bit staging
 * buffers are allocated by the first call and kept); rq_round64_dev enqueues a
 * launch on device buffers like rq_fused_round_joint_dev, with the same
 * variant word (one group per thread in this layout), elected_bits and narrow
 * commit summary. */
#define RQ_LOG_EMPTY64  INT64_MIN
#define RQ_ROUND_COMMIT 1
#define RQ_ROUND_VOTE   2
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
int rq_round64_dev(rq_engine *e, void *stream, int variant, int what,
                   uint32_t G, uint32_t m, uint32_t ld,
                   const uint16_t *old_mask, const uint16_t *new_mask,
                   const int64_t *match, const int64_t *commit,
                   const int64_t *leader_term, const uint32_t *log_off,
                   const int64_t *log_term, const int64_t *log_base,
                   const int64_t *voter_term, const int64_t *voted_for,
                   const int64_t *voter_last_log_term,
                   const int64_t *cand_term, const int64_t *cand_id,
                   int64_t *commit_out, int64_t *voter_term_out,
                   uint16_t *grant_mask, uint16_t *stepdown_mask,
                   uint8_t *elected, int64_t *max_term,
                   uint8_t *status_out, uint64_t *elected_bits,
                   uint16_t *commit_lo16, const int64_t *summary_base,
                   uint32_t *summary_escapes);

/* Wide forms of rq_follower_clamp, rq_append_check and rq_append_check_dev
 * (same rules; int64 planes, 32-bit log offsets). */
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
int rq_append_check64_dev(rq_engine *e, void *stream, uint32_t G, uint32_t n, uint32_t ld,
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

#ifdef __cplusplus
}
#endif
#endif /* RQ_H */
