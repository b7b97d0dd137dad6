/*
 * raft_oracle.h -- CPU restatement of tawawhite/raft (raft.go) quorum logic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing shipped in raft_amd/ may link, load or
 * call this library; it exists so that tests, the smoke check and the
 * cpu_baseline leg of bench.py have an independent checker / CPU timing leg.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * (its tree is raft.go + .gitignore) and no Go toolchain exists in this image,
 * so this restatement is pinned only by known-answer vectors hand-derived from
 * the cited raft.go lines (tests/golden/) and by brute-force properties.
 *
 * Two layers:
 *   rf_*   "reference-shaped": array-of-structs Node/Entry with 64-bit ints
 *          (Go int on amd64), one call per RPC like raft.go.  A Go runtime
 *          panic (index out of range) is reported as a non-zero return code.
 *   orc_*  batch structure-of-arrays entry points with the same argument
 *          layout as the rq_* C-ABI (int32 planes field[p*ld + g]); each one
 *          is implemented by building rf_node objects and calling rf_*.
 */
#ifndef RAFT_ORACLE_H
#define RAFT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* raft.go:7-13 -- iota enum, Leader is the zero value. */
enum { RF_LEADER = 0, RF_FOLLOWER = 1, RF_CANDIDATE = 2 };

/* Go runtime panics surfaced as return codes. */
enum { RF_OK = 0, RF_PANIC_INDEX = 1 };

/* raft.go:71-75 */
typedef struct rf_entry {
    const char *command;
    int64_t index;
    int64_t term_num;
} rf_entry;

/* raft.go:15-69 */
typedef struct rf_node {
    int64_t id;                          /* :17 */
    int64_t node_type;                   /* :20 */
    void (*state_machine)(const char *); /* :23 */
    struct rf_node **peers;              /* :26, includes self (:94-97) */
    int64_t n_peers;
    int64_t current_term;                /* :35 */
    int64_t voted_for;                   /* :39, -1 = null (:86) */
    rf_entry *log;                       /* :44 */
    int64_t log_len, log_cap;
    int64_t commit_index;                /* :51 */
    int64_t last_applied;                /* :56 */
    int64_t *next_index;                 /* :63, NULL unless leader */
    int64_t *match_index;                /* :68, NULL unless leader */
} rf_node;

/* ---- reference-shaped layer (each function cites the lines it follows) ---- */
rf_node *rf_new_node(int64_t id, rf_node **peers, int64_t n_peers,
                     void (*sm)(const char *));          /* raft.go:77-99  */
void rf_free_cluster(rf_node *any_member);
void rf_become_leader(rf_node *n);                       /* raft.go:101-118 */
void rf_become_follower(rf_node *n);                     /* raft.go:120-124 */
void rf_become_candidate(rf_node *n);                    /* raft.go:126-130 */
int rf_append_entries_rpc(rf_node *n, int64_t term, int64_t leader_id,
                          int64_t prev_log_index, int64_t prev_log_term,
                          const rf_entry *new_entries, int64_t n_new,
                          int64_t leader_commit, int64_t *term_result,
                          int *success);                 /* raft.go:132-179 */
int rf_request_vote_rpc(rf_node *n, int64_t term, int64_t candidate_id,
                        int64_t last_log_index, int64_t last_log_term,
                        int64_t *term_result, int *vote_granted); /* :181-210 */
void rf_test_to_abdicate_leadership(rf_node *n, int64_t term);    /* :212-223 */
int64_t rf_min_int(int64_t a, int64_t b);                /* raft.go:226-231 */
int rf_last_entry(const rf_entry *ents, int64_t len, rf_entry *out); /* :234-236 */
void rf_log_append(rf_node *n, const rf_entry *e, int64_t count);

/* Build-defined rules the reference holds state for but never computes
 * (state: raft.go:48-51, 65-68, 114-117; rule: Raft paper Fig. 2). */
int64_t rf_quorum_match_index(const int64_t *match, int64_t n);
int rf_leader_advance_commit(rf_node *leader);
void rf_leader_on_append_reply(rf_node *leader, int64_t slot, int success,
                               int reached_log_check, int64_t prev_log_index,
                               int64_t n_new);
int rf_candidate_tally(rf_node **voters, int64_t n, int64_t cand_term,
                       int64_t cand_id, uint32_t *grant_mask,
                       uint32_t *stepdown_mask, int64_t *max_term,
                       int *elected, uint32_t *panic_mask);

/* ---- batch layer: same layout as the rq_* C-ABI ---- */
#define ORC_ST_LOG_OOB   0x01u  /* commit: N outside the leader's log      */
#define ORC_ST_EMPTY_LOG 0x02u  /* vote: lastEntry() on an empty voter log */
#define ORC_LOG_EMPTY    INT32_MIN

int orc_commit_advance(uint32_t G, uint32_t n, uint32_t ld,
                       const int32_t *match, const int32_t *commit,
                       const int32_t *leader_term, const uint32_t *log_off,
                       const int32_t *log_term, int32_t *commit_out,
                       uint8_t *status_out);
int orc_vote_tally(uint32_t G, uint32_t n, uint32_t ld,
                   const int32_t *voter_term, const int32_t *voted_for,
                   const int32_t *voter_last_log_term, const int32_t *cand_term,
                   const int32_t *cand_id, int32_t *voter_term_out,
                   uint16_t *grant_mask, uint16_t *stepdown_mask,
                   uint8_t *elected, int32_t *max_term, uint8_t *status_out);
int orc_commit_advance_joint(uint32_t G, uint32_t m, uint32_t ld,
                             const uint16_t *old_mask, const uint16_t *new_mask,
                             const int32_t *match, const int32_t *commit,
                             const int32_t *leader_term, const uint32_t *log_off,
                             const int32_t *log_term, int32_t *commit_out,
                             uint8_t *status_out);
int orc_vote_tally_joint(uint32_t G, uint32_t m, uint32_t ld,
                         const uint16_t *old_mask, const uint16_t *new_mask,
                         const int32_t *voter_term, const int32_t *voted_for,
                         const int32_t *voter_last_log_term,
                         const int32_t *cand_term, const int32_t *cand_id,
                         int32_t *voter_term_out, uint16_t *grant_mask,
                         uint16_t *stepdown_mask, uint8_t *elected,
                         int32_t *max_term, uint8_t *status_out);
/* Wide layout: one call = commit (what & 1) and/or vote (what & 2) over int64
 * planes; masks both NULL = every group has the m slots as one configuration;
 * log_base NULL = 0 (window slot i of a group holds log[log_base + i]); an
 * empty voter log is INT64_MIN. */
int orc_round64(int what, uint32_t G, uint32_t m, uint32_t ld,
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
int orc_follower_clamp(uint64_t count, const int32_t *leader_commit,
                       const int32_t *commit, const int32_t *last_new_index,
                       int32_t *commit_out);             /* raft.go:172-176 */
int orc_append_check(uint32_t G, uint32_t n, uint32_t ld,
                     const uint16_t *active_mask,
                     const int32_t *follower_term, const int32_t *leader_term,
                     const int32_t *prev_log_index, const int32_t *prev_log_term,
                     const int32_t *n_new,
                     const uint32_t *flog_off, const int32_t *flog_term,
                     int32_t *follower_term_out, uint16_t *success_mask,
                     uint16_t *stepdown_mask, int32_t *match_index_io,
                     int32_t *next_index_io, uint8_t *status_out); /* :142-153 */

/* :172-176 for the succeeding slots of an AppendEntries batch (in place) */
int orc_append_commit(uint32_t G, uint32_t n, uint32_t ld, const uint16_t *success_mask,
                      const int32_t *leader_commit, const int32_t *last_new_index,
                      const int32_t *n_new, int32_t *follower_commit_io, uint8_t *status_io);
int orc_append_commit64(uint32_t G, uint32_t n, uint32_t ld, const uint16_t *success_mask,
                        const int64_t *leader_commit, const int64_t *last_new_index,
                        const int64_t *n_new, int64_t *follower_commit_io, uint8_t *status_io);

/* wide-layout forms of the two functions above */
int orc_follower_clamp64(uint64_t count, const int64_t *leader_commit,
                         const int64_t *commit, const int64_t *last_new_index,
                         int64_t *commit_out);
int orc_append_check64(uint32_t G, uint32_t n, uint32_t ld,
                       const uint16_t *active_mask,
                       const int64_t *follower_term, const int64_t *leader_term,
                       const int64_t *prev_log_index, const int64_t *prev_log_term,
                       const int64_t *n_new,
                       const uint32_t *flog_off, const int64_t *flog_term,
                       int64_t *follower_term_out, uint16_t *success_mask,
                       uint16_t *stepdown_mask, int64_t *match_index_io,
                       int64_t *next_index_io, uint8_t *status_out);

/* ---- CPU baseline: reference-shaped heap objects, timed ---- */
typedef struct orc_cluster_set orc_cluster_set;
orc_cluster_set *orc_bench_build(uint32_t G, uint32_t n, uint32_t ld,
        const int32_t *match, const int32_t *commit, const int32_t *leader_term,
        const uint32_t *log_off, const int32_t *log_term,
        const int32_t *voter_term, const int32_t *voted_for,
        const int32_t *voter_last_log_term, const int32_t *cand_term,
        const int32_t *cand_id);
/* Runs `passes` fused rounds over all groups on `threads` threads; state is
 * reset (untimed) between passes.  Returns seconds of timed work, <0 on error.
 * Writes a checksum of the decisions so the work cannot be optimised away. */
double orc_bench_run(orc_cluster_set *s, int passes, int threads,
                     uint64_t *checksum);
void orc_bench_free(orc_cluster_set *s);

#ifdef __cplusplus
}
#endif
#endif
