/*
 * raft_oracle.c -- CPU restatement of tawawhite/raft raft.go (see header).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no reference tests/fixtures exist).
 *
 * Every function that follows reference text cites the raft.go lines it
 * restates, including the reference's deviations from the Raft paper, which
 * are kept on purpose: they are the parity contract.
 */
#define _POSIX_C_SOURCE 200809L
#include "raft_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* reference-shaped layer                                                    */
/* ------------------------------------------------------------------------ */

/* raft.go:226-231 */
int64_t rf_min_int(int64_t a, int64_t b) {
    if (a < b) return a;
    return b;
}

/* raft.go:234-236: ents[len(ents)-1]; an empty slice panics in Go. */
int rf_last_entry(const rf_entry *ents, int64_t len, rf_entry *out) {
    if (len <= 0) return RF_PANIC_INDEX;
    *out = ents[len - 1];
    return RF_OK;
}

void rf_log_append(rf_node *n, const rf_entry *e, int64_t count) {
    if (count <= 0) return;
    if (n->log_len + count > n->log_cap) {
        int64_t cap = n->log_cap ? n->log_cap : 4;
        while (cap < n->log_len + count) cap *= 2;
        n->log = (rf_entry *)realloc(n->log, (size_t)cap * sizeof(rf_entry));
        n->log_cap = cap;
    }
    memcpy(n->log + n->log_len, e, (size_t)count * sizeof(rf_entry));
    n->log_len += count;
}

/* raft.go:77-99.  `peers` is the list of already-created members; the new
 * node is appended and every member is re-pointed at the grown list, so the
 * cluster size len(peers) includes self (:94-97). */
rf_node *rf_new_node(int64_t id, rf_node **peers, int64_t n_peers,
                     void (*sm)(const char *)) {
    rf_node *t = (rf_node *)calloc(1, sizeof(rf_node));
    t->id = id;                 /* :80 */
    t->state_machine = sm;      /* :81 */
    t->node_type = RF_FOLLOWER; /* :82 */
    t->current_term = 0;        /* :85 */
    t->voted_for = -1;          /* :86 */
    t->log = NULL;              /* :87 empty log */
    t->log_len = t->log_cap = 0;
    t->commit_index = 0;        /* :88 */
    t->last_applied = 0;        /* :89 */
    rf_node **grown = (rf_node **)malloc((size_t)(n_peers + 1) * sizeof(rf_node *));
    for (int64_t i = 0; i < n_peers; i++) grown[i] = peers[i];
    grown[n_peers] = t;         /* :94 */
    rf_node **old_shared = n_peers > 0 ? peers[0]->peers : NULL;
    for (int64_t i = 0; i <= n_peers; i++) { /* :95-97 */
        grown[i]->peers = grown;
        grown[i]->n_peers = n_peers + 1;
    }
    /* Go's GC would reclaim the previous shared slice; free it unless the
     * caller handed us that very array as `peers`. */
    if (old_shared && old_shared != peers) free(old_shared);
    return t;
}

void rf_free_cluster(rf_node *any) {
    if (!any) return;
    rf_node **ps = any->peers;
    int64_t n = any->n_peers;
    for (int64_t i = 0; i < n; i++) {
        rf_node *x = ps[i];
        free(x->log);
        free(x->next_index);
        free(x->match_index);
        free(x);
    }
    free(ps);
}

/* raft.go:101-118 */
void rf_become_leader(rf_node *t) {
    t->node_type = RF_LEADER; /* :102 */
    free(t->next_index);
    free(t->match_index);
    t->next_index = (int64_t *)malloc((size_t)t->n_peers * sizeof(int64_t));
    for (int64_t i = 0; i < t->n_peers; i++)
        t->next_index[i] = t->log_len + 1; /* :108 */
    t->match_index = (int64_t *)malloc((size_t)t->n_peers * sizeof(int64_t));
    for (int64_t i = 0; i < t->n_peers; i++)
        t->match_index[i] = 0; /* :116, self slot included */
}

/* raft.go:120-124 */
void rf_become_follower(rf_node *t) {
    t->node_type = RF_FOLLOWER;
    free(t->next_index);  t->next_index = NULL;
    free(t->match_index); t->match_index = NULL;
}

/* raft.go:126-130: no term bump, no self-vote. */
void rf_become_candidate(rf_node *t) {
    t->node_type = RF_CANDIDATE;
    free(t->next_index);  t->next_index = NULL;
    free(t->match_index); t->match_index = NULL;
}

/* raft.go:212-223: votedFor is not touched. */
void rf_test_to_abdicate_leadership(rf_node *t, int64_t term) {
    if (term > t->current_term) { /* :219 */
        t->current_term = term;   /* :220 */
        t->node_type = RF_FOLLOWER; /* :221 */
    }
}

/* raft.go:181-210.  last_log_index / last_log_term are accepted and never
 * read, exactly as in the reference; the up-to-date test compares the voter's
 * last log term with the candidate's *term* (:204); votedFor is never
 * written. */
int rf_request_vote_rpc(rf_node *t, int64_t term, int64_t candidate_id,
                        int64_t last_log_index, int64_t last_log_term,
                        int64_t *term_result, int *vote_granted) {
    (void)last_log_index;
    (void)last_log_term;
    rf_test_to_abdicate_leadership(t, term); /* :187 */
    if (term < t->current_term) {            /* :190-192 */
        *term_result = t->current_term;
        *vote_granted = 0;
        return RF_OK;
    }
    int not_yet_voted = t->voted_for == -1;            /* :202 */
    int voted_same_before = t->voted_for == candidate_id; /* :203 */
    rf_entry last;
    if (rf_last_entry(t->log, t->log_len, &last) != RF_OK) { /* :204 panics */
        *term_result = t->current_term;
        *vote_granted = 0;
        return RF_PANIC_INDEX;
    }
    int requester_more_up_to_date = last.term_num <= term; /* :204 */
    if ((not_yet_voted || voted_same_before) && requester_more_up_to_date) {
        *term_result = t->current_term; /* :206 */
        *vote_granted = 1;
        return RF_OK;
    }
    *term_result = t->current_term; /* :209 */
    *vote_granted = 0;
    return RF_OK;
}

/* raft.go:132-179.  Out-of-range slice accesses return RF_PANIC_INDEX at the
 * point Go would panic, with the side effects made so far left in place
 * (Go does not roll back either). */
int rf_append_entries_rpc(rf_node *t, int64_t term, int64_t leader_id,
                          int64_t prev_log_index, int64_t prev_log_term,
                          const rf_entry *new_entries, int64_t n_new,
                          int64_t leader_commit, int64_t *term_result,
                          int *success) {
    (void)leader_id;
    *success = 0;
    rf_test_to_abdicate_leadership(t, term); /* :142 */
    *term_result = t->current_term;
    if (term < t->current_term) return RF_OK; /* :145-147 */
    if (prev_log_index < 0 || prev_log_index >= t->log_len) /* :151 */
        return RF_PANIC_INDEX;
    if (t->log[prev_log_index].term_num != prev_log_term) return RF_OK; /* :151-153 */
    for (int64_t i = 0; i < n_new; i++) { /* :158-167 */
        int index_is_in_range = t->log_len <= new_entries[i].index; /* :159 (sic) */
        if (index_is_in_range) {
            /* :161 evaluates this.log[newEntry.Index] with
             * Index >= len(log): always an index-out-of-range panic, so the
             * comparison and the truncation at :163 are unreachable. */
            if (new_entries[i].index < 0 || new_entries[i].index >= t->log_len)
                return RF_PANIC_INDEX;
        }
    }
    rf_log_append(t, new_entries, n_new); /* :170 unconditional */
    if (leader_commit > t->commit_index) { /* :174 */
        rf_entry last;
        if (rf_last_entry(new_entries, n_new, &last) != RF_OK) /* :175 */
            return RF_PANIC_INDEX;
        t->commit_index = rf_min_int(leader_commit, last.index);
    }
    *term_result = t->current_term; /* :178 */
    *success = 1;
    return RF_OK;
}

/* ---- build-defined rules (absent from raft.go) ---- */

/* Largest N replicated on a majority of the n slots (self included,
 * raft.go:114-117): ascending-sorted match[(n-1)/2], i.e. the
 * (floor(n/2)+1)-th largest value. */
int64_t rf_quorum_match_index(const int64_t *match, int64_t n) {
    int64_t tmp[64];
    if (n <= 0 || n > 64) return 0;
    for (int64_t i = 0; i < n; i++) { /* insertion sort, ascending */
        int64_t v = match[i], j = i;
        while (j > 0 && tmp[j - 1] > v) { tmp[j] = tmp[j - 1]; j--; }
        tmp[j] = v;
    }
    return tmp[(n - 1) / 2];
}

/* Raft paper Fig. 2 leader rule over the state of raft.go:48-51/65-68:
 * if N > commitIndex and log[N].term == currentTerm then commitIndex = N.
 * log[N] is direct slot indexing, the convention of raft.go:151; reading it
 * out of range is an index panic in Go and is evaluated only when
 * N > commitIndex (short-circuit &&). */
int rf_leader_advance_commit(rf_node *l) {
    int64_t N = rf_quorum_match_index(l->match_index, l->n_peers);
    if (N > l->commit_index) {
        if (N < 0 || N >= l->log_len) return RF_PANIC_INDEX;
        if (l->log[N].term_num == l->current_term) l->commit_index = N;
    }
    return RF_OK;
}

/* Leader bookkeeping for one AppendEntries reply (Raft paper Fig. 2; the
 * reference only declares nextIndex/matchIndex, raft.go:60-68).  On success
 * the follower holds everything up to prevLogIndex + n_new.  On a rejection
 * that got as far as the log check (term was current) the leader backs
 * nextIndex off by one, never below 1 (indices start at 1, raft.go:43, :108).
 * A rejection for a stale term changes nothing: the leader is about to step
 * down on the reply's term. */
void rf_leader_on_append_reply(rf_node *l, int64_t slot, int success,
                               int reached_log_check, int64_t prev_log_index,
                               int64_t n_new) {
    if (success) {
        l->match_index[slot] = prev_log_index + n_new;
        l->next_index[slot] = l->match_index[slot] + 1;
    } else if (reached_log_check) {
        if (l->next_index[slot] > 1) l->next_index[slot] -= 1;
    }
}

/* Candidate side: one RequestVoteRPC per slot, the candidate's own slot
 * included; elected iff a strict majority granted and no reply carried a
 * term above the candidate's. */
int rf_candidate_tally(rf_node **voters, int64_t n, int64_t cand_term,
                       int64_t cand_id, uint32_t *grant_mask,
                       uint32_t *stepdown_mask, int64_t *max_term,
                       int *elected, uint32_t *panic_mask) {
    uint32_t gm = 0, sm = 0, pm = 0;
    int64_t mt = INT64_MIN, votes = 0;
    for (int64_t p = 0; p < n; p++) {
        int64_t before = voters[p]->current_term;
        int64_t tr; int granted;
        int rc = rf_request_vote_rpc(voters[p], cand_term, cand_id, 0, 0, &tr, &granted);
        if (rc != RF_OK) pm |= 1u << p;
        if (cand_term > before) sm |= 1u << p; /* :219-221 fired */
        if (granted) { gm |= 1u << p; votes++; }
        if (tr > mt) mt = tr;
    }
    *grant_mask = gm; *stepdown_mask = sm; *max_term = mt; *panic_mask = pm;
    *elected = (2 * votes > n) && (mt == cand_term);
    return RF_OK;
}

/* ------------------------------------------------------------------------ */
/* batch layer                                                               */
/* ------------------------------------------------------------------------ */

#define ORC_MAX_SLOTS 16

static int popcount16(uint32_t x) { int c = 0; while (x) { c += x & 1; x >>= 1; } return c; }

/* One group's commit decision through an rf_node leader. */
static int32_t commit_one(const int64_t *match, int64_t n, int64_t commit,
                          int64_t leader_term, const int32_t *lt, int64_t len,
                          uint8_t *status) {
    rf_entry stack_log[64];
    rf_entry *log = len <= 64 ? stack_log : (rf_entry *)malloc((size_t)len * sizeof(rf_entry));
    for (int64_t i = 0; i < len; i++) { log[i].command = ""; log[i].index = i; log[i].term_num = lt[i]; }
    rf_node L; memset(&L, 0, sizeof L);
    L.node_type = RF_LEADER; L.n_peers = n; L.current_term = leader_term;
    L.commit_index = commit; L.log = log; L.log_len = len;
    L.match_index = (int64_t *)match;
    if (rf_leader_advance_commit(&L) != RF_OK) *status |= ORC_ST_LOG_OOB;
    if (log != stack_log) free(log);
    return (int32_t)L.commit_index;
}

int orc_commit_advance(uint32_t G, uint32_t n, uint32_t ld,
                       const int32_t *match, const int32_t *commit,
                       const int32_t *leader_term, const uint32_t *log_off,
                       const int32_t *log_term, int32_t *commit_out,
                       uint8_t *status_out) {
    if (n < 1 || n > ORC_MAX_SLOTS || ld < G) return -1;
    for (uint32_t g = 0; g < G; g++) {
        int64_t m[ORC_MAX_SLOTS];
        for (uint32_t p = 0; p < n; p++) m[p] = match[(size_t)p * ld + g];
        uint8_t st = 0;
        int32_t c = commit_one(m, n, commit[g], leader_term[g],
                               log_term + log_off[g],
                               (int64_t)log_off[g + 1] - (int64_t)log_off[g], &st);
        commit_out[g] = c;
        if (status_out) status_out[g] = st;
    }
    return 0;
}

/* Build a voter rf_node on the stack from one SoA slot. */
static void voter_init(rf_node *v, rf_entry *slot_log, int32_t term,
                       int32_t voted_for, int32_t last_log_term) {
    memset(v, 0, sizeof *v);
    v->node_type = RF_CANDIDATE; /* any non-Follower role: lets :221 show */
    v->current_term = term;
    v->voted_for = voted_for;
    if (last_log_term == ORC_LOG_EMPTY) { v->log = NULL; v->log_len = 0; }
    else { slot_log->command = ""; slot_log->index = 0; slot_log->term_num = last_log_term;
           v->log = slot_log; v->log_len = 1; }
}

int orc_vote_tally(uint32_t G, uint32_t n, uint32_t ld,
                   const int32_t *voter_term, const int32_t *voted_for,
                   const int32_t *voter_last_log_term, const int32_t *cand_term,
                   const int32_t *cand_id, int32_t *voter_term_out,
                   uint16_t *grant_mask, uint16_t *stepdown_mask,
                   uint8_t *elected, int32_t *max_term, uint8_t *status_out) {
    if (n < 1 || n > ORC_MAX_SLOTS || ld < G) return -1;
    for (uint32_t g = 0; g < G; g++) {
        rf_node nodes[ORC_MAX_SLOTS]; rf_entry logs[ORC_MAX_SLOTS];
        rf_node *vp[ORC_MAX_SLOTS];
        for (uint32_t p = 0; p < n; p++) {
            size_t i = (size_t)p * ld + g;
            voter_init(&nodes[p], &logs[p], voter_term[i], voted_for[i], voter_last_log_term[i]);
            vp[p] = &nodes[p];
        }
        uint32_t gm, sm, pm; int64_t mt; int el;
        rf_candidate_tally(vp, n, cand_term[g], cand_id[g], &gm, &sm, &mt, &el, &pm);
        for (uint32_t p = 0; p < n; p++)
            voter_term_out[(size_t)p * ld + g] = (int32_t)nodes[p].current_term;
        grant_mask[g] = (uint16_t)gm; stepdown_mask[g] = (uint16_t)sm;
        elected[g] = (uint8_t)el; max_term[g] = (int32_t)mt;
        if (status_out) status_out[g] = pm ? ORC_ST_EMPTY_LOG : 0;
    }
    return 0;
}

/* Joint consensus (build-defined): slots 0..m-1, two member masks.  The
 * commit index must be replicated on a majority of EACH configuration; an
 * election needs a strict majority of grants in EACH configuration.  A simple
 * group is old_mask == new_mask.  Slots outside old|new are not evaluated and
 * their voter_term_out is not written. */
static int64_t subset_quorum(const int64_t *match, uint32_t m, uint32_t mask) {
    int64_t sub[ORC_MAX_SLOTS]; int64_t k = 0;
    for (uint32_t p = 0; p < m; p++) if (mask >> p & 1) sub[k++] = match[p];
    return rf_quorum_match_index(sub, k);
}

int orc_commit_advance_joint(uint32_t G, uint32_t m, uint32_t ld,
                             const uint16_t *old_mask, const uint16_t *new_mask,
                             const int32_t *match, const int32_t *commit,
                             const int32_t *leader_term, const uint32_t *log_off,
                             const int32_t *log_term, int32_t *commit_out,
                             uint8_t *status_out) {
    if (m < 1 || m > ORC_MAX_SLOTS || ld < G) return -1;
    uint32_t lim = m >= 16 ? 0xFFFFu : ((1u << m) - 1u);
    for (uint32_t g = 0; g < G; g++) {
        uint32_t om = old_mask[g] & lim, nm = new_mask[g] & lim;
        if (!om || !nm) return -2;
        int64_t mt[ORC_MAX_SLOTS];
        for (uint32_t p = 0; p < m; p++)
            mt[p] = ((om | nm) >> p & 1) ? match[(size_t)p * ld + g] : 0;
        int64_t N = rf_min_int(subset_quorum(mt, m, om), subset_quorum(mt, m, nm));
        /* feed N through the same leader rule as a 1-slot "cluster" */
        uint8_t st = 0;
        int32_t c = commit_one(&N, 1, commit[g], leader_term[g],
                               log_term + log_off[g],
                               (int64_t)log_off[g + 1] - (int64_t)log_off[g], &st);
        commit_out[g] = c;
        if (status_out) status_out[g] = st;
    }
    return 0;
}

int orc_vote_tally_joint(uint32_t G, uint32_t m, uint32_t ld,
                         const uint16_t *old_mask, const uint16_t *new_mask,
                         const int32_t *voter_term, const int32_t *voted_for,
                         const int32_t *voter_last_log_term,
                         const int32_t *cand_term, const int32_t *cand_id,
                         int32_t *voter_term_out, uint16_t *grant_mask,
                         uint16_t *stepdown_mask, uint8_t *elected,
                         int32_t *max_term, uint8_t *status_out) {
    if (m < 1 || m > ORC_MAX_SLOTS || ld < G) return -1;
    uint32_t lim = m >= 16 ? 0xFFFFu : ((1u << m) - 1u);
    for (uint32_t g = 0; g < G; g++) {
        uint32_t om = old_mask[g] & lim, nm = new_mask[g] & lim;
        if (!om || !nm) return -2;
        uint32_t mem = om | nm, gm = 0, sm = 0, pm = 0;
        int64_t mt = INT64_MIN;
        for (uint32_t p = 0; p < m; p++) {
            if (!(mem >> p & 1)) continue;
            size_t i = (size_t)p * ld + g;
            rf_node v; rf_entry e;
            voter_init(&v, &e, voter_term[i], voted_for[i], voter_last_log_term[i]);
            int64_t before = v.current_term, tr; int granted;
            if (rf_request_vote_rpc(&v, cand_term[g], cand_id[g], 0, 0, &tr, &granted) != RF_OK)
                pm |= 1u << p;
            if (cand_term[g] > before) sm |= 1u << p;
            if (granted) gm |= 1u << p;
            if (tr > mt) mt = tr;
            voter_term_out[i] = (int32_t)v.current_term;
        }
        int el = 2 * popcount16(gm & om) > popcount16(om) &&
                 2 * popcount16(gm & nm) > popcount16(nm) && mt == cand_term[g];
        grant_mask[g] = (uint16_t)gm; stepdown_mask[g] = (uint16_t)sm;
        elected[g] = (uint8_t)el; max_term[g] = (int32_t)mt;
        if (status_out) status_out[g] = pm ? ORC_ST_EMPTY_LOG : 0;
    }
    return 0;
}

/* ---- wide layout: the same decisions over int64 planes ------------------- */

/* Leader rule with a windowed log: window slot i holds log[base + i].  The
 * reference-shaped leader is evaluated in the window's own coordinates, as a
 * one-slot cluster whose matchIndex is N - base and whose commitIndex is just
 * below it (N > commitIndex is decided here, on the unshifted values, so no
 * shifted quantity can overflow). */
static int64_t commit_one64(int64_t N, int64_t commit, int64_t leader_term,
                            const int64_t *lt, int64_t len, int64_t base,
                            uint8_t *status) {
    if (!(N > commit)) return commit;
    uint64_t wi = (uint64_t)N - (uint64_t)base;
    if (N < base || wi >= (uint64_t)len) { *status |= ORC_ST_LOG_OOB; return commit; }
    rf_entry ent; ent.command = ""; ent.index = N; ent.term_num = lt[wi];
    int64_t one = 0;
    rf_node L; memset(&L, 0, sizeof L);
    L.node_type = RF_LEADER; L.n_peers = 1; L.current_term = leader_term;
    L.commit_index = -1; L.log = &ent; L.log_len = 1; L.match_index = &one;
    if (rf_leader_advance_commit(&L) != RF_OK) { *status |= ORC_ST_LOG_OOB; return commit; }
    return L.commit_index == 0 ? N : commit;
}

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
                uint8_t *elected, int64_t *max_term, uint8_t *status_out) {
    if (m < 1 || m > ORC_MAX_SLOTS || ld < G) return -1;
    if ((old_mask == NULL) != (new_mask == NULL)) return -1;
    uint32_t lim = m >= 16 ? 0xFFFFu : ((1u << m) - 1u);
    for (uint32_t g = 0; g < G; g++) {
        uint32_t om = old_mask ? (old_mask[g] & lim) : lim;
        uint32_t nm = new_mask ? (new_mask[g] & lim) : lim;
        if (!om || !nm) return -2;
        uint32_t mem = om | nm;
        uint8_t st = 0;
        if (what & 1) {
            int64_t mt[ORC_MAX_SLOTS];
            for (uint32_t p = 0; p < m; p++)
                mt[p] = (mem >> p & 1) ? match[(size_t)p * ld + g] : 0;
            int64_t N = rf_min_int(subset_quorum(mt, m, om), subset_quorum(mt, m, nm));
            commit_out[g] = commit_one64(N, commit[g], leader_term[g], log_term + log_off[g],
                                         (int64_t)log_off[g + 1] - (int64_t)log_off[g],
                                         log_base ? log_base[g] : 0, &st);
        }
        if (what & 2) {
            uint32_t gm = 0, sm = 0, pm = 0;
            int64_t mx = INT64_MIN;
            for (uint32_t p = 0; p < m; p++) {
                if (!(mem >> p & 1)) continue;
                size_t i = (size_t)p * ld + g;
                rf_node v; rf_entry e;
                memset(&v, 0, sizeof v);
                v.node_type = RF_CANDIDATE;
                v.current_term = voter_term[i];
                v.voted_for = voted_for[i];
                if (voter_last_log_term[i] != INT64_MIN) {
                    e.command = ""; e.index = 0; e.term_num = voter_last_log_term[i];
                    v.log = &e; v.log_len = 1;
                }
                int64_t before = v.current_term, tr; int granted;
                if (rf_request_vote_rpc(&v, cand_term[g], cand_id[g], 0, 0, &tr, &granted) != RF_OK)
                    pm |= 1u << p;
                if (cand_term[g] > before) sm |= 1u << p;
                if (granted) gm |= 1u << p;
                if (tr > mx) mx = tr;
                voter_term_out[i] = v.current_term;
            }
            int el = 2 * popcount16(gm & om) > popcount16(om) &&
                     2 * popcount16(gm & nm) > popcount16(nm) && mx == cand_term[g];
            grant_mask[g] = (uint16_t)gm; stepdown_mask[g] = (uint16_t)sm;
            elected[g] = (uint8_t)el; max_term[g] = mx;
            if (pm) st |= ORC_ST_EMPTY_LOG;
        }
        if (status_out) status_out[g] = st;
    }
    return 0;
}

/* raft.go:172-176 with minInt :226-231; may lower commitIndex when the last
 * new entry's index is below it, as the reference does. */
int orc_follower_clamp(uint64_t count, const int32_t *leader_commit,
                       const int32_t *commit, const int32_t *last_new_index,
                       int32_t *commit_out) {
    for (uint64_t i = 0; i < count; i++) {
        int64_t ci = commit[i];
        if (leader_commit[i] > ci) ci = rf_min_int(leader_commit[i], last_new_index[i]);
        commit_out[i] = (int32_t)ci;
    }
    return 0;
}

int orc_follower_clamp64(uint64_t count, const int64_t *leader_commit,
                         const int64_t *commit, const int64_t *last_new_index,
                         int64_t *commit_out) {
    for (uint64_t i = 0; i < count; i++) {
        int64_t ci = commit[i];
        if (leader_commit[i] > ci) ci = rf_min_int(leader_commit[i], last_new_index[i]);
        commit_out[i] = ci;
    }
    return 0;
}

/* raft.go:142-153 for every (group, follower) slot s = p*ld + g: abdication,
 * the term gate and the prevLogIndex/prevLogTerm consistency check (direct
 * slot indexing, :151), then the leader's bookkeeping for the reply.  Follower
 * logs are a CSR arena indexed by slot.  Slots outside active_mask (NULL = all
 * n) receive no RPC and are left untouched. */
/* One body, two element widths: T = int32_t (orc_append_check) and int64_t
 * (orc_append_check64). */
#define ORC_DEFINE_APPEND_CHECK(NAME, T) \
int NAME(uint32_t G, uint32_t n, uint32_t ld, \
                     const uint16_t *active_mask, \
                     const T *follower_term, const T *leader_term, \
                     const T *prev_log_index, const T *prev_log_term, \
                     const T *n_new, \
                     const uint32_t *flog_off, const T *flog_term, \
                     T *follower_term_out, uint16_t *success_mask, \
                     uint16_t *stepdown_mask, T *match_index_io, \
                     T *next_index_io, uint8_t *status_out) { \
    if (n < 1 || n > ORC_MAX_SLOTS || ld < G) return -1; \
    for (uint32_t g = 0; g < G; g++) { \
        uint32_t ok = 0, sd = 0; uint8_t st = 0; \
        uint32_t act = active_mask ? active_mask[g] : 0xFFFFu; \
        rf_node L; memset(&L, 0, sizeof L); \
        int64_t mi[ORC_MAX_SLOTS], ni[ORC_MAX_SLOTS]; \
        L.n_peers = n; L.match_index = mi; L.next_index = ni; \
        for (uint32_t p = 0; p < n; p++) { \
            size_t s = (size_t)p * ld + g; \
            mi[p] = match_index_io ? match_index_io[s] : 0; \
            ni[p] = next_index_io ? next_index_io[s] : 1; \
        } \
        for (uint32_t p = 0; p < n; p++) { \
            if (!(act >> p & 1)) continue; \
            size_t s = (size_t)p * ld + g; \
            int64_t len = (int64_t)flog_off[s + 1] - (int64_t)flog_off[s]; \
            rf_entry stack_log[64]; \
            rf_entry *log = len <= 64 ? stack_log : (rf_entry *)malloc((size_t)len * sizeof(rf_entry)); \
            for (int64_t i = 0; i < len; i++) { log[i].command = ""; log[i].index = i; log[i].term_num = flog_term[flog_off[s] + i]; } \
            rf_node f; memset(&f, 0, sizeof f); \
            f.node_type = RF_CANDIDATE; f.current_term = follower_term[s]; \
            f.voted_for = -1; f.log = log; f.log_len = len; f.log_cap = len; \
            int64_t before = f.current_term, tr; int success; \
            /* no new entries and leaderCommit 0 keep :158-176 inert */ \
            int rc = rf_append_entries_rpc(&f, leader_term[g], 0, prev_log_index[s], \
                                           prev_log_term[s], NULL, 0, 0, &tr, &success); \
            if (rc != RF_OK) st |= ORC_ST_LOG_OOB; \
            if (leader_term[g] > before) sd |= 1u << p; \
            if (success) ok |= 1u << p; \
            follower_term_out[s] = (T)f.current_term; \
            rf_leader_on_append_reply(&L, p, success, leader_term[g] >= before, \
                                      prev_log_index[s], n_new ? n_new[s] : 0); \
            if (f.log != stack_log) free(f.log); \
        } \
        for (uint32_t p = 0; p < n; p++) { \
            size_t s = (size_t)p * ld + g; \
            if (match_index_io) match_index_io[s] = (T)mi[p]; \
            if (next_index_io) next_index_io[s] = (T)ni[p]; \
        } \
        success_mask[g] = (uint16_t)ok; stepdown_mask[g] = (uint16_t)sd; \
        if (status_out) status_out[g] = st; \
    } \
    return 0; \
}

ORC_DEFINE_APPEND_CHECK(orc_append_check, int32_t)
ORC_DEFINE_APPEND_CHECK(orc_append_check64, int64_t)

/* raft.go:172-176 for the slots of an AppendEntries batch whose RPC succeeded
 * (success_mask as orc_append_check wrote it): a commitIndex below leaderCommit
 * becomes minInt(leaderCommit, lastEntry(newEntries).Index) (:175, :226-231).
 * last_new_index[s] is that entry's Index; n_new[s] == 0 (n_new given) means
 * newEntries is empty, so lastEntry indexes ents[-1] (:235): an index panic in
 * Go, reported as ORC_ST_EMPTY_LOG with commitIndex as it was.  Slots without
 * success returned at :146 or :152 and keep their commitIndex. */
#define ORC_DEFINE_APPEND_COMMIT(NAME, T) \
int NAME(uint32_t G, uint32_t n, uint32_t ld, const uint16_t *success_mask, \
         const T *leader_commit, const T *last_new_index, const T *n_new, \
         T *follower_commit_io, uint8_t *status_io) { \
    if (n < 1 || n > ORC_MAX_SLOTS || ld < G) return -1; \
    for (uint32_t g = 0; g < G; g++) { \
        for (uint32_t p = 0; p < n; p++) { \
            if (!(success_mask[g] >> p & 1)) continue; \
            size_t s = (size_t)p * ld + g; \
            if ((int64_t)leader_commit[g] > (int64_t)follower_commit_io[s]) { /* :174 */ \
                rf_entry e = { "", (int64_t)last_new_index[s], 0 }, last; \
                int64_t count = (n_new && n_new[s] == 0) ? 0 : 1; \
                if (rf_last_entry(&e, count, &last) != RF_OK) { /* :175 -> :235 */ \
                    if (status_io) status_io[g] |= ORC_ST_EMPTY_LOG; \
                    continue; \
                } \
                follower_commit_io[s] = (T)rf_min_int((int64_t)leader_commit[g], last.index); \
            } \
        } \
    } \
    return 0; \
}

ORC_DEFINE_APPEND_COMMIT(orc_append_commit, int32_t)
ORC_DEFINE_APPEND_COMMIT(orc_append_commit64, int64_t)

/* ------------------------------------------------------------------------ */
/* CPU baseline: heap-allocated reference-shaped clusters                    */
/* ------------------------------------------------------------------------ */

struct orc_cluster_set {
    uint32_t G, n;
    rf_node **nodes;      /* G*n voter nodes, group-major */
    rf_node **leaders;    /* G leader nodes */
    int64_t *cand_term, *cand_id;
    /* pristine state for the untimed reset between passes */
    int64_t *voter_term0; int64_t *commit0;
};

orc_cluster_set *orc_bench_build(uint32_t G, uint32_t n, uint32_t ld,
        const int32_t *match, const int32_t *commit, const int32_t *leader_term,
        const uint32_t *log_off, const int32_t *log_term,
        const int32_t *voter_term, const int32_t *voted_for,
        const int32_t *voter_last_log_term, const int32_t *cand_term,
        const int32_t *cand_id) {
    orc_cluster_set *s = (orc_cluster_set *)calloc(1, sizeof *s);
    s->G = G; s->n = n;
    s->nodes = (rf_node **)malloc((size_t)G * n * sizeof(rf_node *));
    s->leaders = (rf_node **)malloc((size_t)G * sizeof(rf_node *));
    s->cand_term = (int64_t *)malloc((size_t)G * sizeof(int64_t));
    s->cand_id = (int64_t *)malloc((size_t)G * sizeof(int64_t));
    s->voter_term0 = (int64_t *)malloc((size_t)G * n * sizeof(int64_t));
    s->commit0 = (int64_t *)malloc((size_t)G * sizeof(int64_t));
    for (uint32_t g = 0; g < G; g++) {
        for (uint32_t p = 0; p < n; p++) {
            size_t i = (size_t)p * ld + g;
            rf_node *v = (rf_node *)calloc(1, sizeof(rf_node));
            v->id = p; v->node_type = RF_FOLLOWER; v->n_peers = n;
            v->current_term = voter_term[i]; v->voted_for = voted_for[i];
            if (voter_last_log_term[i] != ORC_LOG_EMPTY) {
                rf_entry e = { "", 0, voter_last_log_term[i] };
                rf_log_append(v, &e, 1);
            }
            s->nodes[(size_t)g * n + p] = v;
            s->voter_term0[(size_t)g * n + p] = voter_term[i];
        }
        rf_node *L = (rf_node *)calloc(1, sizeof(rf_node));
        L->n_peers = n; L->current_term = leader_term[g];
        L->commit_index = commit[g]; L->voted_for = -1;
        uint32_t len = log_off[g + 1] - log_off[g];
        for (uint32_t i = 0; i < len; i++) {
            rf_entry e = { "", i, log_term[log_off[g] + i] };
            rf_log_append(L, &e, 1);
        }
        rf_become_leader(L);
        for (uint32_t p = 0; p < n; p++) L->match_index[p] = match[(size_t)p * ld + g];
        s->leaders[g] = L;
        s->commit0[g] = commit[g];
        s->cand_term[g] = cand_term[g]; s->cand_id[g] = cand_id[g];
    }
    return s;
}

typedef struct { orc_cluster_set *s; uint32_t g0, g1; uint64_t sum; } bench_job;

static void *bench_worker(void *arg) {
    bench_job *j = (bench_job *)arg;
    orc_cluster_set *s = j->s;
    uint64_t sum = 0;
    for (uint32_t g = j->g0; g < j->g1; g++) {
        uint32_t gm, sm, pm; int64_t mt; int el;
        rf_candidate_tally(s->nodes + (size_t)g * s->n, s->n, s->cand_term[g],
                           s->cand_id[g], &gm, &sm, &mt, &el, &pm);
        rf_leader_advance_commit(s->leaders[g]);
        sum += (uint64_t)s->leaders[g]->commit_index * 31u + gm + ((uint64_t)el << 20) + (uint64_t)mt;
    }
    j->sum = sum;
    return NULL;
}

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_bench_run(orc_cluster_set *s, int passes, int threads, uint64_t *checksum) {
    if (!s || passes < 1 || threads < 1 || threads > 1024) return -1.0;
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    bench_job *jobs = (bench_job *)malloc((size_t)threads * sizeof(bench_job));
    double total = 0.0; uint64_t sum = 0;
    for (int it = 0; it < passes; it++) {
        /* untimed reset to the pristine inputs */
        for (size_t i = 0; i < (size_t)s->G * s->n; i++) {
            s->nodes[i]->current_term = s->voter_term0[i];
            s->nodes[i]->node_type = RF_FOLLOWER;
        }
        for (uint32_t g = 0; g < s->G; g++) s->leaders[g]->commit_index = s->commit0[g];
        double t0 = now_s();
        for (int t = 0; t < threads; t++) {
            jobs[t].s = s; jobs[t].sum = 0;
            jobs[t].g0 = (uint32_t)((uint64_t)s->G * (uint64_t)t / (uint64_t)threads);
            jobs[t].g1 = (uint32_t)((uint64_t)s->G * (uint64_t)(t + 1) / (uint64_t)threads);
            if (threads == 1) bench_worker(&jobs[t]);
            else pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
        }
        if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        total += now_s() - t0;
        for (int t = 0; t < threads; t++) sum += jobs[t].sum;
    }
    free(th); free(jobs);
    if (checksum) *checksum = sum;
    return total;
}

void orc_bench_free(orc_cluster_set *s) {
    if (!s) return;
    for (size_t i = 0; i < (size_t)s->G * s->n; i++) { free(s->nodes[i]->log); free(s->nodes[i]); }
    for (uint32_t g = 0; g < s->G; g++) {
        free(s->leaders[g]->log); free(s->leaders[g]->next_index);
        free(s->leaders[g]->match_index); free(s->leaders[g]);
    }
    free(s->nodes); free(s->leaders); free(s->cand_term); free(s->cand_id);
    free(s->voter_term0); free(s->commit0); free(s);
}
