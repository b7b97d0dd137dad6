// raft_host_capi.cpp -- flat C hooks over raft_host.hpp so that the Python
// tests can drive the C++ host the way Go tests inside `package raft` would
// drive the Go host.  Not part of the product ABI (that is include/rq.h).
#include <cstring>
#include <ctime>
#include <memory>

#include "raft_host.hpp"

using namespace raft;

struct rh_cluster {
    uint32_t G = 0, n = 0;
    std::vector<Node *> nodes;  // group-major: nodes[g*n + p]
    std::unique_ptr<BatchedDriver> driver;
    std::unique_ptr<ResidentDriver> resident;
    RoundBuffers last;
    ~rh_cluster() {
        driver.reset();
        resident.reset();
        for (Node *x : nodes) delete x;
    }
};

extern "C" {

// G groups of n members each, created with NewNode like a Go test would.
rh_cluster *rh_cluster_new(uint32_t G, uint32_t n) {
    auto *c = new rh_cluster();
    c->G = G; c->n = n;
    std::vector<Group> groups(G);
    for (uint32_t g = 0; g < G; g++) {
        std::vector<Node *> peers;
        for (uint32_t p = 0; p < n; p++) {
            Node *x = NewNode((int64_t)p, peers, nullptr);
            peers.push_back(x);
            c->nodes.push_back(x);
        }
        groups[g].Members = peers;
    }
    c->driver.reset(new BatchedDriver(std::move(groups), n));
    return c;
}

// Groups of different sizes: group g has sizes[g] members (1..n) in its low slots;
// the remaining slots of the group hold no node.
rh_cluster *rh_cluster_new_sizes(uint32_t G, uint32_t n, const uint32_t *sizes) {
    auto *c = new rh_cluster();
    c->G = G; c->n = n;
    c->nodes.assign((size_t)G * n, nullptr);
    std::vector<Group> groups(G);
    for (uint32_t g = 0; g < G; g++) {
        std::vector<Node *> peers;
        for (uint32_t p = 0; p < sizes[g] && p < n; p++) {
            Node *x = NewNode((int64_t)p, peers, nullptr);
            peers.push_back(x);
            c->nodes[(size_t)g * n + p] = x;
        }
        groups[g].Members = peers;
    }
    c->driver.reset(new BatchedDriver(std::move(groups), n));
    return c;
}

void rh_cluster_free(rh_cluster *c) { delete c; }

static Node *at(rh_cluster *c, uint32_t g, uint32_t p) { return c->nodes[(size_t)g * c->n + p]; }

void rh_node_seed(rh_cluster *c, uint32_t g, uint32_t p, int64_t term, int64_t voted_for,
                  int64_t node_type, int64_t commit_index) {
    at(c, g, p)->SeedForTest(term, voted_for, (NodeType)node_type, commit_index);
}

void rh_node_seed_log(rh_cluster *c, uint32_t g, uint32_t p, const int64_t *terms, int64_t len) {
    at(c, g, p)->SeedLogForTest(std::vector<int64_t>(terms, terms + len));
}

// out[9]: id, nodeType, currentTerm, votedFor, logLen, commitIndex, lastApplied, nPeers, hasLeaderArrays
void rh_node_get(rh_cluster *c, uint32_t g, uint32_t p, int64_t *out) {
    Node::State s = at(c, g, p)->Inspect();
    const int64_t v[9] = {s.id, s.nodeType, s.currentTerm, s.votedFor, s.logLen, s.commitIndex,
                          s.lastApplied, s.nPeers, s.hasLeaderArrays ? 1 : 0};
    std::memcpy(out, v, sizeof v);
}

// next/match: n values each; returns 0 if the node holds no leader arrays.
int rh_node_get_leader_arrays(rh_cluster *c, uint32_t g, uint32_t p, int64_t *next, int64_t *match) {
    Node *x = at(c, g, p);
    if (!x->Inspect().hasLeaderArrays) return 0;
    for (uint32_t i = 0; i < c->n && i < x->NextIndex().size(); i++) { next[i] = x->NextIndex()[i]; match[i] = x->MatchIndex()[i]; }
    return 1;
}

// role: 0 = BecomeLeader, 1 = BecomeFollower, 2 = BecomeCandidate
void rh_node_become(rh_cluster *c, uint32_t g, uint32_t p, int role) {
    Node *x = at(c, g, p);
    if (role == 0) x->BecomeLeader(); else if (role == 1) x->BecomeFollower(); else x->BecomeCandidate();
}

void rh_leader_set_match(rh_cluster *c, uint32_t g, uint32_t p, const int64_t *match) {
    std::vector<int64_t> &m = at(c, g, p)->MatchIndexForTest();
    for (uint32_t i = 0; i < c->n && i < m.size(); i++) m[i] = match[i];
}

// Group roles for the driver; slot -1 = none.
void rh_group_roles(rh_cluster *c, uint32_t g, int leader_slot, int candidate_slot) {
    Group &grp = c->driver->Groups()[g];
    grp.Leader = leader_slot >= 0 ? at(c, g, (uint32_t)leader_slot) : nullptr;
    grp.Candidate = candidate_slot >= 0 ? at(c, g, (uint32_t)candidate_slot) : nullptr;
}

// which slot leads / asks for votes after Apply (-1 = none)
void rh_group_get_roles(rh_cluster *c, uint32_t g, int *leader_slot, int *candidate_slot) {
    Group &grp = c->driver->Groups()[g];
    *leader_slot = *candidate_slot = -1;
    for (uint32_t p = 0; p < c->n && p < grp.Members.size(); p++) {
        if (grp.Members[p] == grp.Leader) *leader_slot = (int)p;
        if (grp.Members[p] == grp.Candidate) *candidate_slot = (int)p;
    }
}

// RPCs: return 0, or 1 where Go would panic (index out of range).
int rh_request_vote(rh_cluster *c, uint32_t g, uint32_t p, int64_t term, int64_t cand, int64_t lli,
                    int64_t llt, int64_t *term_result, int *granted) {
    try {
        auto r = at(c, g, p)->RequestVoteRPC(term, cand, lli, llt);
        *term_result = r.first; *granted = r.second ? 1 : 0;
        return 0;
    } catch (const GoPanic &) { return 1; }
}

int rh_append_entries(rh_cluster *c, uint32_t g, uint32_t p, int64_t term, int64_t leader_id,
                      int64_t prev_index, int64_t prev_term, const int64_t *new_index,
                      const int64_t *new_term, int64_t n_new, int64_t leader_commit,
                      int64_t *term_result, int *success) {
    std::vector<Entry> ents;
    for (int64_t i = 0; i < n_new; i++) ents.push_back(Entry{"x", new_index[i], new_term[i]});
    try {
        auto r = at(c, g, p)->AppendEntriesRPC(term, leader_id, prev_index, prev_term, ents, leader_commit);
        *term_result = r.first; *success = r.second ? 1 : 0;
        return 0;
    } catch (const GoPanic &) { return 1; }
}

// Flatten into the cluster's buffer set; copies out what the caller asks for.
// Planes are [n][G].  Returns the driver code (0, -1 shape, -2 range).
int rh_driver_flatten(rh_cluster *c, int32_t *match, int32_t *commit, int32_t *leader_term,
                      uint32_t *log_off, int32_t *log_term, uint64_t log_term_cap,
                      int32_t *voter_term, int32_t *voted_for, int32_t *last_log_term,
                      int32_t *cand_term, int32_t *cand_id, uint64_t *log_terms_used) {
    RoundBuffers &b = c->last;
    int rc = c->driver->Flatten(b);
    if (rc != 0) return rc;
    const size_t plane = (size_t)b.slots * b.groups * 4, vec = (size_t)b.groups * 4;
    std::memcpy(match, b.match.data(), plane); std::memcpy(voter_term, b.voterTerm.data(), plane);
    std::memcpy(voted_for, b.votedFor.data(), plane); std::memcpy(last_log_term, b.lastLogTerm.data(), plane);
    std::memcpy(commit, b.commit.data(), vec); std::memcpy(leader_term, b.leaderTerm.data(), vec);
    std::memcpy(cand_term, b.candTerm.data(), vec); std::memcpy(cand_id, b.candId.data(), vec);
    std::memcpy(log_off, b.logOff.data(), ((size_t)b.groups + 1) * 4);
    *log_terms_used = b.logTerm.size();
    if (b.logTerm.size() > log_term_cap) return -3;
    std::memcpy(log_term, b.logTerm.data(), b.logTerm.size() * 4);
    return 0;
}

// member masks of the last rh_driver_flatten ([G]); returns 1 if the groups differ in size
int rh_driver_member_masks(rh_cluster *c, uint16_t *mask) {
    std::memcpy(mask, c->last.memberMask.data(), c->last.memberMask.size() * 2);
    return c->last.mixed ? 1 : 0;
}

// Apply outputs (layout of rq_fused_round) to the objects; uses the shapes of the last Flatten.
void rh_driver_apply(rh_cluster *c, const int32_t *commit_out, const int32_t *voter_term_out,
                     const uint16_t *grant_mask, const uint16_t *stepdown_mask,
                     const uint8_t *elected, const int32_t *max_term) {
    RoundBuffers &b = c->last;
    const size_t G = b.groups, plane = (size_t)b.slots * G;
    b.commitOut.assign(commit_out, commit_out + G);
    b.voterTermOut.assign(voter_term_out, voter_term_out + plane);
    b.grantMask.assign(grant_mask, grant_mask + G);
    b.stepdownMask.assign(stepdown_mask, stepdown_mask + G);
    b.elected.assign(elected, elected + G);
    b.maxTerm.assign(max_term, max_term + G);
    c->driver->Apply(b);
}

// Flatten -> rq_fused_round on the GPU -> Apply.  Returns an RQ_* code.
int rh_driver_step(rh_cluster *c, int device) { return c->driver->Step(device); }
// 1 if the last successful step had to use the wide (int64) layout
int rh_driver_last_step_wide(rh_cluster *c) { return c->driver->LastStepWide() ? 1 : 0; }

// Groups the last rh_driver_step flagged (conditions that are Go panics in the
// reference); fills up to cap entries, returns the total.
uint32_t rh_driver_flagged(rh_cluster *c, uint32_t *groups, uint8_t *status, uint32_t cap) {
    const std::vector<GroupStatus> &f = c->driver->Flagged();
    for (uint32_t i = 0; i < f.size() && i < cap; i++) { groups[i] = f[i].group; status[i] = f[i].status; }
    return (uint32_t)f.size();
}
// 1 if ThrowIfFlagged() raised the GoPanic of the first flagged group
int rh_driver_throw_if_flagged(rh_cluster *c) {
    try { c->driver->ThrowIfFlagged(); return 0; } catch (const GoPanic &) { return 1; }
}

// ---- resident driver over the same Node objects and group roles -------------
// (created from the batched driver's current groups; roles set afterwards go
// through rh_resident_roles)
int rh_resident_new(rh_cluster *c, int device) {
    c->resident.reset(new ResidentDriver(c->driver->Groups(), c->n, device));
    return 0;
}
void rh_resident_roles(rh_cluster *c, uint32_t g, int leader_slot, int candidate_slot) {
    Group &grp = c->resident->Groups()[g];
    grp.Leader = leader_slot >= 0 ? at(c, g, (uint32_t)leader_slot) : nullptr;
    grp.Candidate = candidate_slot >= 0 ? at(c, g, (uint32_t)candidate_slot) : nullptr;
    c->resident->MarkDirty(g);
}
void rh_resident_get_roles(rh_cluster *c, uint32_t g, int *leader_slot, int *candidate_slot) {
    Group &grp = c->resident->Groups()[g];
    *leader_slot = *candidate_slot = -1;
    for (uint32_t p = 0; p < c->n && p < grp.Members.size(); p++) {
        if (grp.Members[p] == grp.Leader) *leader_slot = (int)p;
        if (grp.Members[p] == grp.Candidate) *candidate_slot = (int)p;
    }
}
void rh_resident_mark_dirty(rh_cluster *c, uint32_t g) { c->resident->MarkDirty(g); }
// report[3]: groups uploaded, groups changed (downloaded and applied), 1 if the image was rebuilt
int rh_resident_step(rh_cluster *c, uint32_t *report) {
    ResidentDriver::Report r;
    int rc = c->resident->Step(&r);
    if (report) { report[0] = r.uploaded; report[1] = r.changed; report[2] = r.rebuilt ? 1 : 0; }
    return rc;
}
int rh_resident_without_arena(rh_cluster *c) { return c->resident && c->resident->WithoutArena() ? 1 : 0; }
int rh_resident_segments(rh_cluster *c) { return c->resident ? (int)c->resident->Segments() : 0; }
int rh_resident_wide(rh_cluster *c) { return c->resident && c->resident->Wide() ? 1 : 0; }
uint32_t rh_resident_flagged(rh_cluster *c, uint32_t *groups, uint8_t *status, uint32_t cap) {
    const std::vector<GroupStatus> &f = c->resident->Flagged();
    for (uint32_t i = 0; i < f.size() && i < cap; i++) { groups[i] = f[i].group; status[i] = f[i].status; }
    return (uint32_t)f.size();
}

// Timing hook for the resident driver at sizes a Python loop cannot seed: G groups
// of n nodes, every group led by its slot 0 (log of 1..16 entries, one term),
// image resident on `device`.  Each of `steps` steps the host applies one client
// entry to `dirty` randomly chosen groups -- the leader's log grows by one entry
// and a majority's matchIndex moves to it -- marks them dirty and calls Step.
// out[0] = first Step (flatten + upload of the whole image) in ms, out[1] = mean
// microseconds per later Step, out[2] / out[3] = mean groups uploaded / applied
// per step, out[4] = rebuilds among the later steps, out[5] = one
// BatchedDriver::Step on the same objects in ms (0 if skipped).
int rh_resident_bench(uint32_t G, uint32_t n, uint32_t dirty, uint32_t steps, int device, int with_batched,
                      double *out) {
    std::vector<Node *> nodes;
    nodes.reserve((size_t)G * n);
    std::vector<Group> groups(G);
    uint64_t rng = 0x5EED0001ull;
    auto next = [&rng]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
    for (uint32_t g = 0; g < G; g++) {
        std::vector<Node *> peers;
        for (uint32_t p = 0; p < n; p++) {
            Node *x = NewNode((int64_t)p, peers, nullptr);
            peers.push_back(x);
            nodes.push_back(x);
        }
        const uint32_t len = 1 + next() % 16;
        std::vector<int64_t> terms(len, 1);
        for (uint32_t p = 0; p < n; p++) {
            peers[p]->SeedForTest(1, -1, p == 0 ? Leader : Follower, 0);
            peers[p]->SeedLogForTest(p == 0 ? terms : std::vector<int64_t>(1, 1));
        }
        peers[0]->BecomeLeader();
        std::vector<int64_t> &m = peers[0]->MatchIndexForTest();
        for (uint32_t p = 0; p < n; p++) m[p] = next() % len;
        groups[g].Members = peers;
        groups[g].Leader = peers[0];
    }
    auto now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    int rc = 0;
    {
        ResidentDriver rd(groups, n, device);
        double t0 = now();
        rc = rd.Step();
        out[0] = (now() - t0) * 1e3;
        double up = 0, ch = 0, rebuilds = 0, total = 0;
        for (uint32_t s = 0; s < steps && rc == 0; s++) {
            for (uint32_t i = 0; i < dirty; i++) {
                const uint32_t g = next() % G;
                Node *l = rd.Groups()[g].Leader;
                const int64_t idx = l->Inspect().logLen;
                std::vector<Entry> one(1, Entry{"", idx, 1});
                l->AppendForTest(one);
                std::vector<int64_t> &m = l->MatchIndexForTest();
                for (uint32_t p = 0; p < (n / 2) + 1; p++) m[p] = idx;
                rd.MarkDirty(g);
            }
            ResidentDriver::Report r;
            t0 = now();
            rc = rd.Step(&r);
            total += now() - t0;
            up += r.uploaded; ch += r.changed; rebuilds += r.rebuilt ? 1 : 0;
        }
        out[1] = steps ? total / steps * 1e6 : 0;
        out[2] = steps ? up / steps : 0; out[3] = steps ? ch / steps : 0; out[4] = rebuilds;
    }
    out[5] = 0;
    if (rc == 0 && with_batched) {
        BatchedDriver bd(groups, n);
        rc = bd.Step(device);                     // creates the engine, first touch
        double t0 = now();
        if (rc == 0) rc = bd.Step(device);
        out[5] = (now() - t0) * 1e3;
    }
    for (Node *x : nodes) delete x;
    return rc;
}

// Wide-layout Flatten: returns the driver code and the number of log terms; copies nothing
// when an output pointer is null.  Planes are [n][G] int64.
int rh_driver_flatten64(rh_cluster *c, int64_t *match, int64_t *commit, int64_t *leader_term,
                        uint32_t *log_off, int64_t *log_term, uint64_t log_term_cap,
                        int64_t *voter_term, int64_t *voted_for, int64_t *last_log_term,
                        int64_t *cand_term, int64_t *cand_id, uint64_t *log_terms_used) {
    RoundBuffers64 b;
    int rc = c->driver->Flatten(b);
    if (rc != 0) return rc;
    const size_t plane = (size_t)b.slots * b.groups * 8, vec = (size_t)b.groups * 8;
    std::memcpy(match, b.match.data(), plane); std::memcpy(voter_term, b.voterTerm.data(), plane);
    std::memcpy(voted_for, b.votedFor.data(), plane); std::memcpy(last_log_term, b.lastLogTerm.data(), plane);
    std::memcpy(commit, b.commit.data(), vec); std::memcpy(leader_term, b.leaderTerm.data(), vec);
    std::memcpy(cand_term, b.candTerm.data(), vec); std::memcpy(cand_id, b.candId.data(), vec);
    std::memcpy(log_off, b.logOff.data(), ((size_t)b.groups + 1) * 4);
    *log_terms_used = b.logTerm.size();
    if (b.logTerm.size() > log_term_cap) return -3;
    std::memcpy(log_term, b.logTerm.data(), b.logTerm.size() * 8);
    return 0;
}

}  // extern "C"
