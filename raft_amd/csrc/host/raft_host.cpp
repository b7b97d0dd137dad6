// raft_host.cpp -- see raft_host.hpp.
#include "raft_host.hpp"

#include <atomic>
#include <climits>
#include <thread>

#include "../../../include/rq.h"

namespace raft {

int64_t minInt(int64_t a, int64_t b) { return a < b ? a : b; }

const Entry &lastEntry(const std::vector<Entry> &e) {
    if (e.empty()) throw GoPanic("index out of range [-1]");
    return e.back();
}

Node *NewNode(int64_t id, std::vector<Node *> peers, std::function<void(const std::string &)> statemachine) {
    Node *n = new Node();
    n->id = id;
    n->stateMachine = std::move(statemachine);
    n->nodeType = Follower;
    n->votedFor = -1;
    peers.push_back(n);  // raft.go:94
    for (Node *member : peers) member->peers = peers;  // raft.go:95-97
    return n;
}

void Node::BecomeLeader() {
    nodeType = Leader;
    nextIndex.assign(peers.size(), (int64_t)log.size() + 1);  // raft.go:106-109
    matchIndex.assign(peers.size(), 0);                       // raft.go:114-117
    leaderArrays = true;
}

void Node::BecomeFollower() {
    nodeType = Follower;
    nextIndex.clear(); matchIndex.clear(); leaderArrays = false;
}

void Node::BecomeCandidate() {
    nodeType = Candidate;
    nextIndex.clear(); matchIndex.clear(); leaderArrays = false;
}

void Node::testToAbdicateLeadership(int64_t term) {
    if (term > currentTerm) {
        currentTerm = term;
        nodeType = Follower;
    }
}

std::pair<int64_t, bool> Node::RequestVoteRPC(int64_t term, int64_t candidateId, int64_t, int64_t) {
    testToAbdicateLeadership(term);                       // :187
    if (term < currentTerm) return {currentTerm, false};  // :190-192
    const bool notYetVoted = votedFor == -1;              // :202
    const bool votedSameBefore = votedFor == candidateId; // :203
    const bool requesterMoreUpToDate = lastEntry(log).TermNum <= term;  // :204
    return {currentTerm, (notYetVoted || votedSameBefore) && requesterMoreUpToDate};
}

std::pair<int64_t, bool> Node::AppendEntriesRPC(int64_t term, int64_t, int64_t prevLogIndex,
                                                int64_t prevLogTerm, const std::vector<Entry> &newEntries,
                                                int64_t leaderCommit) {
    testToAbdicateLeadership(term);                       // :142
    if (term < currentTerm) return {currentTerm, false};  // :145-147
    if (prevLogIndex < 0 || prevLogIndex >= (int64_t)log.size()) throw GoPanic("index out of range");
    if (log[(size_t)prevLogIndex].TermNum != prevLogTerm) return {currentTerm, false};  // :151-153
    for (const Entry &e : newEntries) {                   // :158-167
        const bool indexIsInRange = (int64_t)log.size() <= e.Index;  // :159, inverted in the reference
        if (indexIsInRange) throw GoPanic("index out of range");     // :161 indexes log[e.Index]
    }
    log.insert(log.end(), newEntries.begin(), newEntries.end());     // :170
    if (leaderCommit > commitIndex)                                  // :174
        commitIndex = minInt(leaderCommit, lastEntry(newEntries).Index);
    return {currentTerm, true};
}

Node::State Node::Inspect() const {
    return State{id, (int64_t)nodeType, currentTerm, votedFor, (int64_t)log.size(), commitIndex,
                 lastApplied, (int64_t)peers.size(), leaderArrays};
}

void Node::SeedForTest(int64_t term, int64_t voted, NodeType type, int64_t commit) {
    currentTerm = term; votedFor = voted; nodeType = type; commitIndex = commit;
}

void Node::SeedLogForTest(const std::vector<int64_t> &terms) {
    log.clear();
    for (size_t i = 0; i < terms.size(); i++) log.push_back(Entry{"", (int64_t)i, terms[i]});
}

// ---- batched driver ---------------------------------------------------------

uint32_t ParallelWorkers(uint32_t groups) {
    if (groups < 32768) return 1;
    const uint32_t hw = std::thread::hardware_concurrency();
    const uint32_t t = hw ? hw : 1;
    return t > 16 ? 16 : t;
}

void ParallelOverGroups(uint32_t groups, const std::function<void(uint32_t, uint32_t, uint32_t)> &fn) {
    const uint32_t workers = ParallelWorkers(groups);
    if (workers <= 1) { fn(0, groups, 0); return; }
    std::vector<std::thread> pool;
    pool.reserve(workers - 1);
    auto bound = [&](uint32_t w) { return (uint32_t)((uint64_t)groups * w / workers); };
    for (uint32_t w = 1; w < workers; w++) pool.emplace_back(fn, bound(w), bound(w + 1), w);
    fn(0, bound(1), 0);
    for (std::thread &t : pool) t.join();
}

// The lowest value of each layout is its empty-log marker and cannot be a value.
static bool narrow(int64_t v, int32_t &out) {
    if (v <= (int64_t)INT32_MIN || v > (int64_t)INT32_MAX) return false;
    out = (int32_t)v;
    return true;
}
static bool narrow(int64_t v, int64_t &out) {
    out = v;
    return true;
}

template <typename T> struct layout_limits;
template <> struct layout_limits<int32_t> {
    static constexpr int32_t kMax = INT32_MAX, kBelowAll = INT32_MIN + 1, kLogEmpty = RQ_LOG_EMPTY;
};
template <> struct layout_limits<int64_t> {
    static constexpr int64_t kMax = INT64_MAX, kBelowAll = INT64_MIN, kLogEmpty = RQ_LOG_EMPTY64;
};

template <typename T>
int BatchedDriver::FlattenT(const std::vector<Group> &groups, uint32_t n, RoundBuffersT<T> &b) {
    typedef layout_limits<T> lim;
    const uint32_t G = (uint32_t)groups.size();
    if (n < 1 || n > RQ_MAX_SLOTS) return kDriverShape;
    b.groups = G; b.slots = n; b.ld = G; b.mixed = false;
    const size_t plane = (size_t)n * G;
    // sizes only: every element is written below (a buffer kept from the last call is reused as it is)
    b.match.resize(plane); b.voterTerm.resize(plane); b.votedFor.resize(plane);
    b.lastLogTerm.resize(plane); b.voterTermOut.resize(plane);
    b.commit.resize(G); b.leaderTerm.resize(G); b.candTerm.resize(G); b.candId.resize(G);
    b.commitOut.resize(G); b.maxTerm.resize(G); b.grantMask.resize(G); b.stepdownMask.resize(G);
    b.elected.resize(G); b.status.resize(G); b.memberMask.resize(G);
    b.logOff.resize((size_t)G + 1);
    // pass 1: the leaders' log lengths, then their offsets in the arena
    std::atomic<int> err{kDriverOk};
    ParallelOverGroups(G, [&](uint32_t g0, uint32_t g1, uint32_t) {
        for (uint32_t g = g0; g < g1; g++) {
            const Group &grp = groups[g];
            const size_t k = grp.Members.size();
            if (k < 1 || k > n) { err.store(kDriverShape); return; }
            b.logOff[g + 1] = grp.Leader ? (uint32_t)grp.Leader->log.size() : 0u;
            if (grp.Leader && grp.Leader->log.size() > 0xFFFFFFFFull) { err.store(kDriverRange); return; }
        }
    });
    if (err.load() != kDriverOk) return err.load();
    uint64_t total = 0;
    b.logOff[0] = 0;
    for (uint32_t g = 0; g < G; g++) {
        total += b.logOff[g + 1];
        if (total > 0xFFFFFFFFull) return kDriverRange;
        b.logOff[g + 1] = (uint32_t)total;
    }
    b.logTerm.resize(total ? (size_t)total : 1);
    if (!total) b.logTerm[0] = 0;
    // pass 2: every group's slots, defaults included
    std::atomic<bool> mixed{false};
    ParallelOverGroups(G, [&](uint32_t g0, uint32_t g1, uint32_t) {
        bool any_small = false;
        for (uint32_t g = g0; g < g1; g++) {
            const Group &grp = groups[g];
            const uint32_t k = (uint32_t)grp.Members.size();
            b.memberMask[g] = (uint16_t)((1u << k) - 1u);
            any_small |= k != n;
            b.commit[g] = lim::kMax;            // no leader: N > commit is never true
            b.leaderTerm[g] = 0;
            b.candTerm[g] = lim::kBelowAll;     // no candidate: no voter's term moves, results are not applied
            b.candId[g] = -1;
            b.commitOut[g] = 0; b.maxTerm[g] = 0; b.grantMask[g] = 0; b.stepdownMask[g] = 0;
            b.elected[g] = 0; b.status[g] = 0;
            for (uint32_t p = 0; p < n; p++) {
                const size_t i = (size_t)p * G + g;
                b.match[i] = 0; b.voterTerm[i] = 0; b.votedFor[i] = -1;
                b.lastLogTerm[i] = lim::kLogEmpty; b.voterTermOut[i] = 0;
            }
            bool ok = true;
            if (const Node *l = grp.Leader) {
                ok = ok && narrow(l->commitIndex, b.commit[g]) && narrow(l->currentTerm, b.leaderTerm[g]);
                for (uint32_t p = 0; p < k && p < l->matchIndex.size(); p++)
                    ok = ok && narrow(l->matchIndex[p], b.match[(size_t)p * G + g]);
                T *dst = b.logTerm.data() + b.logOff[g];
                for (size_t e = 0; e < l->log.size(); e++) ok = ok && narrow(l->log[e].TermNum, dst[e]);
            }
            if (const Node *c = grp.Candidate)
                ok = ok && narrow(c->currentTerm, b.candTerm[g]) && narrow(c->id, b.candId[g]);
            for (uint32_t p = 0; p < k; p++) {
                const Node *m = grp.Members[p];
                const size_t i = (size_t)p * G + g;
                ok = ok && narrow(m->currentTerm, b.voterTerm[i]) && narrow(m->votedFor, b.votedFor[i]);
                if (!m->log.empty())
                    ok = ok && narrow(m->log.back().TermNum, b.lastLogTerm[i]) && b.lastLogTerm[i] != lim::kLogEmpty;
            }
            if (!ok) { err.store(kDriverRange); return; }
        }
        if (any_small) mixed.store(true);
    });
    if (err.load() != kDriverOk) return err.load();
    b.mixed = mixed.load();
    return kDriverOk;
}

template <typename T>
void BatchedDriver::ApplyT(std::vector<Group> &groups, const RoundBuffersT<T> &b) {
    ParallelOverGroups(b.groups, [&](uint32_t g0, uint32_t g1, uint32_t) {
        for (uint32_t g = g0; g < g1; g++) {
            Group &grp = groups[g];
            Node *leader_before = grp.Leader;
            if (grp.Candidate) {
                for (uint32_t p = 0; p < b.slots && p < grp.Members.size(); p++) {
                    Node *m = grp.Members[p];
                    m->currentTerm = b.voterTermOut[(size_t)p * b.ld + g];
                    if ((b.stepdownMask[g] >> p) & 1u) m->nodeType = Follower;  // raft.go:221
                }
                if (b.elected[g]) {
                    grp.Candidate->BecomeLeader();
                    grp.Leader = grp.Candidate;
                    grp.Candidate = nullptr;
                }
            }
            // the round decided the commit rule for the leader the group had when it ran
            if (leader_before) leader_before->commitIndex = b.commitOut[g];
            // a leader a vote request demoted (raft.go:221) leads no more: later rounds must not
            // advance commitIndex for a Follower
            if (grp.Leader && grp.Leader->nodeType != Leader) grp.Leader = nullptr;
        }
    });
}

template <typename T>
void BatchedDriver::CollectFlagged(const RoundBuffersT<T> &b) {
    flagged_.clear();
    for (uint32_t g = 0; g < b.groups; g++)
        if (b.status[g]) flagged_.push_back(GroupStatus{g, b.status[g]});
}

void BatchedDriver::ThrowIfFlagged() const {
    if (flagged_.empty()) return;
    const GroupStatus &f = flagged_.front();
    throw GoPanic("group " + std::to_string(f.group) +
                  ((f.status & RQ_ST_EMPTY_LOG) ? ": index out of range [-1] (lastEntry of an empty log)"
                                                : ": index out of range (log[N] beyond the leader's log)"));
}

int BatchedDriver::Flatten(RoundBuffers &b) const { return FlattenT(groups_, slots_, b); }
int BatchedDriver::Flatten(RoundBuffers64 &b) const { return FlattenT(groups_, slots_, b); }
void BatchedDriver::Apply(const RoundBuffers &b) { ApplyT(groups_, b); }
void BatchedDriver::Apply(const RoundBuffers64 &b) { ApplyT(groups_, b); }

BatchedDriver::~BatchedDriver() {
    if (engine_) rq_engine_destroy(engine_);
}

int BatchedDriver::EnsureEngine(int device, uint32_t groups, uint32_t slots, uint64_t need) {
    if (engine_ && (engine_device_ != device || need > engine_log_cap_ || groups > engine_groups_)) {
        rq_engine_destroy(engine_);
        engine_ = nullptr;
    }
    if (!engine_) {
        engine_log_cap_ = need * 2 + 4096;              // room for the leaders' logs to grow
        engine_groups_ = groups ? groups : 1;
        int rc = rq_engine_create(device, engine_groups_, slots, engine_log_cap_, &engine_);
        if (rc != RQ_OK) { engine_ = nullptr; return rc; }
        engine_device_ = device;
    }
    return RQ_OK;
}

int BatchedDriver::Step(int device) {
    RoundBuffers &b = step_buffers_;
    int rc = Flatten(b);
    if (rc == kDriverRange) {               // some value needs more than 32 bits: wide layout
        RoundBuffers64 &w = step_buffers64_;
        rc = Flatten(w);
        if (rc != kDriverOk) return rc == kDriverRange ? RQ_E_RANGE : RQ_E_ARG;
        rc = EnsureEngine(device, w.groups, w.slots, w.logTerm.size());
        if (rc != RQ_OK) return rc;
        const uint16_t *mask = w.mixed ? w.memberMask.data() : nullptr;
        rc = rq_round64(engine_, RQ_ROUND_COMMIT | RQ_ROUND_VOTE, w.groups, w.slots, w.ld, mask, mask,
                        w.match.data(), w.commit.data(), w.leaderTerm.data(), w.logOff.data(),
                        w.logTerm.data(), nullptr, w.voterTerm.data(), w.votedFor.data(),
                        w.lastLogTerm.data(), w.candTerm.data(), w.candId.data(), w.commitOut.data(),
                        w.voterTermOut.data(), w.grantMask.data(), w.stepdownMask.data(),
                        w.elected.data(), w.maxTerm.data(), w.status.data());
        if (rc != RQ_OK) return rc;
        Apply(w);
        CollectFlagged(w);
        last_step_wide_ = true;
        return RQ_OK;
    }
    if (rc != kDriverOk) return RQ_E_ARG;
    rc = EnsureEngine(device, b.groups, b.slots, b.logTerm.size());
    if (rc != RQ_OK) return rc;
    if (b.mixed)
        rc = rq_fused_round_joint(engine_, b.groups, b.slots, b.ld, b.memberMask.data(), b.memberMask.data(),
                                  b.match.data(), b.commit.data(), b.leaderTerm.data(), b.logOff.data(),
                                  b.logTerm.data(), b.voterTerm.data(), b.votedFor.data(),
                                  b.lastLogTerm.data(), b.candTerm.data(), b.candId.data(),
                                  b.commitOut.data(), b.voterTermOut.data(), b.grantMask.data(),
                                  b.stepdownMask.data(), b.elected.data(), b.maxTerm.data(), b.status.data());
    else
        rc = rq_fused_round(engine_, b.groups, b.slots, b.ld, b.match.data(), b.commit.data(),
                            b.leaderTerm.data(), b.logOff.data(), b.logTerm.data(), b.voterTerm.data(),
                            b.votedFor.data(), b.lastLogTerm.data(), b.candTerm.data(), b.candId.data(),
                            b.commitOut.data(), b.voterTermOut.data(), b.grantMask.data(),
                            b.stepdownMask.data(), b.elected.data(), b.maxTerm.data(), b.status.data());
    if (rc != RQ_OK) return rc;
    Apply(b);
    CollectFlagged(b);
    last_step_wide_ = false;
    return RQ_OK;
}

}  // namespace raft
