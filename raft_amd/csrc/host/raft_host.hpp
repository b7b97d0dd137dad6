// raft_host.hpp -- C++ mirror of the reference's Go surface (package raft in
// tawawhite/raft raft.go) plus the batched driver that calls librq.so.
//
// The build image has no Go toolchain, so the host side that a Go maintainer
// would write (go/raft/*.go) is mirrored here in compiled, tested C++ with the
// same names, argument meaning and error behaviour: a Go runtime panic
// (slice index out of range) is a raft::GoPanic exception.
#pragma once

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

struct rq_engine;

namespace raft {

// raft.go:7-13 -- iota: Leader is the zero value.
enum NodeType : int64_t { Leader = 0, Follower = 1, Candidate = 2 };

// raft.go:71-75
struct Entry {
    std::string Command;
    int64_t Index = 0;
    int64_t TermNum = 0;
};

struct GoPanic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class BatchedDriver;
class ResidentDriver;

// raft.go:15-69.  Fields are private like the reference's unexported fields;
// the batched driver is a friend the way it is a package member in Go.
class Node {
public:
    void BecomeLeader();     // raft.go:101-118
    void BecomeFollower();   // raft.go:120-124
    void BecomeCandidate();  // raft.go:126-130
    // raft.go:132-179
    std::pair<int64_t, bool> AppendEntriesRPC(int64_t term, int64_t leaderId, int64_t prevLogIndex,
                                              int64_t prevLogTerm, const std::vector<Entry> &newEntries,
                                              int64_t leaderCommit);
    // raft.go:181-210
    std::pair<int64_t, bool> RequestVoteRPC(int64_t term, int64_t candidateId, int64_t lastLogIndex,
                                            int64_t lastLogTerm);

    // test access (the Go tests would sit inside the package)
    struct State {
        int64_t id, nodeType, currentTerm, votedFor, logLen, commitIndex, lastApplied, nPeers;
        bool hasLeaderArrays;
    };
    State Inspect() const;
    const std::vector<int64_t> &NextIndex() const { return nextIndex; }
    const std::vector<int64_t> &MatchIndex() const { return matchIndex; }
    std::vector<int64_t> &MatchIndexForTest() { return matchIndex; }
    void SeedForTest(int64_t currentTerm, int64_t votedFor, NodeType type, int64_t commitIndex);
    void SeedLogForTest(const std::vector<int64_t> &terms);
    void AppendForTest(const std::vector<Entry> &entries) { log.insert(log.end(), entries.begin(), entries.end()); }

private:
    friend Node *NewNode(int64_t, std::vector<Node *>, std::function<void(const std::string &)>);
    friend class BatchedDriver;
    friend class ResidentDriver;
    void testToAbdicateLeadership(int64_t term);  // raft.go:212-223

    int64_t id = 0;
    NodeType nodeType = Follower;
    std::function<void(const std::string &)> stateMachine;
    std::vector<Node *> peers;  // includes this node (raft.go:94-97)
    int64_t currentTerm = 0;
    int64_t votedFor = -1;
    std::vector<Entry> log;
    int64_t commitIndex = 0;
    int64_t lastApplied = 0;
    std::vector<int64_t> nextIndex;
    std::vector<int64_t> matchIndex;
    bool leaderArrays = false;  // Go: nextIndex/matchIndex != nil
};

// raft.go:77-99
Node *NewNode(int64_t id, std::vector<Node *> peers, std::function<void(const std::string &)> statemachine);
int64_t minInt(int64_t a, int64_t b);                 // raft.go:226-231
const Entry &lastEntry(const std::vector<Entry> &e);  // raft.go:234-236, GoPanic on empty

// ---- batched driver (no counterpart in the reference) ----------------------

struct Group {
    std::vector<Node *> Members;  // slot order
    Node *Leader = nullptr;       // member currently leading, or null
    Node *Candidate = nullptr;    // member currently asking for votes, or null
};

// Structure-of-arrays image of all groups, layout of include/rq.h: T = int32_t
// is the compact layout, T = int64_t the wide one (rq_round64).
template <typename T>
struct RoundBuffersT {
    uint32_t groups = 0, slots = 0, ld = 0;
    std::vector<T> match, commit, leaderTerm, logTerm;
    std::vector<uint32_t> logOff;
    std::vector<T> voterTerm, votedFor, lastLogTerm, candTerm, candId;
    std::vector<T> commitOut, voterTermOut, maxTerm;
    std::vector<uint16_t> grantMask, stepdownMask;
    std::vector<uint8_t> elected, status;
    // groups smaller than `slots` occupy the low slots; memberMask[g] = (1 << size) - 1
    std::vector<uint16_t> memberMask;
    bool mixed = false;   // some group has fewer than `slots` members: the round needs the masks
};
typedef RoundBuffersT<int32_t> RoundBuffers;
typedef RoundBuffersT<int64_t> RoundBuffers64;

enum : int { kDriverOk = 0, kDriverRange = -2, kDriverShape = -1 };

// Groups are independent, so the drivers walk them in parallel: fn(first, last,
// worker) runs for disjoint group ranges on up to 16 host threads (one call on the
// caller's thread below 32768 groups).  The reference is single-threaded
// (raft.go has no goroutines); this only concerns the flattening of its objects.
void ParallelOverGroups(uint32_t groups, const std::function<void(uint32_t, uint32_t, uint32_t)> &fn);
uint32_t ParallelWorkers(uint32_t groups);

// A group whose round hit a condition that is a Go panic in the reference:
// status carries the RQ_ST_* bits (RQ_ST_EMPTY_LOG: a voter reached
// lastEntry(log) with an empty log, raft.go:204/:235; RQ_ST_LOG_OOB: the leader's
// quorum index lies outside its log, the indexing convention of raft.go:151).
struct GroupStatus {
    uint32_t group;
    uint8_t status;
};

class BatchedDriver {
public:
    // slots = the largest group size; smaller groups are allowed (1..slots members).
    BatchedDriver(std::vector<Group> groups, uint32_t slots) : groups_(std::move(groups)), slots_(slots) {}
    ~BatchedDriver();
    // AoS -> SoA.  A value the layout cannot hold gives kDriverRange and leaves
    // `b` unspecified: outside int32 for the compact layout; for the wide layout
    // only a log term equal to INT64_MIN (the empty-log marker).
    int Flatten(RoundBuffers &b) const;
    int Flatten(RoundBuffers64 &b) const;
    // Applies one round's outputs to the objects: voter terms and roles
    // (raft.go:219-221), leader commitIndex, BecomeLeader for winners.
    void Apply(const RoundBuffers &b);
    void Apply(const RoundBuffers64 &b);
    // Flatten -> one fused round on `device` -> Apply.  The compact layout is
    // tried first; if some value does not fit int32 the round runs in the wide
    // layout instead.  Groups of different sizes run through the member-mask
    // (joint) round with old = new = the group's members.  Returns an RQ_* code.
    //
    // Where the reference would panic for some group (see GroupStatus) the round
    // is still applied for every group -- a flagged group keeps what the reference
    // had done before its panic: voters' terms raised and roles set (raft.go:219-
    // 221), no grant from the panicking voter, no commit advance -- and the groups
    // are listed in Flagged(); ThrowIfFlagged() turns the first one into the
    // GoPanic a per-node call sequence would have raised.
    int Step(int device);
    bool LastStepWide() const { return last_step_wide_; }
    const std::vector<GroupStatus> &Flagged() const { return flagged_; }
    void ThrowIfFlagged() const;
    std::vector<Group> &Groups() { return groups_; }

private:
    int EnsureEngine(int device, uint32_t groups, uint32_t slots, uint64_t log_terms);
    template <typename T> static int FlattenT(const std::vector<Group> &groups, uint32_t n, RoundBuffersT<T> &b);
    template <typename T> static void ApplyT(std::vector<Group> &groups, const RoundBuffersT<T> &b);
    template <typename T> void CollectFlagged(const RoundBuffersT<T> &b);
    std::vector<Group> groups_;
    uint32_t slots_;
    rq_engine *engine_ = nullptr;
    int engine_device_ = -1;
    uint64_t engine_log_cap_ = 0;
    uint32_t engine_groups_ = 0;
    bool last_step_wide_ = false;
    std::vector<GroupStatus> flagged_;
    // Step's flat image, kept between calls (176 MB at 1M x 5: allocating and
    // faulting it in afresh costs more than the round trip over PCIe)
    RoundBuffers step_buffers_;
    RoundBuffers64 step_buffers64_;
};

// ---- resident driver (no counterpart in the reference) ----------------------
//
// BatchedDriver::Step flattens every Node and moves the whole image over PCIe in
// both directions on every call: milliseconds around a 30 us kernel.  The
// resident driver keeps the image in HBM between rounds.  The host tells it
// which groups it touched (MarkDirty, after applying RPCs to their Nodes); Step
// uploads those groups only (rq_groups_scatter_dev), runs the round in place
// (rq_plan_launch), reads the one-bit-per-group "changed" summary the kernel
// wrote and downloads and applies only the groups whose bit is set
// (rq_groups_gather_dev).  The image is laid out in the 64-group tiles of
// rq_round_desc.tile_bytes (the layout the round kernels run fastest on; the
// single-group calls address it like planes).  Staging lives in page-locked
// memory allocated once.
// Compact (int32) layout while every value fits it; a value beyond int32 (Go's int
// is 64-bit, raft.go:35-68) makes the next Step lay the image out afresh in the wide
// (int64) layout, where it stays.
class ResidentDriver {
public:
    struct Report {
        uint32_t uploaded = 0;   // groups scattered to the device before the round
        uint32_t changed = 0;    // groups gathered and applied after it
        bool rebuilt = false;    // the whole image was flattened and uploaded (first step, or a log outgrew its slice)
    };
    ResidentDriver(std::vector<Group> groups, uint32_t slots, int device);
    ~ResidentDriver();
    ResidentDriver(const ResidentDriver &) = delete;
    ResidentDriver &operator=(const ResidentDriver &) = delete;
    void MarkDirty(uint32_t group);
    int Step(Report *report = nullptr);      // an RQ_* code; flagged groups as in BatchedDriver::Step
    const std::vector<GroupStatus> &Flagged() const { return flagged_; }
    // True while the image holds no log arena: every leader's log has Raft's own shape
    // (terms never decrease, none above the leader's term), so the device only needs the
    // index where the leader's term starts and the log's length (rq_round_desc.
    // log_term_start).  A log that loses the shape (the reference appends whatever it is
    // sent, raft.go:158-170) makes the next Step lay the image out afresh with an arena.
    bool WithoutArena() const;
    // segments of the launch's size table: non-zero when the groups differ in size and are kept in
    // size order, so that tiles of one size are decided by the fixed-size code
    uint32_t Segments() const;
    bool Wide() const { return wide_; }      // the image holds int64 elements
    std::vector<Group> &Groups() { return groups_; }

private:
    struct Impl;
    struct LogShape {
        bool shaped = true;      // terms never decrease and none exceeds the leader's term
        int64_t term_start = 0;  // index of the first entry of the leader's term, or the log's length
        uint32_t len = 0;
    };
    // written once over the element type of the image (int32_t or int64_t)
    template <typename T> int RebuildT();
    template <typename T> int StepT(Report *report);
    template <typename T> int FlattenRow(uint32_t g, T *row, std::vector<T> &terms, LogShape *shape = nullptr) const;
    // groups whose row must go up again and groups with status bits are appended to the caller's lists
    template <typename T> void ApplyRow(uint32_t g, const T *row, uint64_t flags, std::vector<uint32_t> &redo,
                                        std::vector<GroupStatus> &flagged);
    LogShape ShapeOf(uint32_t g) const;
    bool wide_ = false;
    std::vector<Group> groups_;
    uint32_t slots_;
    int device_;
    bool mixed_ = false;
    std::vector<uint32_t> dirty_;
    std::vector<uint8_t> dirty_mark_;
    std::vector<GroupStatus> flagged_;
    Impl *impl_ = nullptr;
};

}  // namespace raft
