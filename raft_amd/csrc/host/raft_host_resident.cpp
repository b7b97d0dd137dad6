// raft_host_resident.cpp -- raft::ResidentDriver, see raft_host.hpp.
#include <climits>
#include <cstring>

#include "../../../include/rq.h"
#include "raft_host.hpp"

#include <atomic>
#include <thread>

namespace raft {

namespace {

constexpr uint32_t kMaxRowsPerTransfer = 1u << 18;   // staging rows; more dirty groups than this: full rebuild

// The lowest value of each layout is its empty-log marker and cannot be a value.
bool narrow(int64_t v, int32_t &out) {
    if (v <= (int64_t)INT32_MIN || v > (int64_t)INT32_MAX) return false;
    out = (int32_t)v;
    return true;
}
bool narrow(int64_t v, int64_t &out) {
    if (v == INT64_MIN) return false;
    out = v;
    return true;
}
template <typename T> struct limits;
template <> struct limits<int32_t> { static constexpr int32_t kMax = INT32_MAX, kBelowAll = INT32_MIN + 1, kLogEmpty = RQ_LOG_EMPTY; };
template <> struct limits<int64_t> { static constexpr int64_t kMax = INT64_MAX, kBelowAll = INT64_MIN + 1, kLogEmpty = RQ_LOG_EMPTY64; };

// room a group's arena slice keeps for its log to grow before the image is rebuilt
uint32_t slice_capacity(size_t len) { return (uint32_t)(len + (len / 4 > 4 ? len / 4 : 4)); }

}  // namespace

struct ResidentDriver::Impl {
    rq_engine *e = nullptr;
    rq_plan *plan = nullptr;
    rq_round_desc desc;
    uint32_t G = 0, n = 0, row_cap = 0;
    uint64_t term_cap = 0;              // log terms one scatter can carry
    std::vector<void *> device_allocs, host_allocs;
    // the arena as the host manages it: every group's slice (offset, capacity), free space from `tail` on
    std::vector<uint32_t> off, cap;
    uint64_t tail = 0, arena_entries = 0;
    // staging: one block going up (group ids, term offsets, rows, terms packed back to back for the
    // rows actually sent), one coming down (flags, rows); page-locked mirrors of the same size
    char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    uint64_t *h_cbits = nullptr;
    bool by_start = false;              // no arena: log_term_start + log_len
    // groups of different sizes kept in size order: the launch's segment table, and per 64-group
    // tile the cluster size all of its groups have (0: decided through the member masks)
    std::vector<uint8_t> tile_kind;

    int dalloc(uint64_t bytes, void **out) {
        int rc = rq_device_alloc(e, bytes ? bytes : 4, RQ_MEM_DEFAULT, out);
        if (rc == RQ_OK) device_allocs.push_back(*out);
        return rc;
    }
    template <typename P> int halloc(uint64_t bytes, P **out) {
        void *p = nullptr;
        int rc = rq_host_alloc(e, bytes ? bytes : 4, &p);
        if (rc == RQ_OK) host_allocs.push_back(p);
        *out = static_cast<P *>(p);
        return rc;
    }
    void release() {
        if (plan) rq_plan_destroy(plan);
        plan = nullptr;
        for (void *p : device_allocs) rq_device_free(e, p);
        for (void *p : host_allocs) rq_host_free(e, p);
        device_allocs.clear(); host_allocs.clear();
    }
};

// The rows a step moves are independent of each other, and reading or writing one means
// walking a handful of heap objects that are rarely in cache (a microsecond per row is cache
// misses): from 512 rows on the walk is dealt to host threads, 256 rows or more each,
// fn(first, last, worker) on disjoint row ranges.  Below that one call on the caller's thread.
static uint32_t RowWorkers(uint32_t rows) {
    const uint32_t by_rows = rows / 256;
    const uint32_t cap = ParallelWorkers(1u << 20);       // the drivers' thread limit
    return by_rows < 2 ? 1 : (by_rows < cap ? by_rows : cap);
}
template <typename F>
static void ParallelOverRows(uint32_t rows, uint32_t workers, F fn) {
    if (workers <= 1) { fn(0u, rows, 0u); return; }
    std::vector<std::thread> pool;
    pool.reserve(workers - 1);
    auto bound = [&](uint32_t w) { return (uint32_t)((uint64_t)rows * w / workers); };
    for (uint32_t w = 1; w < workers; w++) pool.emplace_back([&fn, &bound, w]() { fn(bound(w), bound(w + 1), w); });
    fn(0u, bound(1), 0u);
    for (std::thread &t : pool) t.join();
}

ResidentDriver::ResidentDriver(std::vector<Group> groups, uint32_t slots, int device)
    : groups_(std::move(groups)), slots_(slots), device_(device), dirty_mark_(groups_.size(), 0) {}

ResidentDriver::~ResidentDriver() {
    if (impl_) {
        impl_->release();
        if (impl_->e) rq_engine_destroy(impl_->e);
        delete impl_;
    }
}

void ResidentDriver::MarkDirty(uint32_t group) {
    if (group < dirty_mark_.size() && !dirty_mark_[group]) {
        dirty_mark_[group] = 1;
        dirty_.push_back(group);
    }
}

// One group as a scatter row (layout of rq_groups_scatter_dev) plus its leader's
// log terms.  A group without a leader gets commit = INT32_MAX (N > commit is
// never true); a group without a candidate gets a term below every voter's.
ResidentDriver::LogShape ResidentDriver::ShapeOf(uint32_t g) const {
    LogShape s;
    const Node *l = groups_[g].Leader;
    if (!l) return s;
    const std::vector<Entry> &log = l->log;
    s.len = (uint32_t)log.size();
    size_t start = log.size();
    for (size_t i = 0; i < log.size(); i++) {
        if (i && log[i].TermNum < log[i - 1].TermNum) s.shaped = false;
        if (start == log.size() && log[i].TermNum == l->currentTerm) start = i;
    }
    if (!log.empty() && log.back().TermNum > l->currentTerm) s.shaped = false;
    s.term_start = (int64_t)start;
    return s;
}

bool ResidentDriver::WithoutArena() const { return impl_ && impl_->plan && impl_->by_start; }
uint32_t ResidentDriver::Segments() const { return impl_ && impl_->plan ? impl_->desc.n_segments : 0; }

// `shape` non-null: the image keeps no arena -- the log is only measured (ShapeOf), its
// terms are not copied and `terms` stays empty.
template <typename T>
int ResidentDriver::FlattenRow(uint32_t g, T *row, std::vector<T> &terms, LogShape *shape) const {
    const Group &grp = groups_[g];
    const uint32_t n = slots_, k = (uint32_t)grp.Members.size();
    if (k < 1 || k > n || n > RQ_MAX_SLOTS) return RQ_E_ARG;
    T *match = row, *vt = row + n, *vf = row + 2 * n, *ll = row + 3 * n, *tail = row + 4 * n;
    for (uint32_t p = 0; p < n; p++) { match[p] = 0; vt[p] = 0; vf[p] = -1; ll[p] = limits<T>::kLogEmpty; }
    tail[0] = limits<T>::kMax; tail[1] = 0; tail[2] = limits<T>::kBelowAll; tail[3] = -1; tail[4] = 0; tail[6] = 0;
    const uint32_t mask = (1u << k) - 1u;
    tail[5] = (T)(mask | (mask << 16));
    terms.clear();
    if (const Node *l = grp.Leader) {
        if (!narrow(l->commitIndex, tail[0]) || !narrow(l->currentTerm, tail[1])) return RQ_E_RANGE;
        for (uint32_t p = 0; p < k && p < l->matchIndex.size(); p++)
            if (!narrow(l->matchIndex[p], match[p])) return RQ_E_RANGE;
        if (!shape) {
            terms.resize(l->log.size());
            for (size_t i = 0; i < l->log.size(); i++)
                if (!narrow(l->log[i].TermNum, terms[i])) return RQ_E_RANGE;
        }
    }
    if (shape) *shape = ShapeOf(g);
    if (const Node *c = grp.Candidate) {
        if (!narrow(c->currentTerm, tail[2]) || !narrow(c->id, tail[3])) return RQ_E_RANGE;
    }
    for (uint32_t p = 0; p < k; p++) {
        const Node *m = grp.Members[p];
        if (!narrow(m->currentTerm, vt[p]) || !narrow(m->votedFor, vf[p])) return RQ_E_RANGE;
        if (!m->log.empty() && !narrow(m->log.back().TermNum, ll[p])) return RQ_E_RANGE;
    }
    return RQ_OK;
}

// The whole image: every group flattened, arena slices sized with room to grow,
// device buffers, staging and the prepared launch made afresh.
template <typename T>
int ResidentDriver::RebuildT() {
    constexpr uint64_t E = sizeof(T);
    const uint32_t G = (uint32_t)groups_.size(), n = slots_;
    if (G == 0 || n < 1 || n > RQ_MAX_SLOTS) return RQ_E_ARG;
    if (!impl_) impl_ = new Impl();
    Impl &I = *impl_;
    if (!I.e) {
        int rc = rq_engine_create(device_, 64, 1, 64, &I.e);   // the engine only binds the device here
        if (rc != RQ_OK) { I.e = nullptr; return rc; }
    }
    I.release();
    I.G = G; I.n = n;
    const size_t words = 4 * (size_t)n + 7;
    // The image lives in the 64-group tiles of rq_round_desc.tile_bytes: one block for the
    // inputs (and the state updated in place), one for the pure outputs.  Field offsets
    // inside a tile, widest elements first:
    const uint64_t row_t = 64 * E, slots_t = (uint64_t)n * row_t;   // one row / n rows of 64 elements
    const uint64_t o_match = 0, o_vt = slots_t, o_vf = 2 * slots_t, o_ll = 3 * slots_t;
    const uint64_t o_commit = 4 * slots_t, o_lterm = o_commit + row_t, o_cterm = o_lterm + row_t, o_cid = o_cterm + row_t;
    const uint64_t o_tstart = o_cid + row_t;                       // log_term_start (element type), unused with an arena
    const uint64_t o_len = o_tstart + row_t, o_off = o_len + 256;   // 32-bit: log_len, log_off
    const uint64_t o_om = o_off + 256, o_nm = o_om + 128;
    const uint64_t tile_bytes = o_nm + 128;
    const uint64_t oo_max = 0, oo_grant = row_t, oo_step = oo_grant + 128, oo_elected = oo_step + 128, oo_status = oo_elected + 64;
    const uint64_t out_tile_bytes = oo_status + 64 + 64;            // padded to a multiple of 8
    const uint64_t ntiles = ((uint64_t)G + 1 + 63) / 64;         // log_off has G + 1 entries
    std::vector<unsigned char> block(ntiles * tile_bytes, 0);
    auto wT = [&](uint64_t off, uint32_t p, uint32_t g) -> T & {
        return *reinterpret_cast<T *>(block.data() + (uint64_t)(g >> 6) * tile_bytes + off +
                                      ((uint64_t)p * 64 + (g & 63u)) * E);
    };
    auto w32 = [&](uint64_t off, uint32_t g) -> uint32_t & {
        return *reinterpret_cast<uint32_t *>(block.data() + (uint64_t)(g >> 6) * tile_bytes + off + (uint64_t)(g & 63u) * 4);
    };
    auto w16 = [&](uint64_t off, uint32_t g) -> uint16_t & {
        return *reinterpret_cast<uint16_t *>(block.data() + (uint64_t)(g >> 6) * tile_bytes + off + (uint64_t)(g & 63u) * 2);
    };
    std::vector<T> arena;
    std::vector<uint32_t> len(G);
    I.off.assign((size_t)G + 1, 0);
    I.cap.assign(G, 0);
    mixed_ = false;
    // Do all the leaders' logs have Raft's own shape?  Then the device needs no arena.
    std::vector<int64_t> tstart(G);
    std::atomic<bool> all_shaped{true};
    ParallelOverGroups(G, [&](uint32_t g0, uint32_t g1, uint32_t) {
        for (uint32_t g = g0; g < g1; g++) {
            const LogShape s = ShapeOf(g);
            tstart[g] = s.term_start;
            len[g] = s.len;
            if (!s.shaped) all_shaped.store(false);
        }
    });
    I.by_start = all_shaped.load();
    // slice capacities from the leaders' log lengths, then the slices' places in the arena
    for (uint32_t g = 0; g < G && !I.by_start; g++) {
        I.cap[g] = slice_capacity(len[g]);
        const uint64_t next = (uint64_t)I.off[g] + I.cap[g];
        if (next > 0xFFFFFFFFull) return RQ_E_RANGE;
        I.off[g + 1] = (uint32_t)next;
    }
    arena.assign(I.by_start ? 0 : I.off[G], 0);
    // the groups themselves, in parallel (they are independent objects)
    std::atomic<int> err{RQ_OK};
    std::atomic<bool> mixed{false};
    ParallelOverGroups(G, [&](uint32_t g0, uint32_t g1, uint32_t) {
        std::vector<T> row(words), terms;
        for (uint32_t g = g0; g < g1; g++) {
            LogShape shape;
            int rc = FlattenRow<T>(g, row.data(), terms, I.by_start ? &shape : nullptr);
            if (rc == RQ_OK && !I.by_start && terms.size() > I.cap[g]) rc = RQ_E_ARG;
            if (rc == RQ_OK && I.by_start && !narrow(tstart[g], wT(o_tstart, 0, g))) rc = RQ_E_RANGE;
            if (rc != RQ_OK) { err.store(rc); return; }
            for (uint32_t p = 0; p < n; p++) {
                wT(o_match, p, g) = row[p]; wT(o_vt, p, g) = row[n + p];
                wT(o_vf, p, g) = row[2 * n + p]; wT(o_ll, p, g) = row[3 * n + p];
            }
            const T *tail = row.data() + 4 * n;
            wT(o_commit, 0, g) = tail[0]; wT(o_lterm, 0, g) = tail[1];
            wT(o_cterm, 0, g) = tail[2]; wT(o_cid, 0, g) = tail[3];
            w16(o_om, g) = (uint16_t)((uint32_t)tail[5] & 0xFFFFu); w16(o_nm, g) = (uint16_t)((uint32_t)tail[5] >> 16);
            if (groups_[g].Members.size() != n) mixed.store(true);
            if (!I.by_start) {
                len[g] = (uint32_t)terms.size();
                std::memcpy(arena.data() + I.off[g], terms.data(), terms.size() * E);
                w32(o_off, g) = I.off[g];
            }
            w32(o_len, g) = len[g];
        }
    });
    if (err.load() != RQ_OK) return err.load();
    if (!I.by_start) w32(o_off, G) = I.off[G];
    mixed_ = mixed.load();
    // free space behind the slices: logs that outgrow their slice move there
    I.tail = arena.size();
    I.arena_entries = arena.size() + arena.size() / 2 + (1u << 16);
    if (I.arena_entries > 0xFFFFFFFFull) I.arena_entries = 0xFFFFFFFFull;
    if (I.by_start) I.arena_entries = 0;
    arena.resize(I.arena_entries, 0);

    rq_round_desc &d = I.desc;
    std::memset(&d, 0, sizeof d);
    d.struct_size = sizeof d; d.layout = E == 8 ? RQ_LAYOUT_I64 : RQ_LAYOUT_I32; d.what = RQ_ROUND_COMMIT | RQ_ROUND_VOTE;
    d.G = G; d.n = n; d.ld = G; d.log_terms = arena.size();
    d.tile_bytes = tile_bytes; d.out_tile_bytes = out_tile_bytes;
#define RH_TRY(x) do { int rh_rc_ = (x); if (rh_rc_ != RQ_OK) return rh_rc_; } while (0)
    char *in_dev = nullptr, *out_dev = nullptr;
    RH_TRY(I.dalloc(block.size(), (void **)&in_dev));
    RH_TRY(rq_copy_to_device(I.e, in_dev, block.data(), block.size()));
    RH_TRY(I.dalloc(ntiles * out_tile_bytes, (void **)&out_dev));
    d.match = in_dev + o_match; d.voter_term = in_dev + o_vt; d.voted_for = in_dev + o_vf;
    d.voter_last_log_term = in_dev + o_ll;
    d.commit = in_dev + o_commit; d.cand_term = in_dev + o_cterm; d.cand_id = in_dev + o_cid;
    d.log_len = (const uint32_t *)(in_dev + o_len);
    if (I.by_start) {
        d.log_term_start = in_dev + o_tstart;
    } else {
        d.leader_term = in_dev + o_lterm;
        d.log_off = (const uint32_t *)(in_dev + o_off);
        void *arena_dev = nullptr;
        RH_TRY(I.dalloc((uint64_t)arena.size() * E, &arena_dev));
        RH_TRY(rq_copy_to_device(I.e, arena_dev, arena.data(), (uint64_t)arena.size() * E));
        d.log_term = arena_dev;
    }
    I.tile_kind.clear();
    if (mixed_) {
        d.old_mask = (const uint16_t *)(in_dev + o_om);
        d.new_mask = (const uint16_t *)(in_dev + o_nm);
        // A host that keeps its groups in size order gets the fixed-size code for every tile whose
        // groups agree on their size (rq_round_desc.n_segments); any other order, or more runs than
        // the table holds, and every group is decided through its masks as before.
        const uint32_t tiles = (G + 63) / 64;
        std::vector<uint8_t> kind(tiles, 0);
        for (uint32_t t = 0; t < tiles; t++) {
            const size_t k0 = groups_[(size_t)t * 64].Members.size();
            bool same = k0 >= 1 && k0 <= 9;
            for (uint32_t g = t * 64 + 1; same && g < G && g < (t + 1) * 64; g++) same = groups_[g].Members.size() == k0;
            kind[t] = same ? (uint8_t)k0 : 0;
        }
        uint32_t runs = 1;
        for (uint32_t t = 1; t < tiles; t++) runs += kind[t] != kind[t - 1];
        if (runs <= RQ_MAX_SEGMENTS) {
            uint32_t sidx = 0;
            for (uint32_t t = 0; t < tiles; t++) {
                if (t == 0 || kind[t] != kind[t - 1]) {
                    d.seg_first_group[sidx] = t * 64;
                    d.seg_n[sidx] = kind[t];
                    sidx++;
                }
            }
            d.seg_first_group[sidx] = G;
            d.n_segments = sidx;
            I.tile_kind = std::move(kind);
        }
    }
    // state is updated in place between rounds, as the host applies it to its Nodes
    d.commit_out = const_cast<void *>(d.commit);
    d.voter_term_out = const_cast<void *>(d.voter_term);
    d.max_term = out_dev + oo_max;
    d.grant_mask = (uint16_t *)(out_dev + oo_grant);
    d.stepdown_mask = (uint16_t *)(out_dev + oo_step);
    d.elected = (uint8_t *)(out_dev + oo_elected);
    d.status_out = (uint8_t *)(out_dev + oo_status);
    const uint64_t bit_words = ((uint64_t)G + 63) / 64;
    RH_TRY(I.dalloc(bit_words * 8, (void **)&d.changed_bits));
    RH_TRY(rq_plan_round(I.e, &d, &I.plan));

    I.row_cap = G < kMaxRowsPerTransfer ? G : kMaxRowsPerTransfer;
    const uint64_t nrows = I.row_cap;
    I.term_cap = 64ull * nrows < I.arena_entries ? 64ull * nrows : I.arena_entries;
    if (I.by_start) I.term_cap = 0;                  // rows carry no terms
    // going up: group ids and term offsets (32-bit), then rows and terms (element type, 8-byte aligned)
    const uint64_t in_bytes = (nrows * 2 + 2) * 4 + (nrows * words + I.term_cap) * E, out_bytes = nrows * (8 + (n + 2) * E);
    RH_TRY(I.dalloc(in_bytes, (void **)&I.d_in));    RH_TRY(I.halloc(in_bytes, &I.h_in));
    RH_TRY(I.dalloc(out_bytes, (void **)&I.d_out));  RH_TRY(I.halloc(out_bytes, &I.h_out));
    RH_TRY(I.halloc(bit_words * 8, &I.h_cbits));
    for (uint32_t g : dirty_) dirty_mark_[g] = 0;
    dirty_.clear();
    return RQ_OK;
}

// One gathered row back into the objects: voters' terms and roles
// (raft.go:219-221), the leader's commitIndex, BecomeLeader for a winner.
template <typename T>
void ResidentDriver::ApplyRow(uint32_t g, const T *row, uint64_t flags, std::vector<uint32_t> &redo,
                              std::vector<GroupStatus> &flagged) {
    Group &grp = groups_[g];
    const uint32_t stepdown = (uint32_t)(flags >> 16) & 0xFFFFu;
    const bool elected = ((flags >> 32) & 0xFFu) != 0;
    const uint8_t status = (uint8_t)((flags >> 40) & 0xFFu);
    Node *leader_before = grp.Leader;
    if (grp.Candidate) {
        for (uint32_t p = 0; p < slots_ && p < grp.Members.size(); p++) {
            Node *m = grp.Members[p];
            m->currentTerm = row[2 + p];
            if ((stepdown >> p) & 1u) m->nodeType = Follower;
        }
        if (elected) {
            grp.Candidate->BecomeLeader();
            grp.Leader = grp.Candidate;
            grp.Candidate = nullptr;
            redo.push_back(g);            // new leader arrays, no candidate: the row goes up again
        }
    }
    if (leader_before) leader_before->commitIndex = row[0];
    // a leader a vote request demoted (raft.go:221) leads no more: its row goes up again
    // without leader arrays, so that later rounds do not advance commitIndex for a Follower
    if (grp.Leader && grp.Leader->nodeType != Leader) {
        grp.Leader = nullptr;
        redo.push_back(g);
    }
    if (status) flagged.push_back(GroupStatus{g, status});
}

int ResidentDriver::Step(Report *report) {
    // the compact layout until a value does not fit it; the wide one from then on
    if (!wide_) {
        const int rc = StepT<int32_t>(report);
        if (rc != RQ_E_RANGE) return rc;
        wide_ = true;
        if (impl_) { impl_->release(); }          // no prepared launch: the next step lays the image out afresh
    }
    return StepT<int64_t>(report);
}

template <typename T>
int ResidentDriver::StepT(Report *report) {
    constexpr uint64_t E = sizeof(T);
    Report rep;
    flagged_.clear();
    bool need_rebuild = impl_ == nullptr || impl_->plan == nullptr;
    if (!need_rebuild && dirty_.size() > impl_->row_cap) need_rebuild = true;
    if (!need_rebuild && !dirty_.empty()) {
        Impl &I = *impl_;
        const uint32_t n = I.n, r = (uint32_t)dirty_.size();
        const size_t words = 4 * (size_t)n + 7;
        // block going up, packed for r rows: group ids | term offsets (r + 1) | rows | terms
        uint32_t *gids = (uint32_t *)I.h_in, *roff = gids + r;
        T *rows = (T *)(I.h_in + (((uint64_t)r * 2 + 1) * 4 + 7) / 8 * 8), *rterms = rows + (size_t)r * words;
        std::vector<T> terms;
        uint64_t used = 0;
        // Slices that move to free arena space are booked at once (later rows of this batch must
        // not get the same space) and taken back if the batch does not go up after all: a row
        // that cannot be flattened, or a rebuild, must leave the bookkeeping as the device has it.
        struct Moved { uint32_t g, off, cap; };
        std::vector<Moved> moved;
        const uint64_t tail_before = I.tail;
        auto take_back = [&]() {
            for (auto it = moved.rbegin(); it != moved.rend(); ++it) { I.off[it->g] = it->off; I.cap[it->g] = it->cap; }
            I.tail = tail_before;
        };
        // Without an arena a row is self-contained (no slice to book), so the rows are flattened
        // on several threads first; the pass below then only checks them, in order.
        const uint32_t workers = I.by_start ? RowWorkers(r) : 1;
        std::vector<LogShape> shapes;
        std::vector<int> flat_rc;
        if (workers > 1) {
            shapes.resize(r);
            flat_rc.assign(r, RQ_OK);
            ParallelOverRows(r, workers, [&](uint32_t i0, uint32_t i1, uint32_t) {
                std::vector<T> none;
                for (uint32_t i = i0; i < i1; i++)
                    flat_rc[i] = FlattenRow<T>(dirty_[i], rows + (size_t)i * words, none, &shapes[i]);
            });
        }
        for (uint32_t i = 0; i < r && !need_rebuild; i++) {
            const uint32_t g = dirty_[i];
            T *row = rows + (size_t)i * words;
            LogShape shape;
            if (workers > 1) {
                if (flat_rc[i]) { take_back(); return flat_rc[i]; }
                shape = shapes[i];
            } else if (const int frc = FlattenRow<T>(g, row, terms, I.by_start ? &shape : nullptr)) { take_back(); return frc; }
            if (!mixed_ && groups_[g].Members.size() != n) { need_rebuild = true; break; }
            // a group that left the size its segment was launched under: the table is made afresh
            if (!I.tile_kind.empty() && I.tile_kind[g >> 6] && groups_[g].Members.size() != I.tile_kind[g >> 6]) {
                need_rebuild = true; break;
            }
            if (I.by_start) {
                // a log that lost Raft's shape needs the arena back: everything is laid out afresh
                if (!shape.shaped) { need_rebuild = true; break; }
                if (!narrow(shape.term_start, row[4 * n + 6])) { take_back(); return RQ_E_RANGE; }
                gids[i] = g;
                roff[i] = (uint32_t)used;            // lengths only: no terms travel
                used += shape.len;
                if (used > 0xFFFFFFFFull) { need_rebuild = true; break; }
                continue;
            }
            if (terms.size() > I.cap[g]) {
                // the log outgrew its slice: it moves to free space with twice the room;
                // when that runs out everything is laid out afresh
                const uint64_t want = 2 * (uint64_t)terms.size() + 4;
                if (I.tail + want > I.arena_entries) { need_rebuild = true; break; }
                moved.push_back(Moved{g, I.off[g], I.cap[g]});
                I.off[g] = (uint32_t)I.tail;
                I.cap[g] = (uint32_t)want;
                I.tail += want;
            }
            if (used + terms.size() > I.term_cap) { need_rebuild = true; break; }
            row[4 * n + 6] = (T)I.off[g];
            gids[i] = g;
            roff[i] = (uint32_t)used;
            std::memcpy(rterms + used, terms.data(), terms.size() * E);
            used += terms.size();
        }
        if (need_rebuild) take_back();
        if (!need_rebuild) {
            roff[r] = (uint32_t)used;
            const uint64_t bytes = (uint64_t)((const char *)rterms - I.h_in) + (I.by_start ? 0 : used) * E;
            if (const int crc = rq_copy_to_device(I.e, I.d_in, I.h_in, bytes)) { take_back(); return crc; }
            const char *d = I.d_in;
            const int src = rq_groups_scatter_dev(I.e, nullptr, &I.desc, r, (const uint32_t *)d,
                                                  d + ((const char *)rows - I.h_in), (const uint32_t *)(d + ((const char *)roff - I.h_in)),
                                                  d + ((const char *)rterms - I.h_in), nullptr);
            if (src) { take_back(); return src; }
            rep.uploaded = r;
            for (uint32_t g : dirty_) dirty_mark_[g] = 0;
            dirty_.clear();
        }
    }
    if (need_rebuild) {
        RH_TRY(RebuildT<T>());
        rep.rebuilt = true;
        rep.uploaded = (uint32_t)groups_.size();
    }
    Impl &I = *impl_;
    RH_TRY(rq_plan_launch(I.plan, nullptr));
    const uint64_t bit_words = ((uint64_t)I.G + 63) / 64;
    RH_TRY(rq_copy_from_device(I.e, I.h_cbits, I.desc.changed_bits, bit_words * 8));   // ordered after the round
    // the groups the round changed or flagged, in chunks of the staging size; block
    // coming down, packed for r rows: flags (r x 8 bytes) | rows
    uint32_t r = 0;
    uint32_t *gids = (uint32_t *)I.h_in;
    auto flush = [&]() -> int {
        if (r == 0) return RQ_OK;
        RH_TRY(rq_copy_to_device(I.e, I.d_in, I.h_in, (uint64_t)r * 4));
        RH_TRY(rq_groups_gather_dev(I.e, nullptr, &I.desc, r, (const uint32_t *)I.d_in, I.d_out + (uint64_t)r * 8,
                                    (uint64_t *)I.d_out));
        RH_TRY(rq_copy_from_device(I.e, I.h_out, I.d_out, (uint64_t)r * (8 + (I.n + 2) * E)));
        const uint64_t *flags = (const uint64_t *)I.h_out;
        const T *orows = (const T *)(I.h_out + (uint64_t)r * 8);
        const uint32_t workers = RowWorkers(r);
        std::vector<std::vector<uint32_t>> redo(workers);
        std::vector<std::vector<GroupStatus>> flagged(workers);
        ParallelOverRows(r, workers, [&](uint32_t i0, uint32_t i1, uint32_t w) {
            for (uint32_t i = i0; i < i1; i++)
                ApplyRow<T>(gids[i], orows + (size_t)i * (I.n + 2), flags[i], redo[w], flagged[w]);
        });
        for (uint32_t w = 0; w < workers; w++) {          // in row order, as one thread would have left them
            for (uint32_t g : redo[w]) MarkDirty(g);
            flagged_.insert(flagged_.end(), flagged[w].begin(), flagged[w].end());
        }
        rep.changed += r;
        r = 0;
        return RQ_OK;
    };
    for (uint64_t w = 0; w < bit_words; w++) {
        uint64_t bits = I.h_cbits[w];
        while (bits) {
            const uint32_t g = (uint32_t)(w * 64 + (uint64_t)__builtin_ctzll(bits));
            bits &= bits - 1;
            gids[r++] = g;
            if (r == I.row_cap) RH_TRY(flush());
        }
    }
    RH_TRY(flush());
    if (report) *report = rep;
    return RQ_OK;
}
#undef RH_TRY

}  // namespace raft
