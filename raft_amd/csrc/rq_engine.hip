// rq_engine.hip -- the rq_* C-ABI (include/rq.h): engine object, device
// buffers, host<->device staging and launch plumbing.  All compute is in
// rq_kernels.hip; nothing here evaluates a quorum decision on the CPU.
#include "../../include/rq.h"
#include "rq_kernels.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

struct rq_plan {
    rq_engine *engine;
    uint32_t n;
    bool joint;
    bool wide;          // args64 is the live member, else args
    int variant;
    int what;
    rq_round_args args;
    rq_round64_args args64;
    bool segmented;     // segs is a checked table
    rq_round_segs segs;
    // rq_plan_rounds: `images` rounds decided by one grid.  The argument structs of all
    // images live in host memory (many) and, the same bytes, in device memory (table).
    uint32_t images;    // 0: a single round (the fields above)
    void *many;         // rq_round_args[images] or rq_round64_args[images]
    void *table;        // device copy
    uint32_t image_n[RQ_MAX_SEGS];
    bool image_joint[RQ_MAX_SEGS];
};

struct rq_engine {
    int device = 0;
    uint32_t max_groups = 0, max_slots = 0, ld = 0;
    uint64_t max_log_terms = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::atomic<int> last_hip{0};
    char *slab = nullptr;
    char *slab64 = nullptr;   // staging of the 64-bit host-buffer entry points (made on first use)
    // planes [max_slots][ld]
    int32_t *match = nullptr, *voter_term = nullptr, *voted_for = nullptr, *last_log_term = nullptr;
    // per group [ld]
    int32_t *commit = nullptr, *leader_term = nullptr, *cand_term = nullptr, *cand_id = nullptr;
    int32_t *max_term = nullptr;
    uint16_t *grant = nullptr, *stepdown = nullptr, *old_mask = nullptr, *new_mask = nullptr;
    uint8_t *elected = nullptr, *status = nullptr;
    uint32_t *log_off = nullptr;  // [ld + 1]
    int32_t *log_term = nullptr;  // [max_log_terms]
    // grow-only scratch for entry points whose shapes the fixed buffers do not fit
    char *aux = nullptr;
    size_t aux_bytes = 0;
    // resident round (rq_upload / rq_step / rq_download)
    bool resident = false;
    uint32_t res_G = 0, res_n = 0;
    // rq_step(rounds >= kGraphMinRounds): the rounds are captured once into a HIP
    // graph (a launch-bound inner loop when a round is only microseconds long)
    hipGraphExec_t step_graph = nullptr;
    uint32_t graph_G = 0, graph_n = 0, graph_rounds = 0;
};

namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

#define RQ_HIP(e, call)                                    \
    do {                                                   \
        hipError_t rq_err_ = (call);                       \
        if (rq_err_ != hipSuccess) {                       \
            (e)->last_hip.store((int)rq_err_);             \
            return RQ_E_HIP;                               \
        }                                                  \
    } while (0)

int check_shape(const rq_engine *e, uint32_t G, uint32_t n, uint32_t ld) {
    if (!e) return RQ_E_ARG;
    if (n < 1 || n > RQ_MAX_SLOTS || ld < G) return RQ_E_ARG;
    if (G > e->max_groups || n > e->max_slots) return RQ_E_RANGE;
    return RQ_OK;
}

// copy n planes of G elements, host stride ld -> device stride e->ld
template <typename T>
int h2d_planes(rq_engine *e, T *dst, const T *src, uint32_t G, uint32_t n, uint32_t ld) {
    if (G == 0) return RQ_OK;
    RQ_HIP(e, hipMemcpy2DAsync(dst, (size_t)e->ld * sizeof(T), src, (size_t)ld * sizeof(T),
                               (size_t)G * sizeof(T), n, hipMemcpyHostToDevice, e->stream));
    return RQ_OK;
}
template <typename T>
int d2h_planes(rq_engine *e, T *dst, const T *src, uint32_t G, uint32_t n, uint32_t ld) {
    if (G == 0) return RQ_OK;
    RQ_HIP(e, hipMemcpy2DAsync(dst, (size_t)ld * sizeof(T), src, (size_t)e->ld * sizeof(T),
                               (size_t)G * sizeof(T), n, hipMemcpyDeviceToHost, e->stream));
    return RQ_OK;
}
int h2d(rq_engine *e, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return RQ_OK;
    RQ_HIP(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
    return RQ_OK;
}
int d2h(rq_engine *e, void *dst, const void *src, size_t bytes) {
    if (bytes == 0 || dst == nullptr) return RQ_OK;
    RQ_HIP(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
    return RQ_OK;
}

#define RQ_TRY(x) do { int rq_rc_ = (x); if (rq_rc_ != RQ_OK) return rq_rc_; } while (0)

// A host-buffer entry point that fails after it has enqueued copies must not
// return while a DMA into or out of the caller's buffers can still be in
// flight: every exit path that was not disarmed drains the engine's stream.
struct drain_unless_ok {
    rq_engine *e;
    bool armed = true;
    explicit drain_unless_ok(rq_engine *e_) : e(e_) {}
    ~drain_unless_ok() { if (armed) (void)hipStreamSynchronize(e->stream); }
    int ok() { armed = false; return RQ_OK; }
};

// The engine's staging buffers of one element width.  The 32-bit set is part
// of the slab made at create time; the 64-bit set is allocated by the first
// 64-bit host-buffer call and lives until the engine is destroyed.
template <typename T> struct staging {
    T *match, *voter_term, *voted_for, *last_log_term;            // planes [max_slots][ld]
    T *commit, *leader_term, *cand_term, *cand_id, *max_term;     // per group [ld]
    T *log_base;                                                  // per group [ld], 64-bit set only
    T *log_term;                                                  // [max_log_terms]
};

int get_staging(rq_engine *e, staging<int32_t> &s) {
    s = staging<int32_t>{e->match, e->voter_term, e->voted_for, e->last_log_term, e->commit,
                         e->leader_term, e->cand_term, e->cand_id, e->max_term, nullptr, e->log_term};
    return RQ_OK;
}

int get_staging(rq_engine *e, staging<int64_t> &s) {
    const size_t plane = up((size_t)e->max_slots * e->ld * 8), vec = up((size_t)e->ld * 8);
    const size_t arena = up((size_t)e->max_log_terms * 8 + 8);
    if (!e->slab64) {
        const size_t total = 4 * plane + 6 * vec + arena;
        if (hipMalloc((void **)&e->slab64, total) != hipSuccess) { e->slab64 = nullptr; return RQ_E_NOMEM; }
        RQ_HIP(e, hipMemsetAsync(e->slab64, 0, total, e->stream));
    }
    char *p = e->slab64;
    auto take = [&p](size_t bytes) { char *r = p; p += bytes; return (int64_t *)r; };
    s.match = take(plane); s.voter_term = take(plane); s.voted_for = take(plane); s.last_log_term = take(plane);
    s.commit = take(vec); s.leader_term = take(vec); s.cand_term = take(vec); s.cand_id = take(vec);
    s.max_term = take(vec); s.log_base = take(vec);
    s.log_term = take(arena);
    return RQ_OK;
}

// The CSR arena of one call: validates monotone offsets on the host copy the
// caller already holds (the kernel indexes the arena with them).
template <typename T>
int stage_log(rq_engine *e, const staging<T> &s, uint32_t G, const uint32_t *log_off, const T *log_term) {
    if (!log_off || (!log_term && G && log_off[G] != 0)) return RQ_E_ARG;
    for (uint32_t g = 0; g < G; g++)
        if (log_off[g + 1] < log_off[g]) return RQ_E_ARG;
    const uint64_t total = G ? (uint64_t)log_off[G] : 0;
    if (total > e->max_log_terms) return RQ_E_RANGE;
    RQ_TRY(h2d(e, e->log_off, log_off, ((size_t)G + 1) * 4));
    RQ_TRY(h2d(e, s.log_term, log_term, (size_t)total * sizeof(T)));
    return RQ_OK;
}

int check_masks(uint32_t G, uint32_t m, const uint16_t *old_mask, const uint16_t *new_mask) {
    if (!old_mask || !new_mask) return RQ_E_ARG;
    const uint32_t lim = m >= 16 ? 0xFFFFu : ((1u << m) - 1u);
    for (uint32_t g = 0; g < G; g++)
        if (!(old_mask[g] & lim) || !(new_mask[g] & lim)) return RQ_E_MASK;
    return RQ_OK;
}

template <typename T>
struct host_round_t {
    uint32_t G, n, ld;
    bool joint;
    const uint16_t *old_mask, *new_mask;
    // inputs
    const T *match, *commit, *leader_term;
    const uint32_t *log_off;
    const T *log_term;
    const T *log_base;   // 64-bit entry points only; may be null
    const T *voter_term, *voted_for, *last_log_term, *cand_term, *cand_id;
    // outputs
    T *commit_out, *voter_term_out;
    uint16_t *grant_mask, *stepdown_mask;
    uint8_t *elected;
    T *max_term;
    uint8_t *status_out;
};
typedef host_round_t<int32_t> host_round;

template <typename T>
rq_round_args_t<T> device_args(rq_engine *e, const staging<T> &s, uint32_t G, uint32_t m, bool joint) {
    rq_round_args_t<T> a;
    std::memset(&a, 0, sizeof a);
    a.G = G; a.ld = e->ld; a.m = m;
    a.match = s.match; a.commit = s.commit; a.leader_term = s.leader_term;
    a.log_off = e->log_off; a.log_term = s.log_term;
    a.log_terms = e->max_log_terms; a.log_term_bits = 8 * sizeof(T);
    a.voter_term = s.voter_term; a.voted_for = s.voted_for;
    a.last_log_term = s.last_log_term; a.cand_term = s.cand_term; a.cand_id = s.cand_id;
    a.old_mask = joint ? e->old_mask : nullptr;
    a.new_mask = joint ? e->new_mask : nullptr;
    a.commit_out = s.commit; a.voter_term_out = s.voter_term;  // in place on device
    a.grant_mask = e->grant; a.stepdown_mask = e->stepdown; a.elected = e->elected;
    a.max_term = s.max_term; a.status_out = e->status; a.elected_bits = nullptr;
    return a;
}

rq_round_args device_args(rq_engine *e, uint32_t G, uint32_t m, bool joint) {
    staging<int32_t> s;
    (void)get_staging(e, s);
    return device_args(e, s, G, m, joint);
}

template <typename T>
int stage_inputs(rq_engine *e, const staging<T> &s, const host_round_t<T> &h, int what) {
    const size_t E = sizeof(T);
    if (h.joint) {
        RQ_TRY(check_masks(h.G, h.n, h.old_mask, h.new_mask));
        RQ_TRY(h2d(e, e->old_mask, h.old_mask, (size_t)h.G * 2));
        RQ_TRY(h2d(e, e->new_mask, h.new_mask, (size_t)h.G * 2));
    }
    if (what & RQ_WHAT_COMMIT) {
        if (!h.match || !h.commit || !h.leader_term) return RQ_E_ARG;
        RQ_TRY(stage_log(e, s, h.G, h.log_off, h.log_term));
        RQ_TRY(h2d_planes(e, s.match, h.match, h.G, h.n, h.ld));
        RQ_TRY(h2d(e, s.commit, h.commit, (size_t)h.G * E));
        RQ_TRY(h2d(e, s.leader_term, h.leader_term, (size_t)h.G * E));
        if (h.log_base) RQ_TRY(h2d(e, s.log_base, h.log_base, (size_t)h.G * E));
    }
    if (what & RQ_WHAT_VOTE) {
        if (!h.voter_term || !h.voted_for || !h.last_log_term || !h.cand_term || !h.cand_id)
            return RQ_E_ARG;
        RQ_TRY(h2d_planes(e, s.voter_term, h.voter_term, h.G, h.n, h.ld));
        RQ_TRY(h2d_planes(e, s.voted_for, h.voted_for, h.G, h.n, h.ld));
        RQ_TRY(h2d_planes(e, s.last_log_term, h.last_log_term, h.G, h.n, h.ld));
        RQ_TRY(h2d(e, s.cand_term, h.cand_term, (size_t)h.G * E));
        RQ_TRY(h2d(e, s.cand_id, h.cand_id, (size_t)h.G * E));
    }
    return RQ_OK;
}

int stage_inputs(rq_engine *e, const host_round &h, int what) {
    staging<int32_t> s;
    (void)get_staging(e, s);
    return stage_inputs(e, s, h, what);
}

template <typename T>
int fetch_outputs(rq_engine *e, const staging<T> &s, const host_round_t<T> &h, int what) {
    const size_t E = sizeof(T);
    if (what & RQ_WHAT_COMMIT) RQ_TRY(d2h(e, h.commit_out, s.commit, (size_t)h.G * E));
    if (what & RQ_WHAT_VOTE) {
        // Joint rounds: a non-member slot on the device still holds the staged
        // input (the kernel leaves it alone or rewrites the value it loaded), so
        // whole-plane copies hand the caller its own value back for those slots.
        RQ_TRY(d2h_planes(e, h.voter_term_out, s.voter_term, h.G, h.n, h.ld));
        RQ_TRY(d2h(e, h.grant_mask, e->grant, (size_t)h.G * 2));
        RQ_TRY(d2h(e, h.stepdown_mask, e->stepdown, (size_t)h.G * 2));
        RQ_TRY(d2h(e, h.elected, e->elected, (size_t)h.G));
        RQ_TRY(d2h(e, h.max_term, s.max_term, (size_t)h.G * E));
    }
    RQ_TRY(d2h(e, h.status_out, e->status, (size_t)h.G));
    return RQ_OK;
}

template <typename T>
int run_host_round(rq_engine *e, const host_round_t<T> &h, int what) {
    RQ_TRY(check_shape(e, h.G, h.n, h.ld));
    if (!(what & (RQ_WHAT_COMMIT | RQ_WHAT_VOTE)) || (what & ~(RQ_WHAT_COMMIT | RQ_WHAT_VOTE))) return RQ_E_ARG;
    if ((what & RQ_WHAT_COMMIT) && !h.commit_out) return RQ_E_ARG;
    if ((what & RQ_WHAT_VOTE) &&
        (!h.voter_term_out || !h.grant_mask || !h.stepdown_mask || !h.elected || !h.max_term))
        return RQ_E_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    RQ_HIP(e, hipSetDevice(e->device));
    e->resident = false;
    if (h.G == 0) return RQ_OK;
    drain_unless_ok guard(e);
    staging<T> s;
    RQ_TRY(get_staging(e, s));
    RQ_TRY(stage_inputs(e, s, h, what));
    rq_round_args_t<T> a = device_args(e, s, h.G, h.n, h.joint);
    if ((what & RQ_WHAT_COMMIT) && h.log_base) a.log_base = s.log_base;
    RQ_HIP(e, rq_launch_round(a, h.n, h.joint, what, 0, e->stream));
    RQ_TRY(fetch_outputs(e, s, h, what));
    RQ_HIP(e, hipStreamSynchronize(e->stream));
    return guard.ok();
}

// ---- descriptors -> typed kernel arguments ---------------------------------

template <typename T>
int round_args_from_desc(const rq_round_desc &d, rq_round_args_t<T> &a) {
    const uint32_t what = d.what;
    if (!(what & (RQ_WHAT_COMMIT | RQ_WHAT_VOTE)) || (what & ~(uint32_t)(RQ_WHAT_COMMIT | RQ_WHAT_VOTE))) return RQ_E_ARG;
    if (d.n < 1 || d.n > RQ_MAX_SLOTS || (d.ld < d.G && !d.tile_bytes) || d.G > 0x7FFFFFFFu || d.reserved != 0) return RQ_E_ARG;
    if ((d.old_mask == nullptr) != (d.new_mask == nullptr)) return RQ_E_ARG;
    if (what & RQ_WHAT_COMMIT) {
        if (!d.match || !d.commit || !d.commit_out) return RQ_E_ARG;
        if (d.log_term_start) {                       // no arena lookup: index of the term's first entry + log length
            if (!d.log_len) return RQ_E_ARG;
        } else {
            if (!d.leader_term || !d.log_off) return RQ_E_ARG;
            if (!d.log_term && d.log_terms != 0) return RQ_E_ARG;
        }
        if (d.log_terms > 0xFFFFFFFFull) return RQ_E_ARG;
        const uint32_t tb = d.log_term_bits;
        if (tb != 0 && tb != 8 && tb != 16 && tb != 32 && tb != 8 * sizeof(T)) return RQ_E_ARG;
        if (tb > 8 * sizeof(T)) return RQ_E_ARG;
    }
    if (what & RQ_WHAT_VOTE) {
        if (!d.voter_term || !d.voted_for || !d.voter_last_log_term || !d.cand_term || !d.cand_id ||
            !d.voter_term_out || !d.grant_mask || !d.stepdown_mask || !d.elected || !d.max_term)
            return RQ_E_ARG;
    }
    const uint32_t sb = d.summary_bits ? d.summary_bits : 16;
    if (d.commit_summary) {
        if (sb != 16 && sb != 8 && sb != 4) return RQ_E_ARG;
        const uintptr_t p = reinterpret_cast<uintptr_t>(d.commit_summary);
        if ((sb == 16 && (p & 1)) || (sb == 4 && (p & 7))) return RQ_E_ARG;
    }
    if ((d.tile_bytes & 7u) || (d.out_tile_bytes & 7u) || ((d.out_tile_bytes != 0) != (d.tile_bytes != 0))) return RQ_E_ARG;
    std::memset(&a, 0, sizeof a);
    a.G = d.G; a.ld = d.ld; a.m = d.n; a.tile_bytes = d.tile_bytes; a.out_tile_bytes = d.out_tile_bytes;
    a.old_mask = d.old_mask; a.new_mask = d.new_mask;
    a.match = (const T *)d.match; a.commit = (const T *)d.commit; a.leader_term = (const T *)d.leader_term;
    a.log_off = d.log_off; a.log_term = d.log_term;
    a.log_base = (const T *)d.log_base; a.log_term_base = (const T *)d.log_term_base;
    a.log_len = d.log_len;
    a.log_term_start = (const T *)d.log_term_start;
    a.log_terms = d.log_terms;
    a.log_term_bits = d.log_term_bits ? d.log_term_bits : (uint32_t)(8 * sizeof(T));
    a.voter_term = (const T *)d.voter_term; a.voted_for = (const T *)d.voted_for;
    a.last_log_term = (const T *)d.voter_last_log_term;
    a.cand_term = (const T *)d.cand_term; a.cand_id = (const T *)d.cand_id;
    a.commit_out = (T *)d.commit_out; a.voter_term_out = (T *)d.voter_term_out;
    a.grant_mask = d.grant_mask; a.stepdown_mask = d.stepdown_mask; a.elected = d.elected;
    a.max_term = (T *)d.max_term; a.status_out = d.status_out; a.elected_bits = d.elected_bits;
    a.changed_bits = d.changed_bits;
    a.commit_lo16 = (uint16_t *)d.commit_summary; a.summary_base = (const T *)d.summary_base;
    a.summary_escapes = d.summary_escapes; a.summary_bits = sb;
    if (reinterpret_cast<uintptr_t>(d.start_signal) & 7) return RQ_E_ARG;
    a.start_signal = d.start_signal;
    return RQ_OK;
}

// The descriptor as this build knows it: a caller built against 0.2.x headers
// hands over the struct without the segment table.
int full_desc(const rq_round_desc *d, rq_round_desc &full) {
    if (!d || (d->struct_size != sizeof(rq_round_desc) && d->struct_size != RQ_ROUND_DESC_SIZE_V2)) return RQ_E_ARG;
    std::memset(&full, 0, sizeof full);
    std::memcpy(&full, d, d->struct_size);
    full.struct_size = sizeof(rq_round_desc);
    return RQ_OK;
}

int segs_from_desc(const rq_round_desc &d, rq_plan &p) {
    p.segmented = false;
    std::memset(&p.segs, 0, sizeof p.segs);
    if (d.n_segments == 0) return RQ_OK;
    if (d.n_segments > RQ_MAX_SEGMENTS) return RQ_E_ARG;
    static_assert(RQ_MAX_SEGMENTS == RQ_MAX_SEGS, "segment table sizes");
    p.segs.count = d.n_segments;
    if (d.seg_first_group[0] != 0 || d.seg_first_group[d.n_segments] != d.G) return RQ_E_ARG;
    for (uint32_t i = 0; i < d.n_segments; i++) {
        if (d.seg_first_group[i + 1] < d.seg_first_group[i]) return RQ_E_ARG;
        if (i + 1 < d.n_segments && (d.seg_first_group[i + 1] & 63u)) return RQ_E_ARG;
        if (d.seg_n[i] > 9 || d.seg_n[i] > d.n) return RQ_E_ARG;
        p.segs.first_group[i] = d.seg_first_group[i];
        p.segs.n[i] = d.seg_n[i];
    }
    p.segs.first_group[d.n_segments] = d.G;
#ifdef RQ_BUILD_PLAIN
    if ((d.variant & RQ_VARIANT_PLAIN) && !d.old_mask) return RQ_E_ARG;   // that build decides through the masks
#endif
    p.segmented = true;
    return RQ_OK;
}

int plan_from_desc(rq_engine *e, const rq_round_desc *dp, rq_plan &p) {
    rq_round_desc full;
    if (!e) return RQ_E_ARG;
    RQ_TRY(full_desc(dp, full));
    const rq_round_desc *d = &full;
    p.engine = e; p.n = d->n; p.joint = d->old_mask != nullptr; p.variant = d->variant; p.what = (int)d->what;
    p.images = 0; p.many = nullptr; p.table = nullptr;
    std::memset(&p.args, 0, sizeof p.args);
    std::memset(&p.args64, 0, sizeof p.args64);
    RQ_TRY(segs_from_desc(*d, p));
    if (d->layout == RQ_LAYOUT_I32) { p.wide = false; return round_args_from_desc(*d, p.args); }
    if (d->layout == RQ_LAYOUT_I64) { p.wide = true; return round_args_from_desc(*d, p.args64); }
    return RQ_E_ARG;
}

int launch_plan(const rq_plan &p, void *stream) {
    rq_engine *e = p.engine;
    RQ_HIP(e, hipSetDevice(e->device));
    if (p.images) {
        hipError_t err = p.wide
            ? rq_launch_rounds((const rq_round64_args *)p.many, p.image_n, p.image_joint, p.images,
                               (const rq_round64_args *)p.table, p.what, p.variant, (hipStream_t)stream)
            : rq_launch_rounds((const rq_round_args *)p.many, p.image_n, p.image_joint, p.images,
                               (const rq_round_args *)p.table, p.what, p.variant, (hipStream_t)stream);
        if (err == hipErrorInvalidValue) return RQ_E_ARG;
        RQ_HIP(e, err);
        return RQ_OK;
    }
    const rq_round_segs *segs = p.segmented ? &p.segs : nullptr;
    hipError_t err = p.wide ? rq_launch_round(p.args64, p.n, p.joint, p.what, p.variant, (hipStream_t)stream, segs)
                            : rq_launch_round(p.args, p.n, p.joint, p.what, p.variant, (hipStream_t)stream, segs);
    if (err == hipErrorInvalidValue) return RQ_E_ARG;
    RQ_HIP(e, err);
    return RQ_OK;
}

template <typename T>
int follower_clamp_host(rq_engine *e, uint64_t count, const T *leader_commit, const T *commit,
                        const T *last_new_index, T *commit_out) {
    if (!e || !leader_commit || !commit || !last_new_index || !commit_out) return RQ_E_ARG;
    if (count > (uint64_t)e->max_slots * e->ld) return RQ_E_RANGE;
    std::lock_guard<std::mutex> lk(e->mu);
    RQ_HIP(e, hipSetDevice(e->device));
    e->resident = false;
    if (count == 0) return RQ_OK;
    drain_unless_ok guard(e);
    staging<T> s;
    RQ_TRY(get_staging(e, s));
    const size_t bytes = (size_t)count * sizeof(T);
    RQ_TRY(h2d(e, s.match, leader_commit, bytes));
    RQ_TRY(h2d(e, s.voter_term, commit, bytes));
    RQ_TRY(h2d(e, s.voted_for, last_new_index, bytes));
    RQ_HIP(e, rq_launch_follower_clamp(count, s.match, s.voter_term, s.voted_for, s.last_log_term, e->stream));
    RQ_TRY(d2h(e, commit_out, s.last_log_term, bytes));
    RQ_HIP(e, hipStreamSynchronize(e->stream));
    return guard.ok();
}

template <typename T>
int append_args_ok(const rq_append_args_t<T> &a) {
    if (a.n < 1 || a.n > RQ_MAX_SLOTS || (a.ld < a.G && !a.tile_bytes)) return RQ_E_ARG;
    if ((a.tile_bytes & 7u) || (a.out_tile_bytes & 7u) || ((a.tile_bytes != 0) != (a.out_tile_bytes != 0))) return RQ_E_ARG;
    if (!a.follower_term || !a.leader_term || !a.prev_log_index || !a.prev_log_term ||
        !a.flog_off || !a.flog_term || !a.follower_term_out || !a.success_mask || !a.stepdown_mask)
        return RQ_E_ARG;
    const int clamp_ptrs = (a.leader_commit != nullptr) + (a.last_new_index != nullptr) + (a.follower_commit_io != nullptr);
    if (clamp_ptrs != 0 && clamp_ptrs != 3) return RQ_E_ARG;
    return RQ_OK;
}

// h holds the caller's HOST pointers; stages everything into the grow-only aux
// buffer, launches, copies the outputs back.
template <typename T>
int append_check_host(rq_engine *e, const rq_append_args_t<T> &h) {
    if (!e) return RQ_E_ARG;
    RQ_TRY(append_args_ok(h));
    const uint32_t G = h.G, n = h.n, ld = h.ld;
    constexpr size_t E = sizeof(T);
    if (G > e->max_groups) return RQ_E_RANGE;
    if (G == 0) return RQ_OK;
    const size_t slots = (size_t)n * ld;
    for (size_t i = 0; i < slots; i++)
        if (h.flog_off[i + 1] < h.flog_off[i]) return RQ_E_ARG;
    const size_t arena = h.flog_off[slots];
    // device image: 7 planes, offsets, arena, 2 masks (+1 input mask), status, leader terms
    const size_t plane = up(slots * E);
    const size_t total = 7 * plane + up((slots + 1) * 4) + up(arena * E + E) + 3 * up((size_t)G * 2) +
                         up(G) + up((size_t)G * E);
    std::lock_guard<std::mutex> lk(e->mu);
    RQ_HIP(e, hipSetDevice(e->device));
    if (total > e->aux_bytes) {
        if (e->aux) { RQ_HIP(e, hipFree(e->aux)); e->aux = nullptr; e->aux_bytes = 0; }
        if (hipMalloc((void **)&e->aux, total) != hipSuccess) return RQ_E_NOMEM;
        e->aux_bytes = total;
    }
    drain_unless_ok guard(e);
    char *p = e->aux;
    auto take = [&p](size_t bytes) { char *r = p; p += bytes; return r; };
    T *d_ft = (T *)take(plane), *d_pi = (T *)take(plane), *d_pt = (T *)take(plane);
    T *d_nn = (T *)take(plane), *d_mi = (T *)take(plane), *d_ni = (T *)take(plane);
    T *d_fo = (T *)take(plane);
    uint32_t *d_off = (uint32_t *)take(up((slots + 1) * 4));
    T *d_arena = (T *)take(up(arena * E + E));
    uint16_t *d_act = (uint16_t *)take(up((size_t)G * 2));
    uint16_t *d_ok = (uint16_t *)take(up((size_t)G * 2)), *d_sd = (uint16_t *)take(up((size_t)G * 2));
    uint8_t *d_st = (uint8_t *)take(up(G));
    T *d_lt = (T *)take(up((size_t)G * E));
    RQ_TRY(h2d(e, d_ft, h.follower_term, slots * E));
    RQ_TRY(h2d(e, d_pi, h.prev_log_index, slots * E));
    RQ_TRY(h2d(e, d_pt, h.prev_log_term, slots * E));
    if (h.n_new) RQ_TRY(h2d(e, d_nn, h.n_new, slots * E));
    if (h.match_index_io) RQ_TRY(h2d(e, d_mi, h.match_index_io, slots * E));
    if (h.next_index_io) RQ_TRY(h2d(e, d_ni, h.next_index_io, slots * E));
    // slots that receive no RPC keep the caller's term in the output plane
    RQ_TRY(h2d(e, d_fo, h.follower_term, slots * E));
    RQ_TRY(h2d(e, d_off, h.flog_off, (slots + 1) * 4));
    RQ_TRY(h2d(e, d_arena, h.flog_term, arena * E));
    if (h.active_mask) RQ_TRY(h2d(e, d_act, h.active_mask, (size_t)G * 2));
    RQ_TRY(h2d(e, d_lt, h.leader_term, (size_t)G * E));
    rq_append_args_t<T> a = h;
    a.active_mask = h.active_mask ? d_act : nullptr;
    a.follower_term = d_ft; a.leader_term = d_lt; a.prev_log_index = d_pi; a.prev_log_term = d_pt;
    a.n_new = h.n_new ? d_nn : nullptr; a.flog_off = d_off; a.flog_term = d_arena;
    a.flog_terms = arena; a.flog_term_bits = 8 * E;
    a.follower_term_out = d_fo; a.success_mask = d_ok; a.stepdown_mask = d_sd;
    a.match_index_io = h.match_index_io ? d_mi : nullptr;
    a.next_index_io = h.next_index_io ? d_ni : nullptr;
    a.status_out = d_st;
    RQ_HIP(e, rq_launch_append_check(a, nullptr, e->stream));
    RQ_TRY(d2h(e, h.follower_term_out, d_fo, slots * E));
    RQ_TRY(d2h(e, h.success_mask, d_ok, (size_t)G * 2));
    RQ_TRY(d2h(e, h.stepdown_mask, d_sd, (size_t)G * 2));
    RQ_TRY(d2h(e, h.match_index_io, d_mi, slots * E));
    RQ_TRY(d2h(e, h.next_index_io, d_ni, slots * E));
    RQ_TRY(d2h(e, h.status_out, d_st, G));
    RQ_HIP(e, hipStreamSynchronize(e->stream));
    return guard.ok();
}

template <typename T>
int append_check_dev(rq_engine *e, void *stream, const rq_append_desc &d) {
    rq_append_args_t<T> a;
    std::memset(&a, 0, sizeof a);
    a.G = d.G; a.ld = d.ld; a.n = d.n; a.active_mask = d.active_mask;
    a.follower_term = (const T *)d.follower_term; a.leader_term = (const T *)d.leader_term;
    a.prev_log_index = (const T *)d.prev_log_index; a.prev_log_term = (const T *)d.prev_log_term;
    a.n_new = (const T *)d.n_new; a.flog_off = d.flog_off; a.flog_term = d.flog_term;
    a.flog_base = (const T *)d.flog_base; a.flog_term_base = (const T *)d.flog_term_base;
    a.flog_terms = d.flog_terms;
    a.flog_term_bits = d.flog_term_bits ? d.flog_term_bits : (uint32_t)(8 * sizeof(T));
    a.follower_term_out = (T *)d.follower_term_out; a.success_mask = d.success_mask;
    a.stepdown_mask = d.stepdown_mask; a.match_index_io = (T *)d.match_index_io;
    a.next_index_io = (T *)d.next_index_io; a.status_out = d.status_out;
    a.tile_bytes = d.tile_bytes; a.out_tile_bytes = d.out_tile_bytes;
    a.leader_commit = (const T *)d.leader_commit; a.last_new_index = (const T *)d.last_new_index;
    a.follower_commit_io = (T *)d.follower_commit_io;
    if (a.flog_term_bits > 8 * sizeof(T) || d.flog_terms > 0xFFFFFFFFull) return RQ_E_ARG;
    if (!a.flog_term && d.flog_terms == 0) a.flog_term = a.flog_off;   // empty arena: never dereferenced
    RQ_TRY(append_args_ok(a));
    rq_round_args_t<T> r;
    const rq_round_args_t<T> *rp = nullptr;
    if (d.commit_round) {                          // the leader's commit rule in the same launch
        rq_round_desc c;
        RQ_TRY(full_desc(d.commit_round, c));
        if (c.layout != d.layout || c.what != RQ_ROUND_COMMIT || c.n_segments != 0) return RQ_E_ARG;
        RQ_TRY(round_args_from_desc(c, r));
        rp = &r;
    }
    RQ_HIP(e, hipSetDevice(e->device));
    hipError_t err = rq_launch_append_check(a, rp, (hipStream_t)stream);
    if (err == hipErrorInvalidValue) return RQ_E_ARG;
    RQ_HIP(e, err);
    return RQ_OK;
}

template <typename T>
rq_append_args_t<T> make_append_args(uint32_t G, uint32_t n, uint32_t ld, const uint16_t *active_mask,
                                     const T *follower_term, const T *leader_term,
                                     const T *prev_log_index, const T *prev_log_term, const T *n_new,
                                     const uint32_t *flog_off, const T *flog_term,
                                     T *follower_term_out, uint16_t *success_mask,
                                     uint16_t *stepdown_mask, T *match_index_io, T *next_index_io,
                                     uint8_t *status_out) {
    rq_append_args_t<T> h;
    std::memset(&h, 0, sizeof h);
    h.G = G; h.ld = ld; h.n = n; h.active_mask = active_mask;
    h.follower_term = follower_term; h.leader_term = leader_term;
    h.prev_log_index = prev_log_index; h.prev_log_term = prev_log_term; h.n_new = n_new;
    h.flog_off = flog_off; h.flog_term = flog_term; h.follower_term_out = follower_term_out;
    h.success_mask = success_mask; h.stepdown_mask = stepdown_mask;
    h.match_index_io = match_index_io; h.next_index_io = next_index_io; h.status_out = status_out;
    return h;
}

}  // namespace

extern "C" {

int rq_version(void) { return RQ_VERSION; }

const char *rq_strerror(int code) {
    switch (code) {
    case RQ_OK: return "ok";
    case RQ_E_ARG: return "invalid argument";
    case RQ_E_RANGE: return "size exceeds engine capacity";
    case RQ_E_NOMEM: return "out of memory";
    case RQ_E_HIP: return "HIP runtime error (see rq_last_hip_error)";
    case RQ_E_NODEVICE: return "no usable gfx950 device";
    case RQ_E_STATE: return "no resident round (call rq_upload first)";
    case RQ_E_MASK: return "empty joint-consensus configuration mask";
    case RQ_E_COMM: return "RCCL unavailable or collective failed";
    default: return "unknown rq error";
    }
}

int rq_engine_create(int device, uint32_t max_groups, uint32_t max_slots,
                     uint64_t max_log_terms, rq_engine **out) {
    if (!out) return RQ_E_ARG;
    *out = nullptr;
    if (device < 0 || max_groups == 0 || max_groups > 0x7FFFFFFFu - 64 || max_slots < 1 ||
        max_slots > RQ_MAX_SLOTS || max_log_terms > 0xFFFFFFFFull)
        return RQ_E_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count) return RQ_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RQ_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RQ_E_NODEVICE;

    rq_engine *e = new (std::nothrow) rq_engine();
    if (!e) return RQ_E_NOMEM;
    e->device = device;
    e->max_groups = max_groups;
    e->max_slots = max_slots;
    e->max_log_terms = max_log_terms;
    e->ld = (max_groups + 63u) / 64u * 64u;

    const size_t plane = up((size_t)max_slots * e->ld * 4);
    const size_t vec4 = up((size_t)e->ld * 4), vec2 = up((size_t)e->ld * 2), vec1 = up((size_t)e->ld);
    const size_t offs = up(((size_t)e->ld + 1) * 4), arena = up((size_t)max_log_terms * 4 + 4);
    const size_t total = 4 * plane + 5 * vec4 + 4 * vec2 + 2 * vec1 + offs + arena;

    if (hipSetDevice(device) != hipSuccess) { delete e; return RQ_E_NODEVICE; }
    hipError_t err = hipMalloc((void **)&e->slab, total);
    if (err != hipSuccess) { delete e; return RQ_E_NOMEM; }
    err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err != hipSuccess) { (void)hipFree(e->slab); delete e; return RQ_E_HIP; }
    err = hipMemsetAsync(e->slab, 0, total, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) {
        (void)hipStreamDestroy(e->stream); (void)hipFree(e->slab); delete e; return RQ_E_HIP;
    }
    char *p = e->slab;
    auto take = [&p](size_t bytes) { char *r = p; p += bytes; return r; };
    e->match = (int32_t *)take(plane);
    e->voter_term = (int32_t *)take(plane);
    e->voted_for = (int32_t *)take(plane);
    e->last_log_term = (int32_t *)take(plane);
    e->commit = (int32_t *)take(vec4);
    e->leader_term = (int32_t *)take(vec4);
    e->cand_term = (int32_t *)take(vec4);
    e->cand_id = (int32_t *)take(vec4);
    e->max_term = (int32_t *)take(vec4);
    e->grant = (uint16_t *)take(vec2);
    e->stepdown = (uint16_t *)take(vec2);
    e->old_mask = (uint16_t *)take(vec2);
    e->new_mask = (uint16_t *)take(vec2);
    e->elected = (uint8_t *)take(vec1);
    e->status = (uint8_t *)take(vec1);
    e->log_off = (uint32_t *)take(offs);
    e->log_term = (int32_t *)take(arena);
    *out = e;
    return RQ_OK;
}

int rq_engine_destroy(rq_engine *e) {
    if (!e) return RQ_E_ARG;
    int rc = RQ_OK;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (hipSetDevice(e->device) != hipSuccess) rc = RQ_E_HIP;
        if (e->stream) {
            (void)hipStreamSynchronize(e->stream);
            if (hipStreamDestroy(e->stream) != hipSuccess) rc = RQ_E_HIP;
        }
        if (e->slab && hipFree(e->slab) != hipSuccess) rc = RQ_E_HIP;
        if (e->slab64 && hipFree(e->slab64) != hipSuccess) rc = RQ_E_HIP;
        if (e->aux && hipFree(e->aux) != hipSuccess) rc = RQ_E_HIP;
        if (e->step_graph) (void)hipGraphExecDestroy(e->step_graph);
    }
    delete e;
    return rc;
}

int rq_last_hip_error(const rq_engine *e) { return e ? e->last_hip.load() : 0; }
int rq_engine_device(const rq_engine *e) { return e ? e->device : -1; }

int rq_commit_advance(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                      const int32_t *match, const int32_t *commit,
                      const int32_t *leader_term, const uint32_t *log_off,
                      const int32_t *log_term, int32_t *commit_out, uint8_t *status_out) {
    host_round h{};
    h.G = G; h.n = n; h.ld = ld;
    h.match = match; h.commit = commit; h.leader_term = leader_term;
    h.log_off = log_off; h.log_term = log_term;
    h.commit_out = commit_out; h.status_out = status_out;
    return run_host_round(e, h, RQ_WHAT_COMMIT);
}

int rq_vote_tally(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                  const int32_t *voter_term, const int32_t *voted_for,
                  const int32_t *voter_last_log_term, const int32_t *cand_term,
                  const int32_t *cand_id, int32_t *voter_term_out, uint16_t *grant_mask,
                  uint16_t *stepdown_mask, uint8_t *elected, int32_t *max_term,
                  uint8_t *status_out) {
    host_round h{};
    h.G = G; h.n = n; h.ld = ld;
    h.voter_term = voter_term; h.voted_for = voted_for; h.last_log_term = voter_last_log_term;
    h.cand_term = cand_term; h.cand_id = cand_id;
    h.voter_term_out = voter_term_out; h.grant_mask = grant_mask; h.stepdown_mask = stepdown_mask;
    h.elected = elected; h.max_term = max_term; h.status_out = status_out;
    return run_host_round(e, h, RQ_WHAT_VOTE);
}

int rq_fused_round(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                   const int32_t *match, const int32_t *commit, const int32_t *leader_term,
                   const uint32_t *log_off, const int32_t *log_term,
                   const int32_t *voter_term, const int32_t *voted_for,
                   const int32_t *voter_last_log_term, const int32_t *cand_term,
                   const int32_t *cand_id, int32_t *commit_out, int32_t *voter_term_out,
                   uint16_t *grant_mask, uint16_t *stepdown_mask, uint8_t *elected,
                   int32_t *max_term, uint8_t *status_out) {
    host_round h{};
    h.G = G; h.n = n; h.ld = ld;
    h.match = match; h.commit = commit; h.leader_term = leader_term;
    h.log_off = log_off; h.log_term = log_term;
    h.voter_term = voter_term; h.voted_for = voted_for; h.last_log_term = voter_last_log_term;
    h.cand_term = cand_term; h.cand_id = cand_id;
    h.commit_out = commit_out; h.voter_term_out = voter_term_out; h.grant_mask = grant_mask;
    h.stepdown_mask = stepdown_mask; h.elected = elected; h.max_term = max_term;
    h.status_out = status_out;
    return run_host_round(e, h, RQ_WHAT_COMMIT | RQ_WHAT_VOTE);
}

int rq_commit_advance_joint(rq_engine *e, uint32_t G, uint32_t m, uint32_t ld,
                            const uint16_t *old_mask, const uint16_t *new_mask,
                            const int32_t *match, const int32_t *commit,
                            const int32_t *leader_term, const uint32_t *log_off,
                            const int32_t *log_term, int32_t *commit_out, uint8_t *status_out) {
    host_round h{};
    h.G = G; h.n = m; h.ld = ld; h.joint = true; h.old_mask = old_mask; h.new_mask = new_mask;
    h.match = match; h.commit = commit; h.leader_term = leader_term;
    h.log_off = log_off; h.log_term = log_term;
    h.commit_out = commit_out; h.status_out = status_out;
    return run_host_round(e, h, RQ_WHAT_COMMIT);
}

int rq_vote_tally_joint(rq_engine *e, uint32_t G, uint32_t m, uint32_t ld,
                        const uint16_t *old_mask, const uint16_t *new_mask,
                        const int32_t *voter_term, const int32_t *voted_for,
                        const int32_t *voter_last_log_term, const int32_t *cand_term,
                        const int32_t *cand_id, int32_t *voter_term_out, uint16_t *grant_mask,
                        uint16_t *stepdown_mask, uint8_t *elected, int32_t *max_term,
                        uint8_t *status_out) {
    host_round h{};
    h.G = G; h.n = m; h.ld = ld; h.joint = true; h.old_mask = old_mask; h.new_mask = new_mask;
    h.voter_term = voter_term; h.voted_for = voted_for; h.last_log_term = voter_last_log_term;
    h.cand_term = cand_term; h.cand_id = cand_id;
    h.voter_term_out = voter_term_out; h.grant_mask = grant_mask; h.stepdown_mask = stepdown_mask;
    h.elected = elected; h.max_term = max_term; h.status_out = status_out;
    return run_host_round(e, h, RQ_WHAT_VOTE);
}

int rq_fused_round_joint(rq_engine *e, uint32_t G, uint32_t m, uint32_t ld,
                         const uint16_t *old_mask, const uint16_t *new_mask,
                         const int32_t *match, const int32_t *commit,
                         const int32_t *leader_term, const uint32_t *log_off,
                         const int32_t *log_term, const int32_t *voter_term,
                         const int32_t *voted_for, const int32_t *voter_last_log_term,
                         const int32_t *cand_term, const int32_t *cand_id,
                         int32_t *commit_out, int32_t *voter_term_out, uint16_t *grant_mask,
                         uint16_t *stepdown_mask, uint8_t *elected, int32_t *max_term,
                         uint8_t *status_out) {
    host_round h{};
    h.G = G; h.n = m; h.ld = ld; h.joint = true; h.old_mask = old_mask; h.new_mask = new_mask;
    h.match = match; h.commit = commit; h.leader_term = leader_term;
    h.log_off = log_off; h.log_term = log_term;
    h.voter_term = voter_term; h.voted_for = voted_for; h.last_log_term = voter_last_log_term;
    h.cand_term = cand_term; h.cand_id = cand_id;
    h.commit_out = commit_out; h.voter_term_out = voter_term_out; h.grant_mask = grant_mask;
    h.stepdown_mask = stepdown_mask; h.elected = elected; h.max_term = max_term;
    h.status_out = status_out;
    return run_host_round(e, h, RQ_WHAT_COMMIT | RQ_WHAT_VOTE);
}

int rq_follower_clamp(rq_engine *e, uint64_t count, const int32_t *leader_commit,
                      const int32_t *commit, const int32_t *last_new_index,
                      int32_t *commit_out) {
    return follower_clamp_host(e, count, leader_commit, commit, last_new_index, commit_out);
}

int rq_follower_clamp64(rq_engine *e, uint64_t count, const int64_t *leader_commit,
                        const int64_t *commit, const int64_t *last_new_index,
                        int64_t *commit_out) {
    return follower_clamp_host(e, count, leader_commit, commit, last_new_index, commit_out);
}

int rq_append_check(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                    const uint16_t *active_mask, const int32_t *follower_term,
                    const int32_t *leader_term, const int32_t *prev_log_index,
                    const int32_t *prev_log_term, const int32_t *n_new,
                    const uint32_t *flog_off, const int32_t *flog_term,
                    int32_t *follower_term_out, uint16_t *success_mask, uint16_t *stepdown_mask,
                    int32_t *match_index_io, int32_t *next_index_io, uint8_t *status_out) {
    return append_check_host(e, make_append_args(G, n, ld, active_mask, follower_term, leader_term, prev_log_index,
                                             prev_log_term, n_new, flog_off, flog_term, follower_term_out,
                                             success_mask, stepdown_mask, match_index_io, next_index_io,
                                             status_out));
}

int rq_append_check64(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
                    const uint16_t *active_mask, const int64_t *follower_term,
                    const int64_t *leader_term, const int64_t *prev_log_index,
                    const int64_t *prev_log_term, const int64_t *n_new,
                    const uint32_t *flog_off, const int64_t *flog_term,
                    int64_t *follower_term_out, uint16_t *success_mask, uint16_t *stepdown_mask,
                    int64_t *match_index_io, int64_t *next_index_io, uint8_t *status_out) {
    return append_check_host(e, make_append_args(G, n, ld, active_mask, follower_term, leader_term, prev_log_index,
                                             prev_log_term, n_new, flog_off, flog_term, follower_term_out,
                                             success_mask, stepdown_mask, match_index_io, next_index_io,
                                             status_out));
}

int rq_upload(rq_engine *e, uint32_t G, uint32_t n, uint32_t ld,
              const int32_t *match, const int32_t *commit, const int32_t *leader_term,
              const uint32_t *log_off, const int32_t *log_term,
              const int32_t *voter_term, const int32_t *voted_for,
              const int32_t *voter_last_log_term, const int32_t *cand_term,
              const int32_t *cand_id) {
    RQ_TRY(check_shape(e, G, n, ld));
    host_round h{};
    h.G = G; h.n = n; h.ld = ld;
    h.match = match; h.commit = commit; h.leader_term = leader_term;
    h.log_off = log_off; h.log_term = log_term;
    h.voter_term = voter_term; h.voted_for = voted_for; h.last_log_term = voter_last_log_term;
    h.cand_term = cand_term; h.cand_id = cand_id;
    std::lock_guard<std::mutex> lk(e->mu);
    RQ_HIP(e, hipSetDevice(e->device));
    e->resident = false;
    drain_unless_ok guard(e);
    if (G) RQ_TRY(stage_inputs(e, h, RQ_WHAT_COMMIT | RQ_WHAT_VOTE));
    RQ_HIP(e, hipStreamSynchronize(e->stream));
    e->resident = true; e->res_G = G; e->res_n = n;
    return guard.ok();
}

int rq_step(rq_engine *e, uint32_t rounds) {
    if (!e) return RQ_E_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->resident) return RQ_E_STATE;
    RQ_HIP(e, hipSetDevice(e->device));
    if (e->res_G == 0) return RQ_OK;
    rq_round_args a = device_args(e, e->res_G, e->res_n, false);
    constexpr uint32_t kGraphMinRounds = 8;
    if (rounds >= kGraphMinRounds) {
        if (e->step_graph && (e->graph_G != e->res_G || e->graph_n != e->res_n || e->graph_rounds != rounds)) {
            (void)hipGraphExecDestroy(e->step_graph);
            e->step_graph = nullptr;
        }
        if (!e->step_graph) {
            hipGraph_t g = nullptr;
            RQ_HIP(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            hipError_t err = hipSuccess;
            for (uint32_t r = 0; r < rounds && err == hipSuccess; r++)
                err = rq_launch_round(a, e->res_n, false, RQ_WHAT_COMMIT | RQ_WHAT_VOTE, 0, e->stream);
            hipError_t end = hipStreamEndCapture(e->stream, &g);
            if (err == hipSuccess) err = end;
            if (err == hipSuccess) err = hipGraphInstantiate(&e->step_graph, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (err != hipSuccess) { e->step_graph = nullptr; RQ_HIP(e, err); }
            e->graph_G = e->res_G; e->graph_n = e->res_n; e->graph_rounds = rounds;
        }
        RQ_HIP(e, hipGraphLaunch(e->step_graph, e->stream));
    } else {
        for (uint32_t r = 0; r < rounds; r++)
            RQ_HIP(e, rq_launch_round(a, e->res_n, false, RQ_WHAT_COMMIT | RQ_WHAT_VOTE, 0, e->stream));
    }
    RQ_HIP(e, hipStreamSynchronize(e->stream));
    return RQ_OK;
}

int rq_download(rq_engine *e, int32_t *commit_out, int32_t *voter_term_out,
                uint16_t *grant_mask, uint16_t *stepdown_mask, uint8_t *elected,
                int32_t *max_term, uint8_t *status_out) {
    if (!e) return RQ_E_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->resident) return RQ_E_STATE;
    RQ_HIP(e, hipSetDevice(e->device));
    const uint32_t G = e->res_G;
    drain_unless_ok guard(e);
    RQ_TRY(d2h(e, commit_out, e->commit, (size_t)G * 4));
    if (voter_term_out && G) RQ_TRY(d2h_planes(e, voter_term_out, e->voter_term, G, e->res_n, G));
    RQ_TRY(d2h(e, grant_mask, e->grant, (size_t)G * 2));
    RQ_TRY(d2h(e, stepdown_mask, e->stepdown, (size_t)G * 2));
    RQ_TRY(d2h(e, elected, e->elected, (size_t)G));
    RQ_TRY(d2h(e, max_term, e->max_term, (size_t)G * 4));
    RQ_TRY(d2h(e, status_out, e->status, (size_t)G));
    RQ_HIP(e, hipStreamSynchronize(e->stream));
    return guard.ok();
}

// ---- wide layout ----------------------------------------------------------

int rq_round64(rq_engine *e, int what, uint32_t G, uint32_t m, uint32_t ld,
               const uint16_t *old_mask, const uint16_t *new_mask,
               const int64_t *match, const int64_t *commit, const int64_t *leader_term,
               const uint32_t *log_off, const int64_t *log_term, const int64_t *log_base,
               const int64_t *voter_term, const int64_t *voted_for,
               const int64_t *voter_last_log_term, const int64_t *cand_term,
               const int64_t *cand_id, int64_t *commit_out, int64_t *voter_term_out,
               uint16_t *grant_mask, uint16_t *stepdown_mask, uint8_t *elected,
               int64_t *max_term, uint8_t *status_out) {
    if ((old_mask == nullptr) != (new_mask == nullptr)) return RQ_E_ARG;
    host_round_t<int64_t> h{};
    h.G = G; h.n = m; h.ld = ld; h.joint = old_mask != nullptr; h.old_mask = old_mask; h.new_mask = new_mask;
    h.match = match; h.commit = commit; h.leader_term = leader_term;
    h.log_off = log_off; h.log_term = log_term; h.log_base = log_base;
    h.voter_term = voter_term; h.voted_for = voted_for; h.last_log_term = voter_last_log_term;
    h.cand_term = cand_term; h.cand_id = cand_id;
    h.commit_out = commit_out; h.voter_term_out = voter_term_out; h.grant_mask = grant_mask;
    h.stepdown_mask = stepdown_mask; h.elected = elected; h.max_term = max_term;
    h.status_out = status_out;
    return run_host_round(e, h, what);
}

int rq_device_alloc(rq_engine *e, uint64_t bytes, int kind, void **out) {
    if (!e || !out || bytes == 0 || kind < 0 || kind > 3) return RQ_E_ARG;
    *out = nullptr;
    RQ_HIP(e, hipSetDevice(e->device));
    if (kind == RQ_MEM_SIGNAL) {
        if (bytes != 8) return RQ_E_ARG;
        hipError_t err = hipExtMallocWithFlags(out, 8, hipMallocSignalMemory);
        if (err != hipSuccess) { e->last_hip.store((int)err); *out = nullptr; return RQ_E_NOMEM; }
        const uint64_t zero = 0;
        RQ_HIP(e, hipMemcpy(*out, &zero, 8, hipMemcpyHostToDevice));
        return RQ_OK;
    }
    const unsigned flags = kind == RQ_MEM_UNCACHED ? hipDeviceMallocUncached
                         : kind == RQ_MEM_FINEGRAINED ? hipDeviceMallocFinegrained
                                                      : hipDeviceMallocDefault;
    hipError_t err = hipExtMallocWithFlags(out, (size_t)bytes, flags);
    if (err != hipSuccess) { e->last_hip.store((int)err); *out = nullptr; return RQ_E_NOMEM; }
    return RQ_OK;
}

int rq_host_alloc(rq_engine *e, uint64_t bytes, void **out) {
    if (!e || !out || bytes == 0) return RQ_E_ARG;
    *out = nullptr;
    RQ_HIP(e, hipSetDevice(e->device));
    hipError_t err = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (err != hipSuccess) { e->last_hip.store((int)err); *out = nullptr; return RQ_E_NOMEM; }
    return RQ_OK;
}

int rq_host_free(rq_engine *e, void *ptr) {
    if (!e) return RQ_E_ARG;
    if (!ptr) return RQ_OK;
    RQ_HIP(e, hipSetDevice(e->device));
    RQ_HIP(e, hipHostFree(ptr));
    return RQ_OK;
}

int rq_device_free(rq_engine *e, void *ptr) {
    if (!e) return RQ_E_ARG;
    if (!ptr) return RQ_OK;
    RQ_HIP(e, hipSetDevice(e->device));
    RQ_HIP(e, hipFree(ptr));
    return RQ_OK;
}

int rq_copy_to_device(rq_engine *e, void *dst, const void *src, uint64_t bytes) {
    if (!e || (bytes && (!dst || !src))) return RQ_E_ARG;
    RQ_HIP(e, hipSetDevice(e->device));
    if (bytes) RQ_HIP(e, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return RQ_OK;
}

int rq_copy_from_device(rq_engine *e, void *dst, const void *src, uint64_t bytes) {
    if (!e || (bytes && (!dst || !src))) return RQ_E_ARG;
    RQ_HIP(e, hipSetDevice(e->device));
    if (bytes) RQ_HIP(e, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return RQ_OK;
}

// ---- device-pointer launches ----------------------------------------------

int rq_round_dev(rq_engine *e, void *stream, const rq_round_desc *d) {
    rq_plan p;
    RQ_TRY(plan_from_desc(e, d, p));
    return launch_plan(p, stream);
}

int rq_plan_round(rq_engine *e, const rq_round_desc *d, rq_plan **out) {
    if (!out) return RQ_E_ARG;
    *out = nullptr;
    rq_plan tmp;
    RQ_TRY(plan_from_desc(e, d, tmp));
    rq_plan *p = new (std::nothrow) rq_plan(tmp);
    if (!p) return RQ_E_NOMEM;
    *out = p;
    return RQ_OK;
}

int rq_plan_rounds(rq_engine *e, const rq_round_desc *descs, uint32_t count, rq_plan **out) {
    if (!out) return RQ_E_ARG;
    *out = nullptr;
    if (!e || !descs || count < 1 || count > RQ_MAX_SEGMENTS) return RQ_E_ARG;
    rq_plan *p = new (std::nothrow) rq_plan;
    if (!p) return RQ_E_NOMEM;
    std::memset(p, 0, sizeof *p);
    p->engine = e;
    const bool wide = descs[0].layout == RQ_LAYOUT_I64;
    const size_t each = wide ? sizeof(rq_round64_args) : sizeof(rq_round_args);
    p->many = std::malloc(each * count);
    int rc = p->many ? RQ_OK : RQ_E_NOMEM;
    for (uint32_t i = 0; i < count && rc == RQ_OK; i++) {
        // every image: one layout, one decision set, one build; no segment table, no start
        // signal beyond the first image's (a launch starts once)
        rq_plan one;
        rc = plan_from_desc(e, (const rq_round_desc *)((const char *)descs + (size_t)i * descs[0].struct_size), one);
        if (rc != RQ_OK) break;
        const rq_round_desc &d = *(const rq_round_desc *)((const char *)descs + (size_t)i * descs[0].struct_size);
        if (d.struct_size != descs[0].struct_size || d.layout != descs[0].layout || d.what != descs[0].what ||
            d.variant != descs[0].variant || (d.variant & RQ_VARIANT_PLAIN) || one.segmented ||
            (i > 0 && d.start_signal)) { rc = RQ_E_ARG; break; }
        p->image_n[i] = one.n; p->image_joint[i] = one.joint;
        one.args.nt_gather = rq_nt_gather(one.args);          // per image: each has its own arena
        one.args64.nt_gather = rq_nt_gather(one.args64);
        if (one.args.log_term_start) one.args.log_off = nullptr;      // as rq_launch_round does: "no log_off" = "no arena lookup"
        if (one.args64.log_term_start) one.args64.log_off = nullptr;
        if (wide) ((rq_round64_args *)p->many)[i] = one.args64; else ((rq_round_args *)p->many)[i] = one.args;
    }
    if (rc == RQ_OK) {
        p->wide = wide; p->images = count; p->what = (int)descs[0].what; p->variant = descs[0].variant;
        hipError_t err = hipSetDevice(e->device);
        if (err == hipSuccess) err = hipMalloc(&p->table, each * count);
        if (err == hipSuccess) err = hipMemcpy(p->table, p->many, each * count, hipMemcpyHostToDevice);
        if (err != hipSuccess) { e->last_hip.store((int)err); rc = err == hipErrorOutOfMemory ? RQ_E_NOMEM : RQ_E_HIP; }
    }
    if (rc != RQ_OK) {
        if (p->table) (void)hipFree(p->table);
        std::free(p->many);
        delete p;
        return rc;
    }
    *out = p;
    return RQ_OK;
}

int rq_plan_launch(rq_plan *p, void *stream) {
    if (!p) return RQ_E_ARG;
    return launch_plan(*p, stream);
}

int rq_groups_scatter_dev(rq_engine *e, void *stream, const rq_round_desc *d, uint32_t count,
                          const uint32_t *gids, const void *rows, const uint32_t *row_log_off,
                          const void *row_log_terms, uint8_t *row_status) {
    rq_plan p;
    RQ_TRY(plan_from_desc(e, d, p));
    RQ_HIP(e, hipSetDevice(e->device));
    hipError_t err = p.wide
        ? rq_launch_groups_scatter(p.args64, p.n, count, gids, (const int64_t *)rows, row_log_off,
                                   (const int64_t *)row_log_terms, row_status, (hipStream_t)stream)
        : rq_launch_groups_scatter(p.args, p.n, count, gids, (const int32_t *)rows, row_log_off,
                                   (const int32_t *)row_log_terms, row_status, (hipStream_t)stream);
    if (err == hipErrorInvalidValue) return RQ_E_ARG;
    RQ_HIP(e, err);
    return RQ_OK;
}

int rq_groups_gather_dev(rq_engine *e, void *stream, const rq_round_desc *d, uint32_t count,
                         const uint32_t *gids, void *rows_out, uint64_t *flags_out) {
    rq_plan p;
    RQ_TRY(plan_from_desc(e, d, p));
    RQ_HIP(e, hipSetDevice(e->device));
    hipError_t err = p.wide
        ? rq_launch_groups_gather(p.args64, p.n, count, gids, (int64_t *)rows_out, flags_out, (hipStream_t)stream)
        : rq_launch_groups_gather(p.args, p.n, count, gids, (int32_t *)rows_out, flags_out, (hipStream_t)stream);
    if (err == hipErrorInvalidValue) return RQ_E_ARG;
    RQ_HIP(e, err);
    return RQ_OK;
}

int rq_append_check_dev(rq_engine *e, void *stream, const rq_append_desc *d) {
    if (!e || !d || d->struct_size != sizeof(rq_append_desc)) return RQ_E_ARG;
    if (d->layout == RQ_LAYOUT_I32) return append_check_dev<int32_t>(e, stream, *d);
    if (d->layout == RQ_LAYOUT_I64) return append_check_dev<int64_t>(e, stream, *d);
    return RQ_E_ARG;
}

int rq_summary_apply_dev(rq_engine *e, void *stream, const void *gathered, uint64_t msg_stride,
                         uint64_t section_off, uint64_t round_stride, uint32_t rounds,
                         uint32_t ranks, uint32_t groups, uint32_t bits,
                         int32_t *replica, uint32_t *escapes) {
    if (!e || !gathered || !replica) return RQ_E_ARG;
    RQ_HIP(e, hipSetDevice(e->device));
    hipError_t err = rq_launch_summary_apply((const uint8_t *)gathered, msg_stride, section_off, round_stride,
                                             rounds, ranks, groups, bits, replica, escapes, (hipStream_t)stream);
    if (err == hipErrorInvalidValue) return RQ_E_ARG;
    RQ_HIP(e, err);
    return RQ_OK;
}

int rq_plan_destroy(rq_plan *p) {
    if (!p) return RQ_E_ARG;
    if (p->table) {
        (void)hipSetDevice(p->engine->device);
        (void)hipFree(p->table);
    }
    std::free(p->many);
    delete p;
    return RQ_OK;
}

int rq_stream_wait_signal(rq_engine *e, void *stream, const uint64_t *signal, uint64_t value) {
    if (!e || !signal) return RQ_E_ARG;
    RQ_HIP(e, hipSetDevice(e->device));
    RQ_HIP(e, hipStreamWaitValue64((hipStream_t)stream, const_cast<uint64_t *>(signal), value,
                                   hipStreamWaitValueGte, ~0ull));
    return RQ_OK;
}

int rq_stream_sync(rq_engine *e, void *stream) {
    if (!e) return RQ_E_ARG;
    RQ_HIP(e, hipSetDevice(e->device));
    // A round is tens of microseconds long and a thread blocked in
    // hipStreamSynchronize takes 10-20 us to wake: poll for the first two
    // milliseconds, then block.
    const hipStream_t s = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t q;
    while ((q = hipStreamQuery(s)) == hipErrorNotReady) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            q = hipStreamSynchronize(s);
            break;
        }
    }
    RQ_HIP(e, q);
    return RQ_OK;
}

}  // extern "C"
