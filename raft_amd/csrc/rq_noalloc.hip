// rq_noalloc.hip -- rq_fused_round_noalloc (include/rq.h): the host-buffer fused
// round for callers that cannot allocate.
//
// rq_fused_round stages its inputs and outputs through device memory that
// rq_engine_create obtained from hipMalloc.  This variant stages through a fixed
// array that is part of the library's device image instead, so it never calls an
// allocator.  Every process holds one copy of the array per GPU it runs on, until
// it ends.
#include "../../include/rq.h"
#include "rq_kernels.h"

#include <cstring>
#include <mutex>

constexpr size_t kStaticScratchBytes = RQ_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char rq_static_scratch[kStaticScratchBytes];

namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }
#define RQ_TRY(x) do { int rq_rc_ = (x); if (rq_rc_ != RQ_OK) return rq_rc_; } while (0)

constexpr int kMaxDevices = 64;
struct static_scratch_state {
    std::mutex mu[kMaxDevices];          // one call at a time per device: they share the array
    unsigned char *ptr[kMaxDevices] = {};  // the array's address on each device, resolved on first use
};
static_scratch_state &scratch_state() {
    static static_scratch_state s;
    return s;
}

// Device address of this device's copy of the array (the symbol's name is only
// a host-side handle).  `device` must be current.
int static_scratch_ptr(int device, unsigned char **out) {
    static_scratch_state &st = scratch_state();
    if (!st.ptr[device]) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(rq_static_scratch)) != hipSuccess || !p) return RQ_E_HIP;
        st.ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = st.ptr[device];
    return RQ_OK;
}

#define RQ_HIPN(call) do { if ((call) != hipSuccess) return RQ_E_HIP; } while (0)

}  // namespace

extern "C" int rq_fused_round_noalloc(int device, void *stream_, uint32_t G, uint32_t n, uint32_t ld,
                           const int32_t *match, const int32_t *commit, const int32_t *leader_term,
                           const uint32_t *log_off, const int32_t *log_term,
                           const int32_t *voter_term, const int32_t *voted_for,
                           const int32_t *voter_last_log_term, const int32_t *cand_term,
                           const int32_t *cand_id, int32_t *commit_out, int32_t *voter_term_out,
                           uint16_t *grant_mask, uint16_t *stepdown_mask, uint8_t *elected,
                           int32_t *max_term, uint8_t *status_out) {
    if (device < 0 || device >= kMaxDevices || n < 1 || n > RQ_MAX_SLOTS || ld < G) return RQ_E_ARG;
    if (!match || !commit || !leader_term || !log_off || !voter_term || !voted_for || !voter_last_log_term ||
        !cand_term || !cand_id || !commit_out || !voter_term_out || !grant_mask || !stepdown_mask ||
        !elected || !max_term)
        return RQ_E_ARG;
    if (G == 0) return RQ_OK;
    for (uint32_t g = 0; g < G; g++)
        if (log_off[g + 1] < log_off[g]) return RQ_E_ARG;
    const uint64_t terms = log_off[G];
    if (terms && !log_term) return RQ_E_ARG;
    // carve the array; everything that does not fit is refused, nothing grows
    const size_t ldd = ((size_t)G + 63) / 64 * 64;
    const size_t plane = up(n * ldd * 4), vec4 = up(ldd * 4), vec2 = up(ldd * 2), vec1 = up(ldd);
    const size_t offs = up((ldd + 1) * 4), arena = up((size_t)terms * 4 + 4);
    const uint64_t need = 4 * (uint64_t)plane + 5 * vec4 + 2 * vec2 + 2 * vec1 + offs + arena;
    if (need > kStaticScratchBytes) return RQ_E_RANGE;

    hipStream_t stream = (hipStream_t)stream_;
    static_scratch_state &st = scratch_state();
    std::lock_guard<std::mutex> lk(st.mu[device]);
    RQ_HIPN(hipSetDevice(device));
    unsigned char *base = nullptr;
    RQ_TRY(static_scratch_ptr(device, &base));
    struct drain {            // no exit while a copy to or from the caller's buffers is in flight
        hipStream_t s;
        ~drain() { (void)hipStreamSynchronize(s); }
    } drain_on_exit{stream};
    unsigned char *p = base;
    auto take = [&p](size_t bytes) { unsigned char *r = p; p += bytes; return r; };
    int32_t *d_match = (int32_t *)take(plane), *d_vt = (int32_t *)take(plane);
    int32_t *d_vf = (int32_t *)take(plane), *d_ll = (int32_t *)take(plane);
    int32_t *d_commit = (int32_t *)take(vec4), *d_lt = (int32_t *)take(vec4), *d_ct = (int32_t *)take(vec4);
    int32_t *d_ci = (int32_t *)take(vec4), *d_mx = (int32_t *)take(vec4);
    uint16_t *d_gm = (uint16_t *)take(vec2), *d_sm = (uint16_t *)take(vec2);
    uint8_t *d_el = (uint8_t *)take(vec1), *d_st = (uint8_t *)take(vec1);
    uint32_t *d_off = (uint32_t *)take(offs);
    int32_t *d_arena = (int32_t *)take(arena);

    auto planes_in = [&](int32_t *dst, const int32_t *src) {
        return hipMemcpy2DAsync(dst, ldd * 4, src, (size_t)ld * 4, (size_t)G * 4, n, hipMemcpyHostToDevice, stream);
    };
    RQ_HIPN(planes_in(d_match, match));
    RQ_HIPN(planes_in(d_vt, voter_term));
    RQ_HIPN(planes_in(d_vf, voted_for));
    RQ_HIPN(planes_in(d_ll, voter_last_log_term));
    RQ_HIPN(hipMemcpyAsync(d_commit, commit, (size_t)G * 4, hipMemcpyHostToDevice, stream));
    RQ_HIPN(hipMemcpyAsync(d_lt, leader_term, (size_t)G * 4, hipMemcpyHostToDevice, stream));
    RQ_HIPN(hipMemcpyAsync(d_ct, cand_term, (size_t)G * 4, hipMemcpyHostToDevice, stream));
    RQ_HIPN(hipMemcpyAsync(d_ci, cand_id, (size_t)G * 4, hipMemcpyHostToDevice, stream));
    RQ_HIPN(hipMemcpyAsync(d_off, log_off, ((size_t)G + 1) * 4, hipMemcpyHostToDevice, stream));
    if (terms) RQ_HIPN(hipMemcpyAsync(d_arena, log_term, (size_t)terms * 4, hipMemcpyHostToDevice, stream));

    rq_round_args a;
    std::memset(&a, 0, sizeof a);
    a.G = G; a.ld = (uint32_t)ldd; a.m = n;
    a.match = d_match; a.commit = d_commit; a.leader_term = d_lt;
    a.log_off = d_off; a.log_term = d_arena; a.log_terms = terms; a.log_term_bits = 32;
    a.voter_term = d_vt; a.voted_for = d_vf; a.last_log_term = d_ll; a.cand_term = d_ct; a.cand_id = d_ci;
    a.commit_out = d_commit; a.voter_term_out = d_vt;      // in place inside the array
    a.grant_mask = d_gm; a.stepdown_mask = d_sm; a.elected = d_el; a.max_term = d_mx; a.status_out = d_st;
    hipError_t err = rq_launch_round(a, n, false, RQ_WHAT_COMMIT | RQ_WHAT_VOTE, 0, stream);
    if (err == hipErrorInvalidValue) return RQ_E_ARG;
    RQ_HIPN(err);

    RQ_HIPN(hipMemcpyAsync(commit_out, d_commit, (size_t)G * 4, hipMemcpyDeviceToHost, stream));
    RQ_HIPN(hipMemcpy2DAsync(voter_term_out, (size_t)ld * 4, d_vt, ldd * 4, (size_t)G * 4, n,
                             hipMemcpyDeviceToHost, stream));
    RQ_HIPN(hipMemcpyAsync(grant_mask, d_gm, (size_t)G * 2, hipMemcpyDeviceToHost, stream));
    RQ_HIPN(hipMemcpyAsync(stepdown_mask, d_sm, (size_t)G * 2, hipMemcpyDeviceToHost, stream));
    RQ_HIPN(hipMemcpyAsync(elected, d_el, (size_t)G, hipMemcpyDeviceToHost, stream));
    RQ_HIPN(hipMemcpyAsync(max_term, d_mx, (size_t)G * 4, hipMemcpyDeviceToHost, stream));
    if (status_out) RQ_HIPN(hipMemcpyAsync(status_out, d_st, (size_t)G, hipMemcpyDeviceToHost, stream));
    RQ_HIPN(hipStreamSynchronize(stream));
    return RQ_OK;
}

