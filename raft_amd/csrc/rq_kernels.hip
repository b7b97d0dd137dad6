// rq_kernels.hip -- gfx950 (MI355X) kernels of the batched Raft quorum engine,
// 32-bit layout: the quorum round (rq_round_impl.h), the AppendEntries
// consistency batch and the follower commit clamp (rq_append_impl.h)
// instantiated for int32_t.
#include "rq_append_impl.h"
#include "rq_groups_impl.h"

hipError_t rq_launch_round(const rq_round_args &args, uint32_t n, bool joint, int what,
                           int variant, hipStream_t stream, const rq_round_segs *segs) {
    return launch_round_impl<int32_t>(args, n, joint, what, variant, stream, segs);
}

hipError_t rq_launch_rounds(const rq_round_args *args, const uint32_t *n, const bool *joint, uint32_t count,
                            const rq_round_args *table, int what, int variant, hipStream_t stream) {
    return launch_rounds_impl<int32_t>(args, n, joint, count, table, what, variant, stream);
}

hipError_t rq_launch_append_check(const rq_append_args &a, const rq_round_args *commit_round, hipStream_t stream) {
    return launch_append_check_impl<int32_t>(a, commit_round, stream);
}

hipError_t rq_launch_follower_clamp(uint64_t count, const int32_t *leader_commit,
                                    const int32_t *commit, const int32_t *last_new_index,
                                    int32_t *commit_out, hipStream_t stream) {
    return launch_follower_clamp_impl<int32_t>(count, leader_commit, commit, last_new_index, commit_out, stream);
}

hipError_t rq_launch_groups_scatter(const rq_round_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                    const int32_t *rows, const uint32_t *row_off, const int32_t *row_terms,
                                    uint8_t *row_status, hipStream_t stream) {
    return launch_groups_scatter_impl<int32_t>(a, n, count, gids, rows, row_off, row_terms, row_status, stream);
}

hipError_t rq_launch_groups_gather(const rq_round_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                   int32_t *rows_out, uint64_t *flags_out, hipStream_t stream) {
    return launch_groups_gather_impl<int32_t>(a, n, count, gids, rows_out, flags_out, stream);
}

// ---------------------------------------------------------------------------
// Summary consumer (sharded deployment): every rank keeps a replica of every
// rank's commitIndex and adds the gathered per-round advances to it.  One thread
// per (rank, group); offsets of 16 / 8 / 4 bits, all ones = escape: the advance of
// that round is unknown, so nothing of the message is added for the group from
// that round on (the replica entry stays exact up to the round before, and the
// host, which sees the count, fetches the full vectors of the period); escapes
// are counted with one atomic per wave.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void rq_summary_apply_kernel(const uint8_t *msg, uint64_t msg_stride,
                                                               uint64_t section_off, uint64_t round_stride,
                                                               uint32_t rounds, uint32_t groups,
                                                               uint32_t bits, int32_t *replica,
                                                               uint32_t *escapes) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    uint32_t esc = 0;
    if (g < groups) {
        const uint32_t escape = (1u << bits) - 1u;
        uint32_t sum = 0;
        for (uint32_t k = 0; k < rounds; k++) {      // the rounds of one message, oldest first
            const uint8_t *sec = msg + (uint64_t)r * msg_stride + section_off + (uint64_t)k * round_stride;
            uint32_t v;
            if (bits == 16) v = reinterpret_cast<const uint16_t *>(sec)[g];
            else if (bits == 8) v = sec[g];
            else v = (sec[g >> 1] >> (4 * (g & 1))) & 0xFu;   // little-endian words: group g in bits 4*(g%16)
            if (v == escape) { esc++; break; }       // later rounds build on an advance this message does not carry
            sum += v;
        }
        if (sum) replica[(uint64_t)r * groups + g] += (int32_t)sum;
    }
    if (escapes) count_escapes(esc, escapes);
}
}  // namespace

hipError_t rq_launch_summary_apply(const uint8_t *msg, uint64_t msg_stride, uint64_t section_off,
                                   uint64_t round_stride, uint32_t rounds,
                                   uint32_t ranks, uint32_t groups, uint32_t bits,
                                   int32_t *replica, uint32_t *escapes, hipStream_t stream) {
    if (ranks == 0 || groups == 0 || rounds == 0) return hipSuccess;
    if ((bits != 16 && bits != 8 && bits != 4) || ranks > 65535u) return hipErrorInvalidValue;
    if (bits == 16 && ((section_off | msg_stride | round_stride | reinterpret_cast<uintptr_t>(msg)) & 1u)) return hipErrorInvalidValue;
    const dim3 grid((groups + 255u) / 256u, ranks), blk(256);
    hipLaunchKernelGGL(rq_summary_apply_kernel, grid, blk, 0, stream, msg, msg_stride, section_off, round_stride,
                       rounds, groups, bits, replica, escapes);
    return hipGetLastError();
}
