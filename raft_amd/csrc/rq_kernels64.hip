// rq_kernels64.hip -- the same kernels over 64-bit planes: every term, index
// and id is an int64_t, the full range of Go's int on amd64 (raft.go:35-68).
// One group per thread in the round kernels.
#include "rq_append_impl.h"
#include "rq_groups_impl.h"

hipError_t rq_launch_round(const rq_round64_args &args, uint32_t n, bool joint, int what,
                           int variant, hipStream_t stream, const rq_round_segs *segs) {
    return launch_round_impl<int64_t>(args, n, joint, what, variant, stream, segs);
}

hipError_t rq_launch_rounds(const rq_round64_args *args, const uint32_t *n, const bool *joint, uint32_t count,
                            const rq_round64_args *table, int what, int variant, hipStream_t stream) {
    return launch_rounds_impl<int64_t>(args, n, joint, count, table, what, variant, stream);
}

hipError_t rq_launch_append_check(const rq_append64_args &a, const rq_round64_args *commit_round, hipStream_t stream) {
    return launch_append_check_impl<int64_t>(a, commit_round, stream);
}

hipError_t rq_launch_follower_clamp(uint64_t count, const int64_t *leader_commit,
                                    const int64_t *commit, const int64_t *last_new_index,
                                    int64_t *commit_out, hipStream_t stream) {
    return launch_follower_clamp_impl<int64_t>(count, leader_commit, commit, last_new_index, commit_out, stream);
}

hipError_t rq_launch_groups_scatter(const rq_round64_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                    const int64_t *rows, const uint32_t *row_off, const int64_t *row_terms,
                                    uint8_t *row_status, hipStream_t stream) {
    return launch_groups_scatter_impl<int64_t>(a, n, count, gids, rows, row_off, row_terms, row_status, stream);
}

hipError_t rq_launch_groups_gather(const rq_round64_args &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                   int64_t *rows_out, uint64_t *flags_out, hipStream_t stream) {
    return launch_groups_gather_impl<int64_t>(a, n, count, gids, rows_out, flags_out, stream);
}
