// rq_kernels64.hip -- the same kernels over 64-bit planes: every term, index
// and id is an int64_t, the full range of Go's int on amd64 (raft.go:35-68).
// One group per thread in the round kernels.
#include "rq_append_impl.h"

hipError_t rq_launch_round(const rq_round64_args &args, uint32_t n, bool joint, int what,
                           int variant, hipStream_t stream) {
    return launch_round_impl<int64_t>(args, n, joint, what, variant, stream);
}

hipError_t rq_launch_append_check(const rq_append64_args &a, hipStream_t stream) {
    return launch_append_check_impl<int64_t>(a, stream);
}

hipError_t rq_launch_follower_clamp(uint64_t count, const int64_t *leader_commit,
                                    const int64_t *commit, const int64_t *last_new_index,
                                    int64_t *commit_out, hipStream_t stream) {
    return launch_follower_clamp_impl<int64_t>(count, leader_commit, commit, last_new_index, commit_out, stream);
}
