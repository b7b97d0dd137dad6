// rq_kernels.hip -- gfx950 (MI355X) kernels of the batched Raft quorum engine,
// 32-bit layout: the quorum round (rq_round_impl.h), the AppendEntries
// consistency batch and the follower commit clamp (rq_append_impl.h)
// instantiated for int32_t.
#include "rq_append_impl.h"

hipError_t rq_launch_round(const rq_round_args &args, uint32_t n, bool joint, int what,
                           int variant, hipStream_t stream) {
    return launch_round_impl<int32_t>(args, n, joint, what, variant, stream);
}

hipError_t rq_launch_append_check(const rq_append_args &a, hipStream_t stream) {
    return launch_append_check_impl<int32_t>(a, stream);
}

hipError_t rq_launch_follower_clamp(uint64_t count, const int32_t *leader_commit,
                                    const int32_t *commit, const int32_t *last_new_index,
                                    int32_t *commit_out, hipStream_t stream) {
    return launch_follower_clamp_impl<int32_t>(count, leader_commit, commit, last_new_index, commit_out, stream);
}
