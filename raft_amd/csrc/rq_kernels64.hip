// rq_kernels64.hip -- the quorum round over 64-bit planes: every term, index
// and id is an int64_t, the full range of Go's int on amd64 (raft.go:35-68).
// Same kernels as the 32-bit layout (rq_round_impl.h), one group per thread.
#include "rq_round_impl.h"

hipError_t rq_launch_round(const rq_round64_args &args, uint32_t n, bool joint, int what,
                           int variant, hipStream_t stream) {
    return launch_round_impl<int64_t>(args, n, joint, what, variant, stream);
}
