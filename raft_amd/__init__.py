"""raft_amd -- MI355X (gfx950) batched Raft quorum engine.

Drop-in for the per-group quorum arithmetic of tawawhite/raft's raft.go
(RequestVoteRPC tally and leader commitIndex advance), evaluated for very many
independent Raft groups per launch by hand-written HIP kernels in librq.so
behind the C-ABI of include/rq.h.  No CPU implementation lives here.
"""
from ._lib import LIB_PATH, RqError, lib  # noqa: F401
from .engine import (LOG_EMPTY, MAX_SLOTS, ST_EMPTY_LOG, ST_LOG_OOB,  # noqa: F401
                     QuorumEngine, RoundResult, VoteResult)

__version__ = "0.1.0"
