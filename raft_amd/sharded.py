"""Group-sharded deployment: one process per GPU, groups block-partitioned
over ranks, and ONE exchange step per round -- an all-gather of the per-round
summary (commitIndex' per group + the elected decisions packed 1 bit per
group) over torch.distributed (backend "nccl" is RCCL over xGMI on MI355X;
"gloo" runs the same code on CPU tensors for tests).

Raft groups are independent (no rule in raft.go:181-223 or in the commit rule
reads another group's state), so the decision kernels need no collective; the
summary exchange exists so that every rank can route client traffic and
heartbeats for all groups.
"""
import torch
import torch.distributed as dist


def shard_bounds(total_groups, world_size, rank):
    """Block partition: rank r owns groups [r*T/R, (r+1)*T/R)."""
    g0 = total_groups * rank // world_size
    g1 = total_groups * (rank + 1) // world_size
    return g0, g1


def summary_words(groups):
    """int32 words of one rank's summary: G commit values + ceil(G/64) uint64 bit words."""
    g_even = groups + (groups & 1)               # keeps the bit words 8-byte aligned
    return g_even + 2 * ((groups + 63) // 64)


class SummaryExchange:
    """`depth` independent summary slots so that the all-gather of round i can
    run on the communication stream while round i+1 computes into another slot.

    Every rank must use the same `groups` (pad the last shard if needed)."""

    def __init__(self, groups, device, depth=2, group=None):
        self.groups = int(groups)
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.words = summary_words(self.groups)
        self.local = [torch.zeros(self.words, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.all = [torch.zeros((self.world, self.words), dtype=torch.int32, device=self.device)
                    for _ in range(depth)]
        self.work = [None] * depth

    # views handed to the kernel as its commit_out / elected_bits outputs
    def commit_view(self, slot):
        return self.local[slot][: self.groups]

    def bits_view(self, slot):
        g_even = self.groups + (self.groups & 1)
        return self.local[slot][g_even:].view(torch.int64)

    def exchange(self, slot, async_op=True):
        """All-gather slot's summary.  The collective is ordered after the work
        already enqueued on the current stream (the round that produced it)."""
        if self.world == 1:
            self.all[slot][0].copy_(self.local[slot], non_blocking=True)
            return None
        w = dist.all_gather_into_tensor(self.all[slot].view(-1), self.local[slot],
                                        group=self.group, async_op=async_op)
        self.work[slot] = w if async_op else None
        return w

    def wait(self, slot):
        """Make the current stream (and, for gloo, the host) wait for slot's exchange."""
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None

    def wait_all(self):
        for s in range(len(self.work)):
            self.wait(s)

    def gathered(self, slot):
        """(commit [world, groups] int32, elected_bits [world, ceil(groups/64)] int64)."""
        g_even = self.groups + (self.groups & 1)
        a = self.all[slot]
        return a[:, : self.groups], a[:, g_even:].contiguous().view(torch.int64)
