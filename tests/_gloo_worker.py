"""One rank of the CPU (gloo) rehearsal of the sharded deployment: block
partition, per-rank decisions (the oracle stands in for the GPU kernel here),
summary all-gather through raft_amd.sharded.SummaryExchange, pipelined over
two slots, checked on every rank against the unsharded answer."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as o  # noqa: E402
from raft_amd import synth  # noqa: E402
from raft_amd.sharded import SummaryExchange, shard_bounds, summary_words  # noqa: E402


def shard(b, g0, g1):
    s = synth.Batch(b)
    s["G"], s["ld"] = g1 - g0, g1 - g0
    for k in ("match", "voter_term", "voted_for", "last_log_term"):
        s[k] = np.ascontiguousarray(b[k][:, g0:g1])
    for k in ("commit", "leader_term", "cand_term", "cand_id"):
        s[k] = b[k][g0:g1].copy()
    off = b.log_off[g0:g1 + 1].astype(np.int64)
    s["log_term"] = b.log_term[off[0]:off[-1]].copy()       # CSR arena split at a group boundary
    s["log_off"] = (off - off[0]).astype(np.uint32)
    return s


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, rounds = 1000 + 37, 5                 # odd size: exercises the alignment padding
    total = per * world
    ex = SummaryExchange(per, "cpu", depth=2)
    assert ex.words == summary_words(per) and ex.world == world
    expect = []
    for r in range(rounds):
        slot = r % 2
        ex.wait(slot)                          # previous exchange of this slot
        if r >= 2:                             # ... whose result is checked before the slot is reused
            check(ex, slot, expect[r - 2], per, world)
        b = synth.gen_fused(total, 5, seed=900 + r)
        whole = o.fused_round(*b.fused_args())
        expect.append(whole)
        g0, g1 = shard_bounds(total, world, rank)
        assert g1 - g0 == per
        mine = o.fused_round(*shard(b, g0, g1).fused_args())
        assert np.array_equal(mine.commit, whole.commit[g0:g1])     # sharding changes no decision
        assert np.array_equal(mine.elected, whole.elected[g0:g1])
        ex.commit_view(slot).copy_(torch.from_numpy(mine.commit))
        bits = np.packbits(mine.elected, bitorder="little")
        bits = np.concatenate([bits, np.zeros((-bits.size) % 8, np.uint8)]).view(np.int64)
        ex.bits_view(slot).copy_(torch.from_numpy(bits))
        ex.exchange(slot, async_op=True)
    ex.wait_all()
    for r in (rounds - 2, rounds - 1):
        check(ex, r % 2, expect[r], per, world)
    dist.barrier()
    if rank == 0:
        print("GLOO_SHARDED_OK")
    dist.destroy_process_group()


def check(ex, slot, whole, per, world):
    commit, bits = ex.gathered(slot)
    assert commit.shape == (world, per)
    assert np.array_equal(commit.numpy().reshape(-1), whole.commit)
    for r in range(world):
        got = np.unpackbits(bits[r].numpy().view(np.uint8), bitorder="little")[:per]
        assert np.array_equal(got, whole.elected[r * per:(r + 1) * per])


if __name__ == "__main__":
    main()
