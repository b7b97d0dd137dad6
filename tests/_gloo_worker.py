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
from raft_amd.sharded import MODE_BITS, SummaryExchange, shard_bounds, summary_bytes  # noqa: E402


def encode_offsets(off, bits):
    """What the kernel writes for a vector of offsets: all ones where the offset does
    not fit; 16/8 bits per group, or sixteen 4-bit values per little-endian 64-bit word."""
    esc = (1 << bits) - 1
    v = np.where((off >= 0) & (off < esc), off, esc).astype(np.uint64)
    if bits == 16:
        return torch.from_numpy(v.astype(np.uint16).view(np.int16))
    if bits == 8:
        return torch.from_numpy(v.astype(np.uint8))
    pad = np.concatenate([v, np.zeros((-v.size) % 16, np.uint64)]).reshape(-1, 16)
    words = (pad << (4 * np.arange(16, dtype=np.uint64))[None, :]).sum(axis=1, dtype=np.uint64)
    return torch.from_numpy(words.view(np.int64))


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


def run_mode(mode, rank, world, shift=0):
    """u16: offsets from base 0; shift > 0 moves rank 1's commit values past the
    window in one round, which must raise escapes and take the full-vector path.
    u8 / u4: offsets from the round's own input commitIndex (the round's advance);
    shift > 0 lowers rank 1's input commitIndex so that its advances leave the window."""
    per, rounds = 1000 + 37, 5                 # odd size: exercises the alignment padding
    total = per * world
    bits = MODE_BITS[mode]
    ex = SummaryExchange(per, "cpu", depth=2, mode=mode)
    assert ex.nbytes == summary_bytes(per, mode) and ex.world == world
    expect, mine_full, bases = [], [], []
    for r in range(rounds):
        slot = r % 2
        ex.begin(slot)                         # waits for the previous exchange of this slot
        if r >= 2:                             # ... whose result is checked before the slot is reused
            check(ex, slot, expect[r - 2], mine_full[r - 2], bases[r - 2], per, world)
        b = synth.gen_fused(total, 5, seed=900 + r)
        if shift and r == 3 and bits == 16:    # only rank 1's shard leaves the window
            b["commit"][per:2 * per] += shift
            b["match"][:, per:2 * per] += shift   # N is now outside the log: commit stays, but is > 65535
        if shift and r == 3 and bits < 16:
            b["commit"][per:2 * per] -= shift     # advances of rank 1's groups grow by `shift`
        whole = o.fused_round(*b.fused_args())
        expect.append(whole)
        base = None if bits >= 16 else b["commit"][:total].copy()
        bases.append(base)
        g0, g1 = shard_bounds(total, world, rank)
        assert g1 - g0 == per
        mine = o.fused_round(*shard(b, g0, g1).fused_args())
        assert np.array_equal(mine.commit, whole.commit[g0:g1])     # sharding changes no decision
        assert np.array_equal(mine.elected, whole.elected[g0:g1])
        mine_full.append(torch.from_numpy(mine.commit.copy()))
        if mode == "i32":
            ex.commit_view(slot).copy_(torch.from_numpy(mine.commit))
        else:
            off = mine.commit.astype(np.int64) - (0 if base is None else base[g0:g1])
            ex.commit_view(slot).copy_(encode_offsets(off, bits))
        ebits = np.packbits(mine.elected, bitorder="little")
        ebits = np.concatenate([ebits, np.zeros((-ebits.size) % 8, np.uint8)]).view(np.int64)
        ex.bits_view(slot).copy_(torch.from_numpy(ebits))
        ex.exchange(slot, async_op=True)
    ex.wait_all()
    for r in (rounds - 2, rounds - 1):
        check(ex, r % 2, expect[r], mine_full[r], bases[r], per, world)
    dist.barrier()


def run_batched(rank, world, R=3):
    """R consecutive rounds share one all-gather (SummaryExchange(rounds=R))."""
    per, total = 517, 517 * world
    ex = SummaryExchange(per, "cpu", depth=2, mode="u16", rounds=R)
    assert ex.nbytes == R * summary_bytes(per, "u16")
    for batch in range(3):
        slot = batch % 2
        ex.begin(slot)
        wholes = []
        for sub in range(R):
            b = synth.gen_fused(total, 3, seed=40 + batch * R + sub)
            whole = o.fused_round(*b.fused_args())
            wholes.append(whole)
            g0, g1 = shard_bounds(total, world, rank)
            mine = o.fused_round(*shard(b, g0, g1).fused_args())
            ex.commit_view(slot, sub).copy_(torch.from_numpy(mine.commit.astype(np.uint16).view(np.int16)))
            bits = np.packbits(mine.elected, bitorder="little")
            bits = np.concatenate([bits, np.zeros((-bits.size) % 8, np.uint8)]).view(np.int64)
            ex.bits_view(slot, sub).copy_(torch.from_numpy(bits))
        ex.exchange(slot)
        ex.wait(slot)
        for sub in range(R):
            assert np.array_equal(ex.gathered_commit(slot, sub).numpy().reshape(-1), wholes[sub].commit)
            bits = ex.gathered_bits(slot, sub)
            for r in range(world):
                got = np.unpackbits(bits[r].numpy().view(np.uint8), bitorder="little")[:per]
                assert np.array_equal(got, wholes[sub].elected[r * per:(r + 1) * per])
    dist.barrier()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run_batched(rank, world)
    run_mode("i32", rank, world)
    run_mode("u16", rank, world)
    run_mode("u16", rank, world, shift=70000)
    for mode, shift in (("u8", 0), ("u8", 300), ("u4", 0), ("u4", 9)):
        run_mode(mode, rank, world, shift=shift)
    if rank == 0:
        print("GLOO_SHARDED_OK")
    dist.destroy_process_group()


def check(ex, slot, whole, mine_full, base, per, world):
    esc = ex.gathered_escapes(slot)
    off = whole.commit.astype(np.int64) - (0 if base is None else base)
    limit = (1 << ex.bits) - 1
    expect_esc = [int(((off[r * per:(r + 1) * per] < 0) | (off[r * per:(r + 1) * per] >= limit)).sum())
                  for r in range(world)]
    if ex.mode != "i32":
        assert esc.tolist() == expect_esc, (ex.mode, esc.tolist(), expect_esc)
    tbase = None if base is None else torch.from_numpy(base.astype(np.int32).reshape(world, per))
    commit = ex.gathered_commit(slot, 0, base=tbase, full_local_commit=mine_full)
    assert commit.shape == (world, per)
    assert np.array_equal(commit.numpy().reshape(-1), whole.commit)
    bits = ex.gathered_bits(slot)
    for r in range(world):
        got = np.unpackbits(bits[r].numpy().view(np.uint8), bitorder="little")[:per]
        assert np.array_equal(got, whole.elected[r * per:(r + 1) * per])


if __name__ == "__main__":
    main()
