#!/usr/bin/env python3
"""Replicate -> commit at 1M groups x (4 followers + the leader's own slot): the
AppendEntries batch followed by the commit round as two launches, against the batch
deciding the leader's commit rule itself (rq_append_desc.commit_round).  Slot-major planes,
8-bit follower-log and leader-log entries, 6 rotating sets, device-resident.

    python tools/replicate_commit_probe.py [launches=120]
Prints microseconds per replicate+commit step for both forms."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import _lib, synth
from raft_amd.device import DeviceEngine, encode_log_terms

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 120
G, nf, dev = 1 << 20, 4, torch.device("cuda", 0)
nr = nf + 1
eng = DeviceEngine(0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sets = []
for s in range(6):
    b = synth.gen_append(G, nf, seed=900 + s)
    c = synth.gen_fused(G, nr, seed=300 + s)
    match = np.concatenate([b.match_index, c.match[nf:nr]]).astype(np.int32)
    enc, fbits, _ = encode_log_terms(b.flog_term, bits=0)
    arena, bits, _ = encode_log_terms(c.log_term, bits=0)
    t = {k: up(b[k]) for k in ("follower_term", "prev_log_index", "prev_log_term", "n_new", "next_index")}
    t["leader_term"] = up(c.leader_term[:G])
    t["match_index"] = up(match)
    t["flog_off"], t["flog_term"] = up(b.flog_off.view(np.int32)), up(enc)
    t["follower_term_out"] = torch.zeros_like(t["follower_term"])
    t["success_mask"] = torch.zeros(G, dtype=torch.int16, device=dev)
    t["stepdown_mask"] = torch.zeros(G, dtype=torch.int16, device=dev)
    t["status"] = torch.zeros(G, dtype=torch.uint8, device=dev)
    t["commit"], t["commit_out"] = up(c.commit[:G]), torch.zeros(G, dtype=torch.int32, device=dev)
    t["log_off"], t["log_term"] = up(c.log_off.view(np.int32)), up(arena)
    t["round_status"] = torch.zeros(G, dtype=torch.uint8, device=dev)
    d = _lib.RoundDesc()
    d.struct_size = C.sizeof(_lib.RoundDesc)
    d.layout, d.what, d.G, d.n, d.ld = 4, 1, G, nr, G
    d.match, d.commit, d.leader_term = t["match_index"].data_ptr(), t["commit"].data_ptr(), t["leader_term"].data_ptr()
    d.log_off, d.log_term, d.log_terms, d.log_term_bits = t["log_off"].data_ptr(), t["log_term"].data_ptr(), int(arena.size), bits
    d.commit_out, d.status_out = t["commit_out"].data_ptr(), t["round_status"].data_ptr()
    t["_d"], t["_bits"], t["_terms"] = d, fbits, int(enc.size)
    sets.append(t)


def two(i):
    t = sets[i % len(sets)]
    eng.append_check(t, G, nf, G, flog_term_bits=t["_bits"], flog_terms=t["_terms"], wide=False)
    rc = eng._l.rq_round_dev(eng._h, torch.cuda.current_stream().cuda_stream, C.byref(t["_d"]))
    assert rc == 0, rc


def one(i):
    t = sets[i % len(sets)]
    eng.append_check(t, G, nf, G, flog_term_bits=t["_bits"], flog_terms=t["_terms"], wide=False, commit_round=t["_d"])


def timed(fn):
    for i in range(12):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(launches):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches * 1e3


a, b_, a2, b2 = timed(two), timed(one), timed(two), timed(one)
alg = 44 * nf + 4 + 2 + 2 + 2 + 4 * nr + 20
print("replicate + commit, 1M groups x (%d followers + leader), planes: two launches %.2f / %.2f us, one launch %.2f / %.2f us "
      "(%d algorithmic bytes per group as two launches: %.1f %% and %.1f %% of 8 TB/s)"
      % (nf, a, a2, b_, b2, alg, 100 * alg * G / (min(a, a2) * 1e-6) / 8e12, 100 * alg * G / (min(b_, b2) * 1e-6) / 8e12))
eng.close()
