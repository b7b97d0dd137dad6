#!/usr/bin/env python3
"""The AppendEntries consistency batch alone, for profiling: 1M groups x 4 follower
planes (no mask), 8-bit follower-log entries, 6 rotating sets, device-resident.

    python tools/append_probe.py [launches=120] [term_bits=0] [tiles|planes] [clamp|-] [masked]
("masked", planes only: 5 slot planes with every group's own leader slot masked out instead.)
Prints microseconds per launch and the fraction of HBM peak (186 algorithmic
bytes per group; with "clamp" the follower commit rule of raft.go:172-176 runs in the
same launch: 12 more bytes per follower and 4 per group, 238)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import synth
from raft_amd.device import DeviceEngine, TiledAppendBatch, encode_log_terms

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 120
term_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tiled = not (len(sys.argv) > 3 and sys.argv[3] == "planes")
clamp = len(sys.argv) > 4 and sys.argv[4] == "clamp"
masked = len(sys.argv) > 5 and sys.argv[5] == "masked"
G, n, dev = 1 << 20, 5 if masked else 4, torch.device("cuda", 0)
eng = DeviceEngine(0)
sets = []
for s in range(6):
    b = synth.gen_append(G, n, seed=900 + s)
    if clamp:
        b = synth.with_follower_commit(b, seed=s)
    if tiled:
        del b["active_mask"]                       # n follower planes, no mask
        tb = TiledAppendBatch(b, dev, term_bits=term_bits)
        tb.t["_bits"], tb.t["_terms"], tb.t["_keep"] = tb.term_bits, tb.flog_terms, tb
        sets.append(tb.t)
        continue
    enc, bits, _ = encode_log_terms(b.flog_term, bits=term_bits)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t = {k: up(b[k]) for k in ("follower_term", "leader_term", "prev_log_index", "prev_log_term", "n_new",
                               "match_index", "next_index")}
    if clamp:
        t.update({k: up(b[k]) for k in ("leader_commit", "last_new_index", "follower_commit")})
    if masked:
        t["active_mask"] = up(b.active_mask.view(np.int16))
    t["flog_off"] = up(b.flog_off.view(np.int32))
    t["flog_term"] = up(enc)
    t["follower_term_out"] = torch.zeros_like(t["follower_term"])
    t["success_mask"] = torch.zeros(G, dtype=torch.int16, device=dev)
    t["stepdown_mask"] = torch.zeros(G, dtype=torch.int16, device=dev)
    t["status"] = torch.zeros(G, dtype=torch.uint8, device=dev)
    t["_bits"], t["_terms"] = bits, int(enc.size)
    sets.append(t)


def launch(i):
    t = sets[i % len(sets)]
    eng.append_check(t, G, n, G, flog_term_bits=t["_bits"], flog_terms=t["_terms"], wide=False)


for i in range(12):
    launch(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(launches):
    launch(i)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / launches * 1e3
alg = 44 * 4 + 4 + 2 + 2 + 2 + ((12 * 4 + 4) if clamp else 0)
print("append_check%s 1M x 4 followers, %s, %d-bit entries: %.2f us per launch, %.1f %% of 8 TB/s (%d B per group)"
      % (" + commit rule" if clamp else "", "64-group tiles" if tiled else ("5 planes, leader's slot masked" if masked else "planes"), sets[0]["_bits"], us,
         100 * alg * G / (us * 1e-6) / 8e12, alg))
eng.close()
