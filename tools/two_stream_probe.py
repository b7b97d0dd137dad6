#!/usr/bin/env python3
"""Do independent rounds gain from overlapping the end of one launch with the start of the
next?  Consecutive launches on one stream run strictly one after the other; the same 1M x 5
fused rounds issued alternately on K streams may overlap.  Prints microseconds per round by
the host clock for K = 1, 2, 3, 4 (12 rotating sets, 64-group tiles, 8-bit arena entries)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import synth
from raft_amd.device import DeviceBatch, DeviceEngine

G, n, S, steps = 1 << 20, 5, 12, 480
dev = torch.device("cuda", 0)
eng = DeviceEngine(0)
sets = [DeviceBatch(synth.gen_fused(G, n, seed=11 + s, ld=G + 256), dev, log_term_bits=0, tiled=True) for s in range(S)]
for s in sets:
    eng.fused_round(s)
torch.cuda.synchronize()
for K in (1, 2, 3, 4, 1):
    streams = [torch.cuda.Stream() for _ in range(K)]
    def run(count):
        for i in range(count):
            with torch.cuda.stream(streams[i % K]):
                eng.fused_round(sets[i % S])
    run(96)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    print("%d stream(s): %.2f us per round, %.1f %% of 8 TB/s" % (K, us, 100 * 137 * G / (us * 1e-6) / 8e12), flush=True)
eng.close()
