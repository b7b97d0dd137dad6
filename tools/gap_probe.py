#!/usr/bin/env python3
"""Where do the ~100 us between host wall time and kernel time of a 20-launch timed
region go?  Variants of the timed loop of bench.py on 12 rotating sets."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import synth
from raft_amd.device import DeviceBatch, DeviceEngine

dev = torch.device("cuda:0")
eng = DeviceEngine(0)
G, S, K = 1 << 20, 12, 20
sets = [DeviceBatch(synth.gen_fused(G, 5, seed=s, ld=G + 256), dev, log_term_bits=0) for s in range(S)]
for s in sets:
    eng.fused_round(s)
torch.cuda.synchronize()


def run(label, events, poll, pre=None):
    res = []
    for rep in range(5):
        for i in range(5):
            eng.fused_round(sets[i % S])
        torch.cuda.synchronize()
        if pre:
            pre()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        if events:
            e0.record()
        for i in range(K):
            eng.fused_round(sets[i % S])
        if events:
            e1.record()
        else:
            e2.record()
        tq = time.perf_counter()
        if poll:
            ev = e1 if events else e2
            while not ev.query():
                pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k = e0.elapsed_time(e1) * 1e3 / K if events else float("nan")
        res.append(((t1 - t0) * 1e6 / K, k, (tq - t0) * 1e6 / K))
    res.sort()
    w, k, q = res[len(res) // 2]
    print("%-52s wall %.2f us/step  kernel %.2f  enqueue %.2f  gap total %.0f us" % (label, w, k, q, (w - k) * K if k == k else -1), flush=True)


run("events + poll (bench.py)", True, True)
run("events + blocking sync", True, False)
run("no start event, poll on end event", False, True)
run("no start event, blocking sync", False, False)
# K = 1 and K = 100 for the fixed part
for kk in (1, 5, 100):
    K = kk
    run("events + poll, K = %d" % kk, True, True)
