#!/usr/bin/env python3
"""Host-buffer rq_fused_round at 1M x 5 (PCIe in the loop): ms per call from
pageable and from page-locked (rq_host_alloc) buffers, and the raw copy rates
of the same bytes for comparison."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from raft_amd import QuorumEngine, synth


def main():
    G = 1 << 20
    b = synth.gen_fused(G, 5)
    args = b.fused_args()
    in_bytes = sum(np.asarray(a).nbytes for a in args)
    with QuorumEngine(max_groups=G, max_slots=5, max_log_terms=int(b.log_term.size)) as qe:
        for name, a in (("pageable", args), ("pinned", [qe.pinned_like(x) for x in args])):
            qe.fused_round(*a)
            t0 = time.perf_counter()
            for _ in range(10):
                qe.fused_round(*a)
            print("%-8s %.2f ms per call (%.0f MB in)" % (name, (time.perf_counter() - t0) / 10 * 1e3, in_bytes / 1e6), flush=True)
        pin = [qe.pinned_like(x) for x in args]
        out = qe.round_buffers(*pin[0].shape)
        qe.fused_round(*pin, out=out)
        t0 = time.perf_counter()
        for _ in range(10):
            qe.fused_round(*pin, out=out)
        print("pinned inputs and outputs (round_buffers): %.2f ms per call" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
    # raw copies of the same volume through torch (one contiguous buffer each way)
    h = torch.empty(in_bytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(in_bytes, dtype=torch.uint8, device="cuda:0")
    o = torch.empty(36 << 20, dtype=torch.uint8).pin_memory()
    for _ in range(2):
        d.copy_(h, non_blocking=True); o.copy_(d[: o.numel()], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        o.copy_(d[: o.numel()], non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("raw pinned H2D %.1f GB/s (%.2f ms), D2H %.1f GB/s (%.2f ms)" % (
        in_bytes * 10 / (t1 - t0) / 1e9, (t1 - t0) * 100, o.numel() * 10 / (t2 - t1) / 1e9, (t2 - t1) * 100))


if __name__ == "__main__":
    main()
