"""GPU: the benchmark's headline launches (1M x 5, 64-group tiles, 8-bit log-term arena entries, 12
rotating sets) enqueued one by one against the same launches replayed from a HIP graph (captured on
the stream the C-ABI launches into), HIP-event time per launch.

usage: python tools/graph_probe.py [launches_per_graph=120]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                                   # noqa: E402
from raft_amd import synth                     # noqa: E402
from raft_amd.device import DeviceBatch, DeviceEngine   # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 120
G, n, S = 1 << 20, 5, 12
dev = torch.device("cuda:0")
eng = DeviceEngine(0)
hosts = [synth.gen_fused(G, n, seed=synth.DEFAULT_SEED + 104729 * s, ld=G + 256) for s in range(S)]
sets = [DeviceBatch(b, dev, log_term_bits=0, tiled=True) for b in hosts]
eager = bench.timed_launches(torch, dev, lambda i: eng.fused_round(sets[i % S]), 400, 40, 4 * S)
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    for i in range(S):
        eng.fused_round(sets[i % S])
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for i in range(L):
            eng.fused_round(sets[i % S])
replay = bench.timed_launches(torch, dev, lambda i: graph.replay(), 8, 2, 2) / L
ok = bench.verify_against_oracle(sets[0], hosts[0])["ok"]
bpg = bench.alg_bytes_per_group(n) * G
print(json.dumps(dict(eager_us=eager * 1e3, graph_us=replay * 1e3, launches_per_graph=L,
                      frac_eager=bpg / (eager * 1e-3) / 1e9 / bench.HBM_PEAK_GBS,
                      frac_graph=bpg / (replay * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, verified=ok)))
