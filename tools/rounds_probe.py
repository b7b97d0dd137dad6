"""GPU: k independent 1M x 5 batches (the benchmark's headline workload: 64-group tiles, 8-bit
log-term arena entries) decided by ONE grid through rq_plan_rounds, timed with HIP events; the
program a profiler is attached to for the kernel trace of the multi-batch launch.

usage: python tools/rounds_probe.py [launches] [batches_per_launch] [sets]
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


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    G, n = 1 << 20, 5
    dev = torch.device("cuda:0")
    eng = DeviceEngine(0)
    hosts = [synth.gen_fused(G, n, seed=synth.DEFAULT_SEED + 104729 * s, ld=G + 256) for s in range(S)]
    sets = [DeviceBatch(b, dev, log_term_bits=0, tiled=True) for b in hosts]
    plans = [eng.plan_rounds([sets[(j * k + i) % S] for i in range(k)]) for j in range(S // k)]
    P = len(plans)
    ms = bench.timed_launches(torch, dev, lambda i: eng.launch_plan(plans[i % P]), launches, 10, 4 * P)
    ok = all(bench.verify_against_oracle(sets[i], hosts[i])["ok"] for i in (0, k - 1))
    bytes_per_batch = bench.alg_bytes_per_group(n) * G
    print(json.dumps(dict(batches_per_launch=k, launch_us=ms * 1e3, us_per_batch=ms * 1e3 / k,
                          frac=bytes_per_batch * k / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, verified=ok)))
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
