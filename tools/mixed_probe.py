"""GPU: the mixed / joint workload (1M groups of 3-9 peers, 10 % in joint consensus, Zipf logs)
decided three ways -- every group through its member masks with the groups ordered by highest
slot, the same with the groups ordered by cluster size, and the
size order launched with the segment table -- each with the log-term lookup and without
(log_term_start).  Prints one JSON object; set 0 of every variant is checked against the oracle.

usage: python tools/mixed_probe.py [groups] [steps] [sets]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                                   # noqa: E402
from raft_amd import synth                     # noqa: E402
from raft_amd.device import DeviceBatch, DeviceEngine   # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    variant = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
    tiled = os.environ.get("RQ_PROBE_PLANES") is None
    dev = torch.device("cuda:0")
    eng = DeviceEngine(0)
    out = {"groups": G, "steps": steps, "sets": S, "variant": variant, "tiled": tiled}
    for order in (True, "size"):
        first = synth.gen_mixed(G, seed=500, bucketed=order)
        hosts = [first] + [synth.gen_mixed(G, seed=500 + s, bucketed=order, logs_from=first) for s in range(1, S)]
        mem = (first.old_mask | first.new_mask).astype(np.int64)
        members = float(sum((mem >> p) & 1 for p in range(16)).mean())
        bpg = 20 * members + 41
        for use_segs in ((False, True) if order == "size" else (False,)):
            for no_arena in (False, True):
                sets = []
                for i, hb in enumerate(hosts):
                    kw = dict(tiled=tiled)
                    if use_segs:
                        kw["segments"] = synth.segment_table(hb.old_mask, hb.new_mask)
                        assert kw["segments"] is not None
                    if no_arena:
                        kw["term_start"] = synth.term_start_monotone(hb)
                    elif i:
                        kw["share_log_with"] = sets[0]
                    sets.append(DeviceBatch(hb, dev, **kw))
                ms = bench.timed_launches(torch, dev, lambda i: eng.fused_round(sets[i % S], variant=variant), steps, 20, S)
                b = bpg - 4 if no_arena else bpg
                name = ("by_size" if order == "size" else "by_top_slot") + ("_segmented" if use_segs else "") + \
                       ("_term_start" if no_arena else "")
                ok = bench.verify_against_oracle(sets[0], synth.raft_valid(first) if no_arena else first)
                out[name] = dict(kernel_us=round(ms * 1e3, 2), frac=round(b * G / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4), alg_bytes_per_group=b,
                                 verified=ok["ok"], segments=len(kw["segments"][1]) if use_segs else 0)
                print(name, out[name], file=sys.stderr, flush=True)
                del sets
                torch.cuda.empty_cache()
    # one image per cluster size in its own compact tiles + one with member masks, ONE launch
    first = synth.gen_mixed(G, seed=500, bucketed="size")
    hosts = [first] + [synth.gen_mixed(G, seed=500 + s, bucketed="size", logs_from=first) for s in range(1, S)]
    mem = (first.old_mask | first.new_mask).astype(np.int64)
    bpg = 20 * float(sum((mem >> p) & 1 for p in range(16)).mean()) + 41
    for no_arena in (False, True):
        sets, plans, arena = [], [], None
        for hb in hosts:
            imgs = []
            for k, lo, hi, sub in synth.split_by_size(hb):
                kw = dict(tiled=tiled)
                if no_arena:
                    kw["term_start"] = synth.term_start_monotone(sub)
                elif arena is not None:
                    kw["share_log_with"], kw["own_log_off"] = arena, True
                imgs.append(DeviceBatch(sub, dev, **kw))
                if arena is None and not no_arena:
                    arena = imgs[0]
            sets.append(imgs)
            plans.append(eng.plan_rounds(imgs, variant=variant))
        ms = bench.timed_launches(torch, dev, lambda i: eng.launch_plan(plans[i % S]), steps, 20, S)
        ok = all(bench.verify_against_oracle(db, sub)["ok"] for db, (_, _, _, sub) in zip(
            sets[0], synth.split_by_size(synth.raft_valid(first) if no_arena else first)))
        # masks are read for the groups of the member-mask image only
        b = (bpg - 4 if no_arena else bpg)
        name = "images_by_size" + ("_term_start" if no_arena else "")
        out[name] = dict(kernel_us=round(ms * 1e3, 2), frac=round(b * G / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4),
                         alg_bytes_per_group=b, verified=ok, images=len(sets[0]))
        print(name, out[name], file=sys.stderr, flush=True)
        del sets, plans, arena
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
