#!/usr/bin/env python3
"""Times every benchmark configuration that fits one GPU, plus the split
kernels and the host-buffer (PCIe-inclusive) entry point, and writes one JSON
document.  Run on the MI355X box:  python tools/config_report.py [out.json]

All device timings are HIP-event time over back-to-back launches on the
stream the kernels run on, divided by the launch count; input/output sets
rotate so that neither reads nor writes stay in the 256 MB Infinity Cache.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from raft_amd import QuorumEngine, synth
from raft_amd.device import DeviceBatch, DeviceEngine

PEAK = 8.0e12
dev = torch.device("cuda:0")
eng = DeviceEngine(0)


def timed(fn, steps, warmup=10):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps


def rate(name, G, bytes_per_group, sec, extra=None):
    d = dict(config=name, groups=G, us_per_launch=sec * 1e6, decisions_per_s=G / sec,
             alg_bytes_per_group=bytes_per_group, alg_GBps=bytes_per_group * G / sec / 1e9,
             pct_hbm_peak=100 * bytes_per_group * G / sec / PEAK)
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)
    return d


out = []

# C2: 10k groups x 3 peers, commit advance only (latency-bound: 320 KB of state)
b = synth.gen_commit_small()
sets = [DeviceBatch(b, dev) for _ in range(4)]
sec_eager = timed(lambda i: eng.commit_advance(sets[i % 4]), 2000, 100)
# the eager loop is bound by the Python/ctypes launch path; a HIP graph of 200 launches
# (captured on the stream the C-ABI launches into) shows the device-side time per launch
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    for i in range(8):
        eng.commit_advance(sets[i % 4])
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for i in range(200):
            eng.commit_advance(sets[i % 4])
sec = timed(lambda i: graph.replay(), 20, 3) / 200
import oracle_lib
t0 = time.perf_counter()
for _ in range(50):
    oracle_lib.commit_advance(*b.commit_args())
cpu = (time.perf_counter() - t0) / 50
out.append(rate("C2 10k x 3 commit only (latency-bound)", b.G, 4 * 3 + 20, sec,
                dict(cpu_oracle_ns_per_group=cpu / b.G * 1e9, cpu_kind="C restatement, 1 thread",
                     us_per_launch_eager_python=sec_eager * 1e6, timing="HIP graph of 200 launches")))

# C3 and its split kernels: 1M x 5
G, n, S = 1 << 20, 5, 12
hosts = [synth.gen_fused(G, n, seed=77 + s, ld=G + 256) for s in range(S)]
for arena, bits, tiled in (("int32 log-term arena, planes", None, False), ("8-bit log-term arena entries, planes", 0, False),
                           ("8-bit log-term arena entries, 64-group tiles", 0, True)):
    sets = [DeviceBatch(b, dev, log_term_bits=bits, tiled=tiled) for b in hosts]
    for s_ in sets:                       # plans built, buffers touched
        eng.fused_round(s_); eng.commit_advance(s_); eng.vote_tally(s_)
    sec = timed(lambda i: eng.fused_round(sets[i % S]), 300, 30)
    out.append(rate("C3 1M x 5 fused round, " + arena, G, 137, sec))
    sec = timed(lambda i: eng.commit_advance(sets[i % S]), 300, 30)
    out.append(rate("1M x 5 commit advance only, " + arena, G, 40, sec))
    if bits is None or tiled:
        sec = timed(lambda i: eng.vote_tally(sets[i % S]), 300, 30)
        out.append(rate("1M x 5 vote tally only, " + ("64-group tiles" if tiled else "planes"), G, 97, sec))
    del sets
    torch.cuda.empty_cache()
# no arena on the device: the index where the leader's term starts (logs of Raft's own shape)
sets = [DeviceBatch(b, dev, tiled=True, term_start=synth.term_start_monotone(b)) for b in hosts]
for s_ in sets:
    eng.fused_round(s_)
sec = timed(lambda i: eng.fused_round(sets[i % S]), 300, 30)
out.append(rate("C3 1M x 5 fused round, log_term_start instead of the arena, 64-group tiles", G, 133, sec))
del sets
torch.cuda.empty_cache()
del hosts

# host-buffer entry point (PCIe and pageable staging inclusive) on one set
hb = synth.gen_fused(G, n, seed=5)
with QuorumEngine(max_groups=G, max_slots=n, max_log_terms=int(hb.log_term.size)) as qe:
    qe.fused_round(*hb.fused_args())
    t0 = time.perf_counter()
    for _ in range(3):
        qe.fused_round(*hb.fused_args())
    sec = (time.perf_counter() - t0) / 3
out.append(rate("1M x 5 fused round through rq_fused_round (pageable host buffers, PCIe inclusive)", G, 137, sec))
with QuorumEngine(max_groups=G, max_slots=n, max_log_terms=int(hb.log_term.size)) as qe:
    pinned = [qe.pinned_like(a) for a in hb.fused_args()]
    qe.fused_round(*pinned)
    t0 = time.perf_counter()
    for _ in range(3):
        qe.fused_round(*pinned)
    sec = (time.perf_counter() - t0) / 3
out.append(rate("same, inputs in rq_host_alloc (page-locked) memory; outputs pageable", G, 137, sec))
torch.cuda.empty_cache()

# C4 at one GPU: 8M x 5
G8 = 8 << 20
hosts = [synth.gen_fused(G8, n, seed=300 + s, ld=G8 + 256) for s in range(2)]
for arena, bits, tiled in (("int32 log-term arena, planes", None, False), ("8-bit log-term arena entries, planes", 0, False),
                           ("8-bit log-term arena entries, 64-group tiles", 0, True)):
    sets = [DeviceBatch(b, dev, log_term_bits=bits, tiled=tiled) for b in hosts]
    sec = timed(lambda i: eng.fused_round(sets[i % 2]), 40, 5)
    out.append(rate("C4 8M x 5 fused round on ONE GPU (2 sets, 2.8 GB), " + arena, G8, 137, sec))
    del sets
    torch.cuda.empty_cache()
del hosts
torch.cuda.empty_cache()

# C5: 1M mixed 3-9 + joint, Zipf logs (arena shared between the sets): every group through its
# member masks (groups ordered by highest slot), then ordered by cluster size and launched with
# the segment table, then one image per size in ONE launch; tiles throughout
mem = members = None
for order, label in ((True, "ordered by highest slot, every group through its member masks"),
                     ("size", "ordered by cluster size, segment table"),
                     ("images", "one image per cluster size + one with member masks, one launch (rq_plan_rounds)")):
    for bits, arena in ((None, "int32 log-term arena"), (0, "8-bit log-term arena entries")):
        first = synth.gen_mixed(G, seed=500, bucketed=True if order is True else "size")
        hosts = [first] + [synth.gen_mixed(G, seed=500 + s, bucketed=True if order is True else "size", logs_from=first)
                           for s in range(1, 8)]
        mem = (first.old_mask | first.new_mask).astype(np.int64)
        members = float(sum((mem >> p) & 1 for p in range(16)).mean())
        kw = {} if bits is None else dict(log_term_bits=bits)
        if order == "images":
            arena_owner = DeviceBatch(first, dev, tiled=True, **kw)
            isets = [[DeviceBatch(sub, dev, tiled=True, share_log_with=arena_owner, own_log_off=True)
                      for _, _, _, sub in synth.split_by_size(hb)] for hb in hosts]
            plans = [eng.plan_rounds(imgs) for imgs in isets]
            sec = timed(lambda i: eng.launch_plan(plans[i % 8]), 200, 20)
            del isets, plans, arena_owner
        else:
            sets = []
            for hb in hosts:
                sets.append(DeviceBatch(hb, dev, tiled=True, share_log_with=sets[0] if sets else None,
                                        segments=synth.segment_table(hb.old_mask, hb.new_mask) if order == "size" else None,
                                        **({} if sets else kw)))
            sec = timed(lambda i: eng.fused_round(sets[i % 8]), 200, 20)
            del sets
        out.append(rate("C5 1M mixed 3-9 peers + 10%% joint, Zipf logs, %s; %s" % (arena, label), G, 20 * members + 41, sec,
                        dict(mean_member_slots=members, arena_entries=int(first.log_term.size))))
        del hosts, first
        torch.cuda.empty_cache()

# C3 again: k of the 1M x 5 batches decided by ONE grid (rq_plan_rounds)
hosts = [synth.gen_fused(G, 5, seed=3000 + s, ld=G + 256) for s in range(12)]
sets = [DeviceBatch(b, dev, log_term_bits=0, tiled=True) for b in hosts]
for k in (2, 4, 6):
    plans = [eng.plan_rounds([sets[(j * k + i) % 12] for i in range(k)]) for j in range(12 // k)]
    sec = timed(lambda i: eng.launch_plan(plans[i % len(plans)]), 120, 12)
    out.append(rate("C3 1M x 5 fused round, %d batches per launch (rq_plan_rounds), 8-bit arena entries, tiles; per batch" % k,
                    G, 137, sec / k, dict(batches_per_launch=k, us_per_grid=sec * 1e6)))
    del plans
del sets, hosts
torch.cuda.empty_cache()

# AppendEntries consistency batch: 1M groups, 4 followers each
from raft_amd.device import encode_log_terms


def append_sets(n, use_mask, term_bits, window, count=6, commit_rule=False):
    """Device tensors of `count` AppendEntries batches.  n = 5 with the leader's own slot
    masked out (holes in every plane line), or n = 4 follower planes and no mask."""
    out = []
    for s_ in range(count):
        b = synth.gen_append(G, n, seed=900 + s_)
        if window:
            b = synth.window_flogs(b)
        enc, bits, _ = encode_log_terms(b.flog_term, bits=term_bits)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        t = {k: up(b[k]) for k in ("follower_term", "leader_term", "prev_log_index", "prev_log_term", "n_new",
                                   "match_index", "next_index")}
        t["flog_off"] = up(b.flog_off.view(np.int32))
        t["flog_term"] = up(enc)
        if use_mask:
            t["active_mask"] = up(b.active_mask.view(np.int16))
        if window:
            t["flog_base"] = up(b.flog_base)
        if commit_rule:
            cb = synth.with_follower_commit(b, seed=s_)
            t.update({k: up(cb[k]) for k in ("leader_commit", "last_new_index", "follower_commit")})
        t["follower_term_out"] = torch.zeros_like(t["follower_term"])
        t["success_mask"] = torch.zeros(G, dtype=torch.int16, device=dev)
        t["stepdown_mask"] = torch.zeros(G, dtype=torch.int16, device=dev)
        t["status"] = torch.zeros(G, dtype=torch.uint8, device=dev)
        t["_bits"], t["_terms"] = bits, int(enc.size)
        out.append(t)
    return out


# per active follower: term, prevIdx, prevTerm, off, gathered term, n_new, match, next in; term', match', next' out
APPEND_ALG = 44 * 4 + 4 + 2 + 2 + 2
for label, n_planes, use_mask, term_bits, window, commit_rule in (
        ("5 slot planes, leader's slot masked out, int32 follower logs", 5, True, 32, False, False),
        ("5 slot planes, leader's slot masked out, 8-bit follower-log entries", 5, True, 0, False, False),
        ("4 follower planes (no mask, no holes), 8-bit entries", 4, False, 0, False, False),
        ("4 follower planes, 8-bit entries, logs windowed to the suffix around prevLogIndex", 4, False, 0, True, False),
        ("4 follower planes, 8-bit entries, + the follower commit rule (raft.go:172-176) in the same launch",
         4, False, 0, False, True)):
    dab = append_sets(n_planes, use_mask, term_bits, window, commit_rule=commit_rule)
    sec = timed(lambda i: eng.append_check(dab[i % len(dab)], G, n_planes, G, flog_term_bits=dab[i % len(dab)]["_bits"],
                                           flog_terms=dab[i % len(dab)]["_terms"]), 200, 20)
    out.append(rate("AppendEntries consistency batch 1M groups x 4 followers: " + label, G,
                    APPEND_ALG + (12 * 4 + 4 if commit_rule else 0), sec,
                    dict(term_bits=dab[0]["_bits"], arena_entries=dab[0]["_terms"])))
    del dab
    torch.cuda.empty_cache()

# replicate -> commit: the batch + the commit round over its matchIndex array (4 followers + the
# leader's own slot), as two launches and as one (rq_append_desc.commit_round)
import ctypes as C
from raft_amd import _lib
dab = append_sets(4, False, 0, False)
for s_, t in enumerate(dab):
    c = synth.gen_fused(G, 5, seed=300 + s_)
    arena, cbits, _ = encode_log_terms(c.log_term, bits=0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t["match_index"] = torch.cat([t["match_index"], up(c.match[4:5, :G])])
    t["leader_term"] = up(c.leader_term[:G])
    t["commit"], t["commit_out"] = up(c.commit[:G]), torch.zeros(G, dtype=torch.int32, device=dev)
    t["log_off"], t["log_term"] = up(c.log_off.view(np.int32)), up(arena)
    t["round_status"] = torch.zeros(G, dtype=torch.uint8, device=dev)
    d = _lib.RoundDesc()
    d.struct_size = C.sizeof(_lib.RoundDesc)
    d.layout, d.what, d.G, d.n, d.ld = 4, 1, G, 5, G
    d.match, d.commit, d.leader_term = t["match_index"].data_ptr(), t["commit"].data_ptr(), t["leader_term"].data_ptr()
    d.log_off, d.log_term, d.log_terms, d.log_term_bits = t["log_off"].data_ptr(), t["log_term"].data_ptr(), int(arena.size), cbits
    d.commit_out, d.status_out = t["commit_out"].data_ptr(), t["round_status"].data_ptr()
    t["_round"] = d


def two_launches(i):
    t = dab[i % len(dab)]
    eng.append_check(t, G, 4, G, flog_term_bits=t["_bits"], flog_terms=t["_terms"])
    assert eng._l.rq_round_dev(eng._h, torch.cuda.current_stream().cuda_stream, C.byref(t["_round"])) == 0


REPL_ALG = APPEND_ALG + 4 * 5 + 20
out.append(rate("replicate + commit, 1M groups x (4 followers + leader): AppendEntries batch, then the commit round (two launches)",
                G, REPL_ALG, timed(two_launches, 200, 20)))
out.append(rate("replicate + commit, 1M groups x (4 followers + leader): one launch (rq_append_desc.commit_round); bytes as for two",
                G, REPL_ALG, timed(lambda i: eng.append_check(dab[i % len(dab)], G, 4, G, flog_term_bits=dab[i % len(dab)]["_bits"],
                                                              flog_terms=dab[i % len(dab)]["_terms"],
                                                              commit_round=dab[i % len(dab)]["_round"]), 200, 20)))
del dab
torch.cuda.empty_cache()

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "configs.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
