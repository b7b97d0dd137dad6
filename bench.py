#!/usr/bin/env python3
"""Benchmark of the batched Raft quorum engine on MI355X.

A step = one fused round (RequestVote tally + commitIndex advance with the
log-term gather) over this rank's 1M groups x 5 peers, inputs resident in HBM
(by default in 64-group tiles, one input and one output stream per wave; --planes
keeps slot-major planes).  Input/output sets rotate (default 12 sets, ~2.1 GB in total: both the reads
and the 36 MB of writes per step exceed the 256 MB Infinity Cache, which
otherwise absorbs the writes of up to 5 sets and flatters the number).  Before
anything is timed every set's prepared launch is built and one untimed rotation
over all sets is run, so the --warmup / --steps the caller asks for are honoured
exactly and no timed step is the first touch of its buffers.

With N > 1 ranks the groups are sharded (weak scaling: 1M groups per GPU;
--total-groups T instead block-partitions T groups over the ranks, strong
scaling, e.g. 8388608) and the commit/elected summary of every step is
all-gathered over RCCL (one collective per 4 steps by default), overlapped with
the following steps' kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (no WORLD_SIZE in the environment) and with
--gpus N > 1, this process starts the N rank processes itself -- before it
touches torch or the GPU -- and relays rank 0's line.

Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak
METRIC = "quorum decisions/sec at 1M groups x 5 peers (1/8 GPU); % of HBM peak"
UNIT = "decisions/s"
# the round kernels and their argument structure (rq_kernels.h also declares the other kernels' arguments)
KERNEL_SOURCES = ("raft_amd/csrc/rq_round_impl.h", "raft_amd/csrc/rq_kernels.h:rq_round_args_t")


def alg_bytes_per_group(n, what="fused", layout="i32"):
    """Algorithmic bytes per group-round.  int32 layout: commit 4n+20, vote 16n+17;
    int64 layout (every value 8 bytes, log offsets still 4): commit 8n+36, vote 32n+29."""
    e = 8 if layout == "i64" else 4
    commit = e * n + 3 * e + 4 + e   # n match + commit + leaderTerm + gathered term + logOff ; commit'
    vote = 4 * e * n + 3 * e + 5     # 3n planes + candTerm + candId ; n voterTerm' + maxTerm + 2 + 2 + 1
    return {"commit": commit, "vote": vote, "fused": commit + vote}[what]


def kernel_source_hash():
    """Fingerprint of the round kernels' code (comments and white space stripped, so
    rewording a comment does not detach a profile): a PMC traffic figure taken from a
    profile is only attached to a run of the same kernel code."""
    import re
    h = hashlib.sha1()
    for rel in KERNEL_SOURCES:
        rel, _, struct = rel.partition(":")
        with open(os.path.join(ROOT, rel), "r") as f:
            text = f.read()
        if struct:                                   # only this structure's definition
            start = text.index("struct " + struct + " {")
            text = text[start: text.index("\n};", start)]
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--groups", type=int, default=1 << 20, help="groups per GPU")
    ap.add_argument("--total-groups", type=int, default=0,
                    help="strong scaling: this many groups in total, block-partitioned over the ranks "
                         "(8388608 = the 8M-group configuration); overrides --groups")
    ap.add_argument("--peers", type=int, default=5)
    ap.add_argument("--sets", type=int, default=12,
                    help="rotating input+output sets; 12 x ~176 MB keeps reads AND writes out of the 256 MB Infinity Cache")
    ap.add_argument("--variant", type=lambda s: int(s, 0), default=0, help="kernel build selector")
    ap.add_argument("--workload", choices=["fused", "mixed"], default="fused")
    ap.add_argument("--what", choices=["fused", "commit", "vote"], default="fused")
    ap.add_argument("--ld-pad", type=int, default=256,
                    help="plane stride = groups + pad (elements): keeps the slot planes off one power-of-two stride")
    ap.add_argument("--no-exchange", action="store_true", help="N>1: skip the summary all-gather")
    ap.add_argument("--summary", choices=["auto", "u4", "u8", "u16", "i32"], default="auto",
                    help="commit summary on the wire: u4/u8 = the round's commitIndex advance in 4/8 bits "
                         "(offset from the input commitIndex, all ones = escape), u16 = 16-bit offset from 0, "
                         "i32 = the full vector; auto = u8 for the fused workload (advances below 16), u16 "
                         "for the mixed one (logs of up to 4096 entries)")
    ap.add_argument("--rounds-per-exchange", type=int, default=4,
                    help="summaries of this many consecutive rounds travel in one all-gather")
    ap.add_argument("--exchange-impl", choices=["auto", "native", "torch"], default="torch",
                    help="summary all-gather through librq's own RCCL communicator or torch.distributed")
    ap.add_argument("--fence", choices=["event", "signal"], default="signal",
                    help="how the exchange stream learns that a slot's rounds are done: an event recorded "
                         "behind the last of them, or the start signal of the launch that follows "
                         "(rq_round_desc.start_signal)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="initialise torch.distributed and run the summary all-gather even with one rank")
    ap.add_argument("--consume", action="store_true",
                    help="with the exchange: every gathered summary is also decoded on the device "
                         "(commit offsets added to this rank's replica of all ranks' commitIndex)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (wide layout, mixed/joint workload, 32-bit log arena)")
    ap.add_argument("--log-window", type=int, default=-1,
                    help=">= 0: keep only the uncommitted suffix of every log plus this many committed "
                         "entries on the device (log_base window); -1: full logs (the default workload)")
    ap.add_argument("--log-term-bits", type=int, choices=[0, 8, 16, 32], default=0,
                    help="width of the log-term arena entries; 0 = the narrowest the batch's terms fit "
                         "(what a host keeping the arena resident would choose), 32 = the plain int32 arena")
    ap.add_argument("--no-segments", action="store_true",
                    help="mixed workload: groups ordered by highest slot and every group decided through its member "
                         "masks, instead of ordered by cluster size and launched with the segment table")
    ap.add_argument("--planes", action="store_true",
                    help="keep every array as a slot-major plane field[p * ld + g] instead of the default 64-group "
                         "tiles (rq_round_desc.tile_bytes: one input and one output stream per wave)")
    ap.add_argument("--layout", choices=["i32", "i64"], default="i32",
                    help="i64: the wide layout (RQ_LAYOUT_I64 descriptors), values shifted beyond the int32 range")
    ap.add_argument("--no-bits", action="store_true", help="skip the packed elected-bit summary output")
    ap.add_argument("--no-status", action="store_true", help="skip the per-group status byte output")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU work per baseline leg")
    ap.add_argument("--preheat-rotations", type=int, default=32,
                    help="untimed passes over all input sets before --warmup (the first builds every "
                         "prepared launch and touches every buffer)")
    ap.add_argument("--strong-total", type=int, default=8388608,
                    help="2 or 4 ranks: total groups of the additional strong-scaling figure (extra.strong_8m)")
    ap.add_argument("--watchdog-seconds", type=float, default=90.0,
                    help="a rank that retires no step for this long dumps its stacks and exits with code 3 (0 = off)")
    ap.add_argument("--launcher-timeout", type=float, default=540.0,
                    help="--gpus N without a launcher: give up, and kill every rank, after this many seconds")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    return ap.parse_args()


def supervise(procs, deadline_s, out=None):
    """Wait for the rank processes `procs` (each started in a session of its own).  Rank 0's
    stdout (a pipe) is collected; the first rank that exits non-zero, or the deadline, ends the
    run: every rank still alive is killed with its whole process group, so no rank is left
    holding a GPU inside a collective whose peer is gone.  Returns (exit code, rank 0's stdout)."""
    import signal
    import threading
    chunks = []
    reader = None
    if procs[0].stdout is not None:
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    def on_term(signum, frame):
        kill_all()
        os._exit(128 + signum)

    old = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            old[sig] = signal.signal(sig, on_term)
    rc, why = 0, None
    t_end = time.monotonic() + deadline_s
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = 1, "rank(s) failed: %s" % bad
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > t_end:
                rc, why = 1, "no result after %.0f s; ranks still running: %s" % (
                    deadline_s, [r for r, c in enumerate(codes) if c is None])
                break
            time.sleep(0.05)
    finally:
        kill_all()
        for sig, h in old.items():
            signal.signal(sig, h)
    if reader is not None:
        reader.join(timeout=5)
    if why:
        (out or sys.stderr).write("bench.py: %s\n" % why)
    return rc, b"".join(c for c in chunks if c)


def launch_ranks(args):
    """--gpus N without a launcher: this process (which has not touched torch or the
    GPU) starts the N ranks as children with the rendezvous in their environment,
    relays rank 0's JSON line and exits non-zero if any rank does (the other ranks are
    killed at once) or when --launcher-timeout runs out."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out,
                                      start_new_session=True))
    rc, line = supervise(procs, args.launcher_timeout)
    # a failed run relays what rank 0 still wrote: nothing, or the watchdog's line marked "valid": false
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    sys.exit(rc)


class Watchdog:
    """Per-rank progress watchdog.  While armed, beat() must be called at least every `limit_s`
    seconds (every step and every fence stage does); otherwise the rank writes what it was
    doing -- the Python stacks of all threads and the exchange's host-side state -- to stderr
    and leaves with exit code 3, which makes the launcher end the other ranks: a wedged fence
    or collective then costs a minute and leaves a diagnosis instead of running into the
    caller's time limit with nothing."""

    def __init__(self, limit_s, describe=None, out=None, exit_fn=None, last_words=None):
        import threading
        self.limit_s, self.describe, self.last_words = float(limit_s), describe, last_words
        self.out = out or sys.stderr
        self.exit_fn = exit_fn or (lambda code: os._exit(code))
        self.armed, self.last, self.where, self.beats = False, time.monotonic(), "start", 0
        self.fired = False
        if self.limit_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def beat(self, where):
        self.last, self.where = time.monotonic(), where
        self.beats += 1

    def arm(self, where):
        self.beat(where)
        self.armed = True

    def disarm(self):
        self.armed = False

    def _run(self):
        import faulthandler
        while not self.fired:
            time.sleep(min(1.0, self.limit_s / 4))
            idle = time.monotonic() - self.last
            if self.armed and idle > self.limit_s:
                self.fired = True
                self.out.write("bench.py watchdog: rank %s made no progress for %.0f s (last: %s, %d beats)\n" % (
                    os.environ.get("RANK", "0"), idle, self.where, self.beats))
                if self.describe is not None:
                    try:
                        self.out.write("bench.py watchdog: %s\n" % json.dumps(self.describe(), default=str))
                    except Exception as e:          # the diagnosis must not stop the exit
                        self.out.write("bench.py watchdog: state unavailable: %r\n" % (e,))
                self.out.flush()
                try:
                    faulthandler.dump_traceback(file=self.out, all_threads=True)
                except Exception:                   # not a real file (no descriptor): format the stacks here
                    import traceback
                    for tid, frame in sys._current_frames().items():
                        self.out.write("Thread 0x%x:\n%s" % (tid, "".join(traceback.format_stack(frame))))
                self.out.flush()
                if self.last_words is not None:
                    try:
                        self.last_words()
                    except Exception as e:
                        self.out.write("bench.py watchdog: last words failed: %r\n" % (e,))
                self.exit_fn(3)


def cpu_baseline(args, synth):
    """Oracle (C restatement of raft.go: heap Node objects, one RequestVoteRPC-
    shaped call per voter, one sort per leader) timed on the host cores, on a
    bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    G = min(args.groups, 1 << 18)
    b = synth.gen_fused(G, args.peers, seed=synth.DEFAULT_SEED)
    out = {}
    threads_all = max(1, min(16, len(os.sched_getaffinity(0))))
    for label, threads in (("single", 1), ("multi", threads_all)):
        sec1, _ = oracle_lib.cpu_baseline(b, 1, threads)
        passes = max(1, min(2000, int(args.cpu_seconds / max(sec1, 1e-6))))
        sec, _ = oracle_lib.cpu_baseline(b, passes, threads)
        out[label] = dict(value=G * passes / sec, cores=threads, passes=passes, seconds=sec)
    return dict(value=out["multi"]["value"], unit="decisions/s", cores=out["multi"]["cores"],
                kind="port",
                sample="%d groups x %d peers (same generator as the GPU workload), %d passes, %.1f s; "
                       "C restatement of raft.go (Go toolchain absent)" % (
                           G, args.peers, out["multi"]["passes"], out["multi"]["seconds"]),
                single_thread_value=out["single"]["value"], host_cpus=os.cpu_count())


def verify_against_oracle(db, hb, what=3, want_bits=True, status=True):
    """Proof that the timed launches computed the right thing: the outputs DeviceBatch db holds
    after the timed region (every launch rewrites them from constant inputs) against the CPU
    oracle's decisions for the same host batch hb (raft.go:181-223 per voter slot, :151 for the
    looked-up term), bit for bit on every group.  The oracle is the checker here, never the
    thing timed.  Returns the `verified` object of the JSON line."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    G = db.G
    res = db.results()
    kw = dict(G=G)
    if "old_mask" in hb:
        kw.update(old_mask=hb["old_mask"], new_mask=hb["new_mask"])
    if "log_base" in hb:
        kw["log_base"] = hb["log_base"]
    commit_in = (hb["match"], hb["commit"], hb["leader_term"], hb["log_off"], hb["log_term"])
    vote_in = (hb["voter_term"], hb["voted_for"], hb["last_log_term"], hb["cand_term"], hb["cand_id"])
    if db.wide:
        names = ("match", "commit", "leader_term", "log_off", "log_term", "voter_term", "voted_for",
                 "last_log_term", "cand_term", "cand_id")
        exp = o.round64(what, **{k: hb[k] for k in names[:5] if what & 1}, **{k: hb[k] for k in names[5:] if what & 2}, **kw)
    elif what == 3:
        exp = o.fused_round(*commit_in, *vote_in, **kw)
    elif what == 1:
        c, st = o.commit_advance(*commit_in, **kw)
        exp = dict(commit=c, status=st)
    else:
        kw.pop("log_base", None)
        exp = o.vote_tally(*vote_in, **kw)
    bad, checked = [], []

    def same(name, got, want):
        checked.append(name)
        if not np.array_equal(got, want):
            bad.append(name)

    if what & 1:
        same("commit", res["commit"], exp["commit"][:G])
    if what & 2:
        for f in ("grant_mask", "stepdown_mask", "elected", "max_term"):
            same(f, res[f], exp[f][:G])
        gv, ev = res["voter_term"][:, :G], exp["voter_term"][:, :G]
        if "old_mask" in hb:                    # slots outside both configurations are not evaluated
            mem = (hb["old_mask"] | hb["new_mask"]).astype(np.int64)
            mm = ((mem[None, :G] >> np.arange(gv.shape[0])[:, None]) & 1).astype(bool)
            gv, ev = gv[mm], ev[mm]
        same("voter_term", gv, ev)
        if want_bits:
            bits = np.unpackbits(res["elected_bits"].view(np.uint8), bitorder="little")
            same("elected_bits", bits[:G], exp["elected"][:G])
    if status:
        same("status", res["status"], exp["status"][:G])
    return {"vs": "oracle", "groups": G, "ok": not bad, "outputs_checked": checked, "mismatched": bad,
            "layout": "%s, %s, %d-bit log-term arena entries" % (
                "int64" if db.wide else "int32", "64-group tiles" if db.tile_bytes else "slot-major planes",
                db.log_term_bits) + (", log_term_start" if db.term_start else "")}


def timed_launches(torch, device, launch, steps, warmup, preheat):
    """HIP-event time per launch of launch(i) on the current stream: `preheat`
    untimed launches (first touch, prepared launches built), `warmup` more, then
    `steps` timed ones.  The timed launches are bracketed in four runs of steps / 4 by
    events recorded between them and the median run is reported, and Python's cyclic
    garbage collector is held off while launches are being enqueued: a full collection in
    a process that has imported torch takes tens of milliseconds, the queue runs dry
    meanwhile, and whichever figure that landed in read ten times too slow."""
    for i in range(preheat):
        launch(i)
    torch.cuda.synchronize(device)
    for i in range(warmup):
        launch(i)
    torch.cuda.synchronize(device)
    runs = 4 if steps >= 40 else 1
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(runs + 1)]
    bounds = [steps * r // runs for r in range(runs + 1)]
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        marks[0].record()
        for r in range(runs):
            for i in range(bounds[r], bounds[r + 1]):
                launch(i)
            marks[r + 1].record()
        torch.cuda.synchronize(device)
    finally:
        if gc_was_on:
            gc.enable()
    raw = [marks[r].elapsed_time(marks[r + 1]) / (bounds[r + 1] - bounds[r]) for r in range(runs)]
    per = sorted(raw)
    mid = per[(runs - 1) // 2]
    if per[-1] > 2 * mid:
        sys.stderr.write("bench.py: one run of a timed region stalled: ms per launch by run %s, median taken\n"
                         % [round(x, 4) for x in raw])
    return mid


def extras(args, torch, synth, DeviceBatch, eng, device, host_batches, out):
    """Secondary figures, each a few hundred launches: the same fused round in the
    wide (int64) layout -- Go's int is 64-bit, raft.go:35-68 --, with the plain
    32-bit log-term arena, and the mixed 3-9 peer + joint-consensus workload."""
    import numpy as np
    G, n = args.groups, args.peers
    steps, warmup = 120, 12

    def frac(bytes_per_group, ms):
        return bytes_per_group * G / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    tiled = not args.planes
    # plain int32 arena (no narrowing), otherwise the headline workload
    sets = [DeviceBatch(b, device, log_term_bits=32, tiled=tiled) for b in host_batches]
    S = len(sets)
    ms = timed_launches(torch, device, lambda i: eng.fused_round(sets[i % S]), steps, warmup, S)
    out["arena_i32"] = dict(kernel_ms=ms, frac=frac(alg_bytes_per_group(n), ms), input_sets=S,
                            note="log-term arena kept as int32 entries")
    del sets
    torch.cuda.empty_cache()
    # Independent rounds on several streams: launches on ONE stream run strictly one after the
    # other (a 1M-group launch spends ~2 us ramping up and draining: 8M groups in one launch run
    # at 74 %); the same rounds dealt to three streams overlap one launch's end with the next
    # one's start.  Effective time per round by the host clock -- single launches take longer.
    sets = [DeviceBatch(b, device, log_term_bits=args.log_term_bits, tiled=tiled) for b in host_batches]
    streams3 = [torch.cuda.Stream(device=device) for _ in range(3)]

    def dealt(count):
        for i in range(count):
            with torch.cuda.stream(streams3[i % 3]):
                eng.fused_round(sets[i % S])
    dealt(2 * S)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    dealt(240)
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) / 240 * 1e3
    out["three_streams"] = dict(ms_per_round=ms, frac_effective=frac(alg_bytes_per_group(n), ms), input_sets=S,
                                note="the headline rounds dealt to three streams (independent batches): host-clock "
                                     "time per round over 240 launches; launches overlap, so this is not a launch duration")
    # Several batches decided by ONE grid (rq_plan_rounds): k of the headline's sets per launch share
    # one ramp-up and one drain.  HIP-event time per launch / k.
    for k in (4,):
        plans = [eng.plan_rounds([sets[(j * k + i) % S] for i in range(k)]) for j in range(S // k)]
        P = len(plans)
        msk = timed_launches(torch, device, lambda i: eng.launch_plan(plans[i % P]), max(steps // k, 10), max(warmup // k, 2), P)
        out["rounds_in_one_launch"] = dict(
            batches_per_launch=k, launch_ms=msk, ms_per_round=msk / k, frac=frac(alg_bytes_per_group(n), msk / k), input_sets=S,
            note="%d independent 1M x %d batches (the headline's sets) decided by one grid, rq_plan_rounds: HIP-event "
                 "time of the launch / %d" % (k, n, k), verified=verify_against_oracle(sets[k - 1], host_batches[k - 1]))
        del plans
    del sets
    torch.cuda.empty_cache()
    # no arena on the device: the host keeps, per group, the index of the first entry of the
    # leader's term (rq_round_desc.log_term_start); logs of Raft's own shape (terms above the
    # leader's lowered to it)
    sets = [DeviceBatch(b, device, tiled=tiled, term_start=synth.term_start_monotone(b)) for b in host_batches]
    ms = timed_launches(torch, device, lambda i: eng.fused_round(sets[i % S]), steps, warmup, S)
    tsb = alg_bytes_per_group(n) - 4                  # term start + length instead of term, offset and looked-up term
    out["term_start"] = dict(kernel_ms=ms, frac=frac(tsb, ms), input_sets=S, alg_bytes_per_group=tsb,
                             note="log_term_start instead of the log-term lookup (logs non-decreasing, none above "
                                  "the leader's term); nothing of the logs resident")
    del sets
    torch.cuda.empty_cache()
    # the headline workload in slot-major planes (the layout of the host-buffer entry points)
    sets = [DeviceBatch(b, device, log_term_bits=args.log_term_bits, tiled=False) for b in host_batches]
    ms = timed_launches(torch, device, lambda i: eng.fused_round(sets[i % S]), steps, warmup, S)
    out["planes"] = dict(kernel_ms=ms, frac=frac(alg_bytes_per_group(n), ms), input_sets=S,
                         note="slot-major planes field[p * ld + g] instead of 64-group tiles")
    # the two decisions as separate launches (the headline's sets in planes)
    ms = timed_launches(torch, device, lambda i: eng.commit_advance(sets[i % S]), steps, warmup, S)
    out["commit_only"] = dict(kernel_ms=ms, frac=frac(alg_bytes_per_group(n, "commit"), ms),
                              alg_bytes_per_group=alg_bytes_per_group(n, "commit"))
    ms = timed_launches(torch, device, lambda i: eng.vote_tally(sets[i % S]), steps, warmup, S)
    out["vote_only"] = dict(kernel_ms=ms, frac=frac(alg_bytes_per_group(n, "vote"), ms),
                            alg_bytes_per_group=alg_bytes_per_group(n, "vote"))
    del sets
    torch.cuda.empty_cache()

    # AppendEntries consistency batch (raft.go:142-153) + leader bookkeeping: n - 1 follower planes,
    # follower logs in the narrowest entries their terms fit
    from raft_amd.device import encode_log_terms
    asets = []
    for s in range(6):
        ab = synth.gen_append(G, n - 1, seed=900 + s)
        enc, bits, _ = encode_log_terms(ab.flog_term, bits=0)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        t = {k: up(ab[k]) for k in ("follower_term", "leader_term", "prev_log_index", "prev_log_term", "n_new",
                                    "match_index", "next_index")}
        t["flog_off"] = up(ab.flog_off.view(np.int32))
        t["flog_term"] = up(enc)
        t["follower_term_out"] = torch.zeros_like(t["follower_term"])
        t["success_mask"] = torch.zeros(G, dtype=torch.int16, device=device)
        t["stepdown_mask"] = torch.zeros(G, dtype=torch.int16, device=device)
        t["status"] = torch.zeros(G, dtype=torch.uint8, device=device)
        t["_bits"], t["_terms"] = bits, int(enc.size)
        asets.append(t)
    A = len(asets)
    ms = timed_launches(torch, device, lambda i: eng.append_check(
        asets[i % A], G, n - 1, G, flog_term_bits=asets[i % A]["_bits"], flog_terms=asets[i % A]["_terms"]),
        steps, warmup, A)
    abytes = 44 * (n - 1) + 4 + 2 + 2 + 2
    out["append_check"] = dict(kernel_ms=ms, frac=frac(abytes, ms), input_sets=A, alg_bytes_per_group=abytes,
                               note="AppendEntries consistency batch, %d follower planes, %d-bit follower-log entries"
                                    % (n - 1, asets[0]["_bits"]))
    # the same launch also applying the follower's commit rule (raft.go:172-176) to the slots that succeed
    for s, t in enumerate(asets):
        cb = synth.with_follower_commit(synth.gen_append(G, n - 1, seed=900 + s), seed=s)
        t.update({k: up(cb[k]) for k in ("leader_commit", "last_new_index", "follower_commit")})
    ms = timed_launches(torch, device, lambda i: eng.append_check(
        asets[i % A], G, n - 1, G, flog_term_bits=asets[i % A]["_bits"], flog_terms=asets[i % A]["_terms"]),
        steps, warmup, A)
    cbytes = abytes + 12 * (n - 1) + 4
    out["append_check_commit"] = dict(kernel_ms=ms, frac=frac(cbytes, ms), input_sets=A, alg_bytes_per_group=cbytes,
                                      note="AppendEntries batch + follower commit rule in one launch")
    # replicate -> commit: the batch followed by the commit round over the matchIndex array it
    # has just updated (n - 1 followers + the leader's own slot), as two launches and with the
    # leader's commit rule decided inside the batch's launch (rq_append_desc.commit_round)
    import ctypes as C
    from raft_amd import _lib
    for s, t in enumerate(asets):
        for k in ("leader_commit", "last_new_index", "follower_commit"):
            del t[k]
        c = host_batches[s]
        arena, cbits, _ = encode_log_terms(c.log_term, bits=0)
        t["match_index"] = torch.cat([t["match_index"], up(c.match[n - 1:n, :G])])
        t["leader_term"] = up(c.leader_term[:G])
        t["commit"], t["commit_out"] = up(c.commit[:G]), torch.zeros(G, dtype=torch.int32, device=device)
        t["log_off"], t["log_term"] = up(c.log_off.view(np.int32)), up(arena)
        t["round_status"] = torch.zeros(G, dtype=torch.uint8, device=device)
        d = _lib.RoundDesc()
        d.struct_size = C.sizeof(_lib.RoundDesc)
        d.layout, d.what, d.G, d.n, d.ld = 4, 1, G, n, G
        d.match, d.commit, d.leader_term = t["match_index"].data_ptr(), t["commit"].data_ptr(), t["leader_term"].data_ptr()
        d.log_off, d.log_term = t["log_off"].data_ptr(), t["log_term"].data_ptr()
        d.log_terms, d.log_term_bits = int(arena.size), cbits
        d.commit_out, d.status_out = t["commit_out"].data_ptr(), t["round_status"].data_ptr()
        t["_round"] = d

    def two_launches(i):
        t = asets[i % A]
        eng.append_check(t, G, n - 1, G, flog_term_bits=t["_bits"], flog_terms=t["_terms"])
        eng._check(eng._l.rq_round_dev(eng._h, torch.cuda.current_stream().cuda_stream, C.byref(t["_round"])), "rq_round_dev")

    ms2 = timed_launches(torch, device, two_launches, steps, warmup, A)
    ms1 = timed_launches(torch, device, lambda i: eng.append_check(
        asets[i % A], G, n - 1, G, flog_term_bits=asets[i % A]["_bits"], flog_terms=asets[i % A]["_terms"],
        commit_round=asets[i % A]["_round"]), steps, warmup, A)
    rbytes = abytes + alg_bytes_per_group(n, "commit")
    out["replicate_commit"] = dict(kernel_ms=ms1, frac=frac(rbytes, ms1), two_launches_ms=ms2, frac_two_launches=frac(rbytes, ms2),
                                   input_sets=A, alg_bytes_per_group=rbytes,
                                   note="AppendEntries batch + leader commit rule: one launch (commit_round) against two; "
                                        "bytes priced as for the two launches")
    del asets
    torch.cuda.empty_cache()

    # wide layout on the first sets (8 output sets x 66 MB exceed the Infinity Cache)
    whost = [synth.widen(b, index_shift=0, term_shift=1 << 41, id_shift=1 << 42, seed=1000 + s)[0]
             for s, b in enumerate(host_batches[:8])]
    wide = [DeviceBatch(w, device, tiled=tiled) for w in whost]
    S = len(wide)
    ms = timed_launches(torch, device, lambda i: eng.fused_round(wide[i % S]), steps, warmup, S)
    out["i64"] = dict(kernel_ms=ms, frac=frac(alg_bytes_per_group(n, "fused", "i64"), ms), input_sets=S,
                      alg_bytes_per_group=alg_bytes_per_group(n, "fused", "i64"),
                      note="wide layout (RQ_LAYOUT_I64), terms and ids shifted beyond 2^41, int64 arena entries",
                      verified=verify_against_oracle(wide[0], whost[0]))
    del wide
    torch.cuda.empty_cache()
    # the same with the arena as a host would keep it: entries as narrow as the differences from each
    # group's smallest resident term allow (8 bits here), the base term in a plane of its own
    wide = [DeviceBatch(w, device, tiled=tiled, log_term_bits=0, log_term_base=True) for w in whost]
    ms = timed_launches(torch, device, lambda i: eng.fused_round(wide[i % S]), steps, warmup, S)
    out["i64_narrow_arena"] = dict(kernel_ms=ms, frac=frac(alg_bytes_per_group(n, "fused", "i64"), ms), input_sets=S,
                                   alg_bytes_per_group=alg_bytes_per_group(n, "fused", "i64"),
                                   log_term_bits=wide[0].log_term_bits,
                                   note="int64 planes, log-term arena in %d-bit differences from a per-group base term"
                                        % wide[0].log_term_bits)
    del wide, whost
    torch.cuda.empty_cache()

    # mixed 3-9 peers + 10 % joint consensus, Zipf log lengths; one log arena shared by the sets.  The
    # host orders the groups by cluster size (groups in joint consensus last) and says so in the
    # launch descriptor (segment table): the fixed-size segments are decided by the fixed-size code
    # inside the same launch, only the rest through member masks.
    first = synth.gen_mixed(G, seed=500, bucketed="size")
    mhost = [first] + [synth.gen_mixed(G, seed=500 + s, bucketed="size", logs_from=first) for s in range(1, 8)]
    S = len(mhost)
    mem = (first.old_mask | first.new_mask).astype(np.int64)
    members = float(sum((mem >> p) & 1 for p in range(16)).mean())
    bpg = 20 * members + 41
    mnote = "mixed 3-9 peers + 10% joint consensus, Zipf(1.2) logs up to 4096 entries"

    def mixed_sets(**kw):
        sets = []
        for i, mb in enumerate(mhost):
            if "term_start" not in kw and i:
                sets.append(DeviceBatch(mb, device, tiled=tiled, share_log_with=sets[0], **kw))
            else:
                sets.append(DeviceBatch(mb, device, tiled=tiled, **kw))
        return sets

    msets = mixed_sets()
    for db, mb in zip(msets, mhost):
        db.segments = synth.segment_table(mb.old_mask, mb.new_mask)
    ms = timed_launches(torch, device, lambda i: eng.fused_round(msets[i % S]), steps, warmup, S)
    out["mixed"] = dict(kernel_ms=ms, frac=frac(bpg, ms), input_sets=S, alg_bytes_per_group=bpg,
                        mean_member_slots=members, segments=len(msets[0].segments[1]),
                        note=mnote + "; ONE image ordered by cluster size, launched with the segment table",
                        verified=verify_against_oracle(msets[0], first))
    # the same images decided through the member masks alone (no table): what every group cost before
    for db in msets:
        db.segments = None
    ms = timed_launches(torch, device, lambda i: eng.fused_round(msets[i % S]), steps, warmup, S)
    out["mixed_masks_only"] = dict(kernel_ms=ms, frac=frac(bpg, ms), input_sets=S, alg_bytes_per_group=bpg,
                                   note=mnote + "; every group decided through its member masks",
                                   verified=verify_against_oracle(msets[0], first))
    # one image per cluster size, each in its own compact tiles, + one with member masks: ONE launch
    arena = msets[0]
    parts0 = synth.split_by_size(first)
    isets = [[DeviceBatch(sub, device, tiled=tiled, share_log_with=arena, own_log_off=True)
              for _, _, _, sub in synth.split_by_size(mb)] for mb in mhost]
    plans = [eng.plan_rounds(imgs) for imgs in isets]
    ms = timed_launches(torch, device, lambda i: eng.launch_plan(plans[i % S]), steps, warmup, S)
    oks = [verify_against_oracle(db, sub) for db, (_, _, _, sub) in zip(isets[0], parts0)]
    out["mixed_images"] = dict(kernel_ms=ms, frac=frac(bpg, ms), input_sets=S, alg_bytes_per_group=bpg, images=len(parts0),
                               note=mnote + "; one image per cluster size + one with member masks, rq_plan_rounds (one launch)",
                               verified=dict(oks[0], ok=all(v["ok"] for v in oks), groups=G,
                                             mismatched=sorted({m for v in oks for m in v["mismatched"]})))
    del msets, isets, plans, arena
    torch.cuda.empty_cache()
    # the same workload without the log arena (log_term_start): a looked-up term costs a
    # 128-byte line of its own once logs are hundreds of entries long
    msets = [DeviceBatch(mb, device, tiled=tiled, term_start=synth.term_start_monotone(mb),
                         segments=synth.segment_table(mb.old_mask, mb.new_mask)) for mb in mhost]
    ms = timed_launches(torch, device, lambda i: eng.fused_round(msets[i % S]), steps, warmup, S)
    out["mixed_term_start"] = dict(kernel_ms=ms, frac=frac(bpg - 4, ms), input_sets=S, alg_bytes_per_group=bpg - 4,
                                   note="mixed / joint workload (segment table) with log_term_start instead of the lookup")
    del msets
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                       # does not return
    # Libraries (RCCL's version banner, for one) write to stdout; the contract is ONE
    # JSON line there, so everything else is sent to stderr until the line is printed.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("RQ_BENCH_TRACE_AFTER"):       # debugging aid: dump Python stacks if stuck
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RQ_BENCH_TRACE_AFTER"]), exit=True)
    import torch
    import torch.distributed as dist
    from raft_amd import synth
    from raft_amd.device import DeviceBatch, DeviceEngine
    from raft_amd.sharded import SummaryConsumer, SummaryExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("--gpus %d over RCCL needs %d GPUs, this node shows %d "
                         "(--backend gloo rehearses several ranks on one GPU)" % (world, world, ndev))
    dev_index = local_rank % ndev                # one rank per GPU on a real node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_exchange
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # the communicator is built by the first collective (seconds on a full node): here, where
        # no watchdog is armed and nothing is timed
        warm = torch.zeros(1, device=device)
        dist.all_reduce(warm)
        torch.cuda.synchronize(device)

    if args.summary == "auto":
        args.summary = "u8" if args.workload == "fused" else "u16"
    eng = DeviceEngine(dev_index)
    exchange = use_dist and not args.no_exchange
    state = {}                                   # what the watchdog reports

    def degraded_line():
        """The watchdog's last words on rank 0: if the steps without the exchange had already been
        timed when the run got stuck, their figure goes out as a line marked invalid (this rank's
        clock only, nothing sent between ranks), so that a wedged exchange still leaves the
        kernels' figure behind.  The exit code stays 3."""
        if rank != 0:
            return
        done = state.get("main_line")
        if done is not None:
            # the figure of the line was complete; an additional leg got stuck: the line goes out as it
            # was, with the leg's place saying so
            done.setdefault("extra", {})["strong_8m"] = {"error": "the run stopped making progress in phase %r" % (state.get("phase"),)}
            os.write(json_fd, (json.dumps(done) + "\n").encode())
            return
        c = state.get("compute_only_local")
        if c is None:
            return
        line = {"metric": METRIC, "value": c["G"] * world * c["steps"] / c["wall"], "unit": UNIT, "n_gpus": world,
                "steps": c["steps"], "warmup": args.warmup, "ms_per_step": c["wall"] / c["steps"] * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "valid": False,
                "degraded": "the run stopped making progress in phase %r; value and ms_per_step are rank 0's own clock "
                            "over the steps WITHOUT the summary exchange, timed before it" % (state.get("phase"),),
                "config": {"workload": c["workload"], "groups_per_gpu": c["G"], "total_groups": c["G"] * world},
                "roofline": {"bound": "hbm", "kernel_ms": c["kernel_ms"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "achieved": c["bytes_per_group"] * c["G"] / (c["kernel_ms"] * 1e-3) / 1e9,
                             "frac": c["bytes_per_group"] * c["G"] / (c["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": None}}
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    # rank 0 gives up a little earlier than the others: a run that is stuck is stuck on every rank
    # within one exchange period, and rank 0 is the one with last words to say before the launcher,
    # seeing the first rank fail, ends the rest
    dog = Watchdog(args.watchdog_seconds if rank == 0 or args.watchdog_seconds <= 0 else args.watchdog_seconds * 1.25 + 2,
                   lambda: dict(
        step=state.get("step"), phase=state.get("phase"), exchange=state["ex"].debug_state() if state.get("ex") else None),
        last_words=degraded_line)
    # test hook, "rank:step[:leg]": that rank never gets beyond that timed step (with the exchange) of that leg
    hang = os.environ.get("RQ_BENCH_INJECT_HANG", "-1:-1").split(":")
    hang_rank, hang_step, hang_leg = int(hang[0]), int(hang[1]), hang[2] if len(hang) > 2 else "main"

    def measure(G, S, steps, warmup, preheat_rotations, label, rounds_per_exchange=None):
        """Build S rotating input/output sets of G groups on this rank, run the untimed rotations,
        then -- with more than one rank or a forced exchange -- time `steps` steps without the
        exchange, and then the same steps with it (the figure of the line), so that one invocation
        separates the kernels' scaling from the cost of the exchange."""
        n = args.peers
        R = max(1, min(rounds_per_exchange or args.rounds_per_exchange, S))
        while S % R:
            R -= 1                                   # every input set maps to one (slot, round-in-slot)
        depth = S // R
        ex = SummaryExchange(G, device, depth=depth, mode=args.summary, impl=args.exchange_impl,
                             rounds=R, fence=args.fence) if exchange else None
        state["ex"] = ex
        consumer = SummaryConsumer(ex) if (ex is not None and args.consume) else None
        # one run-long counter of groups whose commit offset left the 16-bit summary window
        escape_counter = torch.zeros(2, dtype=torch.int32, device=device) if ex is not None else None
        sets, first_mixed, host_batches, verify_host = [], None, [], None
        for s in range(S):
            seed = synth.DEFAULT_SEED + 7919 * rank + 104729 * s
            if args.workload == "fused":
                b = synth.gen_fused(G, n, seed=seed, ld=G + args.ld_pad)
                if args.log_window >= 0:
                    b = synth.window_logs(b, keep_committed=args.log_window)
            else:                                # the 0.9 GB log arena is generated once and shared
                b = synth.gen_mixed(G, seed=seed, logs_from=first_mixed, bucketed=True if args.no_segments else "size")
                first_mixed = first_mixed or b
            if args.layout == "i64":
                if args.workload == "mixed" or (ex is not None and args.summary == "i32"):
                    raise SystemExit("--layout i64 runs the fused workload with a narrow (u4, u8, u16) summary")
                # indices can only leave the int32 range together with a window base
                b, _ = synth.widen(b, index_shift=(1 << 40) if args.log_window >= 0 else 0,
                                   term_shift=1 << 41, id_shift=1 << 42, seed=seed)
            elif args.workload == "fused":
                host_batches.append(b)
            if s == 0:
                verify_host = b                  # what set 0's outputs are checked against after the timed region
            kw = {}
            if args.workload == "mixed" and sets:
                kw["share_log_with"] = sets[0]
            if args.workload == "mixed" and not args.no_segments:
                kw["segments"] = synth.segment_table(b.old_mask, b.new_mask)
            if args.layout == "i32" and args.workload == "fused":
                kw["log_term_bits"] = args.log_term_bits
            tiled = not args.planes and not (ex is not None and args.summary == "i32")   # full vectors land in plain arrays
            if tiled:
                kw["tiled"] = True
            if ex is not None and args.summary == "i32":
                kw.update(commit_out=ex.commit_view(s // R, s % R), elected_bits=ex.bits_view(s // R, s % R))
            elif ex is not None:
                kw.update(commit_lo16=ex.commit_view(s // R, s % R), elected_bits=ex.bits_view(s // R, s % R),
                          summary_escapes=escape_counter, summary_bits=ex.bits,
                          summary_base="commit" if ex.bits < 16 else None)
            if ex is not None and s % R == 0:
                # the first round of a slot releases the exchange of the slot before it
                kw["start_signal"] = ex.start_signal((s // R - 1) % depth)   # None with event fences
            sets.append(DeviceBatch(b, device, **kw))
        if args.workload == "fused":
            bytes_per_group = alg_bytes_per_group(n, args.what, args.layout)
            workload = "%d groups x %d peers per GPU, fused RequestVote tally + commitIndex advance with log-term gather" % (G, n)
        else:
            om = b.old_mask.astype("int64"); nm = b.new_mask.astype("int64")
            members = sum(((om | nm) >> p) & 1 for p in range(16)).mean()
            bytes_per_group = 20 * members + 37 + 4          # member slots only + two u16 masks
            workload = "%d groups, mixed 3-9 peers + 10%% joint consensus, Zipf log lengths" % G

        debug = bool(os.environ.get("RQ_BENCH_DEBUG"))
        prepared = {}
        stream_handle = torch.cuda.current_stream(device).cuda_stream      # the launches' stream for the whole measurement

        def step(i, with_ex=True):
            s = i % S
            x = ex if with_ex else None
            state["step"] = i
            dog.beat("step")
            if debug:
                sys.stderr.write("rank %d step %d\n" % (rank, i)); sys.stderr.flush()
            if (rank == hang_rank and i == hang_step and label == hang_leg and state.get("phase") == label + ":timed"
                    and (with_ex or ex is None)):
                time.sleep(1e6)                  # test hook: this rank never gets to its next collective
            if x is not None and s % R == 0:
                x.begin(s // R)                  # the slot's previous all-gather must be done
            plan = prepared.get(s)
            if plan is None:             # validated once per set, like a C or Go host that keeps its rq_plan handles
                if args.what == "fused":
                    plan = eng.prepare(sets[s], 3, args.variant, want_bits=not args.no_bits, status=not args.no_status)
                else:
                    plan = eng.prepare(sets[s], 1 if args.what == "commit" else 2, args.variant)
                prepared[s] = plan
            eng.launch_prepared(plan, stream_handle)
            if x is not None and s % R == 0:
                x.round_started((s // R - 1) % depth)
            elif ex is not None and s % R == 0:
                ex.launch_counted((s // R - 1) % depth)   # the launch bumped the slot's signal word all the same
            if x is not None and s % R == R - 1:
                x.filled(s // R)                     # R rounds of summaries in one collective
                if consumer is not None:
                    # device-side decode of the PREVIOUS exchange: its all-gather has had a whole
                    # exchange period to finish, so the compute stream does not stall on it
                    consumer.consume((s // R - 1) % depth)

        def fence(with_ex=True):
            dog.beat("fence: exchange")
            if ex is not None and with_ex:
                ex.wait_all()
            dog.beat("fence: device")
            torch.cuda.synchronize(device)
            dog.beat("fence: barrier")
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(device)
            dog.beat("fence: done")

        def timed(with_ex):
            """`warmup` untimed and `steps` timed steps between fences: (host seconds, kernel ms per step,
            host seconds spent enqueuing)."""
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(); ev1.record()                # creates the events outside the timed region
            state["phase"] = label + ":warmup"
            for i in range(warmup):
                step(i, with_ex)
            fence(with_ex)
            state["phase"] = label + ":timed"
            gc_was_on = gc.isenabled()
            gc.disable()              # a full collection (tens of ms with torch imported) would let the queue run dry
            try:
                t0 = time.perf_counter()
                ev0.record()
                for i in range(steps):
                    step(i, with_ex)
                ev1.record()                              # same stream as the kernels
                t_enqueued = time.perf_counter()
                if ex is not None and with_ex:
                    ex.wait_all()
                eng.stream_sync()                         # polls the kernels' stream: a blocked thread wakes 10-20 us late
                t1 = time.perf_counter()
            finally:
                if gc_was_on:
                    gc.enable()
            fence(with_ex)
            return t1 - t0, ev0.elapsed_time(ev1) / steps, t_enqueued - t0

        # untimed: build every set's prepared launch and touch every buffer once
        dog.arm(label + ":preheat")
        state["phase"] = label + ":preheat"
        preheat = S * max(1, preheat_rotations)
        res = dict(G=G, S=S, R=R, ex=ex, consumer=consumer, sets=sets, host_batches=host_batches, verify_host=verify_host,
                   bytes_per_group=bytes_per_group, workload=workload, preheat=preheat, escape_counter=escape_counter,
                   compute_only=None)
        if ex is not None:
            # first the steps without the exchange (the launches still write their summaries into
            # the exchange's buffers, nothing is sent): should the exchange get stuck afterwards,
            # the kernels' own figure exists already and the watchdog can leave it behind
            for i in range(preheat):
                step(i, False)
            fence(False)
            cw, ck, _ = timed(False)
            res["compute_only"] = dict(wall=cw, kernel_ms=ck)
            if label == "main":
                state["compute_only_local"] = dict(G=G, steps=steps, wall=cw, kernel_ms=ck, workload=workload,
                                                   bytes_per_group=bytes_per_group)
            state["phase"] = label + ":preheat with the exchange"
        for i in range(preheat if ex is None else S * min(4, max(1, preheat_rotations))):
            step(i)
        fence()
        wall, kernel_ms, enq = timed(True)
        res.update(wall=wall, kernel_ms=kernel_ms, enqueue_s=enq)
        dog.disarm()
        state["phase"] = label + ":done"
        return res

    def over_ranks(x, op):
        if not use_dist:
            return float(x)
        tt = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=op)
        return float(tt.item())

    strong = args.total_groups > 0
    if strong:
        from raft_amd.sharded import shard_bounds
        if args.total_groups % world:
            raise SystemExit("--total-groups must be a multiple of the number of ranks")
        g0, g1 = shard_bounds(args.total_groups, world, rank)
        args.groups = g1 - g0
    G, n, S = args.groups, args.peers, max(1, args.sets)
    m = measure(G, S, args.steps, args.warmup, args.preheat_rotations, "main")
    ex, consumer, sets, host_batches, verify_host = m["ex"], m["consumer"], m["sets"], m["host_batches"], m["verify_host"]
    R, preheat, bytes_per_group, workload = m["R"], m["preheat"], m["bytes_per_group"], m["workload"]
    escape_counter = m["escape_counter"]
    log_term_bits = sets[0].log_term_bits
    resident_mb = sum(x.bytes_resident() for x in sets) / 1e6
    kernel_ms, t_enqueue_s = m["kernel_ms"], m["enqueue_s"]
    wall = over_ranks(m["wall"], dist.ReduceOp.MAX) if use_dist else m["wall"]
    kernel_ms_ranks = None
    compute_only = None
    if use_dist:
        kernel_ms_ranks = dict(max=over_ranks(kernel_ms, dist.ReduceOp.MAX), min=over_ranks(kernel_ms, dist.ReduceOp.MIN))
        if m["compute_only"] is not None:
            cw = over_ranks(m["compute_only"]["wall"], dist.ReduceOp.MAX)
            ck = m["compute_only"]["kernel_ms"]
            compute_only = dict(ms_per_step=cw / args.steps * 1e3, value=G * world * args.steps / cw,
                                kernel_ms=dict(max=over_ranks(ck, dist.ReduceOp.MAX), min=over_ranks(ck, dist.ReduceOp.MIN)),
                                note="the same steps without the summary exchange (max over ranks of the host clock)")

    devices = [dev_index]
    if use_dist:
        ids = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([dev_index], dtype=torch.int64, device=device))
        devices = [int(t.item()) for t in ids]

    # outside the timed region: set 0's outputs, as the timed launches left them, against the oracle
    what_bits = {"fused": 3, "commit": 1, "vote": 2}[args.what]
    verified = verify_against_oracle(sets[0], verify_host, what=what_bits,
                                     want_bits=not args.no_bits and args.what != "commit",
                                     status=not args.no_status or args.what != "fused")
    if use_dist:            # every rank checks its own shard; the line carries the AND
        okt = torch.tensor([1 if verified["ok"] else 0], dtype=torch.int32, device=device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        verified["ok"] = bool(okt.item())
        verified["groups"] = verified["groups"] * world
    failed = not verified["ok"]

    summary_escapes = None
    if ex is not None:      # the 16-bit window must have held on this rank for the whole run
        summary_escapes = int(escape_counter[0].item())
    total_groups = G * world
    value = total_groups * args.steps / wall
    achieved = bytes_per_group * G / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    try:
        tj = json.load(open(args.traffic_json))
        key = "%s%s%s_G%d_n%d" % (args.workload, "_masks" if (args.workload == "mixed" and args.no_segments) else "",
                                  "_i64" if args.layout == "i64" else "", G, n)
        ent = tj.get(key)
        if ent is not None:
            if ent.get("kernel_source_hash") == kernel_source_hash():
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_source = "rocprofv3 --pmc profile %s of kernel %s (not measured by this run)" % (
                    ent.get("source"), ent.get("kernel"))
            else:
                traffic_source = "none: the kernel source changed since profile %s was taken" % ent.get("source")
    except (OSError, ValueError):
        pass
    line = {
        "metric": METRIC,
        "value": value, "unit": UNIT, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "int64" if args.layout == "i64" else "int32", "data": "synthetic",
        "config": {"workload": workload, "groups_per_gpu": G, "total_groups": total_groups,
                   "scaling_note": ("strong scaling: %d groups block-partitioned over %d ranks" % (total_groups, world)) if strong
                   else ("weak scaling: %d groups on every GPU; BASELINE's sharded configuration is 8388608 groups in "
                         "total (strong), the same workload as this one only at 8 GPUs" % G),
                   "peers": n, "input_sets": S,
                   "preheat_rounds": preheat,
                   "resident_mb_per_gpu": round(resident_mb, 1),
                   "rccl_ranks": dist.get_world_size() if use_dist else 1,
                   "backend": args.backend if use_dist else None, "devices": devices,
                   "summary_allgather": bool(exchange),
                   "summary_bytes_per_rank": ex.nbytes if ex is not None else 0,
                   "summary_mode": args.summary if ex is not None else None,
                   "summary_impl": ex.impl if ex is not None else None,
                   "summary_consumed_on_device": consumer is not None,
                   "rounds_per_allgather": R if ex is not None else None,
                   "exchange_fence": ex.fence if ex is not None else None,
                   "summary_escapes": summary_escapes, "variant": args.variant, "what": args.what,
                   "ld_pad": args.ld_pad, "log_window": args.log_window, "layout": args.layout,
                   "log_term_bits": log_term_bits, "tiled": bool(getattr(sets[0], "tile_bytes", 0)),
                   "alg_bytes_per_group": bytes_per_group},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms": kernel_ms,
                     "host_enqueue_ms_per_step": t_enqueue_s / args.steps * 1e3,
                     "kernel": "rq_round_fixed<%d,...>" % n if args.workload == "fused" else ("rq_round_joint<16,...>" if args.no_segments else "rq_round_multi<3,...>")},
        "pct_hbm_peak": 100.0 * achieved / HBM_PEAK_GBS,
        "verified": verified,
    }
    if use_dist:
        line["roofline"]["kernel_ms_ranks"] = kernel_ms_ranks
        line.setdefault("extra", {})
        if compute_only is not None:
            line["extra"]["compute_only"] = compute_only
    # 2 or 4 ranks: the sharded configuration itself (8M groups in total, block-partitioned) beside
    # the weak-scaling headline, every rank taking part
    if world in (2, 4) and not strong and args.workload == "fused" and not args.no_extras:
        state["main_line"] = line                 # complete as far as the headline goes: the watchdog's last words from here on
        del sets, host_batches, verify_host
        m["sets"] = m["host_batches"] = m["verify_host"] = None
        if consumer is not None:
            consumer.close()
        if ex is not None:
            ex.close()
        ex = consumer = m["ex"] = m["consumer"] = None
        torch.cuda.empty_cache()
        total8 = args.strong_total
        # 4 sets as 2 exchange slots of 2 rounds, so that a slot's all-gather overlaps the other slot's rounds
        m8 = measure(total8 // world, 4, args.steps, args.warmup, 2, "strong_8m", rounds_per_exchange=2)
        w8 = over_ranks(m8["wall"], dist.ReduceOp.MAX)
        v8 = verify_against_oracle(m8["sets"][0], m8["verify_host"], what=3)
        ok8 = over_ranks(1.0 if v8["ok"] else 0.0, dist.ReduceOp.MIN) == 1.0
        line["extra"]["strong_8m"] = dict(
            total_groups=total8, groups_per_gpu=total8 // world, ms_per_step=w8 / args.steps * 1e3,
            value=total8 * args.steps / w8, kernel_ms=dict(max=over_ranks(m8["kernel_ms"], dist.ReduceOp.MAX),
                                                           min=over_ranks(m8["kernel_ms"], dist.ReduceOp.MIN)),
            input_sets=4, verified=dict(v8, ok=ok8),
            compute_only_ms_per_step=None if m8["compute_only"] is None else
            over_ranks(m8["compute_only"]["wall"], dist.ReduceOp.MAX) / args.steps * 1e3,
            note="BASELINE's sharded configuration: %d groups x %d peers block-partitioned over the ranks (strong scaling)" % (total8, n))
        failed = failed or not ok8
        state["main_line"] = None
        sets, ex, consumer = m8["sets"], m8["ex"], m8["consumer"]
    if rank == 0:
        plain = world == 1 and args.workload == "fused" and args.layout == "i32" and args.what == "fused"
        if plain and not args.no_extras:
            del sets
            torch.cuda.empty_cache()
            # a failure in a secondary figure must not cost the line: what was measured so far stays
            line.setdefault("extra", {})
            try:
                extras(args, torch, synth, DeviceBatch, eng, device, host_batches, line["extra"])
            except Exception as e:
                line["extra"]["error"] = "%s: %s" % (type(e).__name__, e)
            failed = failed or any(isinstance(v, dict) and v.get("verified", {}).get("ok") is False
                                   for v in line["extra"].values())
        if world == 1 and not args.no_cpu_baseline and args.workload == "fused":
            try:
                line["cpu_baseline"] = cpu_baseline(args, synth)
            except Exception as e:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(line), flush=True)

    eng.close()
    if ex is not None:
        ex.close()
    if use_dist:
        dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: outputs of the timed launches differ from the oracle: %s\n" % verified)
        sys.exit(1)


if __name__ == "__main__":
    main()
