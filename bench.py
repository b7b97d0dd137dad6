#!/usr/bin/env python3
"""Benchmark of the batched Raft quorum engine on MI355X.

A step = one fused round (RequestVote tally + commitIndex advance with the
log-term gather) over this rank's 1M groups x 5 peers, inputs resident in HBM.
Input/output sets rotate (default 12 sets, ~2.1 GB in total: both the reads
and the 36 MB of writes per step exceed the 256 MB Infinity Cache, which
otherwise absorbs the writes of up to 5 sets and flatters the number).  With N > 1 ranks the
groups are sharded (weak scaling: 1M groups per GPU; --total-groups T instead
block-partitions T groups over the ranks, strong scaling, e.g. 8388608) and the
commit/elected summary of every step is all-gathered over RCCL (one collective
per 4 steps by default), overlapped with the following steps' kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak


def alg_bytes_per_group(n, what="fused", layout="i32"):
    """Algorithmic bytes per group-round.  int32 layout: commit 4n+20, vote 16n+17;
    int64 layout (every value 8 bytes, log offsets still 4): commit 8n+36, vote 32n+29."""
    e = 8 if layout == "i64" else 4
    commit = e * n + 3 * e + 4 + e   # n match + commit + leaderTerm + gathered term + logOff ; commit'
    vote = 4 * e * n + 3 * e + 5     # 3n planes + candTerm + candId ; n voterTerm' + maxTerm + 2 + 2 + 1
    return {"commit": commit, "vote": vote, "fused": commit + vote}[what]


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
    ap.add_argument("--force-exchange", action="store_true",
                    help="initialise torch.distributed and run the summary all-gather even with one rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--log-window", type=int, default=-1,
                    help=">= 0: keep only the uncommitted suffix of every log plus this many committed "
                         "entries on the device (log_base window); -1: full logs (the default workload)")
    ap.add_argument("--layout", choices=["i32", "i64"], default="i32",
                    help="i64: the wide layout (rq_round64_dev), values shifted beyond the int32 range")
    ap.add_argument("--no-bits", action="store_true", help="skip the packed elected-bit summary output")
    ap.add_argument("--no-status", action="store_true", help="skip the per-group status byte output")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU work per baseline leg")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    return ap.parse_args()


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


def main():
    args = parse()
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
    from raft_amd.sharded import SummaryExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()    # one rank per GPU on a real node
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

    strong = args.total_groups > 0
    if strong:
        from raft_amd.sharded import shard_bounds
        if args.total_groups % world:
            raise SystemExit("--total-groups must be a multiple of the number of ranks")
        g0, g1 = shard_bounds(args.total_groups, world, rank)
        args.groups = g1 - g0
    if args.summary == "auto":
        args.summary = "u8" if args.workload == "fused" else "u16"
    G, n, S = args.groups, args.peers, max(1, args.sets)
    eng = DeviceEngine(dev_index)
    exchange = use_dist and not args.no_exchange
    R = max(1, min(args.rounds_per_exchange, S))
    while S % R:
        R -= 1                                   # every input set maps to one (slot, round-in-slot)
    depth = S // R
    ex = SummaryExchange(G, device, depth=depth, mode=args.summary, impl=args.exchange_impl,
                         rounds=R) if exchange else None

    # one run-long counter of groups whose commit offset left the 16-bit summary window
    escape_counter = torch.zeros(2, dtype=torch.int32, device=device) if ex is not None else None
    sets, first_mixed = [], None
    for s in range(S):
        seed = synth.DEFAULT_SEED + 7919 * rank + 104729 * s
        if args.workload == "fused":
            b = synth.gen_fused(G, n, seed=seed, ld=G + args.ld_pad)
            if args.log_window >= 0:
                b = synth.window_logs(b, keep_committed=args.log_window)
        else:                                # the 0.9 GB log arena is generated once and shared
            b = synth.gen_mixed(G, seed=seed, logs_from=first_mixed)
            first_mixed = first_mixed or b
        if args.layout == "i64":
            if args.workload == "mixed" or (ex is not None and args.summary == "i32"):
                raise SystemExit("--layout i64 runs the fused workload with a narrow (u4, u8, u16) summary")
            # indices can only leave the int32 range together with a window base
            b, _ = synth.widen(b, index_shift=(1 << 40) if args.log_window >= 0 else 0,
                               term_shift=1 << 41, id_shift=1 << 42, seed=seed)
        kw = {}
        if args.workload == "mixed" and sets:
            kw["share_log_with"] = sets[0]
        if ex is not None and args.summary == "i32":
            kw.update(commit_out=ex.commit_view(s // R, s % R), elected_bits=ex.bits_view(s // R, s % R))
        elif ex is not None:
            kw.update(commit_lo16=ex.commit_view(s // R, s % R), elected_bits=ex.bits_view(s // R, s % R),
                      summary_escapes=escape_counter, summary_bits=ex.bits,
                      summary_base="commit" if ex.bits < 16 else None)
        sets.append(DeviceBatch(b, device, **kw))
    if args.workload == "fused":
        bytes_per_group = alg_bytes_per_group(n, args.what, args.layout)
        workload = "%d groups x %d peers per GPU, fused RequestVote tally + commitIndex advance with log-term gather" % (G, n)
    else:
        om = b.old_mask.astype("int64"); nm = b.new_mask.astype("int64")
        members = sum(((om | nm) >> p) & 1 for p in range(16)).mean()
        bytes_per_group = 20 * members + 37 + 4          # member slots only + two u16 masks
        workload = "%d groups, mixed 3-9 peers + 10%% joint consensus, Zipf log lengths" % G
    resident_mb = sum(x.bytes_resident() for x in sets) / 1e6

    def step(i):
        s = i % S
        if ex is not None and s % R == 0:
            ex.begin(s // R)                 # the slot's previous all-gather must be done
        if args.what == "fused":
            eng.fused_round(sets[s], variant=args.variant, want_bits=not args.no_bits, status=not args.no_status)
        elif args.what == "commit":
            eng.commit_advance(sets[s], variant=args.variant)
        else:
            eng.vote_tally(sets[s], variant=args.variant)
        if ex is not None and s % R == R - 1:
            ex.exchange(s // R, async_op=True)   # R rounds of summaries in one collective

    def fence():
        if ex is not None:
            ex.wait_all()
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    fence()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()                              # same stream as the kernels
    if ex is not None:
        ex.wait_all()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # avg launch duration on the kernel's stream

    if use_dist:
        tt = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    summary_escapes = None
    if ex is not None:      # the 16-bit window must have held on this rank for the whole run
        summary_escapes = int(escape_counter[0].item())
    if rank == 0:
        total_groups = G * world
        value = total_groups * args.steps / wall
        achieved = bytes_per_group * G / (kernel_ms * 1e-3) / 1e9
        traffic = None
        try:
            tj = json.load(open(args.traffic_json))
            key = "%s%s_G%d_n%d" % (args.workload, "_i64" if args.layout == "i64" else "", G, n)
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        line = {
            "metric": "quorum decisions/sec at 1M groups x 5 peers (1/8 GPU); % of HBM peak",
            "value": value, "unit": "decisions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "int64" if args.layout == "i64" else "int32", "data": "synthetic",
            "config": {"workload": workload, "groups_per_gpu": G, "peers": n, "input_sets": S,
                       "resident_mb_per_gpu": round(resident_mb, 1),
                       "summary_allgather": bool(exchange),
                       "summary_bytes_per_rank": ex.nbytes if ex is not None else 0,
                       "summary_mode": args.summary if ex is not None else None,
                       "summary_impl": ex.impl if ex is not None else None,
                       "rounds_per_allgather": R if ex is not None else None,
                       "summary_escapes": summary_escapes, "variant": args.variant, "what": args.what, "ld_pad": args.ld_pad, "log_window": args.log_window, "layout": args.layout,
                       "alg_bytes_per_group": bytes_per_group},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kernel_ms,
                         "kernel": "rq_round_fixed<%d,...>" % n if args.workload == "fused" else "rq_round_joint<16,...>"},
            "pct_hbm_peak": 100.0 * achieved / HBM_PEAK_GBS,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "fused":
            line["cpu_baseline"] = cpu_baseline(args, synth)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(line), flush=True)

    eng.close()
    if ex is not None:
        ex.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
