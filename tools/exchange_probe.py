#!/usr/bin/env python3
"""Where does the per-step cost of the summary exchange come from?  One rank, the
bench's 1M x 5 fused round on 12 rotating sets, with the exchange replaced by
progressively more of its parts:
  none   no exchange at all
  fence  the stream fences only (side stream waits for the round, the slot's next
         round waits for an event recorded on the side stream), no work on the side
  copy   fences + a device-to-device copy of the message on the side stream
  rccl   fences + the one-rank RCCL all-gather (torch.distributed)
  record only an event record on the main stream after every R-th round
  sideonly / mainwait / sidewait   one ingredient of the fences at a time
  value  the fences made of hipStreamWriteValue32 / hipStreamWaitValue32 on 8-byte
         signal memory instead of events
  graph-copy / graph-rccl   a whole cycle of 12 rounds and its exchanges captured
         into one graph and replayed
  *-q    the slot's next round skips its wait when the event is already complete
Prints microseconds per step for each."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from raft_amd import synth
from raft_amd.device import DeviceBatch, DeviceEngine


def main():
    G, n, S, R = 1 << 20, 5, 12, 4
    steps, warm = 480, 48
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = DeviceEngine(0)
    sets = [DeviceBatch(synth.gen_fused(G, n, seed=11 + s, ld=G + 256), dev) for s in range(S)]
    nbytes = R * (G + G // 8)
    local = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(S // R)]
    allb = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(S // R)]
    side = torch.cuda.Stream(device=dev)
    done = [torch.cuda.Event() for _ in range(S // R)]
    pending = [False] * (S // R)
    main_stream = torch.cuda.current_stream(dev)

    # prototype of value-based fences: the main stream writes a counter after the
    # round, the side stream waits for it (and the other way round for slot reuse)
    import ctypes as C
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
    hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def signal_word():
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), 8, 0x2)        # hipMallocSignalMemory: 8 bytes exactly
        assert rc == 0, rc
        zero = C.c_uint64(0)
        hip.hipMemcpy(p, C.byref(zero), 8, 1)
        return p.value
    produced = [signal_word() for _ in range(S // R)]             # written by main, waited by side
    consumed = [signal_word() for _ in range(S // R)]             # written by side, waited by main
    seq = [0] * (S // R)

    def run_value(check=False):
        def step(i):
            s = i % S
            slot = s // R
            if s % R == 0 and seq[slot]:
                hip.hipStreamWaitValue32(main_stream.cuda_stream, consumed[slot], seq[slot], 0, 0xFFFFFFFF)
            eng.fused_round(sets[s])
            if s % R == R - 1:
                seq[slot] += 1
                hip.hipStreamWriteValue32(main_stream.cuda_stream, produced[slot], seq[slot], 0)
                hip.hipStreamWaitValue32(side.cuda_stream, produced[slot], seq[slot], 0, 0xFFFFFFFF)
                with torch.cuda.stream(side):
                    dist.all_gather_into_tensor(allb[slot], local[slot])
                hip.hipStreamWriteValue32(side.cuda_stream, consumed[slot], seq[slot], 0)
        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    def run_graph(mode):
        """One whole cycle of S rounds (fences + copy or RCCL all-gather per R rounds)
        captured into a graph and replayed."""
        def cycle():
            for s in range(S):
                slot = s // R
                eng.fused_round(sets[s])
                if s % R == R - 1:
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        if mode == "graph-copy":
                            allb[slot].copy_(local[slot], non_blocking=True)
                        else:
                            dist.all_gather_into_tensor(allb[slot], local[slot])
            torch.cuda.current_stream(dev).wait_stream(side)      # rejoin before the capture ends
        cycle()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cycle()
        for _ in range(warm // S):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // S):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (steps // S * S) * 1e6

    def run(mode):
        if mode == "value":
            return run_value()
        if mode.startswith("graph"):
            return run_graph(mode)
        def step(i):
            s = i % S
            slot = s // R
            if mode != "none" and s % R == 0 and pending[slot]:
                if not (mode.endswith("-q") and done[slot].query()):   # -q: skip the wait once the event is complete
                    main_stream.wait_event(done[slot])
                pending[slot] = False
            eng.fused_round(sets[s])
            if mode == "sideonly":            # unrelated work on the side stream, no dependencies either way
                if s % R == R - 1:
                    with torch.cuda.stream(side):
                        allb[slot].copy_(local[slot], non_blocking=True)
            elif mode == "mainwait":          # the main stream waits for an (already complete) side event
                if s % R == R - 1:
                    done[slot].record(side)
                    main_stream.wait_event(done[slot])
            elif mode == "sidewait":          # the side stream waits for the round, nothing else
                if s % R == R - 1:
                    side.wait_stream(main_stream)
            elif mode == "record":
                if s % R == R - 1:
                    done[slot].record(main_stream)
            elif mode != "none" and s % R == R - 1:
                side.wait_stream(main_stream)
                with torch.cuda.stream(side):
                    if mode == "copy":
                        allb[slot].copy_(local[slot], non_blocking=True)
                    elif mode.startswith("rccl"):
                        dist.all_gather_into_tensor(allb[slot], local[slot])
                    done[slot].record(side)
                pending[slot] = True
        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    for mode in ("none", "sideonly", "mainwait", "sidewait", "record", "fence", "copy", "rccl", "value",
                 "graph-copy", "graph-rccl"):
        print("%-6s %.2f us/step" % (mode, run(mode)), flush=True)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
