"""Group-sharded deployment: one process per GPU, groups block-partitioned
over ranks, and ONE exchange step per round -- an all-gather of the per-round
summary over torch.distributed (backend "nccl" is RCCL over xGMI on MI355X;
"gloo" runs the same code on CPU tensors for tests).

Raft groups are independent (no rule in raft.go:181-223 or in the commit rule
reads another group's state), so the decision kernels need no collective; the
summary exchange exists so that every rank can route client traffic and
heartbeats for all groups.

Summary per group: the elected decision (1 bit) and the new commitIndex.  xGMI
is point-to-point (one ~77 GB/s link per peer and direction), and a round of
1M groups takes ~30 us, so the exchange is sized for the link: by default the
commitIndex travels as an offset from a per-group base in 16, 8 or 4 bits
(modes "u16", "u8", "u4": 2.1, 1.1 or 0.6 bytes per group with the bit); an
offset outside the window is sent as the all-ones escape value, which tells
every receiver to fetch the full 32-bit vector of that round.  With the
round's own input commitIndex as the base the offset is the round's advance,
which is what the narrow modes are meant for.  mode="i32" sends the 32-bit
vector every round.
"""
import torch
import torch.distributed as dist

ESCAPE = 65535
MODE_BITS = {"u16": 16, "u8": 8, "u4": 4, "i32": 32}


def shard_bounds(total_groups, world_size, rank):
    """Block partition: rank r owns groups [r*T/R, (r+1)*T/R)."""
    g0 = total_groups * rank // world_size
    g1 = total_groups * (rank + 1) // world_size
    return g0, g1


def _up8(x):
    return (x + 7) // 8 * 8


def commit_section_bytes(groups, mode):
    bits = MODE_BITS[mode]
    if bits == 4:
        return 8 * ((groups + 15) // 16)       # sixteen groups per 64-bit word
    return groups * bits // 8


def summary_bytes(groups, mode="u16"):
    """Bytes of one rank's summary message: commit section (padded to 8), elected bit words."""
    return _up8(commit_section_bytes(groups, mode)) + 8 * ((groups + 63) // 64)


class SummaryExchange:
    """`depth` independent summary slots so that the all-gather of round i can
    run on the communication stream while round i+1 computes into another slot.
    Every rank must use the same `groups` (pad the last shard if needed)."""

    def __init__(self, groups, device, depth=2, group=None, mode="u16", impl="auto", rounds=1, fence="event"):
        """rounds: summaries of this many consecutive rounds share one message (one
        collective per `rounds` rounds: enqueuing an RCCL collective costs tens of
        microseconds of host time, more than a 1M-group round takes on the GPU).
        impl: "native" = librq's own RCCL communicator and side stream (one C
        call per exchange, a few microseconds of host time), "torch" =
        torch.distributed collectives, "auto" = native on a GPU when every rank
        can create the communicator, else torch.
        fence: how the side stream learns that a slot's rounds have finished.
        "event" = an event recorded on the compute stream behind the last round;
        "signal" (GPU, torch collectives) = the NEXT launch on the compute stream
        reports that it has started, which in an in-order stream means the slot's
        rounds are complete, through a signal word the side stream polls: no marker
        on the compute stream, so the next round starts while the last one drains.
        Protocol with signal fences: the launch that follows a slot's last round
        carries start_signal(slot) in its descriptor; call filled(slot) after the
        slot's last round has been enqueued and round_started(slot) after EVERY
        launch that carries start_signal(slot) (the host counts them); flush() /
        wait_all() fence whatever is still pending with an event."""
        assert mode in MODE_BITS and impl in ("auto", "native", "torch") and fence in ("event", "signal")
        self.bits = MODE_BITS[mode]
        self.groups, self.mode = int(groups), mode
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rounds = int(rounds)
        self.round_bytes = summary_bytes(self.groups, mode)
        self.nbytes = self.rounds * self.round_bytes      # one rank's message
        self._commit_bytes = commit_section_bytes(self.groups, mode)
        self._bits_off = _up8(self._commit_bytes)
        self.local = [torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.all = [torch.zeros((self.world, self.nbytes), dtype=torch.uint8, device=self.device)
                    for _ in range(depth)]
        self.work = [None] * depth
        # On a GPU the collective is issued from a side stream and fenced with one
        # event per slot: torch's async_op=True path costs ~30 us of host time per
        # call on this stack, a blocking call on a side stream ~12 us, and the round
        # itself is only ~30 us.
        self._cuda = self.device.type == "cuda"
        self._gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._comm = None
        if self._cuda and impl != "torch" and dist.is_initialized():
            self._comm = self._native_comm(depth)
            if self._comm is None and impl == "native":
                raise RuntimeError("native RCCL communicator could not be created on every rank")
        self.impl = "native" if self._comm is not None else "torch"
        if self._cuda and self._comm is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._done = [torch.cuda.Event() for _ in range(depth)]
            self._pending = [False] * depth
        self.fence = fence if (self._cuda and self._comm is None) else "event"
        self._eng = None
        if self.fence == "signal":
            import ctypes as C
            from . import _lib
            self._l = _lib.lib()
            h = C.c_void_p()
            rc = self._l.rq_engine_create(self.device.index or 0, 64, 1, 64, C.byref(h))
            if rc != 0:
                raise _lib.RqError(rc, "rq_engine_create")
            self._eng = h
            self._signals = []
            for _ in range(depth):
                p = C.c_void_p()
                if self._l.rq_device_alloc(h, 8, 3, C.byref(p)) != 0:      # RQ_MEM_SIGNAL
                    break
                self._signals.append(p.value)
            # a wait that is satisfied at once: does this stack take stream waits on signal memory at all?
            if len(self._signals) == depth and self._l.rq_stream_wait_signal(h, self._side.cuda_stream,
                                                                             self._signals[0], 0) != 0:
                self._l.rq_device_free(h, self._signals.pop())
            if len(self._signals) < depth:          # no signal memory / stream waits on this device: event fences
                for p in self._signals:
                    self._l.rq_device_free(h, p)
                self._l.rq_engine_destroy(h)
                self._eng, self.fence = None, "event"
        if self.fence == "signal":
            self._launched = [0] * depth           # launches enqueued so far that carry the slot's signal
        self._filled = [False] * depth             # the slot's rounds are enqueued, its exchange is not
        dt = {32: torch.int32, 16: torch.int16, 8: torch.uint8, 4: torch.int64}[self.bits]
        rb = self.round_bytes
        self._commit_views = [[t[r * rb: r * rb + self._commit_bytes].view(dt) for r in range(self.rounds)]
                              for t in self.local]
        self._bits_views = [[t[r * rb + self._bits_off: (r + 1) * rb].view(torch.int64) for r in range(self.rounds)]
                            for t in self.local]
        self._full = None          # lazily allocated [world, groups] int32 for the escape path

    def _native_comm(self, depth):
        """rq_comm_create on every rank with rank 0's unique id; None unless all ranks succeed."""
        import ctypes as C
        from . import _lib
        l = _lib.lib()
        idbuf = (C.c_uint8 * 128)()
        ok = 1
        if self.rank == 0 and l.rq_comm_unique_id(idbuf) != 0:
            ok = 0
        # the id travels as a tensor on the existing process group (works for nccl and gloo)
        t = torch.tensor(list(idbuf) + [ok], dtype=torch.int32, device=self.device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        vals = t.cpu().tolist()
        handle = C.c_void_p()
        if vals[128]:
            idbuf = (C.c_uint8 * 128)(*vals[:128])
            rc = l.rq_comm_create(self.device.index or 0, idbuf, self.rank, self.world, depth, C.byref(handle))
            ok = 1 if rc == 0 else 0
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 1:
            self._l = l
            self._send = [t_.data_ptr() for t_ in self.local]
            self._recv = [t_.data_ptr() for t_ in self.all]
            return handle
        if handle.value:
            l.rq_comm_destroy(handle)
        return None

    def close(self):
        if self._comm is not None:
            self._l.rq_comm_destroy(self._comm)
            self._comm = None
        if self._eng is not None:
            torch.cuda.synchronize(self.device)
            for p in self._signals:
                self._l.rq_device_free(self._eng, p)
            self._l.rq_engine_destroy(self._eng)
            self._eng = None

    def debug_state(self):
        """Host-side bookkeeping of every slot, for a diagnosis when a run is stuck (no device
        access: the device may be what is stuck)."""
        st = dict(impl=self.impl, fence=self.fence, world=self.world, rank=self.rank, rounds=self.rounds,
                  message_bytes=self.nbytes, filled=list(self._filled))
        if self.fence == "signal":
            st["launched_with_signal"] = list(self._launched)
        if self._cuda and self._comm is None:
            st["exchange_pending"] = list(self._pending)
        return st

    def start_signal(self, slot):
        """fence="signal": address of the slot's signal word, for the descriptor of the
        launch that follows the slot's last round; None with event fences."""
        return self._signals[slot] if self.fence == "signal" else None

    def filled(self, slot):
        """The slot's last round has been enqueued on the current stream.  Event
        fences: the exchange is issued now.  Signal fences: it is issued by the
        round_started(slot) that follows."""
        if self.fence == "signal":
            self._filled[slot] = True
        else:
            self.exchange(slot)

    def round_started(self, slot):
        """A launch carrying start_signal(slot) has just been enqueued."""
        if self.fence != "signal":
            return
        self._launched[slot] += 1
        if self._filled[slot]:
            self._filled[slot] = False
            self.exchange(slot, _after_signal=self._launched[slot])

    def launch_counted(self, slot):
        """A launch carrying start_signal(slot) has been enqueued by a caller that is not
        exchanging right now: the word it bumps keeps in step with the host's count."""
        if self.fence == "signal":
            self._launched[slot] += 1

    def flush(self):
        """Issue the exchange of every slot that is filled and still waits for a
        following launch (the end of a run), fenced with an event."""
        for slot, f in enumerate(self._filled):
            if f:
                self._filled[slot] = False
                self.exchange(slot)

    # ---- views handed to the kernel as outputs --------------------------------
    def commit_view(self, slot, sub=0):
        """Round `sub` of the slot: int32 [groups] (mode i32), int16 [groups] holding
        uint16 offsets (u16), uint8 [groups] (u8) or int64 [ceil(groups/16)] words of
        sixteen 4-bit offsets (u4)."""
        return self._commit_views[slot][sub]

    def bits_view(self, slot, sub=0):
        return self._bits_views[slot][sub]

    def begin(self, slot):
        """Call before the round that fills `slot`: waits for the slot's previous exchange."""
        self.wait(slot)

    # ---- the collective ------------------------------------------------------------
    def exchange(self, slot, async_op=True, _after_signal=None):
        """All-gather slot's summary, ordered after the work already enqueued on
        the current stream (the round that produced it), without blocking that
        stream: later rounds overlap with the collective."""
        if self.world == 1 and not dist.is_initialized():
            self.all[slot][0].copy_(self.local[slot], non_blocking=True)
            return
        if self._comm is not None:
            rc = self._l.rq_comm_allgather(self._comm, torch.cuda.current_stream(self.device).cuda_stream,
                                           slot, self._send[slot], self._recv[slot], self.nbytes)
            if rc != 0:
                raise RuntimeError("rq_comm_allgather failed: %d" % rc)
            return
        if self._cuda:
            if _after_signal is not None:
                # a launch enqueued after the slot's rounds bumps the word when it starts
                rc = self._l.rq_stream_wait_signal(self._eng, self._side.cuda_stream, self._signals[slot],
                                                   _after_signal)
                if rc != 0:
                    raise RuntimeError("rq_stream_wait_signal failed: %d" % rc)
            else:
                self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                if self._gloo:
                    # rehearsal on a GPU with the CPU backend: the message is staged through the
                    # host here (torch's gloo backend would stage device tensors through pool
                    # streams and worker threads of its own)
                    host = self.local[slot].to("cpu")          # blocks until the side stream has the message
                    out = torch.empty((self.world, self.nbytes), dtype=torch.uint8)
                    dist.all_gather_into_tensor(out.view(-1), host, group=self.group)
                    self.all[slot].copy_(out)
                else:
                    dist.all_gather_into_tensor(self.all[slot].view(-1), self.local[slot], group=self.group)
                self._done[slot].record(self._side)
            self._pending[slot] = True
            return
        w = dist.all_gather_into_tensor(self.all[slot].view(-1), self.local[slot],
                                        group=self.group, async_op=async_op)
        self.work[slot] = w if async_op else None

    def wait(self, slot):
        """Order the current stream (for gloo: the host) after slot's exchange."""
        if self._filled[slot]:                 # nothing was launched after the slot's rounds: event fence
            self._filled[slot] = False
            self.exchange(slot)
        if self._comm is not None:
            rc = self._l.rq_comm_wait(self._comm, torch.cuda.current_stream(self.device).cuda_stream, slot)
            if rc != 0:
                raise RuntimeError("rq_comm_wait failed: %d" % rc)
            return
        if self._cuda:
            if self._pending[slot]:
                # an exchange issued several rounds ago has normally finished: the wait
                # (a barrier packet that drains the compute stream) is only enqueued if not
                if not self._done[slot].query():
                    torch.cuda.current_stream(self.device).wait_event(self._done[slot])
                self._pending[slot] = False
            return
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None

    def wait_all(self):
        for s in range(len(self.work)):
            self.wait(s)
        if self._comm is not None:          # host-level: the gathered buffers are about to be read
            rc = self._l.rq_comm_sync(self._comm)
            if rc != 0:
                raise RuntimeError("rq_comm_sync failed: %d" % rc)

    # ---- reading a gathered slot --------------------------------------------------
    def _section(self, slot, sub, lo, hi):
        rb = self.round_bytes
        return self.all[slot][:, sub * rb + lo: sub * rb + hi].contiguous()

    def gathered_bits(self, slot, sub=0):
        """elected bits of every rank: int64 [world, ceil(groups/64)]."""
        return self._section(slot, sub, self._bits_off, self.round_bytes).view(torch.int64)

    def _offsets(self, slot, sub):
        """Narrow modes: the gathered offsets as int32 [world, groups]."""
        raw = self._section(slot, sub, 0, self._commit_bytes)
        if self.bits == 16:
            return (raw.view(torch.int16).to(torch.int32) & 0xFFFF).reshape(self.world, self.groups)
        if self.bits == 8:
            return raw.to(torch.int32).reshape(self.world, self.groups)
        b = raw.reshape(self.world, -1).to(torch.int32)          # little-endian words
        nib = torch.stack((b & 0x0F, b >> 4), dim=2).reshape(self.world, -1)
        return nib[:, : self.groups]

    def gathered_escapes(self, slot, sub=0):
        """Groups per rank whose offset left the summary window: int64 [world]
        (a reduction over the gathered message; synchronises if on a GPU)."""
        if self.mode == "i32":
            return torch.zeros(self.world, dtype=torch.int64)
        return (self._offsets(slot, sub) == (1 << self.bits) - 1).sum(dim=1).cpu()

    def gathered_commit(self, slot, sub=0, base=None, full_local_commit=None):
        """commitIndex of every group of every rank: int32 [world, groups].

        Narrow modes: offsets are added to `base` ([world, groups] int32 or None = 0).
        If any rank sent the escape value, `full_local_commit` (this rank's int32
        commit' vector of that round) is all-gathered and used instead; every
        rank sees the same message, so every rank takes this path together."""
        if self.mode == "i32":
            return self._section(slot, sub, 0, self._commit_bytes).view(torch.int32).reshape(self.world, self.groups)
        if int(self.gathered_escapes(slot, sub).sum()) > 0:
            if full_local_commit is None:
                raise RuntimeError("a rank's commit offsets left the summary window: pass full_local_commit")
            if self._full is None:
                self._full = torch.zeros((self.world, self.groups), dtype=torch.int32, device=self.device)
            if dist.is_initialized():
                # one communicator at a time: the native communicator's queued all-gathers
                # are drained before a collective goes out on the process group's own
                if self._comm is not None:
                    self.wait_all()
                dist.all_gather_into_tensor(self._full.view(-1), full_local_commit.contiguous(), group=self.group)
            else:
                self._full[0].copy_(full_local_commit)
            return self._full
        out = self._offsets(slot, sub)
        return out if base is None else out + base


class SummaryConsumer:
    """Consumer side of the exchange without a host synchronisation: this rank's
    replica of EVERY rank's commitIndex ([world, groups] int32 on the device) is
    advanced by each gathered round of advance-coded summaries (narrow modes with
    the round's input commit plane as the base) in a small kernel on the compute
    stream, ordered after the slot's all-gather.  Escape values are counted in a
    device counter (`escapes[0]`) and leave the replica entry alone; a host that
    finds the counter non-zero fetches the full vectors of that period."""

    def __init__(self, ex, initial=None):
        assert ex.bits in (16, 8, 4), "the device consumer decodes the narrow (advance-coded) summaries"
        assert ex.device.type == "cuda", "rq_summary_apply_dev is a GPU kernel"
        import ctypes as C
        from . import _lib
        self._lib = _lib
        self.ex = ex
        self._l = _lib.lib()
        h = C.c_void_p()
        rc = self._l.rq_engine_create(ex.device.index or 0, 64, 1, 64, C.byref(h))
        if rc != 0:
            raise _lib.RqError(rc, "rq_engine_create")
        self._h = h
        self.replica = torch.zeros((ex.world, ex.groups), dtype=torch.int32, device=ex.device)
        if initial is not None:
            self.replica.copy_(initial)
        self.escapes = torch.zeros(2, dtype=torch.int32, device=ex.device)

    def consume(self, slot):
        """Apply every round of `slot`'s gathered message to the replica.  Call after
        ex.exchange(slot); the kernels are ordered after that all-gather on the
        current stream and before the slot's next all-gather."""
        ex = self.ex
        ex.wait(slot)
        stream = torch.cuda.current_stream(ex.device).cuda_stream
        rc = self._l.rq_summary_apply_dev(self._h, stream, ex.all[slot].data_ptr(), ex.nbytes, 0,
                                          ex.round_bytes, ex.rounds, ex.world, ex.groups, ex.bits,
                                          self.replica.data_ptr(), self.escapes.data_ptr())
        if rc != 0:
            raise self._lib.RqError(rc, "rq_summary_apply_dev", self._l.rq_last_hip_error(self._h))

    def close(self):
        if self._h:
            self._l.rq_engine_destroy(self._h)
            self._h = None
