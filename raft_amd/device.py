"""Device-resident rounds: group state lives in HBM (torch tensors are used
only as device allocations) and rq_*_dev launches run on a caller stream.

This is the path a deployment keeps hot between rounds and the one bench.py
times; the host-buffer entry points in engine.py add PCIe copies around the
same kernels.
"""
import numpy as np
import torch

from . import _lib
from ._lib import RqError

VAR_DEFAULT = 0
VAR_NT_LOAD = 0x10
VAR_NT_STORE = 0x20


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def summary_buffer_bytes(groups, bits):
    """Bytes of the narrow commit summary of `groups` groups: 16/8 bits per group, or
    4 bits per group packed sixteen to a 64-bit word."""
    if bits == 4:
        return 8 * ((groups + 15) // 16)
    return groups * bits // 8


def decode_summary(raw, groups, bits):
    """Summary buffer (numpy, any dtype) -> uint16 [groups] of offsets; the escape
    value is (1 << bits) - 1."""
    b = np.ascontiguousarray(raw).view(np.uint8)[: summary_buffer_bytes(groups, bits)]
    if bits == 16:
        return b.view(np.uint16)[:groups].copy()
    if bits == 8:
        return b[:groups].astype(np.uint16)
    nib = np.empty(b.size * 2, np.uint16)
    nib[0::2] = b & 0x0F            # little-endian words: group g sits in bits 4*(g%16)
    nib[1::2] = b >> 4
    return nib[:groups]


class DeviceBatch:
    """One batch of groups uploaded to a GPU plus its output buffers."""

    IN_KEYS = ("match", "commit", "leader_term", "log_off", "log_term",
               "voter_term", "voted_for", "last_log_term", "cand_term", "cand_id")

    def __init__(self, batch, device="cuda:0", in_place=False, commit_out=None, elected_bits=None,
                 share_log_with=None, commit_lo16=None, summary_escapes=None, summary_base=None,
                 summary_bits=16):
        """commit_out / elected_bits: optional preallocated device views (e.g. a
        sharded.SummaryExchange slot) for the kernel to write its summary into.
        share_log_with: another DeviceBatch whose log_off/log_term device arrays
        are reused (the batch must have been generated with the same logs).
        commit_lo16: device buffer for the narrow commit summary in summary_bits
        (16, 8 or 4) bits per group -- the engine must be set to the same width
        (DeviceEngine.set_summary_width).  summary_base: a device tensor, None
        (= 0) or the string "commit" for the batch's own input commit plane (the
        summary is then the round's advance)."""
        self.G, self.n, self.ld = int(batch["G"]), int(batch["n"]), int(batch["ld"])
        self.device = torch.device(device)
        G, n, ld = self.G, self.n, self.ld
        self.joint = "old_mask" in batch
        self.wide = np.asarray(batch["match"]).dtype == np.int64    # int64 planes (rq_round64_dev)
        vdt, tdt = (np.int64, torch.int64) if self.wide else (np.int32, torch.int32)
        host = {}
        for k in self.IN_KEYS:
            if share_log_with is not None and k in ("log_off", "log_term"):
                continue
            a = batch[k]
            if a.dtype == np.uint32:            # torch has no uint32 arithmetic; same bits
                a = a.view(np.int32)
            if a.size == 0:
                a = np.zeros(1, a.dtype)
            host[k] = np.ascontiguousarray(a)
        if "log_base" in batch:                 # windowed log arena
            host["log_base"] = np.ascontiguousarray(batch["log_base"], dtype=vdt)
        if self.joint:
            host["old_mask"] = np.ascontiguousarray(batch["old_mask"].view(np.int16))
            host["new_mask"] = np.ascontiguousarray(batch["new_mask"].view(np.int16))
        outs = [("commit_out", (G,), tdt), ("voter_term_out", (n, ld), tdt),
                ("grant_mask", (G,), torch.int16), ("stepdown_mask", (G,), torch.int16),
                ("elected", (G,), torch.uint8), ("max_term", (G,), tdt),
                ("status", (G,), torch.uint8), ("elected_bits", ((G + 63) // 64,), torch.int64)]
        self.t, o = {}, {}
        for k, a in host.items():
            self.t[k] = torch.from_numpy(a).to(self.device)
        for k, shape, dt in outs:
            o[k] = torch.zeros(shape, dtype=dt, device=self.device)
        if share_log_with is not None:
            self.t["log_off"] = share_log_with.t["log_off"]
            self.t["log_term"] = share_log_with.t["log_term"]
        if in_place:
            o["commit_out"] = self.t["commit"]
            o["voter_term_out"] = self.t["voter_term"]
        if commit_out is not None:
            assert commit_out.numel() == G and commit_out.dtype == tdt
            o["commit_out"] = commit_out
        if elected_bits is not None:
            assert elected_bits.numel() == (G + 63) // 64 and elected_bits.dtype == torch.int64
            o["elected_bits"] = elected_bits
        # narrow (16-bit window) commit summary: only written when a buffer is supplied
        if commit_lo16 is not None:
            assert commit_lo16.numel() * commit_lo16.element_size() >= summary_buffer_bytes(G, summary_bits)
            assert commit_lo16.data_ptr() % 8 == 0 or summary_bits != 4
            assert summary_escapes is not None and summary_escapes.dtype == torch.int32
        o["commit_lo16"] = commit_lo16
        o["summary_escapes"] = summary_escapes
        self.summary_bits = summary_bits
        self.summary_base = self.t["commit"] if isinstance(summary_base, str) else summary_base
        self.o = o
        self._args = {}
        self._plans = []

    def __del__(self):
        for l, plan in getattr(self, "_plans", []):
            l.rq_plan_destroy(plan)

    def bytes_resident(self):
        return sum(x.numel() * x.element_size() for x in list(self.t.values()) + list(self.o.values())
                   if x is not None)

    def results(self):
        """Outputs as numpy arrays with the C-ABI dtypes."""
        torch.cuda.synchronize(self.device)
        o = self.o
        return dict(
            commit=o["commit_out"].cpu().numpy(),
            voter_term=o["voter_term_out"].cpu().numpy(),
            grant_mask=o["grant_mask"].cpu().numpy().view(np.uint16),
            stepdown_mask=o["stepdown_mask"].cpu().numpy().view(np.uint16),
            elected=o["elected"].cpu().numpy(),
            max_term=o["max_term"].cpu().numpy(),
            status=o["status"].cpu().numpy(),
            elected_bits=o["elected_bits"].cpu().numpy().view(np.uint64),
            commit_lo16=None if o["commit_lo16"] is None else decode_summary(
                o["commit_lo16"].cpu().numpy(), self.G, self.summary_bits),
            summary_escapes=None if o["summary_escapes"] is None else int(o["summary_escapes"].cpu()[0]),
        )


class DeviceEngine:
    """rq_engine handle used only for its device binding; buffers are the caller's."""

    def __init__(self, device_index=0):
        import ctypes as C
        self._l = _lib.lib()
        h = C.c_void_p()
        rc = self._l.rq_engine_create(device_index, 64, 1, 64, C.byref(h))
        if rc != 0:
            raise RqError(rc, "rq_engine_create")
        self._h = h

    def close(self):
        if self._h:
            self._l.rq_engine_destroy(self._h)
            self._h = None

    def _check(self, rc, where):
        if rc != 0:
            raise RqError(rc, where, self._l.rq_last_hip_error(self._h))

    def set_summary_width(self, bits):
        """Width of the narrow commit summary written by later launches and plans."""
        self._check(self._l.rq_engine_set_summary_width(self._h, bits), "rq_engine_set_summary_width")
        self._width = bits

    def _match_width(self, b):
        """The engine-wide width must be the batch's when its launch or plan is made."""
        if b.o["commit_lo16"] is not None and getattr(self, "_width", 16) != b.summary_bits:
            self.set_summary_width(b.summary_bits)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def alloc(self, nbytes, kind=0):
        """Raw device memory through the C-ABI (kind: 0 default, 1 fine-grained, 2 uncached)."""
        import ctypes as C
        p = C.c_void_p()
        self._check(self._l.rq_device_alloc(self._h, nbytes, kind, C.byref(p)), "rq_device_alloc")
        return p.value

    def free(self, addr):
        self._check(self._l.rq_device_free(self._h, addr), "rq_device_free")

    def upload(self, addr, array):
        a = np.ascontiguousarray(array)
        self._check(self._l.rq_copy_to_device(self._h, addr, a.ctypes.data, a.nbytes), "rq_copy_to_device")

    def download(self, addr, array):
        """Fill the (contiguous) numpy array from device memory at addr."""
        self._check(self._l.rq_copy_from_device(self._h, array.ctypes.data, addr, array.nbytes),
                    "rq_copy_from_device")

    def fused_round(self, b, variant=VAR_DEFAULT, want_bits=True, status=True):
        """Enqueue one fused round for DeviceBatch b on torch's current stream.
        The argument set is validated once into an rq_plan per (batch, options);
        every later call is a two-argument rq_plan_launch."""
        if b.wide:
            return self.round64(b, 3, variant, want_bits, status)
        key = (variant, want_bits, status, b.summary_bits)
        plan = b._args.get(key)
        if plan is None:
            import ctypes as C
            self._match_width(b)
            t, o = b.t, b.o
            p = lambda x: None if x is None else x.data_ptr()
            plan = C.c_void_p()
            rc = self._l.rq_plan_fused_round(
                self._h, variant, 1 if b.joint else 0, b.G, b.n, b.ld,
                p(t.get("old_mask")), p(t.get("new_mask")),
                p(t["match"]), p(t["commit"]), p(t["leader_term"]), p(t["log_off"]), p(t["log_term"]),
                p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"]), p(t["cand_term"]), p(t["cand_id"]),
                p(o["commit_out"]), p(o["voter_term_out"]), p(o["grant_mask"]), p(o["stepdown_mask"]),
                p(o["elected"]), p(o["max_term"]), p(o["status"]) if status else None,
                p(o["elected_bits"]) if want_bits else None,
                p(o["commit_lo16"]), p(b.summary_base), p(o["summary_escapes"]), p(t.get("log_base")),
                C.byref(plan))
            if rc != 0:
                self._check(rc, "rq_plan_fused_round")
            b._args[key] = plan
            b._plans.append((self._l, plan))
        rc = self._l.rq_plan_launch(plan, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            self._check(rc, "rq_plan_launch")

    def fused_round_direct(self, b, variant=VAR_DEFAULT, want_bits=True, status=True):
        """The same round through rq_fused_round_dev / rq_fused_round_joint_dev (all
        arguments marshalled on every call)."""
        self._match_width(b)
        t, o = b.t, b.o
        p = lambda x: None if x is None else x.data_ptr()
        tail = [p(t["match"]), p(t["commit"]), p(t["leader_term"]), p(t["log_off"]), p(t["log_term"]),
                p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"]), p(t["cand_term"]), p(t["cand_id"]),
                p(o["commit_out"]), p(o["voter_term_out"]), p(o["grant_mask"]), p(o["stepdown_mask"]),
                p(o["elected"]), p(o["max_term"]), p(o["status"]) if status else None,
                p(o["elected_bits"]) if want_bits else None,
                p(o["commit_lo16"]), p(b.summary_base), p(o["summary_escapes"]), p(t.get("log_base"))]
        if b.joint:
            rc = self._l.rq_fused_round_joint_dev(self._h, self._stream(), variant, b.G, b.n, b.ld,
                                                  p(t["old_mask"]), p(t["new_mask"]), *tail)
        else:
            rc = self._l.rq_fused_round_dev(self._h, self._stream(), variant, b.G, b.n, b.ld, *tail)
        if rc != 0:
            self._check(rc, "rq_fused_round_dev")

    def _round64_args(self, b, want_bits, status):
        t, o = b.t, b.o
        p = lambda x: None if x is None else x.data_ptr()
        return [b.G, b.n, b.ld,
                p(t.get("old_mask")), p(t.get("new_mask")),
                p(t["match"]), p(t["commit"]), p(t["leader_term"]), p(t["log_off"]), p(t["log_term"]),
                p(t.get("log_base")),
                p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"]), p(t["cand_term"]), p(t["cand_id"]),
                p(o["commit_out"]), p(o["voter_term_out"]), p(o["grant_mask"]), p(o["stepdown_mask"]),
                p(o["elected"]), p(o["max_term"]), p(o["status"]) if status else None,
                p(o["elected_bits"]) if want_bits else None,
                p(o["commit_lo16"]), p(b.summary_base), p(o["summary_escapes"])]

    def round64(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """Enqueue the commit (what & 1) and/or vote (what & 2) decision for a wide
        (int64) DeviceBatch: validated once into an rq_plan (rq_plan_round64), then
        launched with rq_plan_launch."""
        assert b.wide
        key = ("w", what, variant, want_bits, status, b.summary_bits)
        plan = b._args.get(key)
        if plan is None:
            import ctypes as C
            self._match_width(b)
            plan = C.c_void_p()
            rc = self._l.rq_plan_round64(self._h, variant, what, *self._round64_args(b, want_bits, status),
                                         C.byref(plan))
            if rc != 0:
                self._check(rc, "rq_plan_round64")
            b._args[key] = plan
            b._plans.append((self._l, plan))
        rc = self._l.rq_plan_launch(plan, self._stream())
        if rc != 0:
            self._check(rc, "rq_plan_launch")

    def round64_direct(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """The same through rq_round64_dev (all arguments marshalled on every call)."""
        assert b.wide
        self._match_width(b)
        rc = self._l.rq_round64_dev(self._h, self._stream(), variant, what,
                                    *self._round64_args(b, want_bits, status))
        if rc != 0:
            self._check(rc, "rq_round64_dev")

    def commit_advance(self, b, variant=VAR_DEFAULT):
        if b.wide:
            return self.round64(b, 1, variant)
        t, o = b.t, b.o
        p = lambda x: x.data_ptr()
        rc = self._l.rq_commit_advance_dev(self._h, self._stream(), variant, b.G, b.n, b.ld,
                                           p(t["match"]), p(t["commit"]), p(t["leader_term"]),
                                           p(t["log_off"]), p(t["log_term"]), p(o["commit_out"]),
                                           p(o["status"]),
                                           t["log_base"].data_ptr() if "log_base" in t else None)
        self._check(rc, "rq_commit_advance_dev")

    def vote_tally(self, b, variant=VAR_DEFAULT, want_bits=True):
        if b.wide:
            return self.round64(b, 2, variant, want_bits)
        t, o = b.t, b.o
        p = lambda x: x.data_ptr()
        rc = self._l.rq_vote_tally_dev(self._h, self._stream(), variant, b.G, b.n, b.ld,
                                       p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"]),
                                       p(t["cand_term"]), p(t["cand_id"]), p(o["voter_term_out"]),
                                       p(o["grant_mask"]), p(o["stepdown_mask"]), p(o["elected"]),
                                       p(o["max_term"]), p(o["status"]),
                                       p(o["elected_bits"]) if want_bits else None)
        self._check(rc, "rq_vote_tally_dev")

    def append_check(self, t, G, n, ld):
        """Enqueue the AppendEntries consistency batch on device tensors (rq_append_check_dev,
        or rq_append_check64_dev when the planes are int64).  t: dict of tensors with the
        C-ABI names; active_mask, n_new, match_index, next_index and status may be absent.
        follower_term_out may be the follower_term tensor itself (in-place terms);
        match_index / next_index are updated in place."""
        wide = t["follower_term"].dtype == torch.int64
        p = lambda k: t[k].data_ptr() if t.get(k) is not None else None
        fn = self._l.rq_append_check64_dev if wide else self._l.rq_append_check_dev
        rc = fn(self._h, self._stream(), G, n, ld, p("active_mask"), p("follower_term"), p("leader_term"),
                p("prev_log_index"), p("prev_log_term"), p("n_new"), p("flog_off"), p("flog_term"),
                p("follower_term_out"), p("success_mask"), p("stepdown_mask"), p("match_index"),
                p("next_index"), p("status"))
        if rc != 0:
            self._check(rc, "rq_append_check_dev")
