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


class DeviceBatch:
    """One batch of groups uploaded to a GPU plus its output buffers."""

    IN_KEYS = ("match", "commit", "leader_term", "log_off", "log_term",
               "voter_term", "voted_for", "last_log_term", "cand_term", "cand_id")

    def __init__(self, batch, device="cuda:0", in_place=False, commit_out=None, elected_bits=None):
        """commit_out / elected_bits: optional preallocated device views (e.g. a
        sharded.SummaryExchange slot) for the kernel to write its summary into."""
        self.G, self.n, self.ld = int(batch["G"]), int(batch["n"]), int(batch["ld"])
        self.device = torch.device(device)
        self.t = {}
        for k in self.IN_KEYS:
            a = batch[k]
            if a.dtype == np.uint32:            # torch has no uint32 arithmetic; same bits
                a = a.view(np.int32)
            if a.size == 0:
                a = np.zeros(1, a.dtype)
            self.t[k] = _dev(a, self.device)
        self.joint = "old_mask" in batch
        if self.joint:
            self.t["old_mask"] = _dev(batch["old_mask"].view(np.int16), self.device)
            self.t["new_mask"] = _dev(batch["new_mask"].view(np.int16), self.device)
        G, n, ld = self.G, self.n, self.ld
        o = {}
        o["commit_out"] = self.t["commit"] if in_place else torch.zeros(G, dtype=torch.int32, device=self.device)
        o["voter_term_out"] = self.t["voter_term"] if in_place else torch.zeros((n, ld), dtype=torch.int32, device=self.device)
        o["grant_mask"] = torch.zeros(G, dtype=torch.int16, device=self.device)
        o["stepdown_mask"] = torch.zeros(G, dtype=torch.int16, device=self.device)
        o["elected"] = torch.zeros(G, dtype=torch.uint8, device=self.device)
        o["max_term"] = torch.zeros(G, dtype=torch.int32, device=self.device)
        o["status"] = torch.zeros(G, dtype=torch.uint8, device=self.device)
        o["elected_bits"] = torch.zeros((G + 63) // 64, dtype=torch.int64, device=self.device)
        if commit_out is not None:
            assert commit_out.numel() == G and commit_out.dtype == torch.int32
            o["commit_out"] = commit_out
        if elected_bits is not None:
            assert elected_bits.numel() == (G + 63) // 64 and elected_bits.dtype == torch.int64
            o["elected_bits"] = elected_bits
        self.o = o

    def bytes_resident(self):
        return sum(x.numel() * x.element_size() for x in list(self.t.values()) + list(self.o.values()))

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

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def fused_round(self, b, variant=VAR_DEFAULT, want_bits=True, status=True):
        """Enqueue one fused round for DeviceBatch b on torch's current stream."""
        t, o = b.t, b.o
        p = lambda x: x.data_ptr()
        tail = [p(t["match"]), p(t["commit"]), p(t["leader_term"]), p(t["log_off"]), p(t["log_term"]),
                p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"]), p(t["cand_term"]), p(t["cand_id"]),
                p(o["commit_out"]), p(o["voter_term_out"]), p(o["grant_mask"]), p(o["stepdown_mask"]),
                p(o["elected"]), p(o["max_term"]), p(o["status"]) if status else None,
                p(o["elected_bits"]) if want_bits else None]
        if b.joint:
            rc = self._l.rq_fused_round_joint_dev(self._h, self._stream(), variant, b.G, b.n, b.ld,
                                                  p(t["old_mask"]), p(t["new_mask"]), *tail)
            self._check(rc, "rq_fused_round_joint_dev")
        else:
            rc = self._l.rq_fused_round_dev(self._h, self._stream(), variant, b.G, b.n, b.ld, *tail)
            self._check(rc, "rq_fused_round_dev")

    def commit_advance(self, b, variant=VAR_DEFAULT):
        t, o = b.t, b.o
        p = lambda x: x.data_ptr()
        rc = self._l.rq_commit_advance_dev(self._h, self._stream(), variant, b.G, b.n, b.ld,
                                           p(t["match"]), p(t["commit"]), p(t["leader_term"]),
                                           p(t["log_off"]), p(t["log_term"]), p(o["commit_out"]),
                                           p(o["status"]))
        self._check(rc, "rq_commit_advance_dev")

    def vote_tally(self, b, variant=VAR_DEFAULT, want_bits=True):
        t, o = b.t, b.o
        p = lambda x: x.data_ptr()
        rc = self._l.rq_vote_tally_dev(self._h, self._stream(), variant, b.G, b.n, b.ld,
                                       p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"]),
                                       p(t["cand_term"]), p(t["cand_id"]), p(o["voter_term_out"]),
                                       p(o["grant_mask"]), p(o["stepdown_mask"]), p(o["elected"]),
                                       p(o["max_term"]), p(o["status"]),
                                       p(o["elected_bits"]) if want_bits else None)
        self._check(rc, "rq_vote_tally_dev")
