"""Device-resident rounds: group state lives in HBM (torch tensors are used
only as device allocations) and rq_round_dev launches run on a caller stream.

This is the path a deployment keeps hot between rounds and the one bench.py
times; the host-buffer entry points in engine.py add PCIe copies around the
same kernels.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import RqError, RoundDesc, AppendDesc

VAR_DEFAULT = 0
VAR_PLAIN = 0x1            # default cache policy instead of non-temporal streaming
LAYOUT_I32, LAYOUT_I64 = 4, 8


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


def encode_log_terms(log_term, log_off=None, bits=0, per_group_base=False):
    """The log-term arena in `bits`-wide entries (what a host keeping the arena
    resident stores): 8 / 16 = unsigned, 32 = int32, 64 = int64; bits = 0 picks the
    narrowest of 8, 16 and the array's own width that every entry fits.  With
    per_group_base the entries are differences from the smallest term of their
    group's slice (needs log_off); returns (arena, bits, base or None).  A round
    reads one term per group but HBM hands out 128-byte lines, so the gather's
    traffic shrinks with the entry width."""
    t = np.ascontiguousarray(log_term)
    full = 8 * t.dtype.itemsize
    base = None
    v = t.astype(np.int64)
    if per_group_base:
        off = np.asarray(log_off).astype(np.int64)
        G = off.size - 1
        lens = np.diff(off)
        base = np.zeros(G, np.int64)
        nz = lens > 0
        if v.size:
            base[nz] = np.minimum.reduceat(v[: off[-1]], off[:-1][nz]) if nz.any() else 0
        v = v.copy()
        v[: off[-1]] -= np.repeat(base, lens)
        base = base.astype(t.dtype)
    lo, hi = (int(v.min()), int(v.max())) if v.size else (0, 0)
    if bits == 0:
        bits = 8 if (lo >= 0 and hi < 256) else 16 if (lo >= 0 and hi < 65536) else full
    dt = {8: np.uint8, 16: np.uint16, 32: np.int32, 64: np.int64}[bits]
    if v.size and (lo < np.iinfo(dt).min or hi > np.iinfo(dt).max):
        raise ValueError("log terms do not fit %d-bit arena entries" % bits)
    return v.astype(dt), bits, base


class TiledAppendBatch:
    """An AppendEntries batch (synth.gen_append: planes [n][ld], follower-log CSR over slot
    ids) on the device in the 64-group tiles of rq_append_desc.tile_bytes: one block for the
    inputs and the arrays updated in place, one for the pure outputs.  .t maps the C-ABI names
    to views that start at each array's place inside tile 0 (what the descriptor's pointers
    name), for DeviceEngine.append_check; fetch(name) returns an array as [slots][G] / [G]."""

    def __init__(self, b, device="cuda:0", term_bits=0, in_place_terms=True, extra_in=None, extra_out=None):
        """extra_in: {name: host array [G] or [slots][G]} and extra_out: [(name, slots, numpy dtype)]
        -- further arrays laid into the same two blocks, e.g. the commit round's commit /
        log_off / commit_out, so that ONE resident image serves replication and rounds
        (the round's descriptor then names match_index as its match array)."""
        G, n, ld = int(b["G"]), int(b["n"]), int(b["ld"])
        self.G, self.n = G, n
        self.device = torch.device(device)
        vdt = np.asarray(b["follower_term"]).dtype
        ntiles = (G + 1 + 63) // 64                      # the per-slot offsets have G + 1 entries
        off = np.asarray(b["flog_off"]).astype(np.int64)
        off_ps = np.stack([off[p * ld: p * ld + G + 1] for p in range(n)]).astype(np.uint32).view(np.int32)
        ins = [(k, n, np.asarray(b[k])[:, :G]) for k in ("follower_term", "prev_log_index", "prev_log_term", "n_new",
                                                      "match_index", "next_index", "last_new_index",
                                                      "follower_commit") if k in b]
        if "leader_commit" in b:
            ins.append(("leader_commit", 1, np.asarray(b["leader_commit"])[None, :G]))
        if "flog_base" in b:
            ins.append(("flog_base", n, np.asarray(b["flog_base"])[:, :G]))
        ins.append(("flog_off", n, off_ps))
        ins.append(("leader_term", 1, np.asarray(b["leader_term"])[None, :G]))
        if "active_mask" in b:
            ins.append(("active_mask", 1, np.asarray(b["active_mask"]).view(np.int16)[None, :G]))
        if "flog_term_base" in b:
            ins.append(("flog_term_base", 1, np.asarray(b["flog_term_base"])[None, :G]))
        for k, a in (extra_in or {}).items():
            a = np.asarray(a)
            ins.append((k, a.shape[0] if a.ndim == 2 else 1, a.reshape(a.shape[0] if a.ndim == 2 else 1, -1)))
        outs = [("success_mask", 1, np.dtype(np.int16)), ("stepdown_mask", 1, np.dtype(np.int16)),
                ("status", 1, np.dtype(np.uint8))] + [(k, sl, np.dtype(dt)) for k, sl, dt in (extra_out or [])]
        if not in_place_terms:
            outs.insert(0, ("follower_term_out", n, np.dtype(vdt)))
        ins.sort(key=lambda f: -f[2].dtype.itemsize)
        outs.sort(key=lambda f: -f[2].itemsize)
        self._place, pos = {}, 0
        for k, slots, a in ins:
            self._place[k] = (False, pos, slots, a.dtype)
            pos += slots * 64 * a.dtype.itemsize
        self.tile_bytes = (pos + 255) // 256 * 256
        pos = 0
        for k, slots, dt in outs:
            self._place[k] = (True, pos, slots, dt)
            pos += slots * 64 * dt.itemsize
        self.out_tile_bytes = (pos + 255) // 256 * 256
        block = np.zeros((ntiles, self.tile_bytes), np.uint8)
        for k, slots, a in ins:
            o = self._place[k][1]
            pad = np.zeros((slots, ntiles * 64), a.dtype)
            cols = min(a.shape[1], ntiles * 64)
            pad[:, :cols] = a[:, :cols]
            view = block[:, o: o + slots * 64 * a.dtype.itemsize].view(a.dtype).reshape(ntiles, slots, 64)
            view[...] = pad.reshape(slots, ntiles, 64).transpose(1, 0, 2)
        self._in = torch.from_numpy(block.reshape(-1)).to(self.device)
        oblock = np.zeros((ntiles, self.out_tile_bytes), np.uint8)
        if not in_place_terms:
            # slots that receive no RPC are not written by the batch: they keep the follower's term
            ft = np.asarray(b["follower_term"])[:, :G]
            o = self._place["follower_term_out"][1]
            pad = np.zeros((n, ntiles * 64), ft.dtype)
            pad[:, :G] = ft
            view = oblock[:, o: o + n * 64 * ft.dtype.itemsize].view(ft.dtype).reshape(ntiles, n, 64)
            view[...] = pad.reshape(n, ntiles, 64).transpose(1, 0, 2)
        self._out = torch.from_numpy(oblock.reshape(-1)).to(self.device)
        enc, self.term_bits, _ = encode_log_terms(b["flog_term"], bits=term_bits)
        if enc.size == 0:
            enc = np.zeros(1, enc.dtype)
        self.flog_terms = int(np.asarray(b["flog_term"]).size)
        self.t = {k: (self._out if is_out else self._in)[o:] for k, (is_out, o, _, _) in self._place.items()}
        self.t["flog_term"] = torch.from_numpy(enc).to(self.device)
        if in_place_terms:
            self.t["follower_term_out"] = self.t["follower_term"]
            self._place["follower_term_out"] = self._place["follower_term"]
        self.t["_tile_bytes"], self.t["_out_tile_bytes"] = self.tile_bytes, self.out_tile_bytes
        self.wide = vdt == np.int64

    def store(self, name, a):
        """Overwrite array `name` inside the tiles with host array a ([slots][>= G] or [>= G])."""
        is_out, o, slots, dt = self._place[name]
        blk = self._out if is_out else self._in
        stride = self.out_tile_bytes if is_out else self.tile_bytes
        ntiles = blk.numel() // stride
        a = np.asarray(a, dtype=dt).reshape(slots, -1)[:, : self.G]
        pad = np.zeros((slots, ntiles * 64), dt)
        pad[:, : self.G] = a
        tiles = np.ascontiguousarray(pad.reshape(slots, ntiles, 64).transpose(1, 0, 2)).reshape(ntiles, -1)
        view = blk.view(ntiles, stride)[:, o: o + slots * 64 * dt.itemsize]
        view.copy_(torch.from_numpy(tiles.view(np.uint8)).to(self.device))

    def fetch(self, name):
        is_out, o, slots, dt = self._place[name]
        torch.cuda.synchronize(self.device)
        blk = (self._out if is_out else self._in).cpu().numpy().reshape(-1, self.out_tile_bytes if is_out else self.tile_bytes)
        v = blk[:, o: o + slots * 64 * dt.itemsize].view(dt).reshape(blk.shape[0], slots, 64)
        a = v.transpose(1, 0, 2).reshape(slots, -1)[:, : self.G]
        return a.copy() if slots > 1 else a[0].copy()


class DeviceBatch:
    """One batch of groups uploaded to a GPU plus its output buffers."""

    IN_KEYS = ("match", "commit", "leader_term", "log_off", "log_term",
               "voter_term", "voted_for", "last_log_term", "cand_term", "cand_id")

    def __init__(self, batch, device="cuda:0", in_place=False, commit_out=None, elected_bits=None,
                 share_log_with=None, commit_lo16=None, summary_escapes=None, summary_base=None,
                 summary_bits=16, log_term_bits=None, log_term_base=False, tiled=False, start_signal=None,
                 term_start=False, segments=None, own_log_off=False):
        """commit_out / elected_bits: optional preallocated device views (e.g. a
        sharded.SummaryExchange slot) for the kernel to write its summary into.
        share_log_with: another DeviceBatch whose log_off/log_term device arrays
        are reused (the batch must have been generated with the same logs).
        commit_lo16: device buffer for the narrow commit summary in summary_bits
        (16, 8 or 4) bits per group -- the engine must be set to the same width
        (the width travels in the launch descriptor).  summary_base: a device
        tensor, None (= 0) or the string "commit" for the batch's own input commit
        plane (the summary is then the round's advance).
        log_term_bits: width of the log-term arena entries on the device: None =
        the layout's own element (int32 / int64), 0 = the narrowest that fits, or
        8 / 16 / 32; log_term_base=True stores differences from each group's
        smallest resident term and uploads that base plane (encode_log_terms).
        tiled=True keeps every per-group and per-slot array, inputs and outputs, in
        ONE block of 64-group tiles (rq_round_desc.tile_bytes) instead of separate
        slot-major planes; the log arena and the bit summaries stay plain arrays.
        start_signal: device address of a signal word the launch bumps when it starts
        (everything enqueued before it on its stream has completed by then).
        term_start=True: no log arena on the device at all -- per group the index of the
        first entry of the leader's term and the log length (synth.term_start_form, or
        the two arrays themselves; the batch's logs must be non-decreasing and not
        above the leader's term).
        own_log_off: with share_log_with, share only the arena; this batch's offsets into it
        are its own (synth.split_by_size images).
        segments: (first_group, n) as synth.segment_table returns it -- the groups are
        ordered by cluster size and every launch says so (rq_round_desc.n_segments)."""
        self.start_signal = start_signal
        self.segments = segments
        self.G, self.n, self.ld = int(batch["G"]), int(batch["n"]), int(batch["ld"])
        self.device = torch.device(device)
        G, n, ld = self.G, self.n, self.ld
        self.joint = "old_mask" in batch
        self.wide = np.asarray(batch["match"]).dtype == np.int64    # int64 planes (RQ_LAYOUT_I64)
        vdt, tdt = (np.int64, torch.int64) if self.wide else (np.int32, torch.int32)
        host = {}
        full_bits = 64 if self.wide else 32
        self.log_term_bits = full_bits
        self.log_terms = int(np.asarray(batch["log_term"]).size)
        enc_term = None
        if share_log_with is not None:
            self.log_term_bits, self.log_terms = share_log_with.log_term_bits, share_log_with.log_terms
        elif log_term_bits is not None or log_term_base:
            enc_term, self.log_term_bits, tbase = encode_log_terms(
                batch["log_term"], batch["log_off"], 0 if log_term_bits is None else log_term_bits, log_term_base)
            if tbase is not None:
                host["log_term_base"] = np.ascontiguousarray(tbase, dtype=vdt)
        self.term_start = bool(term_start)
        if term_start:
            from . import synth
            ts, ln = term_start if isinstance(term_start, tuple) else synth.term_start_form(batch)
            host["log_term_start"] = np.ascontiguousarray(ts, dtype=vdt)
            host["log_len"] = np.ascontiguousarray(ln.view(np.int32))
            self.log_terms = 0
        for k in self.IN_KEYS:
            if term_start and k in ("log_off", "log_term", "leader_term"):
                continue                        # not read by a round that knows where the term starts
            if share_log_with is not None and (k == "log_term" or (k == "log_off" and not tiled and not own_log_off)):
                continue                        # (a tiled batch keeps its offsets inside its own tiles)
            a = batch[k]
            if k == "log_term" and enc_term is not None:
                a = enc_term
            if a.dtype == np.uint32:            # torch has no uint32 arithmetic; same bits
                a = a.view(np.int32)
            if a.size == 0:
                a = np.zeros(1, a.dtype)
            host[k] = np.ascontiguousarray(a)
        if "log_len" in batch and not term_start:   # logs with room behind them: length per group, offsets unordered
            host["log_len"] = np.ascontiguousarray(np.asarray(batch["log_len"]).astype(np.uint32).view(np.int32))
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
        self.tile_bytes = 0
        if tiled:
            assert commit_out is None, "a tiled batch keeps commit_out inside its tiles"
            self._tile_layout(host, outs, in_place)
            for k in ("log_term",):
                if k in host:
                    self.t[k] = torch.from_numpy(host[k]).to(self.device)
            o.update(self._tile_out)
            o["elected_bits"] = torch.zeros((G + 63) // 64, dtype=torch.int64, device=self.device)
        else:
            for k, a in host.items():
                self.t[k] = torch.from_numpy(a).to(self.device)
            for k, shape, dt in outs:
                o[k] = torch.zeros(shape, dtype=dt, device=self.device)
        if share_log_with is not None:
            if not tiled and not own_log_off:
                self.t["log_off"] = share_log_with.t["log_off"]
            self.t["log_term"] = share_log_with.t["log_term"]
            if "log_term_base" in share_log_with.t:
                self.t["log_term_base"] = share_log_with.t["log_term_base"]
        if in_place and not tiled:
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

    def _tile_layout(self, host, outs, in_place):
        """Two blocks of ceil((G+1)/64) tiles, one for the input arrays and one for the
        outputs (reads and writes in separate address streams); a tile holds, back to
        back, slots x 64 elements of each of its arrays (widest elements first, so
        everything stays aligned).  self.t / self._tile_out map names to 1-D byte
        views that start at the array's place inside tile 0 of its block (what the
        descriptor's pointers name)."""
        G = self.G
        np_of = {torch.int32: np.int32, torch.int64: np.int64, torch.int16: np.int16, torch.uint8: np.uint8}
        ins, outf = [], []                            # (name, slots, numpy dtype, host array)
        for k, a in host.items():
            if k != "log_term":
                ins.append((k, a.shape[0] if a.ndim == 2 else 1, a.dtype, a))
        for k, shape, dt in outs:
            if k == "elected_bits" or (in_place and k in ("commit_out", "voter_term_out")):
                continue
            outf.append((k, shape[0] if len(shape) == 2 else 1, np.dtype(np_of[dt]), None))
        ntiles = (G + 1 + 63) // 64                   # log_off has G + 1 entries

        def lay(fields):
            fields.sort(key=lambda f: -f[2].itemsize)
            off, place = 0, {}
            for k, slots, dt, _ in fields:
                place[k] = (off, slots, dt)
                off += slots * 64 * dt.itemsize
            return place, (off + 255) // 256 * 256

        in_place_of, self.tile_bytes = lay(ins)
        out_place_of, self.out_tile_bytes = lay(outf)
        block = np.zeros((ntiles, self.tile_bytes), np.uint8)
        for k, slots, dt, a in ins:
            a2 = a.reshape(slots, -1)
            pad = np.zeros((slots, ntiles * 64), dt)
            cols = min(a2.shape[1], ntiles * 64)
            pad[:, :cols] = a2[:, :cols]
            off = in_place_of[k][0]
            view = block[:, off: off + slots * 64 * dt.itemsize].view(dt).reshape(ntiles, slots, 64)
            view[...] = pad.reshape(slots, ntiles, 64).transpose(1, 0, 2)
        self._tile_block = torch.from_numpy(block.reshape(-1)).to(self.device)
        self._out_block = torch.zeros(ntiles * self.out_tile_bytes, dtype=torch.uint8, device=self.device)
        self._tile_fields = {}                        # name -> (block is the output block, offset, slots, dtype)
        self._tile_out = {}
        for k, (off, slots, dt) in in_place_of.items():
            self.t[k] = self._tile_block[off:]
            self._tile_fields[k] = (False, off, slots, dt)
        for k, (off, slots, dt) in out_place_of.items():
            self._tile_out[k] = self._out_block[off:]
            self._tile_fields[k] = (True, off, slots, dt)
        if in_place:
            self._tile_out["commit_out"] = self.t["commit"]
            self._tile_out["voter_term_out"] = self.t["voter_term"]
            self._tile_fields["commit_out"] = self._tile_fields["commit"]
            self._tile_fields["voter_term_out"] = self._tile_fields["voter_term"]

    def _untile(self, blocks, name):
        """Array `name` out of the downloaded tile blocks: [slots, G] for the voter
        terms, [G] for per-group arrays."""
        is_out, off, slots, dt = self._tile_fields[name]
        block = blocks[1] if is_out else blocks[0]
        ntiles = block.shape[0]
        v = block[:, off: off + slots * 64 * dt.itemsize].view(dt).reshape(ntiles, slots, 64)
        a = v.transpose(1, 0, 2).reshape(slots, ntiles * 64)[:, : self.G]
        return a.copy() if name == "voter_term_out" else a[0].copy()

    def __del__(self):
        for l, plan in getattr(self, "_plans", []):
            l.rq_plan_destroy(plan)

    def bytes_resident(self):
        if self.tile_bytes:
            rest = [self.t.get("log_term"), self.o.get("elected_bits"), self.o.get("commit_lo16")]
            return self._tile_block.numel() + self._out_block.numel() + sum(x.numel() * x.element_size() for x in rest if x is not None)
        return sum(x.numel() * x.element_size() for x in list(self.t.values()) + list(self.o.values())
                   if x is not None)

    def results(self):
        """Outputs as numpy arrays with the C-ABI dtypes."""
        torch.cuda.synchronize(self.device)
        o = self.o
        if self.tile_bytes:
            blk = (self._tile_block.cpu().numpy().reshape(-1, self.tile_bytes),
                   self._out_block.cpu().numpy().reshape(-1, self.out_tile_bytes))
            return dict(
                commit=self._untile(blk, "commit_out"), voter_term=self._untile(blk, "voter_term_out"),
                grant_mask=self._untile(blk, "grant_mask").view(np.uint16),
                stepdown_mask=self._untile(blk, "stepdown_mask").view(np.uint16),
                elected=self._untile(blk, "elected"), max_term=self._untile(blk, "max_term"),
                status=self._untile(blk, "status"),
                elected_bits=o["elected_bits"].cpu().numpy().view(np.uint64),
                commit_lo16=None if o["commit_lo16"] is None else decode_summary(
                    o["commit_lo16"].cpu().numpy(), self.G, self.summary_bits),
                summary_escapes=None if o["summary_escapes"] is None else int(o["summary_escapes"].cpu()[0]))
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


class RoundsPlan:
    """An rq_plan over several images (DeviceEngine.plan_rounds): owns the plan handle and keeps
    the images it names alive for as long as it can be launched."""

    def __init__(self, l, handle, batches):
        self._l, self.handle, self.batches = l, handle, batches

    def __del__(self):
        if getattr(self, "handle", None):
            self._l.rq_plan_destroy(self.handle)
            self.handle = None


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

    def stream_sync(self):
        """Host-side wait for torch's current stream through rq_stream_sync (polls first)."""
        self._check(self._l.rq_stream_sync(self._h, self._stream()), "rq_stream_sync")

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

    def describe(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """The rq_round_desc of DeviceBatch b (both layouts, fixed and joint)."""
        t, o = b.t, b.o
        p = lambda x: None if x is None else x.data_ptr()
        commit, vote = bool(what & 1), bool(what & 2)
        d = RoundDesc()
        d.struct_size = C.sizeof(RoundDesc)
        d.layout = LAYOUT_I64 if b.wide else LAYOUT_I32
        d.what, d.variant = what, variant
        d.G, d.n, d.ld = b.G, b.n, b.ld
        d.tile_bytes, d.out_tile_bytes = b.tile_bytes, getattr(b, "out_tile_bytes", 0)
        d.log_term_bits, d.log_terms = b.log_term_bits, b.log_terms
        d.summary_bits = b.summary_bits
        d.old_mask, d.new_mask = p(t.get("old_mask")), p(t.get("new_mask"))
        if commit:
            d.match, d.commit, d.leader_term = p(t["match"]), p(t["commit"]), p(t.get("leader_term"))
            d.log_off, d.log_term = p(t.get("log_off")), p(t.get("log_term"))
            d.log_term_start = p(t.get("log_term_start"))
            d.log_base, d.log_term_base = p(t.get("log_base")), p(t.get("log_term_base"))
            d.log_len = p(t.get("log_len"))
            d.commit_out = p(o["commit_out"])
            d.commit_summary, d.summary_base = p(o["commit_lo16"]), p(b.summary_base)
            d.summary_escapes = p(o["summary_escapes"])
        if vote:
            d.voter_term, d.voted_for, d.voter_last_log_term = p(t["voter_term"]), p(t["voted_for"]), p(t["last_log_term"])
            d.cand_term, d.cand_id = p(t["cand_term"]), p(t["cand_id"])
            d.voter_term_out, d.grant_mask, d.stepdown_mask = p(o["voter_term_out"]), p(o["grant_mask"]), p(o["stepdown_mask"])
            d.elected, d.max_term = p(o["elected"]), p(o["max_term"])
            d.elected_bits = p(o["elected_bits"]) if want_bits else None
        d.status_out = p(o["status"]) if status else None
        d.changed_bits = p(o.get("changed_bits"))
        d.start_signal = b.start_signal
        segs = getattr(b, "segments", None)
        if segs is not None:
            first, kinds = segs
            d.n_segments = len(kinds)
            for i, v in enumerate(first):
                d.seg_first_group[i] = v
            for i, v in enumerate(kinds):
                d.seg_n[i] = v
        return d

    def prepare(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """The prepared launch (rq_plan handle) of the commit (what & 1) and/or vote (what & 2)
        decision for DeviceBatch b: the descriptor is validated once per (batch, options) and the
        plan kept with the batch.  A caller in a tight loop holds the handle and calls
        launch_prepared() -- two arguments per launch, as a C or Go host would."""
        # everything the descriptor bakes in besides the batch's own arrays: a field reassigned
        # after the first launch (a signal word, a summary buffer) gets a plan of its own
        o = b.o
        addr = lambda x: None if x is None else x.data_ptr()
        key = (what, variant, want_bits, status, b.start_signal, addr(o.get("changed_bits")), addr(o.get("commit_lo16")),
               addr(o.get("summary_escapes")), addr(b.summary_base), b.summary_bits, addr(o.get("elected_bits")),
               addr(o.get("commit_out")), None if b.segments is None else (tuple(b.segments[0]), tuple(b.segments[1])))
        plan = b._args.get(key)
        if plan is None:
            d = self.describe(b, what, variant, want_bits, status)
            plan = C.c_void_p()
            rc = self._l.rq_plan_round(self._h, C.byref(d), C.byref(plan))
            if rc != 0:
                self._check(rc, "rq_plan_round")
            b._args[key] = plan
            b._plans.append((self._l, plan))
        return plan

    def launch_prepared(self, plan, stream=None):
        """rq_plan_launch of a handle from prepare() on `stream` (a raw stream handle; None =
        torch's current stream)."""
        rc = self._l.rq_plan_launch(plan, torch.cuda.current_stream().cuda_stream if stream is None else stream)
        if rc != 0:
            self._check(rc, "rq_plan_launch")

    def round(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """Enqueue the decision for DeviceBatch b on torch's current stream (prepare() on first
        use, then a two-argument rq_plan_launch)."""
        self.launch_prepared(self.prepare(b, what, variant, want_bits, status))

    def plan_rounds(self, batches, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """ONE prepared launch over several DeviceBatch images (rq_plan_rounds): each keeps its
        own arrays, cluster size or member masks and tiling.  Returns a RoundsPlan for
        launch_plan(); it keeps the batches (and so their device buffers) alive."""
        descs = (RoundDesc * len(batches))()
        for i, b in enumerate(batches):
            descs[i] = self.describe(b, what, variant, want_bits, status)
        plan = C.c_void_p()
        rc = self._l.rq_plan_rounds(self._h, descs, len(batches), C.byref(plan))
        if rc != 0:
            self._check(rc, "rq_plan_rounds")
        return RoundsPlan(self._l, plan, list(batches))

    def launch_plan(self, plan):
        rc = self._l.rq_plan_launch(plan.handle, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            self._check(rc, "rq_plan_launch")

    def round_direct(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        """The same round through rq_round_dev (descriptor validated on every call)."""
        d = self.describe(b, what, variant, want_bits, status)
        rc = self._l.rq_round_dev(self._h, self._stream(), C.byref(d))
        if rc != 0:
            self._check(rc, "rq_round_dev")

    def fused_round(self, b, variant=VAR_DEFAULT, want_bits=True, status=True):
        return self.round(b, 3, variant, want_bits, status)

    def commit_advance(self, b, variant=VAR_DEFAULT):
        return self.round(b, 1, variant)

    def vote_tally(self, b, variant=VAR_DEFAULT, want_bits=True):
        return self.round(b, 2, variant, want_bits)

    # names kept for the wide-layout tests
    def round64(self, b, what=3, variant=VAR_DEFAULT, want_bits=True, status=True):
        assert b.wide
        return self.round(b, what, variant, want_bits, status)

    def append_check(self, t, G, n, ld, flog_term_bits=0, flog_terms=None, wide=None, commit_round=None):
        """Enqueue the AppendEntries consistency batch on device tensors
        (rq_append_check_dev; int32 or int64 planes).  t: dict of tensors with the C-ABI
        names; active_mask, n_new, match_index, next_index, status, flog_base and
        flog_term_base may be absent; leader_commit + last_new_index + follower_commit
        (all three or none) add the follower's commit clamp of raft.go:172-176.
        commit_round: a RoundDesc of the commit round over the same image (its match is
        match_index), decided in this launch.  follower_term_out may be the follower_term tensor
        itself (in-place terms); match_index / next_index are updated in place.
        flog_term_bits: entry width of the flog_term tensor (0 = the planes' element)."""
        wide = t["follower_term"].dtype == torch.int64 if wide is None else wide
        p = lambda k: t[k].data_ptr() if t.get(k) is not None else None
        d = AppendDesc()
        d.struct_size = C.sizeof(AppendDesc)
        d.layout = LAYOUT_I64 if wide else LAYOUT_I32
        d.G, d.n, d.ld = G, n, ld
        d.tile_bytes, d.out_tile_bytes = t.get("_tile_bytes", 0), t.get("_out_tile_bytes", 0)
        d.flog_term_bits = flog_term_bits
        d.flog_terms = t["flog_term"].numel() if flog_terms is None else flog_terms
        d.active_mask = p("active_mask")
        d.follower_term, d.leader_term = p("follower_term"), p("leader_term")
        d.prev_log_index, d.prev_log_term, d.n_new = p("prev_log_index"), p("prev_log_term"), p("n_new")
        d.flog_off, d.flog_term = p("flog_off"), p("flog_term")
        d.flog_base, d.flog_term_base = p("flog_base"), p("flog_term_base")
        d.follower_term_out = p("follower_term_out")
        d.success_mask, d.stepdown_mask = p("success_mask"), p("stepdown_mask")
        d.match_index_io, d.next_index_io = p("match_index"), p("next_index")
        d.status_out = p("status")
        d.leader_commit, d.last_new_index = p("leader_commit"), p("last_new_index")
        d.follower_commit_io = p("follower_commit")
        if commit_round is not None:        # a RoundDesc: the leader's commit rule decided in the same launch
            d.commit_round = C.addressof(commit_round)
        rc = self._l.rq_append_check_dev(self._h, self._stream(), C.byref(d))
        if rc != 0:
            self._check(rc, "rq_append_check_dev")
