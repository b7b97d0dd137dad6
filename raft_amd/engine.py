"""numpy-facing wrapper over the rq_* C-ABI (host-buffer entry points).

Mirrors the batched driver a Go host would write around cgo: flat int32
slot-major planes in, decision vectors out.  All arithmetic happens in
librq.so's HIP kernels; this file only checks shapes and marshals pointers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RqError, i32p, u8p, u16p, u32p

ST_LOG_OOB = 0x01
ST_EMPTY_LOG = 0x02
LOG_EMPTY = np.iinfo(np.int32).min
LOG_EMPTY64 = np.iinfo(np.int64).min
ROUND_COMMIT, ROUND_VOTE = 1, 2
MAX_SLOTS = 16


def _plane(a, name):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.ndim != 2:
        raise ValueError("%s must be a [slots][ld] int32 array" % name)
    return a


def _vec(a, dtype, name, G):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.ndim != 1 or a.shape[0] < G:
        raise ValueError("%s must be a 1-D %s array of at least %d" % (name, np.dtype(dtype).name, G))
    return a


def _p(a, t):
    return a.ctypes.data_as(t)


class VoteResult:
    __slots__ = ("voter_term", "grant_mask", "stepdown_mask", "elected", "max_term", "status")


class RoundResult(VoteResult):
    __slots__ = ("commit",)


class QuorumEngine:
    """One engine = one GPU, one stream (rq_engine_create .. rq_engine_destroy)."""

    def __init__(self, max_groups, max_slots=5, max_log_terms=None, device=0):
        self._l = _lib.lib()
        if max_log_terms is None:
            max_log_terms = 16 * max_groups
        h = C.c_void_p()
        rc = self._l.rq_engine_create(device, max_groups, max_slots, max_log_terms, C.byref(h))
        if rc != 0:
            raise RqError(rc, "rq_engine_create")
        self._h = h
        self.max_groups, self.max_slots, self.max_log_terms = max_groups, max_slots, max_log_terms

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self._l.rq_host_free(self._h, p)
            self._pinned = []
            self._l.rq_engine_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _check(self, rc, where):
        if rc != 0:
            raise RqError(rc, where, self._l.rq_last_hip_error(self._h))

    def pinned_like(self, array):
        """A copy of `array` in page-locked host memory obtained from rq_host_alloc
        (numpy view over it; freed with the engine)."""
        a = np.ascontiguousarray(array)
        p = C.c_void_p()
        self._check(self._l.rq_host_alloc(self._h, max(a.nbytes, 8), C.byref(p)), "rq_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_uint8 * max(a.nbytes, 8)).from_address(p.value)
        out = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
        out[...] = a
        return out

    # -- helpers ----------------------------------------------------------
    @staticmethod
    def _commit_in(match, commit, leader_term, log_off, log_term, G):
        match = _plane(match, "match")
        n, ld = match.shape
        G = ld if G is None else G
        commit = _vec(commit, np.int32, "commit", G)
        leader_term = _vec(leader_term, np.int32, "leader_term", G)
        log_off = _vec(log_off, np.uint32, "log_off", G + 1)
        log_term = np.ascontiguousarray(log_term, dtype=np.int32)
        if log_term.size == 0:
            log_term = np.zeros(1, np.int32)
        return match, n, ld, G, commit, leader_term, log_off, log_term

    @staticmethod
    def _vote_in(voter_term, voted_for, last_log_term, cand_term, cand_id, G):
        voter_term = _plane(voter_term, "voter_term")
        n, ld = voter_term.shape
        G = ld if G is None else G
        voted_for = _plane(voted_for, "voted_for")
        last_log_term = _plane(last_log_term, "voter_last_log_term")
        if voted_for.shape != (n, ld) or last_log_term.shape != (n, ld):
            raise ValueError("vote planes must share one [slots][ld] shape")
        cand_term = _vec(cand_term, np.int32, "cand_term", G)
        cand_id = _vec(cand_id, np.int32, "cand_id", G)
        return voter_term, voted_for, last_log_term, n, ld, G, cand_term, cand_id

    @staticmethod
    def _vote_out(n, ld, G, voter_term):
        r = {}
        r["voter_term"] = voter_term.copy()  # non-member / padding lanes keep inputs
        r["grant_mask"] = np.zeros(G, np.uint16)
        r["stepdown_mask"] = np.zeros(G, np.uint16)
        r["elected"] = np.zeros(G, np.uint8)
        r["max_term"] = np.zeros(G, np.int32)
        r["status"] = np.zeros(G, np.uint8)
        return r

    @staticmethod
    def _pack_vote(res, o):
        res.voter_term, res.grant_mask, res.stepdown_mask = o["voter_term"], o["grant_mask"], o["stepdown_mask"]
        res.elected, res.max_term, res.status = o["elected"], o["max_term"], o["status"]
        return res

    @staticmethod
    def _masks(old_mask, new_mask, G):
        return _vec(old_mask, np.uint16, "old_mask", G), _vec(new_mask, np.uint16, "new_mask", G)

    # -- entry points -------------------------------------------------------
    def commit_advance(self, match, commit, leader_term, log_off, log_term, G=None,
                       old_mask=None, new_mask=None):
        match, n, ld, G, commit, leader_term, log_off, log_term = self._commit_in(
            match, commit, leader_term, log_off, log_term, G)
        out = np.zeros(G, np.int32)
        st = np.zeros(G, np.uint8)
        tail = (_p(match, i32p), _p(commit, i32p), _p(leader_term, i32p), _p(log_off, u32p),
                _p(log_term, i32p), _p(out, i32p), _p(st, u8p))
        if old_mask is None:
            rc = self._l.rq_commit_advance(self._h, G, n, ld, *tail)
            self._check(rc, "rq_commit_advance")
        else:
            om, nm = self._masks(old_mask, new_mask, G)
            rc = self._l.rq_commit_advance_joint(self._h, G, n, ld, _p(om, u16p), _p(nm, u16p), *tail)
            self._check(rc, "rq_commit_advance_joint")
        return out, st

    def vote_tally(self, voter_term, voted_for, last_log_term, cand_term, cand_id, G=None,
                   old_mask=None, new_mask=None):
        vt, vf, ll, n, ld, G, ct, ci = self._vote_in(voter_term, voted_for, last_log_term,
                                                     cand_term, cand_id, G)
        o = self._vote_out(n, ld, G, vt)
        tail = (_p(vt, i32p), _p(vf, i32p), _p(ll, i32p), _p(ct, i32p), _p(ci, i32p),
                _p(o["voter_term"], i32p), _p(o["grant_mask"], u16p), _p(o["stepdown_mask"], u16p),
                _p(o["elected"], u8p), _p(o["max_term"], i32p), _p(o["status"], u8p))
        if old_mask is None:
            self._check(self._l.rq_vote_tally(self._h, G, n, ld, *tail), "rq_vote_tally")
        else:
            om, nm = self._masks(old_mask, new_mask, G)
            self._check(self._l.rq_vote_tally_joint(self._h, G, n, ld, _p(om, u16p), _p(nm, u16p), *tail),
                        "rq_vote_tally_joint")
        return self._pack_vote(VoteResult(), o)

    def round_buffers(self, n, ld, G=None, pinned=True):
        """Reusable output arrays for fused_round(out=...), by default in page-locked
        memory: copies into fresh pageable arrays cost more than the round itself
        (1M x 5: 6.0 ms per call against 3.6 ms with these)."""
        G = ld if G is None else G
        mk = self.pinned_like if pinned else (lambda a: a)
        r = RoundResult()
        r.commit = mk(np.zeros(G, np.int32)); r.voter_term = mk(np.zeros((n, ld), np.int32))
        r.grant_mask = mk(np.zeros(G, np.uint16)); r.stepdown_mask = mk(np.zeros(G, np.uint16))
        r.elected = mk(np.zeros(G, np.uint8)); r.max_term = mk(np.zeros(G, np.int32))
        r.status = mk(np.zeros(G, np.uint8))
        return r

    def fused_round(self, match, commit, leader_term, log_off, log_term,
                    voter_term, voted_for, last_log_term, cand_term, cand_id, G=None,
                    old_mask=None, new_mask=None, out=None):
        """out: a RoundResult from round_buffers() to write into (and return)
        instead of fresh arrays; padding columns of its voter_term are not touched."""
        match, n, ld, G, commit, leader_term, log_off, log_term = self._commit_in(
            match, commit, leader_term, log_off, log_term, G)
        vt, vf, ll, n2, ld2, _, ct, ci = self._vote_in(voter_term, voted_for, last_log_term,
                                                       cand_term, cand_id, G)
        if (n2, ld2) != (n, ld):
            raise ValueError("commit and vote planes must share one [slots][ld] shape")
        if out is not None:
            if out.voter_term.shape != (n, ld) or out.commit.shape[0] < G:
                raise ValueError("out does not fit this batch")
            o = dict(voter_term=out.voter_term, grant_mask=out.grant_mask, stepdown_mask=out.stepdown_mask,
                     elected=out.elected, max_term=out.max_term, status=out.status)
            cout = out.commit
        else:
            o = self._vote_out(n, ld, G, vt)
            cout = np.zeros(G, np.int32)
        tail = (_p(match, i32p), _p(commit, i32p), _p(leader_term, i32p), _p(log_off, u32p),
                _p(log_term, i32p),
                _p(vt, i32p), _p(vf, i32p), _p(ll, i32p), _p(ct, i32p), _p(ci, i32p),
                _p(cout, i32p), _p(o["voter_term"], i32p), _p(o["grant_mask"], u16p),
                _p(o["stepdown_mask"], u16p), _p(o["elected"], u8p), _p(o["max_term"], i32p),
                _p(o["status"], u8p))
        if old_mask is None:
            self._check(self._l.rq_fused_round(self._h, G, n, ld, *tail), "rq_fused_round")
        else:
            om, nm = self._masks(old_mask, new_mask, G)
            self._check(self._l.rq_fused_round_joint(self._h, G, n, ld, _p(om, u16p), _p(nm, u16p), *tail),
                        "rq_fused_round_joint")
        res = self._pack_vote(out if out is not None else RoundResult(), o)
        res.commit = cout
        return res

    def follower_clamp(self, leader_commit, commit, last_new_index, wide=False):
        """wide=True: int64 arrays through rq_follower_clamp64."""
        dt = np.int64 if wide else np.int32
        lc = np.ascontiguousarray(leader_commit, dtype=dt).ravel()
        ci = np.ascontiguousarray(commit, dtype=dt).ravel()
        ln = np.ascontiguousarray(last_new_index, dtype=dt).ravel()
        if not (lc.size == ci.size == ln.size):
            raise ValueError("follower_clamp arrays must have equal length")
        out = np.zeros(lc.size, dt)
        if wide:
            vp = lambda a: C.c_void_p(a.ctypes.data)
            self._check(self._l.rq_follower_clamp64(self._h, lc.size, vp(lc), vp(ci), vp(ln), vp(out)),
                        "rq_follower_clamp64")
            return out
        self._check(self._l.rq_follower_clamp(self._h, lc.size, _p(lc, i32p), _p(ci, i32p),
                                              _p(ln, i32p), _p(out, i32p)), "rq_follower_clamp")
        return out

    def append_check(self, follower_term, leader_term, prev_log_index, prev_log_term,
                     flog_off, flog_term, n_new=None, match_index=None, next_index=None,
                     active_mask=None, G=None, wide=False):
        """AppendEntries consistency batch.  Returns a dict with follower_term,
        success_mask, stepdown_mask, status and (if given) updated match_index /
        next_index copies.  wide=True: int64 planes through rq_append_check64."""
        if wide:
            return self._append_check64(follower_term, leader_term, prev_log_index, prev_log_term,
                                        flog_off, flog_term, n_new, match_index, next_index, active_mask, G)
        ft = _plane(follower_term, "follower_term")
        n, ld = ft.shape
        G = ld if G is None else G
        pi = _plane(prev_log_index, "prev_log_index"); pt = _plane(prev_log_term, "prev_log_term")
        lt = _vec(leader_term, np.int32, "leader_term", G)
        off = _vec(flog_off, np.uint32, "flog_off", n * ld + 1)
        arena = np.ascontiguousarray(flog_term, dtype=np.int32)
        if arena.size == 0:
            arena = np.zeros(1, np.int32)
        opt = lambda a, name: None if a is None else _plane(a, name).copy()
        nn, mi, ni = opt(n_new, "n_new"), opt(match_index, "match_index"), opt(next_index, "next_index")
        am = None if active_mask is None else _vec(active_mask, np.uint16, "active_mask", G)
        out = dict(follower_term=ft.copy(), success_mask=np.zeros(G, np.uint16),
                   stepdown_mask=np.zeros(G, np.uint16), status=np.zeros(G, np.uint8),
                   match_index=mi, next_index=ni)
        pp = lambda a, t: None if a is None else _p(a, t)
        rc = self._l.rq_append_check(self._h, G, n, ld, pp(am, u16p), _p(ft, i32p), _p(lt, i32p),
                                     _p(pi, i32p), _p(pt, i32p), pp(nn, i32p), _p(off, u32p),
                                     _p(arena, i32p), _p(out["follower_term"], i32p),
                                     _p(out["success_mask"], u16p), _p(out["stepdown_mask"], u16p),
                                     pp(mi, i32p), pp(ni, i32p), _p(out["status"], u8p))
        self._check(rc, "rq_append_check")
        return out

    def _append_check64(self, follower_term, leader_term, prev_log_index, prev_log_term,
                        flog_off, flog_term, n_new, match_index, next_index, active_mask, G):
        c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        ft = c64(follower_term)
        n, ld = ft.shape
        G = ld if G is None else G
        pi, pt, lt = c64(prev_log_index), c64(prev_log_term), _vec(leader_term, np.int64, "leader_term", G)
        off = _vec(flog_off, np.uint32, "flog_off", n * ld + 1)
        arena = c64(flog_term)
        if arena.size == 0:
            arena = np.zeros(1, np.int64)
        opt = lambda a: None if a is None else c64(a).copy()
        nn, mi, ni = opt(n_new), opt(match_index), opt(next_index)
        am = None if active_mask is None else _vec(active_mask, np.uint16, "active_mask", G)
        out = dict(follower_term=ft.copy(), success_mask=np.zeros(G, np.uint16),
                   stepdown_mask=np.zeros(G, np.uint16), status=np.zeros(G, np.uint8),
                   match_index=mi, next_index=ni)
        vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        rc = self._l.rq_append_check64(self._h, G, n, ld, vp(am), vp(ft), vp(lt), vp(pi), vp(pt), vp(nn),
                                       vp(off), vp(arena), vp(out["follower_term"]), vp(out["success_mask"]),
                                       vp(out["stepdown_mask"]), vp(mi), vp(ni), vp(out["status"]))
        self._check(rc, "rq_append_check64")
        return out

    # -- wide layout (int64 planes) -------------------------------------------
    def round64(self, what=ROUND_COMMIT | ROUND_VOTE, match=None, commit=None, leader_term=None,
                log_off=None, log_term=None, voter_term=None, voted_for=None, last_log_term=None,
                cand_term=None, cand_id=None, G=None, old_mask=None, new_mask=None, log_base=None):
        """rq_round64: the commit and/or vote decision over int64 planes.  Returns a
        RoundResult whose fields of the side that was not selected are None."""
        def plane(a, name):
            a = np.ascontiguousarray(a, dtype=np.int64)
            if a.ndim != 2:
                raise ValueError("%s must be a [slots][ld] int64 array" % name)
            return a
        first = match if what & ROUND_COMMIT else voter_term
        n, ld = plane(first, "planes").shape
        G = ld if G is None else G
        vp = C.c_void_p
        ptr = lambda a: None if a is None else vp(a.ctypes.data)
        keep = []

        def vec(a, dt, name, count):
            a = _vec(a, dt, name, count); keep.append(a); return a
        res = RoundResult()
        for f in RoundResult.__slots__ + VoteResult.__slots__:
            setattr(res, f, None)
        ins = [None] * 11
        outs = [None] * 7
        res.status = np.zeros(G, np.uint8)
        if what & ROUND_COMMIT:
            m = plane(match, "match"); keep.append(m)
            lt = np.ascontiguousarray(log_term, dtype=np.int64)
            if lt.size == 0:
                lt = np.zeros(1, np.int64)
            keep.append(lt)
            res.commit = np.zeros(G, np.int64)
            ins[0:6] = [m, vec(commit, np.int64, "commit", G), vec(leader_term, np.int64, "leader_term", G),
                        vec(log_off, np.uint32, "log_off", G + 1), lt,
                        None if log_base is None else vec(log_base, np.int64, "log_base", G)]
            outs[0] = res.commit
        if what & ROUND_VOTE:
            vt, vf, ll = plane(voter_term, "voter_term"), plane(voted_for, "voted_for"), plane(last_log_term, "voter_last_log_term")
            if not (vt.shape == vf.shape == ll.shape == (n, ld)):
                raise ValueError("all planes must share one [slots][ld] shape")
            keep += [vt, vf, ll]
            ins[6:11] = [vt, vf, ll, vec(cand_term, np.int64, "cand_term", G), vec(cand_id, np.int64, "cand_id", G)]
            res.voter_term = vt.copy()
            res.grant_mask = np.zeros(G, np.uint16); res.stepdown_mask = np.zeros(G, np.uint16)
            res.elected = np.zeros(G, np.uint8); res.max_term = np.zeros(G, np.int64)
            outs[1:6] = [res.voter_term, res.grant_mask, res.stepdown_mask, res.elected, res.max_term]
        outs[6] = res.status
        om = None if old_mask is None else _vec(old_mask, np.uint16, "old_mask", G)
        nm = None if new_mask is None else _vec(new_mask, np.uint16, "new_mask", G)
        rc = self._l.rq_round64(self._h, what, G, n, ld, ptr(om), ptr(nm), *[ptr(a) for a in ins],
                                *[ptr(a) for a in outs])
        self._check(rc, "rq_round64")
        return res

    # -- device-resident trio ----------------------------------------------
    def upload(self, match, commit, leader_term, log_off, log_term,
               voter_term, voted_for, last_log_term, cand_term, cand_id, G=None):
        match, n, ld, G, commit, leader_term, log_off, log_term = self._commit_in(
            match, commit, leader_term, log_off, log_term, G)
        vt, vf, ll, n2, ld2, _, ct, ci = self._vote_in(voter_term, voted_for, last_log_term,
                                                       cand_term, cand_id, G)
        if (n2, ld2) != (n, ld):
            raise ValueError("commit and vote planes must share one [slots][ld] shape")
        self._check(self._l.rq_upload(self._h, G, n, ld, _p(match, i32p), _p(commit, i32p),
                                      _p(leader_term, i32p), _p(log_off, u32p), _p(log_term, i32p),
                                      _p(vt, i32p), _p(vf, i32p), _p(ll, i32p), _p(ct, i32p),
                                      _p(ci, i32p)), "rq_upload")
        self._res = (G, n)

    def step(self, rounds=1):
        self._check(self._l.rq_step(self._h, rounds), "rq_step")

    def download(self):
        G, n = self._res
        res = RoundResult()
        res.commit = np.zeros(G, np.int32)
        res.voter_term = np.zeros((n, G), np.int32)
        res.grant_mask = np.zeros(G, np.uint16)
        res.stepdown_mask = np.zeros(G, np.uint16)
        res.elected = np.zeros(G, np.uint8)
        res.max_term = np.zeros(G, np.int32)
        res.status = np.zeros(G, np.uint8)
        self._check(self._l.rq_download(self._h, _p(res.commit, i32p), _p(res.voter_term, i32p),
                                        _p(res.grant_mask, u16p), _p(res.stepdown_mask, u16p),
                                        _p(res.elected, u8p), _p(res.max_term, i32p),
                                        _p(res.status, u8p)), "rq_download")
        return res
