"""ctypes binding of oracle/librq_oracle.so -- the CPU restatement of raft.go.

Test infrastructure only: tests, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this; nothing under raft_amd/ does.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "librq_oracle.so")

i32p = C.POINTER(C.c_int32); u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16); u8p = C.POINTER(C.c_uint8)
i64p = C.POINTER(C.c_int64)
u32 = C.c_uint32


class Entry(C.Structure):           # rf_entry  (raft.go:71-75)
    _fields_ = [("command", C.c_char_p), ("index", C.c_int64), ("term_num", C.c_int64)]


class Node(C.Structure):            # rf_node   (raft.go:15-69)
    pass


Node._fields_ = [
    ("id", C.c_int64), ("node_type", C.c_int64), ("state_machine", C.c_void_p),
    ("peers", C.POINTER(C.POINTER(Node))), ("n_peers", C.c_int64),
    ("current_term", C.c_int64), ("voted_for", C.c_int64),
    ("log", C.POINTER(Entry)), ("log_len", C.c_int64), ("log_cap", C.c_int64),
    ("commit_index", C.c_int64), ("last_applied", C.c_int64),
    ("next_index", i64p), ("match_index", i64p),
]
NodeP = C.POINTER(Node)
LEADER, FOLLOWER, CANDIDATE = 0, 1, 2
OK, PANIC_INDEX = 0, 1
ST_LOG_OOB, ST_EMPTY_LOG = 1, 2

_l = None


def lib():
    global _l
    if _l is None:
        if not os.path.exists(PATH):
            raise ImportError("oracle/librq_oracle.so not built; run `make`")
        l = C.CDLL(PATH)
        l.rf_new_node.argtypes = [C.c_int64, C.POINTER(NodeP), C.c_int64, C.c_void_p]
        l.rf_new_node.restype = NodeP
        l.rf_free_cluster.argtypes = [NodeP]; l.rf_free_cluster.restype = None
        for f in ("rf_become_leader", "rf_become_follower", "rf_become_candidate"):
            getattr(l, f).argtypes = [NodeP]; getattr(l, f).restype = None
        l.rf_append_entries_rpc.argtypes = [NodeP, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                            C.POINTER(Entry), C.c_int64, C.c_int64, i64p,
                                            C.POINTER(C.c_int)]
        l.rf_request_vote_rpc.argtypes = [NodeP, C.c_int64, C.c_int64, C.c_int64, C.c_int64, i64p,
                                          C.POINTER(C.c_int)]
        l.rf_test_to_abdicate_leadership.argtypes = [NodeP, C.c_int64]
        l.rf_test_to_abdicate_leadership.restype = None
        l.rf_min_int.argtypes = [C.c_int64, C.c_int64]; l.rf_min_int.restype = C.c_int64
        l.rf_last_entry.argtypes = [C.POINTER(Entry), C.c_int64, C.POINTER(Entry)]
        l.rf_log_append.argtypes = [NodeP, C.POINTER(Entry), C.c_int64]; l.rf_log_append.restype = None
        l.rf_quorum_match_index.argtypes = [i64p, C.c_int64]; l.rf_quorum_match_index.restype = C.c_int64
        l.rf_leader_advance_commit.argtypes = [NodeP]
        l.orc_commit_advance.argtypes = [u32, u32, u32, i32p, i32p, i32p, u32p, i32p, i32p, u8p]
        l.orc_vote_tally.argtypes = [u32, u32, u32, i32p, i32p, i32p, i32p, i32p, i32p, u16p, u16p,
                                     u8p, i32p, u8p]
        l.orc_commit_advance_joint.argtypes = [u32, u32, u32, u16p, u16p, i32p, i32p, i32p, u32p,
                                               i32p, i32p, u8p]
        l.orc_vote_tally_joint.argtypes = [u32, u32, u32, u16p, u16p, i32p, i32p, i32p, i32p, i32p,
                                           i32p, u16p, u16p, u8p, i32p, u8p]
        l.orc_follower_clamp.argtypes = [C.c_uint64, i32p, i32p, i32p, i32p]
        l.orc_append_check.argtypes = [u32, u32, u32, u16p, i32p, i32p, i32p, i32p, i32p, u32p, i32p,
                                       i32p, u16p, u16p, i32p, i32p, u8p]
        l.orc_follower_clamp64.argtypes = [C.c_uint64] + [C.c_void_p] * 4
        l.orc_append_commit.argtypes = [u32, u32, u32] + [C.c_void_p] * 6
        l.orc_append_commit64.argtypes = [u32, u32, u32] + [C.c_void_p] * 6
        l.orc_append_check64.argtypes = [u32, u32, u32] + [C.c_void_p] * 14
        l.orc_round64.argtypes = [C.c_int, u32, u32, u32] + [C.c_void_p] * 20
        l.orc_bench_build.argtypes = [u32, u32, u32, i32p, i32p, i32p, u32p, i32p, i32p, i32p, i32p,
                                      i32p, i32p]
        l.orc_bench_build.restype = C.c_void_p
        l.orc_bench_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        l.orc_bench_run.restype = C.c_double
        l.orc_bench_free.argtypes = [C.c_void_p]; l.orc_bench_free.restype = None
        _l = l
    return _l


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a, t):
    return a.ctypes.data_as(t)


class Result(dict):
    __getattr__ = dict.__getitem__


def commit_advance(match, commit, leader_term, log_off, log_term, G=None, old_mask=None, new_mask=None,
                   log_base=None):
    """log_base: the arena is a window (slot i of a group holds log[base + i]).  The
    reference-shaped leader is then evaluated in the window's own coordinates:
    matchIndex and commitIndex shifted down by base (the quorum index and the
    N > commitIndex test are shift-invariant), the result shifted back."""
    if log_base is not None:
        base = _c(log_base, np.int32).astype(np.int64)
        m64 = _c(match, np.int32).astype(np.int64)
        m64[:, : base.size] -= base[None, :]              # planes may be wider (ld > G)
        c64 = _c(commit, np.int32).astype(np.int64)
        c64[: base.size] -= base[: c64.size]
        assert np.abs(m64).max(initial=0) < 2 ** 31 and np.abs(c64).max(initial=0) < 2 ** 31
        out, st = commit_advance(m64.astype(np.int32), c64.astype(np.int32), leader_term, log_off, log_term,
                                 G, old_mask, new_mask)
        return (out.astype(np.int64) + base[: out.size]).astype(np.int32), st
    match = _c(match, np.int32); n, ld = match.shape; G = ld if G is None else G
    commit = _c(commit, np.int32); leader_term = _c(leader_term, np.int32)
    log_off = _c(log_off, np.uint32); log_term = _c(log_term, np.int32)
    if log_term.size == 0:
        log_term = np.zeros(1, np.int32)
    out = np.zeros(G, np.int32); st = np.zeros(G, np.uint8)
    tail = (_p(match, i32p), _p(commit, i32p), _p(leader_term, i32p), _p(log_off, u32p),
            _p(log_term, i32p), _p(out, i32p), _p(st, u8p))
    if old_mask is None:
        rc = lib().orc_commit_advance(G, n, ld, *tail)
    else:
        om = _c(old_mask, np.uint16); nm = _c(new_mask, np.uint16)
        rc = lib().orc_commit_advance_joint(G, n, ld, _p(om, u16p), _p(nm, u16p), *tail)
    assert rc == 0, rc
    return out, st


def vote_tally(voter_term, voted_for, last_log_term, cand_term, cand_id, G=None, old_mask=None, new_mask=None):
    vt = _c(voter_term, np.int32); n, ld = vt.shape; G = ld if G is None else G
    vf = _c(voted_for, np.int32); ll = _c(last_log_term, np.int32)
    ct = _c(cand_term, np.int32); ci = _c(cand_id, np.int32)
    r = Result(voter_term=vt.copy(), grant_mask=np.zeros(G, np.uint16), stepdown_mask=np.zeros(G, np.uint16),
               elected=np.zeros(G, np.uint8), max_term=np.zeros(G, np.int32), status=np.zeros(G, np.uint8))
    tail = (_p(vt, i32p), _p(vf, i32p), _p(ll, i32p), _p(ct, i32p), _p(ci, i32p),
            _p(r.voter_term, i32p), _p(r.grant_mask, u16p), _p(r.stepdown_mask, u16p),
            _p(r.elected, u8p), _p(r.max_term, i32p), _p(r.status, u8p))
    if old_mask is None:
        rc = lib().orc_vote_tally(G, n, ld, *tail)
    else:
        om = _c(old_mask, np.uint16); nm = _c(new_mask, np.uint16)
        rc = lib().orc_vote_tally_joint(G, n, ld, _p(om, u16p), _p(nm, u16p), *tail)
    assert rc == 0, rc
    return r


def fused_round(match, commit, leader_term, log_off, log_term, voter_term, voted_for, last_log_term,
                cand_term, cand_id, G=None, old_mask=None, new_mask=None, log_base=None):
    """The oracle has no fused form: a fused round IS the two decisions side by side."""
    c, st_c = commit_advance(match, commit, leader_term, log_off, log_term, G, old_mask, new_mask, log_base)
    r = vote_tally(voter_term, voted_for, last_log_term, cand_term, cand_id, G, old_mask, new_mask)
    r["commit"] = c
    r["status"] = r["status"] | st_c
    return r


def round64(what, match=None, commit=None, leader_term=None, log_off=None, log_term=None,
            voter_term=None, voted_for=None, last_log_term=None, cand_term=None, cand_id=None,
            G=None, old_mask=None, new_mask=None, log_base=None):
    """orc_round64: the wide-layout decisions (what: 1 = commit, 2 = vote, 3 = both)."""
    c64 = lambda a: None if a is None else _c(a, np.int64)
    first = c64(match if what & 1 else voter_term)
    n, ld = first.shape; G = ld if G is None else G
    vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    lt = c64(log_term)
    if lt is not None and lt.size == 0:
        lt = np.zeros(1, np.int64)
    ins = [c64(match), c64(commit), c64(leader_term), None if log_off is None else _c(log_off, np.uint32),
           lt, c64(log_base), c64(voter_term), c64(voted_for), c64(last_log_term), c64(cand_term),
           c64(cand_id)]
    r = Result(commit=None, voter_term=None, grant_mask=None, stepdown_mask=None, elected=None,
               max_term=None, status=np.zeros(G, np.uint8))
    if what & 1:
        r["commit"] = np.zeros(G, np.int64)
    if what & 2:
        r["voter_term"] = ins[6].copy()
        r["grant_mask"] = np.zeros(G, np.uint16); r["stepdown_mask"] = np.zeros(G, np.uint16)
        r["elected"] = np.zeros(G, np.uint8); r["max_term"] = np.zeros(G, np.int64)
    om = None if old_mask is None else _c(old_mask, np.uint16)
    nm = None if new_mask is None else _c(new_mask, np.uint16)
    outs = [r.commit, r.voter_term, r.grant_mask, r.stepdown_mask, r.elected, r.max_term, r.status]
    rc = lib().orc_round64(what, G, n, ld, vp(om), vp(nm), *[vp(a) for a in ins], *[vp(a) for a in outs])
    assert rc == 0, rc
    return r


def follower_clamp(leader_commit, commit, last_new_index):
    lc = _c(leader_commit, np.int32).ravel(); ci = _c(commit, np.int32).ravel()
    ln = _c(last_new_index, np.int32).ravel()
    out = np.zeros(lc.size, np.int32)
    assert lib().orc_follower_clamp(lc.size, _p(lc, i32p), _p(ci, i32p), _p(ln, i32p), _p(out, i32p)) == 0
    return out


def append_check(follower_term, leader_term, prev_log_index, prev_log_term, flog_off, flog_term,
                 n_new=None, match_index=None, next_index=None, active_mask=None, G=None):
    ft = _c(follower_term, np.int32); n, ld = ft.shape; G = ld if G is None else G
    pi = _c(prev_log_index, np.int32); pt = _c(prev_log_term, np.int32); lt = _c(leader_term, np.int32)
    off = _c(flog_off, np.uint32); arena = _c(flog_term, np.int32)
    if arena.size == 0:
        arena = np.zeros(1, np.int32)
    opt = lambda a: None if a is None else _c(a, np.int32).copy()
    nn, mi, ni = opt(n_new), opt(match_index), opt(next_index)
    am = None if active_mask is None else _c(active_mask, np.uint16)
    out = Result(follower_term=ft.copy(), success_mask=np.zeros(G, np.uint16),
                 stepdown_mask=np.zeros(G, np.uint16), status=np.zeros(G, np.uint8),
                 match_index=mi, next_index=ni)
    pp = lambda a, t: None if a is None else _p(a, t)
    rc = lib().orc_append_check(G, n, ld, pp(am, u16p), _p(ft, i32p), _p(lt, i32p), _p(pi, i32p),
                                _p(pt, i32p), pp(nn, i32p), _p(off, u32p), _p(arena, i32p),
                                _p(out.follower_term, i32p), _p(out.success_mask, u16p),
                                _p(out.stepdown_mask, u16p), pp(mi, i32p), pp(ni, i32p),
                                _p(out.status, u8p))
    assert rc == 0, rc
    return out


def append_commit(res, leader_commit, last_new_index, follower_commit, n_new=None, G=None):
    """raft.go:172-176 for the succeeding slots of an append_check result `res`: returns
    the followers' commitIndex planes afterwards and ORs the panic bit into res.status.
    Element type from follower_commit (int32 or int64)."""
    dt = np.asarray(follower_commit).dtype
    fc = _c(follower_commit, dt).copy(); n, ld = fc.shape; G = ld if G is None else G
    lc, li = _c(leader_commit, dt), _c(last_new_index, dt)
    nn = None if n_new is None else _c(n_new, dt)
    vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    fn = lib().orc_append_commit64 if dt == np.int64 else lib().orc_append_commit
    assert fn(G, n, ld, vp(res.success_mask), vp(lc), vp(li), vp(nn), vp(fc), vp(res.status)) == 0
    return fc


def follower_clamp64(leader_commit, commit, last_new_index):
    lc = _c(leader_commit, np.int64).ravel(); ci = _c(commit, np.int64).ravel()
    ln = _c(last_new_index, np.int64).ravel()
    out = np.zeros(lc.size, np.int64)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    assert lib().orc_follower_clamp64(lc.size, vp(lc), vp(ci), vp(ln), vp(out)) == 0
    return out


def append_check64(follower_term, leader_term, prev_log_index, prev_log_term, flog_off, flog_term,
                   n_new=None, match_index=None, next_index=None, active_mask=None, G=None):
    ft = _c(follower_term, np.int64); n, ld = ft.shape; G = ld if G is None else G
    pi = _c(prev_log_index, np.int64); pt = _c(prev_log_term, np.int64); lt = _c(leader_term, np.int64)
    off = _c(flog_off, np.uint32); arena = _c(flog_term, np.int64)
    if arena.size == 0:
        arena = np.zeros(1, np.int64)
    opt = lambda a: None if a is None else _c(a, np.int64).copy()
    nn, mi, ni = opt(n_new), opt(match_index), opt(next_index)
    am = None if active_mask is None else _c(active_mask, np.uint16)
    out = Result(follower_term=ft.copy(), success_mask=np.zeros(G, np.uint16),
                 stepdown_mask=np.zeros(G, np.uint16), status=np.zeros(G, np.uint8),
                 match_index=mi, next_index=ni)
    vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    rc = lib().orc_append_check64(G, n, ld, vp(am), vp(ft), vp(lt), vp(pi), vp(pt), vp(nn), vp(off), vp(arena),
                                  vp(out.follower_term), vp(out.success_mask), vp(out.stepdown_mask),
                                  vp(mi), vp(ni), vp(out.status))
    assert rc == 0, rc
    return out


def cpu_baseline(batch, passes, threads):
    """Times the reference-shaped heap-object path (one RequestVoteRPC-shaped
    call per voter, one sort per leader) on `threads` host threads.
    Returns (seconds, checksum)."""
    b = batch
    arrs = [_c(b["match"], np.int32), _c(b["commit"], np.int32), _c(b["leader_term"], np.int32),
            _c(b["log_off"], np.uint32), _c(b["log_term"], np.int32), _c(b["voter_term"], np.int32),
            _c(b["voted_for"], np.int32), _c(b["last_log_term"], np.int32), _c(b["cand_term"], np.int32),
            _c(b["cand_id"], np.int32)]
    types = [i32p, i32p, i32p, u32p, i32p, i32p, i32p, i32p, i32p, i32p]
    n, ld = arrs[0].shape
    h = lib().orc_bench_build(b["G"], n, ld, *[_p(a, t) for a, t in zip(arrs, types)])
    try:
        cs = C.c_uint64(0)
        sec = lib().orc_bench_run(h, passes, threads, C.byref(cs))
    finally:
        lib().orc_bench_free(h)
    return sec, cs.value
