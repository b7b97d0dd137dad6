"""ctypes binding of raft_amd/libraft_host.so: the C++ mirror of the
reference's Go surface (Node, NewNode, Become*, *RPC) and the BatchedDriver."""
import ctypes as C
import os

import numpy as np

from raft_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "raft_amd", "libraft_host.so")
i64p = C.POINTER(C.c_int64); i32p = C.POINTER(C.c_int32); u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16); u8p = C.POINTER(C.c_uint8)
LEADER, FOLLOWER, CANDIDATE = 0, 1, 2
_l = None


def lib():
    global _l
    if _l is None:
        _lib.lib()                      # librq.so (and the HIP runtime) first
        l = C.CDLL(PATH)
        l.rh_cluster_new.restype = C.c_void_p
        l.rh_cluster_new.argtypes = [C.c_uint32, C.c_uint32]
        l.rh_cluster_new_sizes.restype = C.c_void_p
        l.rh_cluster_new_sizes.argtypes = [C.c_uint32, C.c_uint32, u32p]
        l.rh_driver_member_masks.argtypes = [C.c_void_p, u16p]
        l.rh_cluster_free.argtypes = [C.c_void_p]; l.rh_cluster_free.restype = None
        l.rh_node_seed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_int64] * 4
        l.rh_node_seed.restype = None
        l.rh_node_seed_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, i64p, C.c_int64]
        l.rh_node_seed_log.restype = None
        l.rh_node_get.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, i64p]; l.rh_node_get.restype = None
        l.rh_node_get_leader_arrays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, i64p, i64p]
        l.rh_node_become.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]; l.rh_node_become.restype = None
        l.rh_leader_set_match.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, i64p]; l.rh_leader_set_match.restype = None
        l.rh_group_roles.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int]; l.rh_group_roles.restype = None
        l.rh_group_get_roles.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.rh_group_get_roles.restype = None
        l.rh_request_vote.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_int64] * 4 + [i64p, C.POINTER(C.c_int)]
        l.rh_append_entries.argtypes = ([C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_int64] * 4 +
                                        [i64p, i64p, C.c_int64, C.c_int64, i64p, C.POINTER(C.c_int)])
        l.rh_driver_flatten.argtypes = [C.c_void_p, i32p, i32p, i32p, u32p, i32p, C.c_uint64,
                                        i32p, i32p, i32p, i32p, i32p, C.POINTER(C.c_uint64)]
        l.rh_driver_apply.argtypes = [C.c_void_p, i32p, i32p, u16p, u16p, u8p, i32p]
        l.rh_driver_apply.restype = None
        l.rh_driver_step.argtypes = [C.c_void_p, C.c_int]
        l.rh_driver_last_step_wide.argtypes = [C.c_void_p]
        l.rh_driver_flatten64.argtypes = [C.c_void_p, i64p, i64p, i64p, u32p, i64p, C.c_uint64,
                                          i64p, i64p, i64p, i64p, i64p, C.POINTER(C.c_uint64)]
        l.rh_driver_flagged.argtypes = [C.c_void_p, u32p, u8p, C.c_uint32]
        l.rh_driver_flagged.restype = C.c_uint32
        l.rh_driver_throw_if_flagged.argtypes = [C.c_void_p]
        l.rh_resident_new.argtypes = [C.c_void_p, C.c_int]
        l.rh_resident_roles.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int]; l.rh_resident_roles.restype = None
        l.rh_resident_get_roles.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.rh_resident_get_roles.restype = None
        l.rh_resident_mark_dirty.argtypes = [C.c_void_p, C.c_uint32]; l.rh_resident_mark_dirty.restype = None
        l.rh_resident_step.argtypes = [C.c_void_p, u32p]
        l.rh_resident_without_arena.argtypes = [C.c_void_p]
        l.rh_resident_segments.argtypes = [C.c_void_p]
        l.rh_resident_wide.argtypes = [C.c_void_p]
        l.rh_resident_flagged.argtypes = [C.c_void_p, u32p, u8p, C.c_uint32]
        l.rh_resident_flagged.restype = C.c_uint32
        _l = l
    return _l


FIELDS = ("id", "nodeType", "currentTerm", "votedFor", "logLen", "commitIndex", "lastApplied", "nPeers",
          "hasLeaderArrays")


class Cluster:
    def __init__(self, G, n, sizes=None):
        """sizes: members per group (1..n each); None = every group has n members."""
        self.l = lib(); self.G, self.n = G, n
        if sizes is None:
            self.h = C.c_void_p(self.l.rh_cluster_new(G, n))
        else:
            a = np.ascontiguousarray(sizes, dtype=np.uint32)
            assert a.size == G
            self.h = C.c_void_p(self.l.rh_cluster_new_sizes(G, n, a.ctypes.data_as(u32p)))

    resident = False

    def close(self):
        if self.h:
            self.l.rh_cluster_free(self.h); self.h = None

    def use_resident(self, device=0):
        """From here on roles / get_roles / step go to a ResidentDriver made from the
        batched driver's current groups, and every helper that changes a Node marks
        its group dirty (what a host does after applying an RPC)."""
        assert self.l.rh_resident_new(self.h, device) == 0
        self.resident = True

    def _touch(self, g):
        if self.resident:
            self.l.rh_resident_mark_dirty(self.h, g)

    def step_resident(self):
        rep = (C.c_uint32 * 3)()
        rc = self.l.rh_resident_step(self.h, rep)
        return rc, dict(uploaded=rep[0], changed=rep[1], rebuilt=bool(rep[2]))

    def segments(self):
        """Segments of the resident launch's size table (0: none)."""
        return int(self.l.rh_resident_segments(self.h))

    def without_arena(self):
        """True while the resident image keeps no log arena (log_term_start + log_len)."""
        return bool(self.l.rh_resident_without_arena(self.h))

    def resident_wide(self):
        """True once the resident image holds int64 elements (some value did not fit int32)."""
        return bool(self.l.rh_resident_wide(self.h))

    def flagged(self):
        """(group, status) pairs of the last step: conditions that are Go panics in the reference."""
        fn = self.l.rh_resident_flagged if self.resident else self.l.rh_driver_flagged
        n = fn(self.h, None, None, 0)
        g = np.zeros(max(n, 1), np.uint32); st = np.zeros(max(n, 1), np.uint8)
        fn(self.h, g.ctypes.data_as(u32p), st.ctypes.data_as(u8p), n)
        return list(zip(g[:n].tolist(), st[:n].tolist()))

    def throw_if_flagged(self):
        return bool(self.l.rh_driver_throw_if_flagged(self.h))

    def seed(self, g, p, term, voted_for, role, commit=0, log_terms=None):
        self._touch(g)
        self.l.rh_node_seed(self.h, g, p, term, voted_for, role, commit)
        if log_terms is not None:
            a = (C.c_int64 * max(1, len(log_terms)))(*log_terms)
            self.l.rh_node_seed_log(self.h, g, p, a, len(log_terms))

    def get(self, g, p):
        out = (C.c_int64 * 9)()
        self.l.rh_node_get(self.h, g, p, out)
        return dict(zip(FIELDS, list(out)))

    def leader_arrays(self, g, p):
        nx = (C.c_int64 * self.n)(); mt = (C.c_int64 * self.n)()
        if not self.l.rh_node_get_leader_arrays(self.h, g, p, nx, mt):
            return None
        return list(nx), list(mt)

    def become(self, g, p, role):
        self._touch(g)
        self.l.rh_node_become(self.h, g, p, role)

    def set_match(self, g, p, match):
        self._touch(g)
        self.l.rh_leader_set_match(self.h, g, p, (C.c_int64 * self.n)(*match))

    def roles(self, g, leader=-1, candidate=-1):
        self.l.rh_group_roles(self.h, g, leader, candidate)
        if self.resident:
            self.l.rh_resident_roles(self.h, g, leader, candidate)

    def get_roles(self, g):
        a, b = C.c_int(), C.c_int()
        fn = self.l.rh_resident_get_roles if self.resident else self.l.rh_group_get_roles
        fn(self.h, g, C.byref(a), C.byref(b))
        return a.value, b.value

    def request_vote(self, g, p, term, cand, lli=0, llt=0):
        tr = C.c_int64(); gr = C.c_int()
        self._touch(g)
        rc = self.l.rh_request_vote(self.h, g, p, term, cand, lli, llt, C.byref(tr), C.byref(gr))
        return rc, tr.value, bool(gr.value)

    def append_entries(self, g, p, term, leader, prev_index, prev_term, new, leader_commit):
        ni = (C.c_int64 * max(1, len(new)))(*[e[0] for e in new])
        nt = (C.c_int64 * max(1, len(new)))(*[e[1] for e in new])
        tr = C.c_int64(); ok = C.c_int()
        self._touch(g)
        rc = self.l.rh_append_entries(self.h, g, p, term, leader, prev_index, prev_term, ni, nt, len(new),
                                      leader_commit, C.byref(tr), C.byref(ok))
        return rc, tr.value, bool(ok.value)

    def flatten(self, log_cap=1 << 20):
        G, n = self.G, self.n
        plane = lambda: np.zeros((n, G), np.int32)
        vec = lambda: np.zeros(G, np.int32)
        b = dict(match=plane(), commit=vec(), leader_term=vec(), log_off=np.zeros(G + 1, np.uint32),
                 log_term=np.zeros(log_cap, np.int32), voter_term=plane(), voted_for=plane(),
                 last_log_term=plane(), cand_term=vec(), cand_id=vec())
        used = C.c_uint64()
        p = lambda a, t: a.ctypes.data_as(t)
        rc = self.l.rh_driver_flatten(self.h, p(b["match"], i32p), p(b["commit"], i32p), p(b["leader_term"], i32p),
                                      p(b["log_off"], u32p), p(b["log_term"], i32p), log_cap,
                                      p(b["voter_term"], i32p), p(b["voted_for"], i32p),
                                      p(b["last_log_term"], i32p), p(b["cand_term"], i32p),
                                      p(b["cand_id"], i32p), C.byref(used))
        b["log_term"] = b["log_term"][: used.value]
        if rc == 0:
            mask = np.zeros(G, np.uint16)
            b["mixed"] = bool(self.l.rh_driver_member_masks(self.h, mask.ctypes.data_as(u16p)))
            b["member_mask"] = mask
        return rc, b

    def flatten64(self, log_cap=1 << 20):
        """The wide-layout image of the objects (int64 planes)."""
        G, n = self.G, self.n
        plane = lambda: np.zeros((n, G), np.int64)
        vec = lambda: np.zeros(G, np.int64)
        b = dict(match=plane(), commit=vec(), leader_term=vec(), log_off=np.zeros(G + 1, np.uint32),
                 log_term=np.zeros(log_cap, np.int64), voter_term=plane(), voted_for=plane(),
                 last_log_term=plane(), cand_term=vec(), cand_id=vec())
        used = C.c_uint64()
        p = lambda a, t: a.ctypes.data_as(t)
        rc = self.l.rh_driver_flatten64(self.h, p(b["match"], i64p), p(b["commit"], i64p), p(b["leader_term"], i64p),
                                        p(b["log_off"], u32p), p(b["log_term"], i64p), log_cap,
                                        p(b["voter_term"], i64p), p(b["voted_for"], i64p),
                                        p(b["last_log_term"], i64p), p(b["cand_term"], i64p),
                                        p(b["cand_id"], i64p), C.byref(used))
        b["log_term"] = b["log_term"][: used.value]
        return rc, b

    def last_step_wide(self):
        return bool(self.l.rh_driver_last_step_wide(self.h))

    def apply(self, r):
        p = lambda a, dt, t: np.ascontiguousarray(a, dtype=dt).ctypes.data_as(t)
        keep = [np.ascontiguousarray(r["commit"], np.int32), np.ascontiguousarray(r["voter_term"], np.int32),
                np.ascontiguousarray(r["grant_mask"], np.uint16), np.ascontiguousarray(r["stepdown_mask"], np.uint16),
                np.ascontiguousarray(r["elected"], np.uint8), np.ascontiguousarray(r["max_term"], np.int32)]
        self.l.rh_driver_apply(self.h, keep[0].ctypes.data_as(i32p), keep[1].ctypes.data_as(i32p),
                               keep[2].ctypes.data_as(u16p), keep[3].ctypes.data_as(u16p),
                               keep[4].ctypes.data_as(u8p), keep[5].ctypes.data_as(i32p))

    def step(self, device=0):
        return self.l.rh_driver_step(self.h, device)
