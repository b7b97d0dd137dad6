"""CPU tests: the oracle (C restatement of raft.go) against the hand-derived
known-answer vectors in tests/golden/kats.json and against brute force.

PARITY UNPINNED by the reference itself: tawawhite/raft has no tests or
fixtures; every vector cites the raft.go lines it was derived from.
"""
import ctypes as C
import itertools
import json
import os

import numpy as np
import pytest

import oracle_lib as o

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))


class Cluster:
    """Reference-shaped nodes built with rf_new_node (raft.go:77-99)."""

    def __init__(self, n):
        self.l = o.lib()
        self.nodes = []
        arr = (o.NodeP * 0)()
        for i in range(n):
            arr = (o.NodeP * len(self.nodes))(*self.nodes)
            self.nodes.append(self.l.rf_new_node(i, arr, len(self.nodes), None))

    def seed_log(self, i, terms):
        for k, t in enumerate(terms):
            e = o.Entry(b"", k, t)
            self.l.rf_log_append(self.nodes[i], C.byref(e), 1)

    def close(self):
        self.l.rf_free_cluster(self.nodes[0])


def _voter(cur, voted_for, llt, role, log_terms=None):
    c = Cluster(1)
    nd = c.nodes[0].contents
    nd.current_term, nd.voted_for, nd.node_type = cur, voted_for, role
    c.seed_log(0, [llt] if log_terms is None else log_terms)
    return c


def _request_vote(c, term, cand, lli=0, llt=0):
    tr = C.c_int64(); gr = C.c_int()
    rc = c.l.rf_request_vote_rpc(c.nodes[0], term, cand, lli, llt, C.byref(tr), C.byref(gr))
    return rc, tr.value, bool(gr.value)


def test_new_node_shares_peer_list_including_self():
    c = Cluster(3)
    try:
        for nd in c.nodes:
            assert nd.contents.n_peers == 3                       # raft.go:94-97
            assert nd.contents.node_type == o.FOLLOWER            # :82
            assert nd.contents.voted_for == -1 and nd.contents.current_term == 0
            assert nd.contents.log_len == 0                       # :87
        assert C.addressof(c.nodes[0].contents.peers.contents) == C.addressof(c.nodes[2].contents.peers.contents)
        c.seed_log(1, [0, 1, 1])
        c.l.rf_become_leader(c.nodes[1])
        L = c.nodes[1].contents
        assert L.node_type == o.LEADER
        assert [L.next_index[i] for i in range(3)] == [4, 4, 4]   # :108 len(log)+1
        assert [L.match_index[i] for i in range(3)] == [0, 0, 0]  # :116
        c.l.rf_become_candidate(c.nodes[1])
        assert L.node_type == o.CANDIDATE and not L.match_index and L.current_term == 0
        c.l.rf_become_follower(c.nodes[1])
        assert L.node_type == o.FOLLOWER and not L.next_index
    finally:
        c.close()


@pytest.mark.parametrize("k", KATS["vote"], ids=lambda k: k["id"])
def test_request_vote_kats(k):
    v, r, e = k["voter"], k["req"], k["expect"]
    c = _voter(v["cur"], v["voted_for"], v["last_log_term"], v["role"])
    try:
        rc, tr, gr = _request_vote(c, r["term"], r["cand"], r["last_log_index"], r["last_log_term"])
        nd = c.nodes[0].contents
        assert rc == o.OK
        assert (tr, gr, nd.current_term, nd.node_type) == (e["term_result"], e["grant"], e["cur"], e["role"])
        assert nd.voted_for == v["voted_for"]                      # never written (Q2)
    finally:
        c.close()


def test_request_vote_ignores_candidate_log_arguments():
    for lli, llt in itertools.product([-5, 0, 7, 1 << 40], repeat=2):
        c = _voter(2, -1, 2, o.FOLLOWER)
        try:
            assert _request_vote(c, 3, 1, lli, llt) == (o.OK, 3, True)
        finally:
            c.close()


@pytest.mark.parametrize("k", KATS["vote_sequences"], ids=lambda k: k["id"])
def test_request_vote_sequences(k):
    v = k["voter"]
    c = _voter(v["cur"], v["voted_for"], v["last_log_term"], v["role"])
    try:
        for call in k["calls"]:
            rc, tr, gr = _request_vote(c, call["req"]["term"], call["req"]["cand"])
            assert (rc, tr, gr) == (o.OK, call["expect"]["term_result"], call["expect"]["grant"])
        nd = c.nodes[0].contents
        assert (nd.current_term, nd.voted_for, nd.node_type) == (k["final"]["cur"], k["final"]["voted_for"], k["final"]["role"])
    finally:
        c.close()


@pytest.mark.parametrize("k", KATS["vote_empty_log"], ids=lambda k: k["id"])
def test_request_vote_empty_log(k):
    v, e = k["voter"], k["expect"]
    c = _voter(v["cur"], v["voted_for"], 0, v["role"], log_terms=[])
    try:
        rc, tr, gr = _request_vote(c, k["req"]["term"], k["req"]["cand"])
        nd = c.nodes[0].contents
        assert (rc == o.PANIC_INDEX) == e["panic"]
        assert (nd.current_term, nd.node_type) == (e["cur"], e["role"])
        if not e["panic"]:
            assert (tr, gr) == (e["term_result"], e["grant"])
    finally:
        c.close()


@pytest.mark.parametrize("k", KATS["quorum"], ids=lambda k: str(k["match"]))
def test_quorum_index_kats(k):
    m = (C.c_int64 * len(k["match"]))(*k["match"])
    assert o.lib().rf_quorum_match_index(m, len(k["match"])) == k["expect"]


def _bruteforce_quorum(match):
    """Largest N such that a strict majority of slots has match >= N."""
    n = len(match)
    best = None
    for N in sorted(set(match)):
        if 2 * sum(1 for x in match if x >= N) > n:
            best = N
    return best


def test_quorum_index_matches_bruteforce_definition():
    rng = np.random.default_rng(7)
    l = o.lib()
    for n in range(1, 17):
        for _ in range(300):
            match = [int(x) for x in rng.integers(-3, 6, n)]
            m = (C.c_int64 * n)(*match)
            assert l.rf_quorum_match_index(m, n) == _bruteforce_quorum(match), (n, match)
    for n in range(1, 8):           # exhaustive over a small alphabet
        for match in itertools.product(range(3), repeat=n):
            m = (C.c_int64 * n)(*match)
            assert l.rf_quorum_match_index(m, n) == _bruteforce_quorum(match)


@pytest.mark.parametrize("i", range(len(KATS["commit"])))
def test_commit_kats_node_and_batch(i):
    k = KATS["commit"][i]
    n = len(k["match"])
    # reference-shaped node
    c = Cluster(n)
    try:
        c.seed_log(0, k["log_terms"])
        c.l.rf_become_leader(c.nodes[0])
        L = c.nodes[0].contents
        L.current_term, L.commit_index = k["leader_term"], k["commit"]
        for p, mv in enumerate(k["match"]):
            L.match_index[p] = mv
        rc = c.l.rf_leader_advance_commit(c.nodes[0])
        assert (rc == o.PANIC_INDEX) == k["oob"]
        assert L.commit_index == k["expect"]
    finally:
        c.close()
    # batch form
    match = np.array(k["match"], np.int32).reshape(n, 1)
    out, st = o.commit_advance(match, [k["commit"]], [k["leader_term"]], [0, len(k["log_terms"])],
                               np.array(k["log_terms"], np.int32))
    assert out[0] == k["expect"] and bool(st[0] & o.ST_LOG_OOB) == k["oob"]


@pytest.mark.parametrize("i", range(len(KATS["tally"])))
def test_tally_kats(i):
    k = KATS["tally"][i]
    n = k["n"]
    vt = np.array([[v["cur"]] for v in k["voters"]], np.int32)
    vf = np.array([[v["voted_for"]] for v in k["voters"]], np.int32)
    ll = np.array([[v["last_log_term"]] for v in k["voters"]], np.int32)
    r = o.vote_tally(vt, vf, ll, [k["term"]], [k["cand"]])
    assert bin(r.grant_mask[0]).count("1") == k["expect"]["votes"]
    assert bool(r.elected[0]) == k["expect"]["elected"]
    assert r.max_term[0] == k["expect"]["max_term"]
    assert n == vt.shape[0]


def test_vote_kats_through_batch_form():
    ks = KATS["vote"]
    vt = np.array([[k["voter"]["cur"] for k in ks]], np.int32)
    vf = np.array([[k["voter"]["voted_for"] for k in ks]], np.int32)
    ll = np.array([[k["voter"]["last_log_term"] for k in ks]], np.int32)
    r = o.vote_tally(vt, vf, ll, [k["req"]["term"] for k in ks], [k["req"]["cand"] for k in ks])
    for g, k in enumerate(ks):
        e = k["expect"]
        assert r.voter_term[0, g] == e["cur"] and r.max_term[g] == e["term_result"], k["id"]
        assert bool(r.grant_mask[g] & 1) == e["grant"], k["id"]
        assert bool(r.stepdown_mask[g] & 1) == (k["req"]["term"] > k["voter"]["cur"]), k["id"]
        assert bool(r.elected[g]) == e["grant"], k["id"]      # n = 1: one grant is a majority


@pytest.mark.parametrize("k", KATS["clamp"], ids=lambda k: str(k["expect"]))
def test_follower_clamp_kats(k):
    assert o.follower_clamp([k["leader_commit"]], [k["commit"]], [k["last_new_index"]])[0] == k["expect"]
    assert o.lib().rf_min_int(3, 9) == 3 and o.lib().rf_min_int(9, 3) == 3 and o.lib().rf_min_int(4, 4) == 4


@pytest.mark.parametrize("k", KATS["append"], ids=lambda k: k["id"])
def test_append_entries_kats(k):
    c = Cluster(1)
    try:
        c.seed_log(0, k["log_terms"])
        nd = c.nodes[0].contents
        nd.current_term = k["cur"]
        ents = (o.Entry * max(1, len(k["new"])))(*[o.Entry(b"x", i, t) for i, t in k["new"]])
        tr = C.c_int64(); ok = C.c_int()
        rc = c.l.rf_append_entries_rpc(c.nodes[0], k["term"], 0, k["prev_index"], k["prev_term"],
                                       ents, len(k["new"]), k["leader_commit"], C.byref(tr), C.byref(ok))
        e = k["expect"]
        assert (rc == o.PANIC_INDEX) == e["panic"]
        if not e["panic"]:
            assert (tr.value, bool(ok.value)) == (e["term_result"], e["success"])
        if "log_len" in e:
            assert nd.log_len == e["log_len"]
        if "commit" in e:
            assert nd.commit_index == e["commit"]
        if "cur" in e:
            assert nd.current_term == e["cur"]
    finally:
        c.close()


def test_config1_single_group_election_and_commit_on_cpu():
    """Three in-process nodes: election through RequestVoteRPC, then replication
    through AppendEntriesRPC inside its non-panicking domain (logs pre-seeded
    with a slot-0 sentinel, new entries overwrite-by-duplicate only), then the
    leader commit rule over matchIndex."""
    c = Cluster(3)
    try:
        l = c.l
        for i in range(3):
            c.seed_log(i, [0])                                  # sentinel at slot 0
        cand = c.nodes[0]
        l.rf_become_candidate(cand)
        cand.contents.current_term += 1                         # host does the bump; :126-130 does not
        votes = 0
        for nd in c.nodes:
            tr = C.c_int64(); gr = C.c_int()
            assert l.rf_request_vote_rpc(nd, cand.contents.current_term, 0, 0, 0, C.byref(tr), C.byref(gr)) == o.OK
            assert tr.value == 1
            votes += gr.value
        assert votes == 3
        l.rf_become_leader(cand)
        L = cand.contents
        # the leader takes one client entry at slot 1 in its own log
        e = o.Entry(b"set x", 1, 1)
        l.rf_log_append(cand, C.byref(e), 1)
        L.match_index[0] = 1
        # followers must already hold a slot at the new index (see :159-161): pre-extend, then replicate
        for p in (1, 2):
            c.seed_log(p, [0])
            dup = o.Entry(b"set x", 1, 1)
            tr = C.c_int64(); ok = C.c_int()
            rc = l.rf_append_entries_rpc(c.nodes[p], 1, 0, 0, 0, C.byref(dup), 1, 0, C.byref(tr), C.byref(ok))
            assert (rc, tr.value, ok.value) == (o.OK, 1, 1)
            L.match_index[p] = 1
        assert l.rf_leader_advance_commit(cand) == o.OK
        assert L.commit_index == 1                               # log[1].term == currentTerm == 1
        # next heartbeat carries leaderCommit; followers clamp to the last new index
        for p in (1, 2):
            dup = o.Entry(b"set x", 1, 1)
            tr = C.c_int64(); ok = C.c_int()
            assert l.rf_append_entries_rpc(c.nodes[p], 1, 0, 0, 0, C.byref(dup), 1, 1, C.byref(tr), C.byref(ok)) == o.OK
            assert c.nodes[p].contents.commit_index == 1
    finally:
        c.close()


def test_batch_oracle_equals_numpy_closed_form():
    """The batch oracle (built from rf_* calls) against the closed form of the
    grant predicate and a numpy sort, on adversarial inputs."""
    from raft_amd import synth
    for n in (1, 2, 3, 4, 5, 7, 9, 16):
        b = synth.gen_adversarial(4000, n, seed=100 + n)
        r = o.fused_round(*b.fused_args())
        vt, vf, ll = b.voter_term.astype(np.int64), b.voted_for.astype(np.int64), b.last_log_term.astype(np.int64)
        t, cid = b.cand_term.astype(np.int64)[None, :], b.cand_id.astype(np.int64)[None, :]
        empty = ll == synth.LOG_EMPTY
        reach = t >= vt
        grant = reach & ((vf == -1) | (vf == cid)) & ~empty & (ll <= t)
        bits = (1 << np.arange(n))[:, None]
        assert np.array_equal(r.grant_mask, (grant * bits).sum(0).astype(np.uint16))
        assert np.array_equal(r.stepdown_mask, ((t > vt) * bits).sum(0).astype(np.uint16))
        assert np.array_equal(r.voter_term, np.maximum(vt, t).astype(np.int32))
        mx = np.maximum(vt, t).max(0)
        assert np.array_equal(r.max_term, mx.astype(np.int32))
        assert np.array_equal(r.elected.astype(bool), (2 * grant.sum(0) > n) & (mx == t[0]))
        N = np.sort(b.match.astype(np.int64), axis=0)[(n - 1) // 2]
        lens = np.diff(b.log_off.astype(np.int64))
        adv = N > b.commit
        inb = (N >= 0) & (N < lens)
        idx = np.where(adv & inb, b.log_off[:-1].astype(np.int64) + N, 0)
        tat = b.log_term[idx] if b.log_term.size else np.zeros_like(idx)
        exp = np.where(adv & inb & (tat == b.leader_term), N, b.commit)
        assert np.array_equal(r.commit, exp.astype(np.int32))
        st = np.where(adv & ~inb, 1, 0) | np.where((reach & empty).any(0), 2, 0)
        assert np.array_equal(r.status, st.astype(np.uint8))


def test_append_check_batch_semantics():
    """Hand-checked (group, follower) cases for the AppendEntries consistency
    batch: A2 (stale leader, raft.go:145-147), A3 (term mismatch, :151-153), a
    success, and an out-of-range prevLogIndex (:151 index panic)."""
    # one group, four follower slots, each with log terms [0, 1] (sentinel + one entry)
    n, G = 4, 1
    flog_off = np.arange(0, 2 * n + 1, 2, dtype=np.uint32)
    flog = np.tile(np.array([0, 1], np.int32), n)
    ft = np.array([[3], [1], [1], [0]], np.int32)        # follower terms
    pi = np.array([[1], [1], [1], [2]], np.int32)        # prevLogIndex
    pt = np.array([[1], [2], [1], [1]], np.int32)        # prevLogTerm
    nn = np.array([[1], [1], [2], [1]], np.int32)
    mi = np.zeros((n, G), np.int32); ni = np.full((n, G), 3, np.int32)
    r = o.append_check(ft, [2], pi, pt, flog_off, flog, n_new=nn, match_index=mi, next_index=ni)
    assert r.success_mask[0] == 0b0100                   # only slot 2 passes
    assert r.stepdown_mask[0] == 0b1110                  # term 2 > 1, 1, 0
    assert r.follower_term[:, 0].tolist() == [3, 2, 2, 2]
    assert r.status[0] == o.ST_LOG_OOB                   # slot 3: prevLogIndex 2 >= len 2
    assert r.match_index[:, 0].tolist() == [0, 0, 3, 0]  # success: prev + n_new
    assert r.next_index[:, 0].tolist() == [3, 2, 4, 2]   # stale: unchanged; mismatch/oob: back off
    # active_mask: slots outside it are untouched
    r = o.append_check(ft, [2], pi, pt, flog_off, flog, n_new=nn, match_index=mi, next_index=ni,
                       active_mask=[0b0100])
    assert r.follower_term[:, 0].tolist() == [3, 1, 2, 0] and r.success_mask[0] == 0b0100
    assert r.next_index[:, 0].tolist() == [3, 3, 4, 3] and r.status[0] == 0


def test_joint_oracle_equals_numpy_closed_form():
    """Joint-consensus oracle (member masks, majority in both configurations)
    against an independent numpy formulation, random masks and adversarial values."""
    from raft_amd import synth
    rng = np.random.default_rng(23)
    for m in (1, 2, 3, 5, 8, 11, 16):
        G = 3000
        b = synth.gen_adversarial(G, m, seed=700 + m)
        lim = (1 << m) - 1
        om = rng.integers(1, lim + 1, G).astype(np.uint16)
        nm = np.where(rng.random(G) < 0.5, om, rng.integers(1, lim + 1, G)).astype(np.uint16)
        r = o.fused_round(*b.fused_args(), old_mask=om, new_mask=nm)
        slots = np.arange(m)[:, None]
        in_old = ((om[None, :].astype(np.int64) >> slots) & 1).astype(bool)
        in_new = ((nm[None, :].astype(np.int64) >> slots) & 1).astype(bool)
        member = in_old | in_new
        # commit: min of the two configurations' quorum indices
        N = np.minimum(synth.masked_quorum_index(b.match, om), synth.masked_quorum_index(b.match, nm))
        lens = np.diff(b.log_off.astype(np.int64))
        adv = N > b.commit
        inb = (N >= 0) & (N < lens)
        idx = np.where(adv & inb, b.log_off[:-1].astype(np.int64) + N, 0)
        tat = b.log_term[idx] if b.log_term.size else np.zeros_like(idx)
        assert np.array_equal(r.commit, np.where(adv & inb & (tat == b.leader_term), N, b.commit).astype(np.int32))
        # vote: predicate on members only, majority in both configurations
        vt, vf, ll = (x.astype(np.int64) for x in (b.voter_term, b.voted_for, b.last_log_term))
        t, cid = b.cand_term.astype(np.int64)[None, :], b.cand_id.astype(np.int64)[None, :]
        empty = ll == synth.LOG_EMPTY
        grant = member & (t >= vt) & ((vf == -1) | (vf == cid)) & ~empty & (ll <= t)
        bits = (1 << np.arange(m))[:, None]
        assert np.array_equal(r.grant_mask, (grant * bits).sum(0).astype(np.uint16))
        assert np.array_equal(r.stepdown_mask, ((member & (t > vt)) * bits).sum(0).astype(np.uint16))
        mx = np.where(member, np.maximum(vt, t), np.iinfo(np.int64).min).max(0)
        assert np.array_equal(r.max_term, mx.astype(np.int32))
        el = (2 * (grant & in_old).sum(0) > in_old.sum(0)) & (2 * (grant & in_new).sum(0) > in_new.sum(0)) & (mx == t[0])
        assert np.array_equal(r.elected.astype(bool), el)
        assert np.array_equal(r.voter_term[member], np.maximum(vt, t)[member].astype(np.int32))
        assert np.array_equal(r.voter_term[~member], b.voter_term[~member])


def test_term_start_rule_equals_the_lookup_on_raft_shaped_logs():
    """On logs whose terms never decrease and never exceed the leader's term,
    log[N].TermNum == currentTerm is the same as N >= (index of the term's first
    entry): the oracle, which looks the term up, against the rule in numpy; and the
    closed form the bench uses for its generated logs against a scan of the arena."""
    from raft_amd import synth
    for b0 in (synth.gen_fused(20011, 5, seed=31), synth.gen_mixed(9001, seed=32),
               synth.window_logs(synth.gen_fused(7000, 3, seed=33), keep_committed=2)):
        b = synth.raft_valid(b0)
        G = int(b["G"])
        kw = {}
        if "old_mask" in b:
            kw = dict(old_mask=b.old_mask, new_mask=b.new_mask)
            N = np.minimum(synth.masked_quorum_index(b.match[:, :G], b.old_mask),
                           synth.masked_quorum_index(b.match[:, :G], b.new_mask)).astype(np.int64)
        else:
            N = synth.quorum_index(b.match[:, :G]).astype(np.int64)
        if "log_base" in b:
            kw["log_base"] = b.log_base
        exp, st = o.commit_advance(*b.commit_args(), **kw)
        start, ln = synth.term_start_form(b)
        base = b["log_base"].astype(np.int64) if "log_base" in b else 0
        commit = b.commit[:G].astype(np.int64)
        in_log = (N >= base) & (N < base + ln)
        rule = np.where((N > commit) & in_log & (N >= start), N, commit)
        assert np.array_equal(rule, exp.astype(np.int64))
        assert np.array_equal((st & 1).astype(bool), (N > commit) & ~in_log)
        if "log_base" not in b0:
            s2, l2 = synth.term_start_monotone(b0)
            assert np.array_equal(s2, start) and np.array_equal(l2, ln)


@pytest.mark.parametrize("dt", [np.int32, np.int64])
def test_follower_commit_rule_after_a_successful_append(dt):
    """raft.go:172-176 as orc_append_commit restates it for a batch: only slots whose RPC
    succeeded are touched; (leaderCommit, commitIndex, last new Index) -> commitIndex as in
    the clamp's known answers, the decrease included; no new entries where leaderCommit is
    above commitIndex is the lastEntry panic (:175, :235): status bit, commitIndex kept."""
    check = o.append_check64 if dt == np.int64 else o.append_check
    G = 5
    ft = np.array([[2, 2, 2, 2, 9]], dt)                       # the last follower is ahead of its leader: no success
    args = (ft, np.full(G, 2, dt), np.zeros((1, G), dt), np.ones((1, G), dt),
            np.arange(G + 1, dtype=np.uint32), np.ones(G, dt))
    lc, ci, li = np.array([5, 5, 3, 5, 5], dt), np.array([[3, 3, 3, 4, 1]], dt), np.array([[4, 9, 9, 2, 4]], dt)
    res = check(*args, n_new=np.ones((1, G), dt))
    assert res.success_mask.tolist() == [1, 1, 1, 1, 0]
    out = o.append_commit(res, lc, li, ci, n_new=np.ones((1, G), dt))
    assert out.tolist() == [[4, 5, 3, 2, 1]] and not res.status.any()
    res = check(*args, n_new=np.array([[0, 1, 0, 0, 0]], dt))
    out = o.append_commit(res, lc, li, ci, n_new=np.array([[0, 1, 0, 0, 0]], dt))
    assert out.tolist() == [[3, 5, 3, 4, 1]] and res.status.tolist() == [2, 0, 0, 2, 0]


def test_batch_commit_rule_equals_per_node_append_entries_rpc():
    """The batch form (orc_append_check + orc_append_commit) against the per-node restatement
    of AppendEntriesRPC fed real entries, inside the domain where raft.go:158-167 does not
    panic (every new entry's Index below the follower's log length; heartbeats included):
    same success, same term, same commitIndex, and a panic exactly where the batch flags one."""
    rng = np.random.default_rng(11)
    G = 400
    lens = rng.integers(1, 9, G)
    logs = [rng.integers(1, 4, l).cumsum() // 2 + 1 for l in lens]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    ft = rng.integers(1, 5, (1, G)).astype(np.int32)
    lt = rng.integers(1, 5, G).astype(np.int32)
    pi = np.array([[rng.integers(0, l) for l in lens]], np.int32)
    pt = np.array([[logs[g][pi[0, g]] + (rng.random() < 0.25) for g in range(G)]], np.int32)
    nn = rng.integers(0, 3, (1, G)).astype(np.int32)
    li = np.array([[rng.integers(0, l) for l in lens]], np.int32)       # Index of the last new entry: inside the log
    lc = rng.integers(0, 10, G).astype(np.int32)
    ci = rng.integers(0, 8, (1, G)).astype(np.int32)
    res = o.append_check(ft, lt, pi, pt, off, np.concatenate(logs).astype(np.int32), n_new=nn)
    fc = o.append_commit(res, lc, li, ci, n_new=nn)
    seen = set()
    for g in range(G):
        c = Cluster(1)
        try:
            c.seed_log(0, [int(t) for t in logs[g]])
            nd = c.nodes[0].contents
            nd.current_term, nd.commit_index = int(ft[0, g]), int(ci[0, g])
            k = int(nn[0, g])
            ents = (o.Entry * max(1, k))(*[o.Entry(b"x", int(li[0, g]), 1) for _ in range(k)])
            tr = C.c_int64(); ok = C.c_int()
            rc = c.l.rf_append_entries_rpc(c.nodes[0], int(lt[g]), 0, int(pi[0, g]), int(pt[0, g]), ents, k,
                                           int(lc[g]), C.byref(tr), C.byref(ok))
            flagged = bool(res.status[g] & o.ST_EMPTY_LOG)
            assert (rc == o.PANIC_INDEX) == flagged, g
            # a panic at :175 comes after every check passed: the batch reports such a slot as succeeding
            assert bool(res.success_mask[g] & 1) == (bool(ok.value) or flagged), g
            assert nd.current_term == res.follower_term[0, g] and nd.commit_index == fc[0, g], g
            seen.add((bool(ok.value), flagged, int(fc[0, g]) != int(ci[0, g])))
        finally:
            c.close()
    assert {(True, False, True), (True, False, False), (False, False, False), (False, True, False)} <= seen


def test_size_order_helpers_describe_the_batch_truthfully():
    """synth.segment_table / split_by_size / reorder (host-side layout helpers of the one-launch
    forms): the table covers [0, G) in 64-aligned steps and only calls a tile a size-n tile when all of
    its groups are plain n-peer clusters; the per-size images together are the batch (the oracle on
    each image equals the oracle on the whole); reordering permutes the oracle's decisions."""
    from raft_amd import synth
    for G, frac in ((1, 0.1), (64, 0.0), (4097, 0.1), (20000, 0.3)):
        b = synth.gen_mixed(G, seed=9 + G, bucketed="size", joint_frac=frac)
        first, kinds = synth.segment_table(b.old_mask, b.new_mask)
        assert first[0] == 0 and first[-1] == G and len(first) == len(kinds) + 1 and len(kinds) <= 16
        assert all(x % 64 == 0 for x in first[1:-1]) and all(first[i] < first[i + 1] for i in range(len(kinds)))
        assert all(kinds[i] != kinds[i + 1] for i in range(len(kinds) - 1))
        for lo, hi, k in zip(first[:-1], first[1:], kinds):
            if k:
                assert (b.old_mask[lo:hi] == (1 << k) - 1).all() and (b.new_mask[lo:hi] == (1 << k) - 1).all()
        whole = o.fused_round(*b.fused_args(), G=G, old_mask=b.old_mask, new_mask=b.new_mask)
        covered = 0
        for k, lo, hi, sub in synth.split_by_size(b):
            assert lo == covered and sub.G == hi - lo and sub.n == (k or b.n)
            covered = hi
            kw = {} if k else dict(old_mask=sub.old_mask, new_mask=sub.new_mask)
            part = o.fused_round(*sub.fused_args(), G=sub.G, **kw)
            for f in ("commit", "grant_mask", "stepdown_mask", "elected", "max_term", "status"):
                assert np.array_equal(part[f], whole[f][lo:hi]), (G, k, f)
        assert covered == G
        order = np.random.default_rng(G).permutation(G)
        r = synth.reorder(b, order)
        moved = o.fused_round(*r.fused_args(), G=G, old_mask=r.old_mask, new_mask=r.new_mask)
        for f in ("commit", "grant_mask", "elected", "status"):
            assert np.array_equal(moved[f], whole[f][order]), f
    # an order that needs more than 16 segments has no table
    b = synth.gen_mixed(5000, seed=3, bucketed=False, joint_frac=0.0)
    assert synth.segment_table(b.old_mask, b.new_mask) in (None, ([0, 5000], [0]))
