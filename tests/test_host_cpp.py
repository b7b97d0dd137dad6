"""The C++ host (raft_amd/csrc/host): the reference's Go surface mirrored in
compiled code (Go toolchain absent) and the BatchedDriver above the C-ABI.
CPU tests cover the RPC methods against the hand-derived KATs and the
flatten/apply logic against the oracle; the GPU test runs a whole Step()."""
import json
import os

import numpy as np
import pytest

import host_lib as h
import oracle_lib as o
from raft_amd import synth

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))


def test_new_node_and_roles():
    c = h.Cluster(1, 3)
    try:
        for p in range(3):
            s = c.get(0, p)
            assert (s["id"], s["nodeType"], s["currentTerm"], s["votedFor"], s["logLen"], s["nPeers"]) == \
                (p, h.FOLLOWER, 0, -1, 0, 3)                       # raft.go:80-97
        c.seed(0, 1, 0, -1, h.FOLLOWER, log_terms=[0, 1, 1])
        c.become(0, 1, 0)
        nx, mt = c.leader_arrays(0, 1)
        assert nx == [4, 4, 4] and mt == [0, 0, 0]                   # :108, :116
        c.become(0, 1, 2)
        assert c.get(0, 1)["nodeType"] == h.CANDIDATE and c.leader_arrays(0, 1) is None
        assert c.get(0, 1)["currentTerm"] == 0                       # no term bump (:126-130)
    finally:
        c.close()


@pytest.mark.parametrize("k", KATS["vote"], ids=lambda k: k["id"])
def test_request_vote_kats(k):
    v, r, e = k["voter"], k["req"], k["expect"]
    c = h.Cluster(1, 1)
    try:
        c.seed(0, 0, v["cur"], v["voted_for"], v["role"], log_terms=[v["last_log_term"]])
        rc, tr, gr = c.request_vote(0, 0, r["term"], r["cand"], r["last_log_index"], r["last_log_term"])
        s = c.get(0, 0)
        assert (rc, tr, gr, s["currentTerm"], s["nodeType"], s["votedFor"]) == \
            (0, e["term_result"], e["grant"], e["cur"], e["role"], v["voted_for"])
    finally:
        c.close()


@pytest.mark.parametrize("k", KATS["vote_empty_log"], ids=lambda k: k["id"])
def test_request_vote_empty_log_panics_like_go(k):
    c = h.Cluster(1, 1)
    try:
        c.seed(0, 0, k["voter"]["cur"], k["voter"]["voted_for"], k["voter"]["role"], log_terms=[])
        rc, tr, gr = c.request_vote(0, 0, k["req"]["term"], k["req"]["cand"])
        assert (rc == 1) == k["expect"]["panic"]
        assert c.get(0, 0)["currentTerm"] == k["expect"]["cur"]
    finally:
        c.close()


@pytest.mark.parametrize("k", KATS["append"], ids=lambda k: k["id"])
def test_append_entries_kats(k):
    c = h.Cluster(1, 1)
    try:
        c.seed(0, 0, k["cur"], -1, h.FOLLOWER, log_terms=k["log_terms"])
        rc, tr, ok = c.append_entries(0, 0, k["term"], 0, k["prev_index"], k["prev_term"], k["new"], k["leader_commit"])
        e = k["expect"]
        assert (rc == 1) == e["panic"]
        s = c.get(0, 0)
        if not e["panic"]:
            assert (tr, ok) == (e["term_result"], e["success"])
        for key, field in (("log_len", "logLen"), ("commit", "commitIndex"), ("cur", "currentTerm")):
            if key in e:
                assert s[field] == e[key]
    finally:
        c.close()


def _build(G, n, seed, term_shift=0, raft_shaped=False):
    """A cluster whose object state equals a synthetic batch (every term moved up
    by term_shift); returns (cluster, batch) -- the batch is returned unshifted.
    raft_shaped: no log term above its leader's term (synth.raft_valid)."""
    b = synth.gen_fused(G, n, seed=seed)
    if raft_shaped:
        b = synth.raft_valid(b)
        # a candidate that wins leads with its own log: none of its terms above its term either
        b["last_log_term"] = np.minimum(b["last_log_term"], b["cand_term"][None, :]).astype(np.int32)
    c = h.Cluster(G, n)
    if term_shift:
        c0, b0 = c, b
        b = synth.Batch(b0)
        for k in ("voter_term", "last_log_term", "log_term", "leader_term", "cand_term"):
            b[k] = b0[k].astype(np.int64) + term_shift
        _seed_objects(c, b, n, G)
        # the unshifted planes must describe the objects too (see the end of _seed_objects)
        for k in ("voter_term", "last_log_term"):
            b0[k] = (b[k] - term_shift).astype(np.int32)
        return c, b0
    _seed_objects(c, b, n, G)
    return c, b


def _seed_objects(c, b, n, G):
    for g in range(G):
        lead = int(b.cand_id[g] + 1) % n                       # some member other than the candidate leads
        lo, hi = int(b.log_off[g]), int(b.log_off[g + 1])
        for p in range(n):
            logs = [int(b.last_log_term[p, g])]
            role = h.FOLLOWER
            if p == lead:
                logs = [int(x) for x in b.log_term[lo:hi]]
            c.seed(g, p, int(b.voter_term[p, g]), int(b.voted_for[p, g]), role, 0, logs)
        c.become(g, lead, 0)
        c.seed(g, lead, int(b.leader_term[g]), int(b.voted_for[lead, g]), h.LEADER, int(b.commit[g]))
        c.set_match(g, lead, [int(x) for x in b.match[:, g]])
        cand = int(b.cand_id[g])
        c.seed(g, cand, int(b.cand_term[g]), int(b.voted_for[cand, g]), h.CANDIDATE, 0)
        c.roles(g, leader=lead, candidate=cand)
        # the batch's voter planes must describe the objects: the leader's own slot carries its state
        b.voter_term[lead, g] = b.leader_term[g]
        b.last_log_term[lead, g] = b.log_term[hi - 1]
        b.voter_term[cand, g] = b.cand_term[g]


def test_flatten_matches_the_objects_and_apply_matches_the_oracle():
    G, n = 300, 5
    c, b = _build(G, n, seed=17)
    try:
        rc, f = c.flatten()
        assert rc == 0
        for k in ("match", "commit", "leader_term", "log_off", "log_term", "voter_term", "voted_for",
                  "last_log_term", "cand_term", "cand_id"):
            assert np.array_equal(f[k], b[k]), k
        r = o.fused_round(*b.fused_args())                  # oracle stands in for the GPU on CPU
        c.apply(r)
        for g in range(G):
            lead, cand = int(b.cand_id[g] + 1) % n, int(b.cand_id[g])
            for p in range(n):
                s = c.get(g, p)
                assert s["currentTerm"] == r.voter_term[p, g]
                if (r.stepdown_mask[g] >> p) & 1:
                    assert s["nodeType"] == h.FOLLOWER
            assert c.get(g, lead)["commitIndex"] == r.commit[g]
            new_lead, new_cand = c.get_roles(g)
            if r.elected[g]:
                assert new_lead == cand and new_cand == -1
                assert c.get(g, cand)["nodeType"] == h.LEADER and c.leader_arrays(g, cand) is not None
            else:
                # a leader the candidate's term demoted (raft.go:221) leads no more
                demoted = (r.stepdown_mask[g] >> lead) & 1
                assert new_lead == (-1 if demoted else lead) and new_cand == cand
                assert c.get(g, lead)["nodeType"] == (h.FOLLOWER if demoted else h.LEADER)
    finally:
        c.close()


def test_flatten_and_apply_on_several_threads():
    """At 32768 groups and more the driver walks the groups on several host threads
    (two passes: log lengths, then everything); the flat image and the applied
    objects must be what the single-threaded walk gives."""
    G, n = 33000, 5
    c, b = _build(G, n, seed=23)
    try:
        rc, f = c.flatten()
        assert rc == 0
        for k in ("match", "commit", "leader_term", "log_off", "log_term", "voter_term", "voted_for",
                  "last_log_term", "cand_term", "cand_id"):
            assert np.array_equal(f[k], b[k]), k
        r = o.fused_round(*b.fused_args())
        c.apply(r)
        for g in list(range(0, G, 97)) + [G - 1]:
            lead, cand = int(b.cand_id[g] + 1) % n, int(b.cand_id[g])
            for p in range(n):
                assert c.get(g, p)["currentTerm"] == r.voter_term[p, g]
            assert c.get(g, lead)["commitIndex"] == r.commit[g]
            new_lead, new_cand = c.get_roles(g)
            demoted = (r.stepdown_mask[g] >> lead) & 1
            assert (new_lead, new_cand) == ((cand, -1) if r.elected[g] else (-1 if demoted else lead, cand))
    finally:
        c.close()


def test_flatten_rejects_values_outside_int32():
    c = h.Cluster(1, 3)
    try:
        c.seed(0, 0, 1 << 40, -1, h.FOLLOWER, log_terms=[0])
        rc, _ = c.flatten()
        assert rc == -2
    finally:
        c.close()


def test_groups_without_leader_or_candidate_are_inert():
    G, n = 4, 3
    c = h.Cluster(G, n)
    try:
        for g in range(G):
            for p in range(n):
                c.seed(g, p, 5 + p, -1, h.FOLLOWER, 0, [0, 1])
        rc, f = c.flatten()
        assert rc == 0
        r = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                          f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
        assert not r.elected.any() and not r.grant_mask.any() and not r.stepdown_mask.any()
        assert np.array_equal(r.voter_term, f["voter_term"]) and np.array_equal(r.commit, f["commit"])
    finally:
        c.close()


@pytest.mark.gpu
def test_step_on_gpu_equals_per_node_reference_calls():
    """BatchedDriver.Step() (one GPU launch) against the same round done one
    RequestVoteRPC at a time on the oracle's reference-shaped nodes."""
    G, n = 2000, 5
    c, b = _build(G, n, seed=23)
    try:
        r = o.fused_round(*b.fused_args())
        assert c.step(0) == 0
        for g in range(G):
            lead, cand = int(b.cand_id[g] + 1) % n, int(b.cand_id[g])
            for p in range(n):
                assert c.get(g, p)["currentTerm"] == r.voter_term[p, g]
            assert c.get(g, lead)["commitIndex"] == r.commit[g]
            assert (c.get_roles(g)[0] == cand) == bool(r.elected[g])
        assert c.step(0) == 0                      # a second round on the applied state
    finally:
        c.close()


def test_values_beyond_int32_need_the_wide_flatten():
    SH = 1 << 40
    c, b = _build(50, 3, seed=5, term_shift=SH)
    try:
        rc, _ = c.flatten()
        assert rc == -2                                   # the compact layout cannot hold these terms
        rc, f = c.flatten64()
        assert rc == 0
        assert np.array_equal(f["voter_term"], b.voter_term.astype(np.int64) + SH)
        assert np.array_equal(f["last_log_term"], b.last_log_term.astype(np.int64) + SH)
        assert np.array_equal(f["cand_term"], b.cand_term.astype(np.int64) + SH)
        assert np.array_equal(f["leader_term"], b.leader_term.astype(np.int64) + SH)
        assert np.array_equal(f["log_term"], b.log_term.astype(np.int64) + SH)
        assert np.array_equal(f["match"], b.match) and np.array_equal(f["voted_for"], b.voted_for)
    finally:
        c.close()


@pytest.mark.gpu
def test_step_falls_back_to_the_wide_layout():
    """Terms beyond int32: Step() runs rq_round64 and applies the same decisions the
    compact round gives on the unshifted state."""
    G, n, SH = 2000, 5, 1 << 40
    c, b = _build(G, n, seed=29, term_shift=SH)
    try:
        r = o.fused_round(*b.fused_args())
        assert c.step(0) == 0 and c.last_step_wide()
        for g in range(G):
            lead, cand = int(b.cand_id[g] + 1) % n, int(b.cand_id[g])
            for p in range(n):
                assert c.get(g, p)["currentTerm"] == int(r.voter_term[p, g]) + SH
            assert c.get(g, lead)["commitIndex"] == r.commit[g]
            assert (c.get_roles(g)[0] == cand) == bool(r.elected[g])
    finally:
        c.close()


def _build_mixed(G, seed):
    """Groups of 3..9 members whose object state equals a gen_mixed batch without joint
    configurations; returns (cluster, batch, sizes)."""
    b = synth.gen_mixed(G, seed=seed, joint_frac=0.0)
    sizes = np.array([bin(int(m)).count("1") for m in b.old_mask[:G]], np.uint32)
    n = int(sizes.max())
    c = h.Cluster(G, n, sizes=sizes)
    for g in range(G):
        k = int(sizes[g])
        lead, cand = (int(b.cand_id[g]) + 1) % k, int(b.cand_id[g]) % k
        lo, hi = int(b.log_off[g]), int(b.log_off[g + 1])
        for p in range(k):
            logs = [int(b.last_log_term[p, g])] if b.last_log_term[p, g] != synth.LOG_EMPTY else []
            if p == lead:
                logs = [int(x) for x in b.log_term[lo:hi]]
            c.seed(g, p, int(b.voter_term[p, g]), int(b.voted_for[p, g]), h.FOLLOWER, 0, logs)
        c.become(g, lead, 0)
        c.seed(g, lead, int(b.leader_term[g]), int(b.voted_for[lead, g]), h.LEADER, int(b.commit[g]))
        c.set_match(g, lead, [int(x) for x in b.match[:k, g]])
        c.seed(g, cand, int(b.cand_term[g]), int(b.voted_for[cand, g]), h.CANDIDATE, 0)
        c.roles(g, leader=lead, candidate=cand)
    return c, b, sizes


def test_flatten_groups_of_different_sizes():
    G = 400
    c, b, sizes = _build_mixed(G, seed=6)
    try:
        rc, f = c.flatten(log_cap=1 << 22)
        assert rc == 0 and f["mixed"]
        assert np.array_equal(f["member_mask"], (1 << sizes.astype(np.int64)) - 1)
        n = f["match"].shape[0]
        mem = (np.arange(n)[:, None] < sizes[None, :])
        assert np.array_equal(f["match"][mem], b.match[:n, :G][mem])
        assert (f["match"][~mem] == 0).all() and (f["voted_for"][~mem] == -1).all()
        assert (f["last_log_term"][~mem] == synth.LOG_EMPTY).all()
        assert np.array_equal(f["log_off"], b.log_off[:G + 1])
        for g in range(G):
            k = int(sizes[g])
            assert c.get_roles(g) == ((int(b.cand_id[g]) + 1) % k, int(b.cand_id[g]) % k)
            assert len(c.leader_arrays(g, c.get_roles(g)[0])[1]) == c.n
    finally:
        c.close()


def test_apply_with_groups_of_different_sizes():
    """Flatten -> the oracle's member-mask round -> Apply, all on the CPU."""
    G = 300
    c, b, sizes = _build_mixed(G, seed=7)
    try:
        rc, f = c.flatten(log_cap=1 << 22)
        assert rc == 0
        r = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                          f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"],
                          old_mask=f["member_mask"], new_mask=f["member_mask"])
        roles_before = [c.get_roles(g) for g in range(G)]
        c.apply(r)
        for g in range(G):
            lead, cand = roles_before[g]
            for p in range(int(sizes[g])):
                assert c.get(g, p)["currentTerm"] == r.voter_term[p, g]
            assert c.get(g, lead)["commitIndex"] == r.commit[g]
            assert (c.get_roles(g)[0] == cand) == bool(r.elected[g])
            if r.elected[g]:
                assert len(c.leader_arrays(g, cand)[0]) == c.n      # hook pads to n; node holds sizes[g]
    finally:
        c.close()


@pytest.mark.gpu
def test_step_with_groups_of_different_sizes():
    """Step() on 3..9-member groups runs the member-mask round; the objects end up where
    the oracle's joint round (old = new = members) on the flattened image says."""
    G = 1500
    c, b, sizes = _build_mixed(G, seed=8)
    try:
        rc, f = c.flatten(log_cap=1 << 23)
        assert rc == 0 and f["mixed"]
        r = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                          f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"],
                          old_mask=f["member_mask"], new_mask=f["member_mask"])
        roles_before = [c.get_roles(g) for g in range(G)]
        assert c.step(0) == 0 and not c.last_step_wide()
        for g in range(G):
            lead, cand = roles_before[g]
            for p in range(int(sizes[g])):
                assert c.get(g, p)["currentTerm"] == r.voter_term[p, g]
            assert c.get(g, lead)["commitIndex"] == r.commit[g]
            assert (c.get_roles(g)[0] == cand) == bool(r.elected[g])
        assert 0.05 < r.elected.mean() < 0.95
    finally:
        c.close()


def _scenario_three_node_groups(step):
    """64 three-node groups through a whole term: election (batched tally), one client
    entry replicated with AppendEntriesRPC inside its non-panicking domain (slot-0
    sentinel, followers pre-extended, raft.go:158-161), commit advance (batched), and
    the follower clamp on the next heartbeat.  `step(cluster)` runs one batched round."""
    G, n = 64, 3
    c = h.Cluster(G, n)
    try:
        for g in range(G):
            for p in range(n):
                c.seed(g, p, 0, -1, h.FOLLOWER, 0, [0])
            c.become(g, 0, 2)                                   # BecomeCandidate: no term bump (:126-130)
            c.seed(g, 0, 1, -1, h.CANDIDATE, 0)                 # ... the host bumps the term
            c.roles(g, leader=-1, candidate=0)
        step(c)
        for g in range(G):
            assert c.get_roles(g) == (0, -1)                    # elected, no longer a candidate
            assert [c.get(g, p)["currentTerm"] for p in range(n)] == [1, 1, 1]
            assert c.get(g, 0)["nodeType"] == h.LEADER and c.get(g, 1)["nodeType"] == h.FOLLOWER
            nx, mt = c.leader_arrays(g, 0)
            assert nx == [2, 2, 2] and mt == [0, 0, 0]          # :108, :116 (log holds the sentinel only)
        # one client entry at slot 1 of every leader; followers that are reachable replicate it
        for g in range(G):
            c.seed(g, 0, 1, -1, h.LEADER, 0, [0, 1])
            reach = {0: (1, 2), 1: (1,), 2: (2,), 3: ()}[g % 4]
            match = [1, 0, 0]
            for p in reach:
                c.seed(g, p, 1, -1, h.FOLLOWER, 0, [0, 0])     # pre-extended: slot 1 exists (:159)
                rc, tr, ok = c.append_entries(g, p, 1, 0, 0, 0, [(1, 1)], 0)
                assert (rc, tr, ok) == (0, 1, True)
                match[p] = 1
            c.set_match(g, 0, match)
        step(c)
        for g in range(G):
            expect = 0 if g % 4 == 3 else 1                     # a majority of 3 needs one follower
            assert c.get(g, 0)["commitIndex"] == expect
            assert c.get_roles(g) == (0, -1)
        # heartbeat carrying leaderCommit: followers clamp to min(leaderCommit, last new index) (:172-176)
        for g in range(G):
            lc = c.get(g, 0)["commitIndex"]
            for p in {0: (1, 2), 1: (1,), 2: (2,), 3: ()}[g % 4]:
                rc, tr, ok = c.append_entries(g, p, 1, 0, 0, 0, [(1, 1)], lc)
                assert (rc, ok) == (0, True) and c.get(g, p)["commitIndex"] == lc
    finally:
        c.close()


def _oracle_step(c):
    rc, f = c.flatten()
    assert rc == 0 and not f["mixed"]
    c.apply(o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                          f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"]))


def test_three_node_groups_through_a_term_on_cpu():
    _scenario_three_node_groups(_oracle_step)


@pytest.mark.gpu
def test_three_node_groups_through_a_term_on_gpu():
    def gpu_step(c):
        assert c.step(0) == 0
    _scenario_three_node_groups(gpu_step)


@pytest.mark.gpu
def test_three_node_groups_through_a_term_on_the_resident_driver():
    """The same scenario with the image resident in HBM: after the first step only
    the groups whose Nodes were touched go up and only changed groups come back."""
    reports = []

    def resident_step(c):
        if not c.resident:
            c.use_resident(0)
        rc, rep = c.step_resident()
        assert rc == 0 and not c.flagged()
        reports.append(rep)
    _scenario_three_node_groups(resident_step)
    assert reports[0]["rebuilt"] and reports[0]["changed"] == 64           # every group elected its candidate
    assert not reports[1]["rebuilt"] and reports[1]["uploaded"] == 64      # every leader took a client entry
    assert reports[1]["changed"] == 48                                     # commitIndex moved where a follower replied


@pytest.mark.gpu
def test_step_reports_groups_where_the_reference_panics():
    """A cluster fresh from NewNode has empty logs (raft.go:87): RequestVoteRPC panics
    at lastEntry (raft.go:204, :235) after it has raised the voter's term.  The batched
    step applies what the reference had done before the panic and lists the group; a
    leader whose quorum index lies outside its log (log[N], the convention of
    raft.go:151) is listed too and keeps its commitIndex."""
    for resident in (False, True):
        G, n = 6, 3
        c = h.Cluster(G, n)
        try:
            for g in range(G):
                for p in range(n):
                    logs = [] if g < 2 else [0, 1]                  # groups 0, 1: straight from NewNode
                    c.seed(g, p, 0, -1, h.FOLLOWER, 0, logs)
            for g in (0, 1, 2):                                     # candidates at term 1
                c.become(g, 0, 2)
                c.seed(g, 0, 1, -1, h.CANDIDATE, 0, [] if g < 2 else [0, 1])
                c.roles(g, leader=-1, candidate=0)
            for g in (3, 4):                                        # leaders; group 4's followers claim index 7
                c.become(g, 0, 0)
                c.seed(g, 0, 1, -1, h.LEADER, 0, [0, 1])
                c.set_match(g, 0, [7, 7, 7] if g == 4 else [1, 1, 0])
                c.roles(g, leader=0, candidate=-1)
            # the per-node reference calls panic for the NewNode groups
            probe = h.Cluster(1, n)
            assert probe.request_vote(0, 1, 1, 0)[0] == 1
            probe.close()
            if resident:
                c.use_resident(0)
                rc, _ = c.step_resident()
            else:
                rc = c.step(0)
            assert rc == 0
            assert sorted(c.flagged()) == [(0, 2), (1, 2), (4, 1)]
            if not resident:
                assert c.throw_if_flagged()
            for g in (0, 1):                                        # term raised, role set, nobody elected
                assert [c.get(g, p)["currentTerm"] for p in range(n)] == [1, 1, 1]
                assert c.get_roles(g) == (-1, 0)
            assert c.get_roles(2) == (0, -1)                        # the group with logs elects as usual
            assert c.get(3, 0)["commitIndex"] == 1 and c.get(4, 0)["commitIndex"] == 0
        finally:
            c.close()


@pytest.mark.gpu
def test_resident_driver_tracks_the_oracle_over_incremental_steps():
    """2000 five-node groups resident on the device.  Every step the host touches a few
    hundred groups (longer leader logs, new matchIndex, a new candidate term) and marks
    them dirty; the driver must leave the objects where the oracle's round over a fresh
    flatten of ALL objects puts them, while moving only the touched and changed rows."""
    G, n = 2000, 5
    c, b = _build(G, n, seed=41)
    rng = np.random.default_rng(9)
    try:
        c.use_resident(0)
        for g in range(G):
            c.roles(g, *c.get_roles(g))                            # same roles in both drivers
        for it in range(5):
            if it:
                touched = rng.choice(G, 300, replace=False)
                for g in touched.tolist():
                    lead, cand = c.get_roles(g)
                    if lead >= 0:
                        s = c.get(g, lead)
                        # it 2: some logs outgrow their slice and move inside the arena;
                        # it 3: one log outgrows the arena's free space, everything is laid out afresh
                        grow = 40 if (it == 2 and g % 10 == 0) else int(rng.integers(0, 3))
                        if it == 3 and g == int(touched[0]):
                            grow = 90000
                        terms = [1 + (i * int(s["currentTerm"])) // (s["logLen"] + grow) for i in range(s["logLen"] + grow)]
                        c.seed(g, lead, s["currentTerm"], s["votedFor"], h.LEADER, s["commitIndex"], terms)
                        c.set_match(g, lead, [int(x) for x in rng.integers(0, s["logLen"] + grow, n)])
                    if cand >= 0:
                        s = c.get(g, cand)
                        c.seed(g, cand, s["currentTerm"] + int(rng.integers(0, 2)), s["votedFor"], h.CANDIDATE, s["commitIndex"])
            rc, f = c.flatten(log_cap=1 << 22)
            assert rc == 0
            exp = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                                f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
            roles_before = [c.get_roles(g) for g in range(G)]
            rc, rep = c.step_resident()
            assert rc == 0
            assert rep["rebuilt"] == (it in (0, 3))                 # first image; a log that outgrew the arena
            if it in (1, 2, 4):
                assert 300 <= rep["uploaded"] <= 300 + G            # touched groups (+ groups that elected last step)
            moved = (exp.commit != f["commit"]) | (exp.stepdown_mask != 0) | (exp.elected != 0) | (exp.status != 0)
            assert rep["changed"] == int(moved.sum())
            assert sorted(c.flagged()) == sorted((int(g), int(exp.status[g])) for g in np.nonzero(exp.status)[0])
            for g in range(G):
                lead, cand = roles_before[g]
                if cand >= 0:
                    for p in range(n):
                        assert c.get(g, p)["currentTerm"] == exp.voter_term[p, g], (it, g, p)
                    assert (c.get_roles(g)[0] == cand) == bool(exp.elected[g]), (it, g)
                if lead >= 0:
                    assert c.get(g, lead)["commitIndex"] == exp.commit[g], (it, g)
            for g in range(G):                                      # keep the batched driver's roles (used by flatten) in step
                lead, cand = c.get_roles(g)
                c.l.rh_group_roles(c.h, g, lead, cand)
    finally:
        c.close()


@pytest.mark.gpu
def test_resident_driver_with_groups_of_different_sizes_in_size_order():
    """3..9-member groups resident on the device.  Kept in size order the driver launches with the
    size table (tiles of one size run the fixed-size code); a group that changes its size is a
    rebuild.  The objects end up where the oracle's member-mask round on a fresh flatten puts them."""
    G = 3000
    c, b, sizes = _build_mixed(G, seed=12)
    assert (np.diff(sizes.astype(np.int64)) >= 0).all()              # gen_mixed without joint groups is in size order
    rng = np.random.default_rng(4)
    try:
        c.use_resident(0)
        for g in range(G):
            c.roles(g, *c.get_roles(g))
        for it in range(3):
            if it:
                for g in rng.choice(G, 200, replace=False).tolist():
                    lead, cand = c.get_roles(g)
                    k = int(sizes[g])
                    if lead >= 0:
                        s = c.get(g, lead)
                        c.set_match(g, lead, [int(x) for x in rng.integers(0, max(1, s["logLen"]), k)])
                    if cand >= 0:
                        s = c.get(g, cand)
                        c.seed(g, cand, s["currentTerm"] + 1, s["votedFor"], h.CANDIDATE, s["commitIndex"])
            rc, f = c.flatten(log_cap=1 << 23)
            assert rc == 0 and f["mixed"]
            exp = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                                f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"],
                                old_mask=f["member_mask"], new_mask=f["member_mask"])
            roles_before = [c.get_roles(g) for g in range(G)]
            rc, rep = c.step_resident()
            assert rc == 0 and rep["rebuilt"] == (it == 0)
            assert 7 <= c.segments() <= 16                             # sizes 3..9 + the tiles where two sizes meet
            for g in range(G):
                lead, cand = roles_before[g]
                if cand >= 0:
                    for p in range(int(sizes[g])):
                        assert c.get(g, p)["currentTerm"] == exp.voter_term[p, g], (it, g, p)
                    assert (c.get_roles(g)[0] == cand) == bool(exp.elected[g]), (it, g)
                if lead >= 0:
                    assert c.get(g, lead)["commitIndex"] == exp.commit[g], (it, g)
            for g in range(G):
                lead, cand = c.get_roles(g)
                c.l.rh_group_roles(c.h, g, lead, cand)
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [False, True])
def test_resident_driver_random_schedule(wide):
    """(wide: every term beyond 2^40, so the image is laid out in int64 elements from the first step.)
    The resident driver under a random schedule (RQ_FUZZ_SEED / RQ_FUZZ_ITERS: seed and number
    of steps, for soak runs): a group count that does not fill its last tile, every step a random
    number of groups touched -- logs growing a little, a lot (moves inside the arena, now and then
    a full re-layout), matchIndex redrawn, candidate terms raised -- and after every step the
    objects where the oracle's round over a fresh flatten of all objects puts them."""
    import os
    G, n = 3001, 5
    seed = int(os.environ.get("RQ_FUZZ_SEED", 5))
    steps = int(os.environ.get("RQ_FUZZ_ITERS", 8))
    SH = (1 << 40) if wide else 0
    c, b = _build(G, n, seed=100 + seed, term_shift=SH)
    rng = np.random.default_rng(seed)
    rebuilt = 0
    try:
        c.use_resident(0)
        for g in range(G):
            c.roles(g, *c.get_roles(g))
        for it in range(steps):
            if it:
                touched = rng.choice(G, int(rng.integers(0, 400)), replace=False)
                for g in touched.tolist():
                    lead, cand = c.get_roles(g)
                    if lead >= 0:
                        s = c.get(g, lead)
                        u = rng.random()
                        grow = int(rng.integers(0, 3)) if u < 0.9 else int(rng.integers(20, 60)) if u < 0.999 else 5000
                        ln = s["logLen"] + grow
                        terms = [SH + 1 + (i * int(s["currentTerm"] - SH)) // ln for i in range(ln)]
                        c.seed(g, lead, s["currentTerm"], s["votedFor"], h.LEADER, s["commitIndex"], terms)
                        c.set_match(g, lead, [int(x) for x in rng.integers(0, ln, n)])
                    if cand >= 0:
                        s = c.get(g, cand)
                        c.seed(g, cand, s["currentTerm"] + int(rng.integers(0, 2)), s["votedFor"], h.CANDIDATE, s["commitIndex"])
            if wide:
                rc, f = c.flatten64(log_cap=1 << 25)
                assert rc == 0, (it, "the test's own flat image is too small")
                exp = o.round64(3, f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                                f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
            else:
                rc, f = c.flatten(log_cap=1 << 25)
                assert rc == 0, (it, "the test's own flat image is too small")
                exp = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                                    f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
            roles_before = [c.get_roles(g) for g in range(G)]
            rc, rep = c.step_resident()
            assert rc == 0, it
            assert c.resident_wide() == wide
            rebuilt += int(rep["rebuilt"])
            moved = (exp.commit != f["commit"]) | (exp.stepdown_mask != 0) | (exp.elected != 0) | (exp.status != 0)
            assert rep["changed"] == int(moved.sum()), it
            for g in range(G):
                lead, cand = roles_before[g]
                if cand >= 0:
                    for p in range(n):
                        assert c.get(g, p)["currentTerm"] == exp.voter_term[p, g], (it, g, p)
                    assert (c.get_roles(g)[0] == cand) == bool(exp.elected[g]), (it, g)
                if lead >= 0:
                    assert c.get(g, lead)["commitIndex"] == exp.commit[g], (it, g)
            for g in range(G):
                lead, cand = c.get_roles(g)
                c.l.rh_group_roles(c.h, g, lead, cand)
        assert rebuilt >= 1
    finally:
        c.close()


@pytest.mark.gpu
def test_resident_driver_switches_to_the_wide_layout():
    """A resident image in the compact layout meets its first value beyond int32 (one candidate's
    term, Go's int is 64-bit): that step lays the image out afresh in int64 elements and decides
    like the wide oracle; later steps stay wide and move only touched rows again."""
    G, n = 1500, 5
    c, b = _build(G, n, seed=77)
    try:
        c.use_resident(0)
        for g in range(G):
            c.roles(g, *c.get_roles(g))
        rc, rep = c.step_resident()
        assert rc == 0 and rep["rebuilt"] and not c.resident_wide()
        for g in range(G):                                          # the batched driver's roles (used by flatten) in step
            c.l.rh_group_roles(c.h, g, *c.get_roles(g))
        g_big = next(g for g in range(G) if c.get_roles(g)[1] >= 0)
        for it in range(2):
            if it == 0:
                cand = c.get_roles(g_big)[1]
                s = c.get(g_big, cand)
                c.seed(g_big, cand, (1 << 40) + 5, s["votedFor"], h.CANDIDATE, s["commitIndex"])
            else:
                for g in (3, 4, 5):
                    lead = c.get_roles(g)[0]
                    if lead >= 0:
                        c.set_match(g, lead, [1] * n)
            rc, f = c.flatten64(log_cap=1 << 22)
            assert rc == 0
            exp = o.round64(3, f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                            f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
            roles_before = [c.get_roles(g) for g in range(G)]
            rc, rep = c.step_resident()
            assert rc == 0 and c.resident_wide()
            assert rep["rebuilt"] == (it == 0)
            for g in range(G):
                lead, cand = roles_before[g]
                if cand >= 0:
                    for p in range(n):
                        assert c.get(g, p)["currentTerm"] == exp.voter_term[p, g], (it, g, p)
                    assert (c.get_roles(g)[0] == cand) == bool(exp.elected[g]), (it, g)
                if lead >= 0:
                    assert c.get(g, lead)["commitIndex"] == exp.commit[g], (it, g)
            for g in range(G):
                lead, cand = c.get_roles(g)
                c.l.rh_group_roles(c.h, g, lead, cand)
        # every voter of that group saw the huge term
        assert all(c.get(g_big, p)["currentTerm"] == (1 << 40) + 5 for p in range(n))
    finally:
        c.close()


@pytest.mark.gpu
def test_resident_driver_without_a_log_arena():
    """All leaders' logs have Raft's own shape (terms never decrease, none above the
    leader's term): the resident image holds no log arena, only the index where each
    leader's term starts and the log's length, and touched groups travel without their
    terms.  The objects must still end where the oracle (which looks terms up) puts
    them.  A log that loses the shape brings the arena back at the next step."""
    G, n = 1500, 5
    c, b = _build(G, n, seed=43, raft_shaped=True)
    rng = np.random.default_rng(11)
    try:
        c.use_resident(0)
        for g in range(G):
            c.roles(g, *c.get_roles(g))
        for it in range(5):
            if it:
                touched = rng.choice(G, 200, replace=False)
                for j, g in enumerate(touched.tolist()):
                    lead, cand = c.get_roles(g)
                    if lead >= 0:
                        s = c.get(g, lead)
                        grow = 200 if (it == 2 and j < 5) else int(rng.integers(0, 4))
                        new_len = s["logLen"] + grow
                        terms = [1 + (i * int(s["currentTerm"])) // new_len for i in range(new_len)]   # 1 .. currentTerm
                        if it == 4 and j == 0:
                            terms[0] = terms[-1] + 5          # a log no Raft leader would hold: decreasing terms
                        c.seed(g, lead, s["currentTerm"], s["votedFor"], h.LEADER, s["commitIndex"], terms)
                        c.set_match(g, lead, [int(x) for x in rng.integers(0, new_len + 2, n)])    # some beyond the log
                    if cand >= 0:
                        s = c.get(g, cand)
                        c.seed(g, cand, s["currentTerm"] + int(rng.integers(0, 2)), s["votedFor"], h.CANDIDATE, s["commitIndex"])
            rc, f = c.flatten(log_cap=1 << 22)
            assert rc == 0
            exp = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                                f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
            roles_before = [c.get_roles(g) for g in range(G)]
            rc, rep = c.step_resident()
            assert rc == 0
            assert rep["rebuilt"] == (it in (0, 4))               # first image; the log that lost the shape
            assert c.without_arena() == (it < 4)
            moved = (exp.commit != f["commit"]) | (exp.stepdown_mask != 0) | (exp.elected != 0) | (exp.status != 0)
            assert rep["changed"] == int(moved.sum())
            assert sorted(c.flagged()) == sorted((int(g), int(exp.status[g])) for g in np.nonzero(exp.status)[0])
            for g in range(G):
                lead, cand = roles_before[g]
                if cand >= 0:
                    for p in range(n):
                        assert c.get(g, p)["currentTerm"] == exp.voter_term[p, g], (it, g, p)
                    assert (c.get_roles(g)[0] == cand) == bool(exp.elected[g]), (it, g)
                if lead >= 0:
                    assert c.get(g, lead)["commitIndex"] == exp.commit[g], (it, g)
            for g in range(G):
                lead, cand = c.get_roles(g)
                c.l.rh_group_roles(c.h, g, lead, cand)
    finally:
        c.close()


@pytest.mark.gpu
def test_resident_driver_moves_many_rows_on_several_threads():
    """Thousands of touched groups per step: the rows are flattened and applied on several host
    threads (from 512 rows on).  Every object must still end where the oracle puts it, the
    groups be reported in order, and a re-marked group (a new leader) go up again."""
    G, n = 12000, 5
    c, b = _build(G, n, seed=47, raft_shaped=True)
    rng = np.random.default_rng(13)
    try:
        c.use_resident(0)
        for g in range(G):
            c.roles(g, *c.get_roles(g))
        for it in range(4):
            if it:
                touched = rng.choice(G, 6000, replace=False)
                for g in touched.tolist():
                    lead, cand = c.get_roles(g)
                    if lead >= 0:
                        s = c.get(g, lead)
                        new_len = s["logLen"] + int(rng.integers(0, 4))
                        terms = [1 + (i * int(s["currentTerm"])) // new_len for i in range(new_len)]
                        c.seed(g, lead, s["currentTerm"], s["votedFor"], h.LEADER, s["commitIndex"], terms)
                        c.set_match(g, lead, [int(x) for x in rng.integers(0, new_len + 2, n)])
                    if cand >= 0:
                        s = c.get(g, cand)
                        c.seed(g, cand, s["currentTerm"] + int(rng.integers(0, 2)), s["votedFor"], h.CANDIDATE, s["commitIndex"])
            rc, f = c.flatten(log_cap=1 << 23)
            assert rc == 0
            exp = o.fused_round(f["match"], f["commit"], f["leader_term"], f["log_off"], f["log_term"],
                                f["voter_term"], f["voted_for"], f["last_log_term"], f["cand_term"], f["cand_id"])
            roles_before = [c.get_roles(g) for g in range(G)]
            rc, rep = c.step_resident()
            assert rc == 0
            assert rep["rebuilt"] == (it == 0) and c.without_arena()
            if it:
                assert rep["uploaded"] >= 6000            # the touched rows + the ones the last step re-marked
            moved = (exp.commit != f["commit"]) | (exp.stepdown_mask != 0) | (exp.elected != 0) | (exp.status != 0)
            assert rep["changed"] == int(moved.sum())
            if it == 1:
                assert rep["changed"] >= 2048             # the apply side ran on several threads too
            assert c.flagged() == [(int(g), int(exp.status[g])) for g in np.nonzero(exp.status)[0]]   # in group order
            for g in range(G):
                lead, cand = roles_before[g]
                if cand >= 0:
                    for p in range(n):
                        assert c.get(g, p)["currentTerm"] == exp.voter_term[p, g], (it, g, p)
                    assert (c.get_roles(g)[0] == cand) == bool(exp.elected[g]), (it, g)
                if lead >= 0:
                    assert c.get(g, lead)["commitIndex"] == exp.commit[g], (it, g)
            for g in range(G):
                lead, cand = c.get_roles(g)
                c.l.rh_group_roles(c.h, g, lead, cand)
    finally:
        c.close()
