#!/usr/bin/env python3
"""Writes tests/golden/kats.json: known-answer vectors for the quorum path.

The reference (tawawhite/raft) ships no tests, fixtures or golden files and no
Go toolchain exists in this image, so these vectors are HAND-DERIVED by
reading raft.go; each carries the lines it pins.  Nothing here runs reference
code.  Roles use the reference's iota values: Leader=0, Follower=1,
Candidate=2 (raft.go:7-13).

Run:  python tests/golden/make_kats.py
"""
import json
import os

L, F, C = 0, 1, 2
ANY = 0  # lastLogIndex / lastLogTerm are never read by RequestVoteRPC (raft.go:181-210)


def v(id_, voter, req, expect, pins):
    cur, voted_for, llt, role = voter
    term, cand, lli, lltc = req
    tr, grant, cur2, role2 = expect
    return dict(id=id_, voter=dict(cur=cur, voted_for=voted_for, last_log_term=llt, role=role),
                req=dict(term=term, cand=cand, last_log_index=lli, last_log_term=lltc),
                expect=dict(term_result=tr, grant=grant, cur=cur2, role=role2), pins=pins)


VOTE = [
    v("V1", (1, -1, 1, F), (2, 7, ANY, ANY), (2, True, 2, F), "219-221 raise term; 202 null vote; 204 1<=2"),
    v("V2", (3, -1, 1, F), (2, 7, ANY, ANY), (3, False, 3, F), "190-192 stale term rejected"),
    v("V3", (2, 4, 1, F), (2, 7, ANY, ANY), (2, False, 2, F), "202-203 voted for another"),
    v("V4", (2, 7, 2, F), (2, 7, ANY, ANY), (2, True, 2, F), "203 same candidate; 204 uses <="),
    v("V5", (2, -1, 3, F), (2, 7, ANY, ANY), (2, False, 2, F), "204 voter last log term 3 > term 2"),
    v("V6", (2, -1, 2, F), (3, 1, 0, 0), (3, True, 3, F), "204 compares with TERM: a stale candidate log (0,0) is still granted"),
    v("V7", (1, 4, 1, F), (5, 7, ANY, ANY), (5, False, 5, F), "219-222 never clears votedFor on a new term"),
    v("V8", (2, -1, 1, F), (2, 8, ANY, ANY), (2, True, 2, F), "205-207 never records the vote: V1's voter grants a second candidate in term 2"),
    v("V9", (2, -1, 1, L), (3, 7, ANY, ANY), (3, True, 3, F), "221 leader steps down on higher term"),
    v("V10", (3, 3, 3, C), (3, 5, ANY, ANY), (3, False, 3, C), "equal term: candidate keeps its role, voted for itself"),
    v("V11", (2, -1, 1, L), (2, 9, ANY, ANY), (2, True, 2, L), "219 strict >: equal term leaves a leader a leader, and it still grants"),
    v("V12", (2, -1, 2, F), (2, -1, ANY, ANY), (2, True, 2, F), "202-203 candidate id -1 collides with the null sentinel"),
]

# A voter asked twice in a row (state carried between calls).
VOTE_SEQUENCES = [
    dict(id="S1", voter=dict(cur=1, voted_for=-1, last_log_term=1, role=F),
         calls=[dict(req=dict(term=2, cand=7), expect=dict(term_result=2, grant=True)),
                dict(req=dict(term=2, cand=8), expect=dict(term_result=2, grant=True))],
         final=dict(cur=2, voted_for=-1, role=F), pins="V1 then V8 on one node: double grant in a term"),
]

# Empty voter log: lastEntry panics (raft.go:204 -> 235) unless 190-192 returned first.
VOTE_EMPTY_LOG = [
    dict(id="E1", voter=dict(cur=1, voted_for=-1, role=F), req=dict(term=2, cand=7),
         expect=dict(panic=True, cur=2, role=F), pins="204/235 index panic after 219-221 already ran"),
    dict(id="E2", voter=dict(cur=3, voted_for=-1, role=F), req=dict(term=2, cand=7),
         expect=dict(panic=False, term_result=3, grant=False, cur=3, role=F), pins="190-192 returns before lastEntry"),
]

# Quorum index (build-defined: ascending-sorted match[(n-1)/2]; n counts self, raft.go:114-117).
QUORUM = [
    dict(match=[7, 3, 9, 5, 0], expect=5),
    dict(match=[0, 0, 4], expect=0),
    dict(match=[0, 0, 0], expect=0, pins="fresh leader, 114-117 zero init"),
    dict(match=[5, 5, 2, 2], expect=2),
    dict(match=[8, 3], expect=3),
    dict(match=[6], expect=6),
    dict(match=[1, 2, 3, 4, 5, 6, 7], expect=4),
    dict(match=[9, 1, 8, 2, 7, 3, 6, 4, 5], expect=5),
    dict(match=[4, 4, 4, 1, 1, 1], expect=1),
    dict(match=[3, 9, 9, 9, 1, 0, 2, 9], expect=3),
]

# Leader commit rule: commit' = (N > commit && log[N].term == leaderTerm) ? N : commit,
# log indexed directly by slot (raft.go:151).
LOG6 = [0, 1, 1, 2, 2, 3]           # log[i].TermNum, slot 0 is a sentinel
COMMIT = [
    dict(match=[7, 3, 9, 5, 0], commit=4, leader_term=3, log_terms=LOG6, expect=5, oob=False, pins="N=5, term_at(5)=3 == leaderTerm"),
    dict(match=[7, 3, 9, 5, 0], commit=4, leader_term=4, log_terms=LOG6, expect=4, oob=False, pins="term gate fails"),
    dict(match=[7, 3, 9, 5, 0], commit=5, leader_term=3, log_terms=LOG6, expect=5, oob=False, pins="needs N > commit"),
    dict(match=[0, 0, 4], commit=0, leader_term=2, log_terms=LOG6, expect=0, oob=False, pins="no majority beyond 0"),
    dict(match=[0, 0, 0], commit=0, leader_term=1, log_terms=LOG6, expect=0, oob=False, pins="fresh leader"),
    dict(match=[9, 9, 9], commit=2, leader_term=3, log_terms=LOG6, expect=2, oob=True, pins="N=9 outside the 6-slot log: index panic in Go, status bit here"),
    dict(match=[9, 9, 9], commit=9, leader_term=3, log_terms=LOG6, expect=9, oob=False, pins="N > commit is false, log[N] never evaluated"),
    dict(match=[2, 2, 2], commit=1, leader_term=1, log_terms=[], expect=1, oob=True, pins="empty leader log"),
    dict(match=[3, 4, 5], commit=-1, leader_term=2, log_terms=LOG6, expect=4, oob=False, pins="negative commit"),
]

# Candidate tally: elected = 2*votes > n && max termResult == candTerm.
def _voters(grants, n, t, higher=0):
    vs = []
    for p in range(n):
        if p < higher:
            vs.append(dict(cur=t + 1, voted_for=-1, last_log_term=0))
        elif p < higher + grants:
            vs.append(dict(cur=t - 1, voted_for=-1, last_log_term=0))
        else:
            vs.append(dict(cur=t, voted_for=99, last_log_term=0))
    return vs


TALLY = [
    dict(n=5, term=4, cand=2, voters=_voters(3, 5, 4), expect=dict(votes=3, elected=True, max_term=4)),
    dict(n=5, term=4, cand=2, voters=_voters(2, 5, 4), expect=dict(votes=2, elected=False, max_term=4)),
    dict(n=4, term=4, cand=2, voters=_voters(2, 4, 4), expect=dict(votes=2, elected=False, max_term=4)),
    dict(n=4, term=4, cand=2, voters=_voters(3, 4, 4), expect=dict(votes=3, elected=True, max_term=4)),
    dict(n=3, term=4, cand=2, voters=_voters(2, 3, 4), expect=dict(votes=2, elected=True, max_term=4)),
    dict(n=3, term=4, cand=2, voters=_voters(1, 3, 4), expect=dict(votes=1, elected=False, max_term=4)),
    dict(n=5, term=4, cand=2, voters=_voters(4, 5, 4, higher=1), expect=dict(votes=4, elected=False, max_term=5),
         pins="a reply with a higher term voids the election"),
    dict(n=1, term=1, cand=0, voters=_voters(1, 1, 1), expect=dict(votes=1, elected=True, max_term=1)),
]

# Follower clamp (raft.go:172-176, 226-231).
CLAMP = [
    dict(leader_commit=5, commit=3, last_new_index=4, expect=4),
    dict(leader_commit=5, commit=3, last_new_index=9, expect=5),
    dict(leader_commit=3, commit=3, last_new_index=9, expect=3, pins="174 strict >"),
    dict(leader_commit=5, commit=4, last_new_index=2, expect=2, pins="175 can LOWER commitIndex"),
]

# AppendEntriesRPC behaviours (raft.go:132-179) on a node whose log is pre-seeded.
APPEND = [
    dict(id="A1", log_terms=[], cur=0, term=1, prev_index=0, prev_term=0, new=[], leader_commit=0,
         expect=dict(panic=True), pins="151 indexes an empty log (NewNode :87)"),
    dict(id="A2", log_terms=[0, 1], cur=3, term=2, prev_index=1, prev_term=1, new=[], leader_commit=0,
         expect=dict(panic=False, term_result=3, success=False), pins="145-147"),
    dict(id="A3", log_terms=[0, 1], cur=1, term=1, prev_index=1, prev_term=2, new=[], leader_commit=0,
         expect=dict(panic=False, term_result=1, success=False), pins="151-153 term mismatch"),
    dict(id="A4", log_terms=[0, 1], cur=1, term=1, prev_index=1, prev_term=1, new=[[2, 1]], leader_commit=0,
         expect=dict(panic=True), pins="159-161 new index >= len(log) always panics"),
    dict(id="A5", log_terms=[0, 1, 1], cur=1, term=1, prev_index=1, prev_term=1, new=[[1, 1]], leader_commit=2,
         expect=dict(panic=False, term_result=1, success=True, log_len=4, commit=1),
         pins="170 appends a duplicate; 175 min(leaderCommit, last new index)"),
    dict(id="A6", log_terms=[0, 1], cur=1, term=1, prev_index=1, prev_term=1, new=[], leader_commit=1,
         expect=dict(panic=True), pins="175 lastEntry(empty newEntries) on a heartbeat"),
    dict(id="A7", log_terms=[0, 1], cur=1, term=1, prev_index=1, prev_term=1, new=[], leader_commit=0,
         expect=dict(panic=False, term_result=1, success=True, log_len=2, commit=0), pins="heartbeat, nothing to commit"),
    dict(id="A8", log_terms=[0, 1], cur=1, term=4, prev_index=2, prev_term=1, new=[], leader_commit=0,
         expect=dict(panic=True, cur=4), pins="142 raises the term before 151 panics"),
]

OUT = dict(
    source="hand-derived from tawawhite/raft raft.go; the reference ships no tests (parity unpinned by the reference)",
    roles=dict(Leader=L, Follower=F, Candidate=C),
    vote=VOTE, vote_sequences=VOTE_SEQUENCES, vote_empty_log=VOTE_EMPTY_LOG,
    quorum=QUORUM, commit=COMMIT, tally=TALLY, clamp=CLAMP, append=APPEND,
)

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")
    with open(path, "w") as f:
        json.dump(OUT, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)
