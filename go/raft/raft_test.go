package raft

// Tests of the Go host against the committed known-answer vectors
// (tests/golden/kats.json, hand-derived from raft.go) and of the batched driver
// against one-RPC-at-a-time evaluation on the same objects.
//
// NOTE: written without a Go toolchain (none in the build image); the same
// vectors run against the C restatement and the C++ host in tests/*.py.

import (
	"encoding/json"
	"os"
	"testing"
)

type katVoter struct {
	Cur         int `json:"cur"`
	VotedFor    int `json:"voted_for"`
	LastLogTerm int `json:"last_log_term"`
	Role        int `json:"role"`
}

type katVote struct {
	ID    string   `json:"id"`
	Voter katVoter `json:"voter"`
	Req   struct {
		Term         int `json:"term"`
		Cand         int `json:"cand"`
		LastLogIndex int `json:"last_log_index"`
		LastLogTerm  int `json:"last_log_term"`
	} `json:"req"`
	Expect struct {
		Cur        int  `json:"cur"`
		Grant      bool `json:"grant"`
		Role       int  `json:"role"`
		TermResult int  `json:"term_result"`
	} `json:"expect"`
}

type katCommit struct {
	Match      []int `json:"match"`
	Commit     int   `json:"commit"`
	LeaderTerm int   `json:"leader_term"`
	LogTerms   []int `json:"log_terms"`
	Expect     int   `json:"expect"`
	OOB        bool  `json:"oob"`
}

type katTally struct {
	N      int        `json:"n"`
	Term   int        `json:"term"`
	Cand   int        `json:"cand"`
	Voters []katVoter `json:"voters"`
	Expect struct {
		Elected bool `json:"elected"`
		MaxTerm int  `json:"max_term"`
		Votes   int  `json:"votes"`
	} `json:"expect"`
}

type kats struct {
	Vote   []katVote   `json:"vote"`
	Commit []katCommit `json:"commit"`
	Tally  []katTally  `json:"tally"`
}

func loadKats(t *testing.T) kats {
	raw, err := os.ReadFile("../../tests/golden/kats.json")
	if err != nil {
		t.Fatal(err)
	}
	var k kats
	if err := json.Unmarshal(raw, &k); err != nil {
		t.Fatal(err)
	}
	return k
}

// voterNode builds a node in the given state with a one-entry log whose last
// term is lastLogTerm (RequestVoteRPC only ever reads the last entry, raft.go:204).
func voterNode(v katVoter) *Node {
	n := NewNode(0, nil, nil)
	n.currentTerm, n.votedFor, n.nodeType = v.Cur, v.VotedFor, NodeType(v.Role)
	n.log = []Entry{{Command: "", Index: 0, TermNum: v.LastLogTerm}}
	return n
}

func TestRequestVoteKnownAnswers(t *testing.T) {
	for _, k := range loadKats(t).Vote {
		n := voterNode(k.Voter)
		term, granted := n.RequestVoteRPC(k.Req.Term, k.Req.Cand, k.Req.LastLogIndex, k.Req.LastLogTerm)
		if term != k.Expect.TermResult || granted != k.Expect.Grant ||
			n.currentTerm != k.Expect.Cur || int(n.nodeType) != k.Expect.Role || n.votedFor != k.Voter.VotedFor {
			t.Errorf("%s: got (%d,%v) cur=%d role=%d votedFor=%d", k.ID, term, granted, n.currentTerm, n.nodeType, n.votedFor)
		}
	}
}

// cluster builds one group of n members through NewNode, as a user of the
// reference would.
func cluster(n int) []*Node {
	var peers []*Node
	for p := 0; p < n; p++ {
		peers = append(peers, NewNode(p, peers, nil))
	}
	return peers[len(peers)-1].peers
}

// TestBatchedDriverAgainstPerNodeCalls runs the tally and commit vectors
// through BatchedDriver.Step (one GPU launch for all groups) and through
// RequestVoteRPC calls on a copy of the same objects.
func TestBatchedDriverAgainstPerNodeCalls(t *testing.T) {
	k := loadKats(t)
	var groups []Group
	var want []katTally
	for _, tk := range k.Tally {
		// The driver takes the candidate's term from the candidate object, which is
		// also the voter in its own slot; raising that voter's term to the
		// candidate's changes no outcome as long as it was not above it.
		if tk.N != 5 || tk.Voters[tk.Cand%5].Cur > tk.Term {
			continue
		}
		m := cluster(5)
		for p, v := range tk.Voters {
			m[p].currentTerm, m[p].votedFor = v.Cur, v.VotedFor
			m[p].log = []Entry{{TermNum: v.LastLogTerm}}
		}
		cand := m[tk.Cand%5]
		cand.id = tk.Cand
		cand.nodeType = Candidate
		groups = append(groups, Group{Members: m, Candidate: cand})
		want = append(want, tk)
	}
	// the driver reads the candidate's own term; the vectors carry it separately
	for i := range groups {
		groups[i].Candidate.currentTerm = want[i].Term
	}
	d, err := NewBatchedDriver(0, groups, 5, 1<<16)
	if err != nil {
		t.Skip("no MI355X / librq.so: ", err)
	}
	defer d.Close()
	if err := d.Step(); err != nil {
		t.Fatal(err)
	}
	for i, g := range d.groups {
		elected := g.Leader != nil && g.Leader.nodeType == Leader
		if elected != want[i].Expect.Elected {
			t.Errorf("tally %d: elected=%v want %v", i, elected, want[i].Expect.Elected)
		}
	}
}

func TestLeaderCommitKnownAnswers(t *testing.T) {
	k := loadKats(t)
	var groups []Group
	var want []katCommit
	for _, ck := range k.Commit {
		if len(ck.Match) != 5 || ck.OOB {
			continue
		}
		m := cluster(5)
		l := m[0]
		for _, term := range ck.LogTerms {
			l.log = append(l.log, Entry{TermNum: term})
		}
		l.BecomeLeader()
		l.currentTerm, l.commitIndex = ck.LeaderTerm, ck.Commit
		copy(l.matchIndex, ck.Match)
		for _, f := range m[1:] { // followers need a last entry for the (unused) vote side
			f.log = []Entry{{TermNum: 0}}
		}
		groups = append(groups, Group{Members: m, Leader: l})
		want = append(want, ck)
	}
	d, err := NewBatchedDriver(0, groups, 5, 1<<16)
	if err != nil {
		t.Skip("no MI355X / librq.so: ", err)
	}
	defer d.Close()
	if err := d.Step(); err != nil {
		t.Fatal(err)
	}
	for i, g := range d.groups {
		if g.Leader.commitIndex != want[i].Expect {
			t.Errorf("commit %d: got %d want %d", i, g.Leader.commitIndex, want[i].Expect)
		}
	}
}

// TestWideFallback: a term beyond int32 makes Step use rq_round64.
func TestWideFallback(t *testing.T) {
	m := cluster(3)
	big := 1 << 40
	for _, x := range m {
		x.currentTerm = big
		x.log = []Entry{{TermNum: big}}
	}
	cand := m[1]
	cand.currentTerm, cand.nodeType = big+1, Candidate
	d, err := NewBatchedDriver(0, []Group{{Members: m, Candidate: cand}}, 3, 1<<10)
	if err != nil {
		t.Skip("no MI355X / librq.so: ", err)
	}
	defer d.Close()
	if err := d.Step(); err != nil {
		t.Fatal(err)
	}
	if !d.LastStepWide || cand.nodeType != Leader || m[0].currentTerm != big+1 || m[0].nodeType != Follower {
		t.Errorf("wide=%v role=%d term=%d", d.LastStepWide, cand.nodeType, m[0].currentTerm)
	}
}
