// Package raft keeps the type and method surface of tawawhite/raft (raft.go)
// and adds a batched driver (batched_driver.go) that evaluates the per-group
// quorum decisions of many groups at once on an MI355X through librq.so.
//
// NOTE: this package could not be compiled in the build environment (no Go
// toolchain). Its logic is mirrored and tested in C (oracle/raft_oracle.c)
// and through the C-ABI (tests/test_gpu_parity.py).
package raft

// NodeType is the role of a node. Leader is the zero value, as in the
// reference (raft.go:7-13).
type NodeType int

const (
	Leader NodeType = iota
	Follower
	Candidate
)

// Entry is one log record (raft.go:71-75).
type Entry struct {
	Command string
	Index   int
	TermNum int
}

// Node holds the per-server state of raft.go:15-69. Fields stay unexported, so
// the batched driver lives in this package.
type Node struct {
	id           int
	nodeType     NodeType
	stateMachine func(string)
	peers        []*Node // every member of the cluster, this node included

	currentTerm int
	votedFor    int // -1 means "none"
	log         []Entry

	commitIndex int
	lastApplied int

	nextIndex  []int
	matchIndex []int
}

// NewNode creates a follower with an empty log and registers it with the
// peers given, so that afterwards every member shares one peer list that
// includes the new node (raft.go:77-99).
func NewNode(id int, peers []*Node, statemachine func(string)) *Node {
	n := &Node{
		id:           id,
		stateMachine: statemachine,
		nodeType:     Follower,
		votedFor:     -1,
		log:          []Entry{},
	}
	all := append(peers, n)
	for _, member := range all {
		member.peers = all
	}
	return n
}

// BecomeLeader switches role and sizes the leader-only arrays over all
// members, self slot included (raft.go:101-118).
func (n *Node) BecomeLeader() {
	n.nodeType = Leader
	size := len(n.peers)
	n.nextIndex = make([]int, size)
	n.matchIndex = make([]int, size)
	for i := 0; i < size; i++ {
		n.nextIndex[i] = len(n.log) + 1
	}
}

// BecomeFollower / BecomeCandidate only change the role and drop the leader
// arrays; a candidate does not bump its term here (raft.go:120-130).
func (n *Node) BecomeFollower() {
	n.nodeType = Follower
	n.nextIndex, n.matchIndex = nil, nil
}

func (n *Node) BecomeCandidate() {
	n.nodeType = Candidate
	n.nextIndex, n.matchIndex = nil, nil
}

// abdicateIfStale is testToAbdicateLeadership (raft.go:212-223).
func (n *Node) abdicateIfStale(term int) {
	if term > n.currentTerm {
		n.currentTerm = term
		n.nodeType = Follower
	}
}

// RequestVoteRPC answers one vote request exactly as raft.go:181-210 does:
// lastLogIndex and lastLogTerm are accepted and ignored, the up-to-date test
// compares this node's last log term with the candidate's term, and votedFor
// is never written.
func (n *Node) RequestVoteRPC(term, candidateId, lastLogIndex, lastLogTerm int) (int, bool) {
	n.abdicateIfStale(term)
	if term < n.currentTerm {
		return n.currentTerm, false
	}
	free := n.votedFor == -1 || n.votedFor == candidateId
	upToDate := n.log[len(n.log)-1].TermNum <= term // panics on an empty log, like lastEntry
	return n.currentTerm, free && upToDate
}

// AppendEntriesRPC keeps the reference's observable behaviour (raft.go:132-179)
// including its index panics: the range test at raft.go:159 is inverted, so a new
// entry whose Index is at or past len(log) indexes out of range and panics.
func (n *Node) AppendEntriesRPC(term, leaderId, prevLogIndex, prevLogTerm int,
	newEntries []Entry, leaderCommit int) (int, bool) {
	n.abdicateIfStale(term)
	if term < n.currentTerm {
		return n.currentTerm, false
	}
	if n.log[prevLogIndex].TermNum != prevLogTerm {
		return n.currentTerm, false
	}
	for _, e := range newEntries {
		if len(n.log) <= e.Index {
			_ = n.log[e.Index] // index out of range, as in the reference
		}
	}
	n.log = append(n.log, newEntries...)
	if leaderCommit > n.commitIndex {
		last := newEntries[len(newEntries)-1].Index
		if leaderCommit < last {
			n.commitIndex = leaderCommit
		} else {
			n.commitIndex = last
		}
	}
	return n.currentTerm, true
}
