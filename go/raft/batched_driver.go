package raft

import (
	"errors"
	"math"
	"strconv"
)

// Group is one Raft group as the driver sees it: its members in slot order,
// the member currently leading (nil if none) and the member currently asking
// for votes (nil if none).
type Group struct {
	Members   []*Node
	Leader    *Node
	Candidate *Node
}

// word is the element type of a layout: int32 (compact) or int64 (wide, the
// full range of Go's int on amd64).
type word interface{ ~int32 | ~int64 }

// roundBuffers is the structure-of-arrays image of all groups:
// plane[p*ld+g] for slot p of group g.
type roundBuffers[T word] struct {
	groups, slots, ld int
	match, commit, leaderTerm []T
	logOff                    []uint32
	logTerm                   []T
	voterTerm, votedFor, lastLogTerm, candTerm, candID []T
	commitOut, voterTermOut, maxTerm                   []T
	grantMask, stepdownMask                            []uint16
	elected, status                                    []uint8
	// groups smaller than slots occupy the low slots; memberMask[g] = 1<<size - 1
	memberMask []uint16
	mixed      bool // some group has fewer than slots members: the round needs the masks
}

// limits of a layout: its largest value, a candidate term no voter reacts to,
// and the empty-log marker (RQ_LOG_EMPTY / RQ_LOG_EMPTY64), which is the one
// value of the element type that cannot be stored as a log term.
type layout[T word] struct {
	max, belowAll, logEmpty T
	fits                    func(v int) bool
}

var compact = layout[int32]{math.MaxInt32, math.MinInt32 + 1, math.MinInt32,
	func(v int) bool { return v > math.MinInt32 && v <= math.MaxInt32 }}
var wide = layout[int64]{math.MaxInt64, math.MinInt64, math.MinInt64,
	func(v int) bool { return true }}

var errRange = errors.New("raft: value outside the range of the engine layout")

// GroupStatus names a group whose round hit a condition that is a runtime panic
// in the per-node methods: Status carries the RQ_ST_* bits (2: a voter reached
// lastEntry(log) with an empty log, raft.go:204/:235; 1: the leader's quorum index
// lies outside its log, the indexing convention of raft.go:151).
type GroupStatus struct {
	Group  int
	Status uint8
}

// PanicError is returned by Step when some groups were flagged. The round has
// been applied for every group; a flagged group keeps what the reference had
// done before its panic (voters' terms raised and roles set, raft.go:219-221, no
// grant from the panicking voter, no commit advance).
type PanicError struct{ Groups []GroupStatus }

func (e *PanicError) Error() string {
	return "raft: " + strconv.Itoa(len(e.Groups)) + " group(s) hit a condition that panics in the per-node methods"
}

func flagged(status []uint8) error {
	var fl []GroupStatus
	for g, st := range status {
		if st != 0 {
			fl = append(fl, GroupStatus{g, st})
		}
	}
	if fl == nil {
		return nil
	}
	return &PanicError{fl}
}

// BatchedDriver evaluates one quorum round for every registered group in one
// GPU launch and applies the decisions to the Go objects.
type BatchedDriver struct {
	eng    *rqEngine
	groups []Group
	slots  int
	// LastStepWide reports whether the last Step needed the 64-bit layout.
	LastStepWide bool
}

func NewBatchedDriver(device int, groups []Group, slots int, maxLogTerms uint64) (*BatchedDriver, error) {
	eng, err := newRqEngine(device, uint32(len(groups)), uint32(slots), maxLogTerms)
	if err != nil {
		return nil, err
	}
	return &BatchedDriver{eng: eng, groups: groups, slots: slots}, nil
}

// flatten builds the SoA image in layout L. A group without a leader gets
// commit = the layout's maximum (N > commit is never true); a group without a
// candidate gets the layout's belowAll term (no voter's term moves and the
// vote outputs of such a group are not applied).
func flatten[T word](d *BatchedDriver, L layout[T]) (*roundBuffers[T], error) {
	G, n := len(d.groups), d.slots
	b := &roundBuffers[T]{groups: G, slots: n, ld: G}
	plane := func() []T { return make([]T, n*G) }
	b.match, b.voterTerm, b.votedFor, b.lastLogTerm, b.voterTermOut = plane(), plane(), plane(), plane(), plane()
	vec := func() []T { return make([]T, G) }
	b.commit, b.leaderTerm, b.candTerm, b.candID, b.commitOut, b.maxTerm = vec(), vec(), vec(), vec(), vec(), vec()
	b.grantMask, b.stepdownMask = make([]uint16, G), make([]uint16, G)
	b.elected, b.status = make([]uint8, G), make([]uint8, G)
	b.logOff = make([]uint32, G+1)
	b.memberMask = make([]uint16, G)
	put := func(dst *T, v int) bool {
		if !L.fits(v) {
			return false
		}
		*dst = T(v)
		return true
	}
	for g, grp := range d.groups {
		k := len(grp.Members)
		if k < 1 || k > n || n > 16 {
			return nil, errors.New("raft: a group needs 1..slots members (slots <= 16)")
		}
		b.memberMask[g] = uint16(1)<<uint(k) - 1
		if k != n {
			b.mixed = true
		}
		b.commit[g], b.candTerm[g] = L.max, L.belowAll
		if l := grp.Leader; l != nil {
			if !put(&b.commit[g], l.commitIndex) || !put(&b.leaderTerm[g], l.currentTerm) {
				return nil, errRange
			}
			for p := 0; p < k && p < len(l.matchIndex); p++ {
				if !put(&b.match[p*G+g], l.matchIndex[p]) {
					return nil, errRange
				}
			}
			for _, e := range l.log {
				var t T
				if !put(&t, e.TermNum) {
					return nil, errRange
				}
				b.logTerm = append(b.logTerm, t)
			}
		}
		b.logOff[g+1] = uint32(len(b.logTerm))
		if c := grp.Candidate; c != nil {
			if !put(&b.candTerm[g], c.currentTerm) || !put(&b.candID[g], c.id) {
				return nil, errRange
			}
		}
		for p, m := range grp.Members {
			i := p*G + g
			if !put(&b.voterTerm[i], m.currentTerm) || !put(&b.votedFor[i], m.votedFor) {
				return nil, errRange
			}
			b.lastLogTerm[i] = L.logEmpty
			if len(m.log) > 0 {
				if !put(&b.lastLogTerm[i], m.log[len(m.log)-1].TermNum) || b.lastLogTerm[i] == L.logEmpty {
					return nil, errRange
				}
			}
		}
	}
	if len(b.logTerm) == 0 {
		b.logTerm = []T{0}
	}
	return b, nil
}

// apply writes one round's decisions back: voters' terms and roles
// (raft.go:219-221), the leader's commitIndex, and BecomeLeader for every
// candidate that won.
func apply[T word](d *BatchedDriver, b *roundBuffers[T]) {
	G := b.groups
	for g, grp := range d.groups {
		if grp.Candidate != nil {
			for p, m := range grp.Members {
				m.currentTerm = int(b.voterTermOut[p*G+g])
				if b.stepdownMask[g]>>uint(p)&1 == 1 {
					m.nodeType = Follower
				}
			}
			if b.elected[g] == 1 {
				grp.Candidate.BecomeLeader()
				d.groups[g].Leader, d.groups[g].Candidate = grp.Candidate, nil
			}
		}
		if grp.Leader != nil {
			grp.Leader.commitIndex = int(b.commitOut[g])
		}
	}
}

// Step runs one fused round and applies it. The compact layout is tried first;
// if some value does not fit int32 the round runs in the wide layout. Groups
// of different sizes run through the member-mask round (old = new = members).
// A *PanicError lists the groups where the per-node methods would have panicked
// (a cluster straight from NewNode has empty logs, raft.go:87, and panics in
// RequestVoteRPC); the round is applied either way.
//
// Step flattens every Node and moves the whole image over PCIe both ways: fine
// for thousands of groups, milliseconds for a million. ResidentDriver keeps the
// image in HBM and moves only the groups the host touched.
func (d *BatchedDriver) Step() error {
	b, err := flatten(d, compact)
	if err == errRange {
		w, err := flatten(d, wide)
		if err != nil {
			return err
		}
		if err := d.eng.round64(w); err != nil {
			return err
		}
		apply(d, w)
		d.LastStepWide = true
		return flagged(w.status)
	}
	if err != nil {
		return err
	}
	if err := d.eng.fusedRound(b); err != nil {
		return err
	}
	apply(d, b)
	d.LastStepWide = false
	return flagged(b.status)
}

// Close releases the GPU engine.
func (d *BatchedDriver) Close() { d.eng.close() }
