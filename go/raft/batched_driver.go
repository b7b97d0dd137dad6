package raft

import (
	"errors"
	"math"
	"runtime"
	"strconv"
	"sync"
	"sync/atomic"
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
	// Step's flat images, kept between calls (176 MB at 1M x 5: allocating and
	// zeroing them afresh costs more than the round trip over PCIe)
	buf32 roundBuffers[int32]
	buf64 roundBuffers[int64]
}

// parallelOverGroups runs fn(first, last) for disjoint group ranges on up to 16
// goroutines (one call on the caller's goroutine below 32768 groups). Groups are
// independent objects; the reference itself has no goroutines (raft.go), this
// only concerns the flattening of its objects.
func parallelOverGroups(groups int, fn func(g0, g1 int)) {
	workers := 1
	if groups >= 32768 {
		workers = runtime.GOMAXPROCS(0)
		if workers > 16 {
			workers = 16
		}
	}
	if workers <= 1 {
		fn(0, groups)
		return
	}
	var wg sync.WaitGroup
	for w := 0; w < workers; w++ {
		wg.Add(1)
		go func(g0, g1 int) {
			defer wg.Done()
			fn(g0, g1)
		}(groups*w/workers, groups*(w+1)/workers)
	}
	wg.Wait()
}

// sized returns s with length n, reusing its memory when it is large enough.
func sized[E any](s []E, n int) []E {
	if cap(s) >= n {
		return s[:n]
	}
	return make([]E, n)
}

func NewBatchedDriver(device int, groups []Group, slots int, maxLogTerms uint64) (*BatchedDriver, error) {
	eng, err := newRqEngine(device, uint32(len(groups)), uint32(slots), maxLogTerms)
	if err != nil {
		return nil, err
	}
	return &BatchedDriver{eng: eng, groups: groups, slots: slots}, nil
}

// flatten builds the SoA image in layout L into b (a buffer kept from the last
// call is reused as it is: every element is written). A group without a leader
// gets commit = the layout's maximum (N > commit is never true); a group without
// a candidate gets the layout's belowAll term (no voter's term moves and the vote
// outputs of such a group are not applied). Two passes over the groups, both in
// parallel: the leaders' log lengths (then their offsets in the arena), then
// every group's slots.
func flatten[T word](d *BatchedDriver, L layout[T], b *roundBuffers[T]) error {
	G, n := len(d.groups), d.slots
	if n < 1 || n > 16 {
		return errors.New("raft: a group needs 1..slots members (slots <= 16)")
	}
	b.groups, b.slots, b.ld, b.mixed = G, n, G, false
	b.match, b.voterTerm, b.votedFor = sized(b.match, n*G), sized(b.voterTerm, n*G), sized(b.votedFor, n*G)
	b.lastLogTerm, b.voterTermOut = sized(b.lastLogTerm, n*G), sized(b.voterTermOut, n*G)
	b.commit, b.leaderTerm, b.candTerm, b.candID = sized(b.commit, G), sized(b.leaderTerm, G), sized(b.candTerm, G), sized(b.candID, G)
	b.commitOut, b.maxTerm = sized(b.commitOut, G), sized(b.maxTerm, G)
	b.grantMask, b.stepdownMask, b.memberMask = sized(b.grantMask, G), sized(b.stepdownMask, G), sized(b.memberMask, G)
	b.elected, b.status = sized(b.elected, G), sized(b.status, G)
	b.logOff = sized(b.logOff, G+1)
	var shapeErr, rangeErr, mixed atomic.Bool
	parallelOverGroups(G, func(g0, g1 int) {
		for g := g0; g < g1; g++ {
			grp := &d.groups[g]
			if k := len(grp.Members); k < 1 || k > n {
				shapeErr.Store(true)
				return
			}
			b.logOff[g+1] = 0
			if grp.Leader != nil {
				if len(grp.Leader.log) > math.MaxUint32 {
					rangeErr.Store(true)
					return
				}
				b.logOff[g+1] = uint32(len(grp.Leader.log))
			}
		}
	})
	if shapeErr.Load() {
		return errors.New("raft: a group needs 1..slots members (slots <= 16)")
	}
	if rangeErr.Load() {
		return errRange
	}
	total := uint64(0)
	b.logOff[0] = 0
	for g := 0; g < G; g++ {
		total += uint64(b.logOff[g+1])
		if total > math.MaxUint32 {
			return errRange
		}
		b.logOff[g+1] = uint32(total)
	}
	if total == 0 {
		b.logTerm = sized(b.logTerm, 1)
		b.logTerm[0] = 0
	} else {
		b.logTerm = sized(b.logTerm, int(total))
	}
	put := func(dst *T, v int) bool {
		if !L.fits(v) {
			return false
		}
		*dst = T(v)
		return true
	}
	parallelOverGroups(G, func(g0, g1 int) {
		anySmall := false
		for g := g0; g < g1; g++ {
			grp := &d.groups[g]
			k := len(grp.Members)
			b.memberMask[g] = uint16(1)<<uint(k) - 1
			anySmall = anySmall || k != n
			b.commit[g], b.leaderTerm[g], b.candTerm[g], b.candID[g] = L.max, 0, L.belowAll, -1
			b.commitOut[g], b.maxTerm[g], b.grantMask[g], b.stepdownMask[g], b.elected[g], b.status[g] = 0, 0, 0, 0, 0, 0
			for p := 0; p < n; p++ {
				i := p*G + g
				b.match[i], b.voterTerm[i], b.votedFor[i], b.lastLogTerm[i], b.voterTermOut[i] = 0, 0, -1, L.logEmpty, 0
			}
			ok := true
			if l := grp.Leader; l != nil {
				ok = ok && put(&b.commit[g], l.commitIndex) && put(&b.leaderTerm[g], l.currentTerm)
				for p := 0; p < k && p < len(l.matchIndex); p++ {
					ok = ok && put(&b.match[p*G+g], l.matchIndex[p])
				}
				dst := b.logTerm[b.logOff[g]:]
				for e := range l.log {
					ok = ok && put(&dst[e], l.log[e].TermNum)
				}
			}
			if c := grp.Candidate; c != nil {
				ok = ok && put(&b.candTerm[g], c.currentTerm) && put(&b.candID[g], c.id)
			}
			for p, m := range grp.Members {
				i := p*G + g
				ok = ok && put(&b.voterTerm[i], m.currentTerm) && put(&b.votedFor[i], m.votedFor)
				if len(m.log) > 0 {
					ok = ok && put(&b.lastLogTerm[i], m.log[len(m.log)-1].TermNum) && b.lastLogTerm[i] != L.logEmpty
				}
			}
			if !ok {
				rangeErr.Store(true)
				return
			}
		}
		if anySmall {
			mixed.Store(true)
		}
	})
	if rangeErr.Load() {
		return errRange
	}
	b.mixed = mixed.Load()
	return nil
}

// apply writes one round's decisions back: voters' terms and roles
// (raft.go:219-221), the leader's commitIndex, and BecomeLeader for every
// candidate that won. Groups are independent, so the walk is parallel.
func apply[T word](d *BatchedDriver, b *roundBuffers[T]) {
	G := b.groups
	parallelOverGroups(G, func(g0, g1 int) {
		for g := g0; g < g1; g++ {
			grp := &d.groups[g]
			leaderBefore := grp.Leader
			if grp.Candidate != nil {
				for p, m := range grp.Members {
					m.currentTerm = int(b.voterTermOut[p*G+g])
					if b.stepdownMask[g]>>uint(p)&1 == 1 {
						m.nodeType = Follower
					}
				}
				if b.elected[g] == 1 {
					grp.Candidate.BecomeLeader()
					grp.Leader, grp.Candidate = grp.Candidate, nil
				}
			}
			if leaderBefore != nil {
				leaderBefore.commitIndex = int(b.commitOut[g])
			}
			// a leader a vote request demoted (raft.go:221) leads no more
			if grp.Leader != nil && grp.Leader.nodeType != Leader {
				grp.Leader = nil
			}
		}
	})
}

// Step runs one fused round and applies it. The compact layout is tried first;
// if some value does not fit int32 the round runs in the wide layout. Groups
// of different sizes run through the member-mask round (old = new = members).
// A *PanicError lists the groups where the per-node methods would have panicked
// (a cluster straight from NewNode has empty logs, raft.go:87, and panics in
// RequestVoteRPC); the round is applied either way.
//
// Step flattens every Node and moves the whole image over PCIe both ways: fine
// for thousands of groups, tens of milliseconds for a million. A resident
// driver keeps the image in HBM and moves only the groups the host touched: its
// cgo calls are in resident_driver.go, its logic in the C++ host
// (raft_amd/csrc/host/raft_host_resident.cpp).
func (d *BatchedDriver) Step() error {
	err := flatten(d, compact, &d.buf32)
	if err == errRange {
		if err := flatten(d, wide, &d.buf64); err != nil {
			return err
		}
		if err := d.eng.round64(&d.buf64); err != nil {
			return err
		}
		apply(d, &d.buf64)
		d.LastStepWide = true
		return flagged(d.buf64.status)
	}
	if err != nil {
		return err
	}
	if err := d.eng.fusedRound(&d.buf32); err != nil {
		return err
	}
	apply(d, &d.buf32)
	d.LastStepWide = false
	return flagged(d.buf32.status)
}

// Close releases the GPU engine.
func (d *BatchedDriver) Close() { d.eng.close() }
