package raft

import (
	"errors"
	"math"
)

// Group is one Raft group as the driver sees it: its members in slot order,
// the member currently leading (nil if none) and the member currently asking
// for votes (nil if none).
type Group struct {
	Members   []*Node
	Leader    *Node
	Candidate *Node
}

// roundBuffers is the structure-of-arrays image of all groups:
// plane[p*ld+g] for slot p of group g.
type roundBuffers struct {
	groups, slots, ld int
	match, commit, leaderTerm []int32
	logOff                    []uint32
	logTerm                   []int32
	voterTerm, votedFor, lastLogTerm, candTerm, candID []int32
	commitOut, voterTermOut, maxTerm                   []int32
	grantMask, stepdownMask                            []uint16
	elected, status                                    []uint8
}

const logEmpty = math.MinInt32 // RQ_LOG_EMPTY

// BatchedDriver evaluates one quorum round for every registered group in one
// GPU launch and applies the decisions to the Go objects.
type BatchedDriver struct {
	eng    *rqEngine
	groups []Group
	slots  int
}

func NewBatchedDriver(device int, groups []Group, slots int, maxLogTerms uint64) (*BatchedDriver, error) {
	eng, err := newRqEngine(device, uint32(len(groups)), uint32(slots), maxLogTerms)
	if err != nil {
		return nil, err
	}
	return &BatchedDriver{eng: eng, groups: groups, slots: slots}, nil
}

func narrow(v int) (int32, error) {
	if v < math.MinInt32+1 || v > math.MaxInt32 {
		return 0, errors.New("raft: value outside the int32 range of the batched engine")
	}
	return int32(v), nil
}

// flatten builds the SoA image. A group without a leader gets commit =
// MaxInt32 (N > commit is never true); a group without a candidate gets
// candTerm = MinInt32+1 (below every term: no step-down, no grant).
func (d *BatchedDriver) flatten() (*roundBuffers, error) {
	G, n := len(d.groups), d.slots
	b := &roundBuffers{groups: G, slots: n, ld: G}
	plane := func() []int32 { return make([]int32, n*G) }
	b.match, b.voterTerm, b.votedFor, b.lastLogTerm, b.voterTermOut = plane(), plane(), plane(), plane(), plane()
	vec := func() []int32 { return make([]int32, G) }
	b.commit, b.leaderTerm, b.candTerm, b.candID, b.commitOut, b.maxTerm = vec(), vec(), vec(), vec(), vec(), vec()
	b.grantMask, b.stepdownMask = make([]uint16, G), make([]uint16, G)
	b.elected, b.status = make([]uint8, G), make([]uint8, G)
	b.logOff = make([]uint32, G+1)
	var err error
	for g, grp := range d.groups {
		if len(grp.Members) != n {
			return nil, errors.New("raft: group size differs from the driver's slot count")
		}
		b.commit[g], b.candTerm[g] = math.MaxInt32, math.MinInt32+1
		if l := grp.Leader; l != nil {
			if b.commit[g], err = narrow(l.commitIndex); err != nil {
				return nil, err
			}
			if b.leaderTerm[g], err = narrow(l.currentTerm); err != nil {
				return nil, err
			}
			for p := 0; p < n && p < len(l.matchIndex); p++ {
				if b.match[p*G+g], err = narrow(l.matchIndex[p]); err != nil {
					return nil, err
				}
			}
			for _, e := range l.log {
				t, err := narrow(e.TermNum)
				if err != nil {
					return nil, err
				}
				b.logTerm = append(b.logTerm, t)
			}
		}
		b.logOff[g+1] = uint32(len(b.logTerm))
		if c := grp.Candidate; c != nil {
			if b.candTerm[g], err = narrow(c.currentTerm); err != nil {
				return nil, err
			}
			if b.candID[g], err = narrow(c.id); err != nil {
				return nil, err
			}
		}
		for p, m := range grp.Members {
			i := p*G + g
			if b.voterTerm[i], err = narrow(m.currentTerm); err != nil {
				return nil, err
			}
			if b.votedFor[i], err = narrow(m.votedFor); err != nil {
				return nil, err
			}
			b.lastLogTerm[i] = logEmpty
			if len(m.log) > 0 {
				if b.lastLogTerm[i], err = narrow(m.log[len(m.log)-1].TermNum); err != nil {
					return nil, err
				}
			}
		}
	}
	if len(b.logTerm) == 0 {
		b.logTerm = []int32{0}
	}
	return b, nil
}

// Step runs one fused round and applies it: voters' terms and roles
// (raft.go:219-221), the leader's commitIndex, and BecomeLeader for every
// candidate that won.
func (d *BatchedDriver) Step() error {
	b, err := d.flatten()
	if err != nil {
		return err
	}
	if err := d.eng.fusedRound(b); err != nil {
		return err
	}
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
	return nil
}

// Close releases the GPU engine.
func (d *BatchedDriver) Close() { d.eng.close() }
