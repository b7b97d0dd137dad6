package raft

/*
#include <string.h>
#include "rq.h"
*/
import "C"

import (
	"errors"
	"math"
	"math/bits"
	"unsafe"
)

// ResidentDriver keeps the structure-of-arrays image of all groups in HBM
// between rounds. The host tells it which groups it touched (MarkDirty, after
// applying RPCs to their Nodes); Step uploads those groups only
// (rq_groups_scatter_dev), runs the round in place (rq_plan_launch), reads the
// one-bit-per-group "changed" summary the kernel wrote and downloads and applies
// only the groups whose bit is set (rq_groups_gather_dev). All buffers C sees
// are C memory (rq_host_alloc / rq_device_alloc): no Go pointer crosses cgo.
// Compact (int32) layout; a value beyond it makes Step return errRange and the
// caller falls back to BatchedDriver.
type ResidentDriver struct {
	eng    *rqEngine
	groups []Group
	slots  int
	mixed  bool
	built  bool
	// byStart: every leader's log has Raft's own shape (terms never decrease, none above the
	// leader's term), so the image holds no log arena, only the index where each leader's term
	// starts and the log's length (rq_round_desc.log_term_start); a log that loses the shape
	// (the reference appends whatever it is sent, raft.go:158-170) brings the arena back
	byStart bool

	desc C.rq_round_desc
	plan *C.rq_plan
	// the arena as the host manages it: every group's slice, free space from tail on
	off, capacity []uint32
	tail, arena   uint64
	// staging: one block up, one block down, page-locked mirrors of both
	dIn, dOut, hIn, hOut unsafe.Pointer
	hBits                unsafe.Pointer
	rowCap               int
	termCap              uint64

	dirty     []int
	dirtyMark []bool
	dev, host []unsafe.Pointer
}

const maxRowsPerTransfer = 1 << 18

func NewResidentDriver(device int, groups []Group, slots int) (*ResidentDriver, error) {
	eng, err := newRqEngine(device, 64, 1, 64) // the engine only binds the device here
	if err != nil {
		return nil, err
	}
	return &ResidentDriver{eng: eng, groups: groups, slots: slots, dirtyMark: make([]bool, len(groups))}, nil
}

// MarkDirty tells the driver that the host changed a Node of group g.
func (d *ResidentDriver) MarkDirty(g int) {
	if g >= 0 && g < len(d.dirtyMark) && !d.dirtyMark[g] {
		d.dirtyMark[g] = true
		d.dirty = append(d.dirty, g)
	}
}

func sliceCapacity(n int) uint32 {
	room := n / 4
	if room < 4 {
		room = 4
	}
	return uint32(n + room)
}

func put32(dst *int32, v int) bool {
	if v <= math.MinInt32 || v > math.MaxInt32 { // MinInt32 is the empty-log marker
		return false
	}
	*dst = int32(v)
	return true
}

// flattenRow writes group g as one scatter row (4n+7 words, layout of
// rq_groups_scatter_dev; the last word, the arena offset, is set by the caller)
// and returns its leader's log terms.
// logShape measures group g's leader log: shaped (terms never decrease, none above the
// leader's term), the index of the first entry of the leader's term (or the length), the length.
func (d *ResidentDriver) logShape(g int) (shaped bool, start int32, length uint32) {
	l := d.groups[g].Leader
	if l == nil {
		return true, 0, 0
	}
	shaped, first := true, len(l.log)
	for i, e := range l.log {
		if i > 0 && e.TermNum < l.log[i-1].TermNum {
			shaped = false
		}
		if first == len(l.log) && e.TermNum == l.currentTerm {
			first = i
		}
	}
	if len(l.log) > 0 && l.log[len(l.log)-1].TermNum > l.currentTerm {
		shaped = false
	}
	return shaped, int32(first), uint32(len(l.log))
}

// WithoutArena reports whether the resident image currently holds no log arena.
func (d *ResidentDriver) WithoutArena() bool { return d.built && d.byStart }

func (d *ResidentDriver) flattenRow(g int, row []int32) ([]int32, error) {
	grp, n := d.groups[g], d.slots
	k := len(grp.Members)
	if k < 1 || k > n || n > 16 {
		return nil, errors.New("raft: a group needs 1..slots members (slots <= 16)")
	}
	match, vt, vf, ll, tail := row[:n], row[n:2*n], row[2*n:3*n], row[3*n:4*n], row[4*n:]
	for p := 0; p < n; p++ {
		match[p], vt[p], vf[p], ll[p] = 0, 0, -1, math.MinInt32
	}
	tail[0], tail[1], tail[2], tail[3], tail[4] = math.MaxInt32, 0, math.MinInt32+1, -1, 0
	mask := uint32(1)<<uint(k) - 1
	tail[5] = int32(mask | mask<<16)
	var terms []int32
	if l := grp.Leader; l != nil {
		if !put32(&tail[0], l.commitIndex) || !put32(&tail[1], l.currentTerm) {
			return nil, errRange
		}
		for p := 0; p < k && p < len(l.matchIndex); p++ {
			if !put32(&match[p], l.matchIndex[p]) {
				return nil, errRange
			}
		}
		if !d.byStart { // without an arena the log is only measured (logShape), not copied
			terms = make([]int32, len(l.log))
			for i, e := range l.log {
				if !put32(&terms[i], e.TermNum) {
					return nil, errRange
				}
			}
		}
	}
	if c := grp.Candidate; c != nil {
		if !put32(&tail[2], c.currentTerm) || !put32(&tail[3], c.id) {
			return nil, errRange
		}
	}
	for p, m := range grp.Members {
		if !put32(&vt[p], m.currentTerm) || !put32(&vf[p], m.votedFor) {
			return nil, errRange
		}
		if len(m.log) > 0 && !put32(&ll[p], m.log[len(m.log)-1].TermNum) {
			return nil, errRange
		}
	}
	return terms, nil
}

func (d *ResidentDriver) check(rc C.int, where string) error {
	if rc != C.RQ_OK {
		return errors.New(where + ": " + C.GoString(C.rq_strerror(rc)))
	}
	return nil
}

func (d *ResidentDriver) dalloc(bytes uint64) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	if bytes == 0 {
		bytes = 4
	}
	err := d.check(C.rq_device_alloc(d.eng.h, C.uint64_t(bytes), C.RQ_MEM_DEFAULT, &p), "rq_device_alloc")
	if err == nil {
		d.dev = append(d.dev, p)
	}
	return p, err
}

func (d *ResidentDriver) halloc(bytes uint64) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	if bytes == 0 {
		bytes = 4
	}
	err := d.check(C.rq_host_alloc(d.eng.h, C.uint64_t(bytes), &p), "rq_host_alloc")
	if err == nil {
		d.host = append(d.host, p)
	}
	return p, err
}

func (d *ResidentDriver) release() {
	if d.plan != nil {
		C.rq_plan_destroy(d.plan)
		d.plan = nil
	}
	for _, p := range d.dev {
		C.rq_device_free(d.eng.h, p)
	}
	for _, p := range d.host {
		C.rq_host_free(d.eng.h, p)
	}
	d.dev, d.host, d.built = nil, nil, false
}

// upload copies a Go slice's bytes into fresh device memory through a
// page-locked C buffer (cgo must not hand Go memory to an asynchronous copy).
func upload[T any](d *ResidentDriver, src []T) (unsafe.Pointer, error) {
	var zero T
	bytes := uint64(len(src)) * uint64(unsafe.Sizeof(zero))
	dp, err := d.dalloc(bytes)
	if err != nil || bytes == 0 {
		return dp, err
	}
	var hp unsafe.Pointer
	if err := d.check(C.rq_host_alloc(d.eng.h, C.uint64_t(bytes), &hp), "rq_host_alloc"); err != nil {
		return nil, err
	}
	defer C.rq_host_free(d.eng.h, hp)
	copy(unsafe.Slice((*T)(hp), len(src)), src)
	return dp, d.check(C.rq_copy_to_device(d.eng.h, dp, hp, C.uint64_t(bytes)), "rq_copy_to_device")
}

// rebuild lays the whole image out afresh: every group flattened, arena slices
// with room to grow and free space behind them, device buffers, staging, plan.
func (d *ResidentDriver) rebuild() error {
	G, n := len(d.groups), d.slots
	if G == 0 {
		return errors.New("raft: no groups")
	}
	d.release()
	words := 4*n + 7
	plane := func() []int32 { return make([]int32, n*G) }
	match, vt, vf, ll := plane(), plane(), plane(), plane()
	commit, lterm, cterm, cid := make([]int32, G), make([]int32, G), make([]int32, G), make([]int32, G)
	length := make([]uint32, G)
	om, nm := make([]uint16, G), make([]uint16, G)
	d.off, d.capacity = make([]uint32, G+1), make([]uint32, G)
	var arena []int32
	row := make([]int32, words)
	d.mixed = false
	tstart := make([]int32, G)
	d.byStart = true
	for g := 0; g < G; g++ {
		shaped, st, ln := d.logShape(g)
		tstart[g], length[g] = st, ln
		if !shaped {
			d.byStart = false
		}
	}
	for g := 0; g < G; g++ {
		terms, err := d.flattenRow(g, row)
		if err != nil {
			return err
		}
		for p := 0; p < n; p++ {
			match[p*G+g], vt[p*G+g], vf[p*G+g], ll[p*G+g] = row[p], row[n+p], row[2*n+p], row[3*n+p]
		}
		tail := row[4*n:]
		commit[g], lterm[g], cterm[g], cid[g] = tail[0], tail[1], tail[2], tail[3]
		om[g], nm[g] = uint16(uint32(tail[5])), uint16(uint32(tail[5])>>16)
		if len(d.groups[g].Members) != n {
			d.mixed = true
		}
		if d.byStart {
			continue
		}
		length[g] = uint32(len(terms))
		d.capacity[g] = sliceCapacity(len(terms))
		next := uint64(d.off[g]) + uint64(d.capacity[g])
		if next > math.MaxUint32 {
			return errRange
		}
		d.off[g+1] = uint32(next)
		arena = append(arena, terms...)
		arena = append(arena, make([]int32, int(next)-len(arena))...)
	}
	d.tail = uint64(len(arena))
	d.arena = d.tail + d.tail/2 + 1<<16
	if d.arena > math.MaxUint32 {
		d.arena = math.MaxUint32
	}
	if d.byStart {
		d.arena = 0
	}
	arena = append(arena, make([]int32, int(d.arena)-len(arena))...)

	C.memset(unsafe.Pointer(&d.desc), 0, C.sizeof_rq_round_desc)
	dd := &d.desc
	dd.struct_size, dd.layout = C.sizeof_rq_round_desc, C.RQ_LAYOUT_I32
	dd.what = C.RQ_ROUND_COMMIT | C.RQ_ROUND_VOTE
	dd.G, dd.n, dd.ld, dd.log_terms = C.uint32_t(G), C.uint32_t(n), C.uint32_t(G), C.uint64_t(len(arena))
	var err error
	up := func(dst *unsafe.Pointer, src []int32) {
		if err == nil {
			*dst, err = upload(d, src)
		}
	}
	up(&dd.match, match)
	up(&dd.voter_term, vt)
	up(&dd.voted_for, vf)
	up(&dd.voter_last_log_term, ll)
	up(&dd.commit, commit)
	up(&dd.cand_term, cterm)
	up(&dd.cand_id, cid)
	if d.byStart {
		up(&dd.log_term_start, tstart)
	} else {
		up(&dd.leader_term, lterm)
		up(&dd.log_term, arena)
	}
	if err != nil {
		return err
	}
	var p unsafe.Pointer
	if !d.byStart {
		if p, err = upload(d, d.off); err != nil {
			return err
		}
		dd.log_off = (*C.uint32_t)(p)
	}
	if p, err = upload(d, length); err != nil {
		return err
	}
	dd.log_len = (*C.uint32_t)(p)
	if d.mixed {
		if p, err = upload(d, om); err != nil {
			return err
		}
		dd.old_mask = (*C.uint16_t)(p)
		if p, err = upload(d, nm); err != nil {
			return err
		}
		dd.new_mask = (*C.uint16_t)(p)
	}
	dd.commit_out, dd.voter_term_out = dd.commit, dd.voter_term // state is updated in place between rounds
	bitWords := uint64(G+63) / 64
	outs := []struct {
		dst   *unsafe.Pointer
		bytes uint64
	}{{(*unsafe.Pointer)(unsafe.Pointer(&dd.grant_mask)), uint64(G) * 2},
		{(*unsafe.Pointer)(unsafe.Pointer(&dd.stepdown_mask)), uint64(G) * 2},
		{(*unsafe.Pointer)(unsafe.Pointer(&dd.elected)), uint64(G)},
		{&dd.max_term, uint64(G) * 4},
		{(*unsafe.Pointer)(unsafe.Pointer(&dd.status_out)), uint64(G)},
		{(*unsafe.Pointer)(unsafe.Pointer(&dd.changed_bits)), bitWords * 8}}
	for _, o := range outs {
		if *o.dst, err = d.dalloc(o.bytes); err != nil {
			return err
		}
	}
	if err = d.check(C.rq_plan_round(d.eng.h, dd, &d.plan), "rq_plan_round"); err != nil {
		return err
	}
	d.rowCap = G
	if d.rowCap > maxRowsPerTransfer {
		d.rowCap = maxRowsPerTransfer
	}
	d.termCap = 64 * uint64(d.rowCap)
	if d.termCap > d.arena {
		d.termCap = d.arena
	}
	inBytes := (uint64(d.rowCap)*uint64(words+2) + 1 + d.termCap) * 4
	outBytes := uint64(d.rowCap) * uint64(8+(n+2)*4)
	for _, a := range []struct {
		dst    *unsafe.Pointer
		bytes  uint64
		device bool
	}{{&d.dIn, inBytes, true}, {&d.hIn, inBytes, false}, {&d.dOut, outBytes, true}, {&d.hOut, outBytes, false},
		{&d.hBits, bitWords * 8, false}} {
		if a.device {
			*a.dst, err = d.dalloc(a.bytes)
		} else {
			*a.dst, err = d.halloc(a.bytes)
		}
		if err != nil {
			return err
		}
	}
	for _, g := range d.dirty {
		d.dirtyMark[g] = false
	}
	d.dirty, d.built = d.dirty[:0], true
	return nil
}

// applyRow writes one gathered row back into the objects: voters' terms and
// roles (raft.go:219-221), the leader's commitIndex, BecomeLeader for a winner.
func (d *ResidentDriver) applyRow(g int, row []int32, flags uint64, fl *[]GroupStatus) {
	grp := &d.groups[g]
	stepdown, elected, status := uint32(flags>>16)&0xFFFF, (flags>>32)&0xFF != 0, uint8(flags>>40)
	leaderBefore := grp.Leader
	if grp.Candidate != nil {
		for p, m := range grp.Members {
			m.currentTerm = int(row[2+p])
			if stepdown>>uint(p)&1 == 1 {
				m.nodeType = Follower
			}
		}
		if elected {
			grp.Candidate.BecomeLeader()
			grp.Leader, grp.Candidate = grp.Candidate, nil
			d.MarkDirty(g) // new leader arrays, no candidate: the row goes up again
		}
	}
	if leaderBefore != nil {
		leaderBefore.commitIndex = int(row[0])
	}
	if status != 0 {
		*fl = append(*fl, GroupStatus{g, status})
	}
}

// Step runs one fused round on the resident image. Only dirty groups go up,
// only changed groups come back. A *PanicError lists flagged groups as in
// BatchedDriver.Step.
func (d *ResidentDriver) Step() error {
	n := d.slots
	words := 4*n + 7
	needRebuild := !d.built || len(d.dirty) > d.rowCap
	if !needRebuild && len(d.dirty) > 0 {
		r := len(d.dirty)
		// block going up, packed for r rows: group ids | term offsets (r+1) | rows | terms
		in := unsafe.Slice((*int32)(d.hIn), (d.rowCap*(words+2)+1)+int(d.termCap))
		gids, roff := in[:r], in[r:2*r+1]
		rows := in[2*r+1 : 2*r+1+r*words]
		rterms := in[2*r+1+r*words:]
		used := 0
		for i, g := range d.dirty {
			row := rows[i*words : (i+1)*words]
			terms, err := d.flattenRow(g, row)
			if err != nil {
				return err
			}
			if !d.mixed && len(d.groups[g].Members) != n {
				needRebuild = true
				break
			}
			if d.byStart { // lengths only: no terms travel; the row's last word is the term's start
				shaped, st, ln := d.logShape(g)
				if !shaped {
					needRebuild = true
					break
				}
				row[4*n+6] = st
				gids[i], roff[i] = int32(g), int32(used)
				used += int(ln)
				continue
			}
			if len(terms) > int(d.capacity[g]) { // the log outgrew its slice: it moves to free space
				want := 2*uint64(len(terms)) + 4
				if d.tail+want > d.arena {
					needRebuild = true
					break
				}
				d.off[g], d.capacity[g] = uint32(d.tail), uint32(want)
				d.tail += want
			}
			if uint64(used+len(terms)) > d.termCap {
				needRebuild = true
				break
			}
			row[4*n+6] = int32(d.off[g])
			gids[i], roff[i] = int32(g), int32(used)
			copy(rterms[used:], terms)
			used += len(terms)
		}
		if !needRebuild {
			roff[r] = int32(used)
			sent := used
			if d.byStart {
				sent = 0
			}
			bytes := uint64(r*(words+2)+1+sent) * 4
			if err := d.check(C.rq_copy_to_device(d.eng.h, d.dIn, d.hIn, C.uint64_t(bytes)), "rq_copy_to_device"); err != nil {
				return err
			}
			at := func(words int) unsafe.Pointer { return unsafe.Add(d.dIn, words*4) }
			rc := C.rq_groups_scatter_dev(d.eng.h, nil, &d.desc, C.uint32_t(r), (*C.uint32_t)(d.dIn),
				at(2*r+1), (*C.uint32_t)(at(r)), at(2*r+1+r*words), nil)
			if err := d.check(rc, "rq_groups_scatter_dev"); err != nil {
				return err
			}
			for _, g := range d.dirty {
				d.dirtyMark[g] = false
			}
			d.dirty = d.dirty[:0]
		}
	}
	if needRebuild {
		if err := d.rebuild(); err != nil {
			return err
		}
	}
	if err := d.check(C.rq_plan_launch(d.plan, nil), "rq_plan_launch"); err != nil {
		return err
	}
	G := len(d.groups)
	bitWords := (G + 63) / 64
	rc := C.rq_copy_from_device(d.eng.h, d.hBits, unsafe.Pointer(d.desc.changed_bits), C.uint64_t(bitWords*8))
	if err := d.check(rc, "rq_copy_from_device"); err != nil { // ordered after the round
		return err
	}
	var fl []GroupStatus
	gids := unsafe.Slice((*uint32)(d.hIn), d.rowCap)
	r := 0
	flush := func() error {
		if r == 0 {
			return nil
		}
		if err := d.check(C.rq_copy_to_device(d.eng.h, d.dIn, d.hIn, C.uint64_t(r*4)), "rq_copy_to_device"); err != nil {
			return err
		}
		// block coming down, packed for r rows: flags (r x 8 bytes) | rows
		rc := C.rq_groups_gather_dev(d.eng.h, nil, &d.desc, C.uint32_t(r), (*C.uint32_t)(d.dIn),
			unsafe.Add(d.dOut, r*8), (*C.uint64_t)(d.dOut))
		if err := d.check(rc, "rq_groups_gather_dev"); err != nil {
			return err
		}
		bytes := uint64(r) * uint64(8+(n+2)*4)
		if err := d.check(C.rq_copy_from_device(d.eng.h, d.hOut, d.dOut, C.uint64_t(bytes)), "rq_copy_from_device"); err != nil {
			return err
		}
		flags := unsafe.Slice((*uint64)(d.hOut), r)
		orows := unsafe.Slice((*int32)(unsafe.Add(d.hOut, r*8)), r*(n+2))
		for i := 0; i < r; i++ {
			d.applyRow(int(gids[i]), orows[i*(n+2):(i+1)*(n+2)], flags[i], &fl)
		}
		r = 0
		return nil
	}
	words64 := unsafe.Slice((*uint64)(d.hBits), bitWords)
	for w, b := range words64 {
		for b != 0 {
			gids[r] = uint32(w*64 + bits.TrailingZeros64(b))
			b &= b - 1
			if r++; r == d.rowCap {
				if err := flush(); err != nil {
					return err
				}
			}
		}
	}
	if err := flush(); err != nil {
		return err
	}
	if fl != nil {
		return &PanicError{fl}
	}
	return nil
}

// Close releases device and page-locked memory and the engine.
func (d *ResidentDriver) Close() {
	d.release()
	d.eng.close()
}
