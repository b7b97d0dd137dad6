package raft

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raft_amd -lrq -Wl,-rpath,${SRCDIR}/../../raft_amd
#include "rq.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"
)

// rqEngine wraps one rq_engine (one GPU, one stream).
type rqEngine struct{ h *C.rq_engine }

func newRqEngine(device int, maxGroups, maxSlots uint32, maxLogTerms uint64) (*rqEngine, error) {
	var h *C.rq_engine
	rc := C.rq_engine_create(C.int(device), C.uint32_t(maxGroups), C.uint32_t(maxSlots),
		C.uint64_t(maxLogTerms), &h)
	if rc != C.RQ_OK {
		return nil, errors.New("rq_engine_create: " + C.GoString(C.rq_strerror(rc)))
	}
	e := &rqEngine{h: h}
	runtime.SetFinalizer(e, func(e *rqEngine) { e.close() })
	return e, nil
}

func (e *rqEngine) close() {
	if e.h != nil {
		C.rq_engine_destroy(e.h)
		e.h = nil
	}
}

// Slice -> C pointer; an empty slice (a driver with no groups) is NULL, which the
// entry points accept together with G = 0.
func i32(p []int32) *C.int32_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.int32_t)(unsafe.Pointer(&p[0]))
}
func u32(p []uint32) *C.uint32_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&p[0]))
}
func u16(p []uint16) *C.uint16_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.uint16_t)(unsafe.Pointer(&p[0]))
}
func u8(p []uint8) *C.uint8_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&p[0]))
}
func i64(p []int64) *C.int64_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.int64_t)(unsafe.Pointer(&p[0]))
}

// fusedRound passes only flat scalar slices (no Go pointers inside them), which
// is what the cgo pointer rules require; librq copies before it returns.
func (e *rqEngine) fusedRound(b *roundBuffers[int32]) error {
	var rc C.int
	if b.mixed {
		rc = C.rq_fused_round_joint(e.h, C.uint32_t(b.groups), C.uint32_t(b.slots), C.uint32_t(b.ld),
			u16(b.memberMask), u16(b.memberMask),
			i32(b.match), i32(b.commit), i32(b.leaderTerm), u32(b.logOff), i32(b.logTerm),
			i32(b.voterTerm), i32(b.votedFor), i32(b.lastLogTerm), i32(b.candTerm), i32(b.candID),
			i32(b.commitOut), i32(b.voterTermOut), u16(b.grantMask), u16(b.stepdownMask),
			u8(b.elected), i32(b.maxTerm), u8(b.status))
	} else {
		rc = C.rq_fused_round(e.h, C.uint32_t(b.groups), C.uint32_t(b.slots), C.uint32_t(b.ld),
			i32(b.match), i32(b.commit), i32(b.leaderTerm), u32(b.logOff), i32(b.logTerm),
			i32(b.voterTerm), i32(b.votedFor), i32(b.lastLogTerm), i32(b.candTerm), i32(b.candID),
			i32(b.commitOut), i32(b.voterTermOut), u16(b.grantMask), u16(b.stepdownMask),
			u8(b.elected), i32(b.maxTerm), u8(b.status))
	}
	if rc != C.RQ_OK {
		return errors.New("rq_fused_round: " + C.GoString(C.rq_strerror(rc)))
	}
	return nil
}

// round64 is the same round in the wide layout (rq_round64): member masks only
// when the groups differ in size, full logs (no window base), both decisions in
// one launch.
func (e *rqEngine) round64(b *roundBuffers[int64]) error {
	var mask *C.uint16_t
	if b.mixed {
		mask = u16(b.memberMask)
	}
	rc := C.rq_round64(e.h, C.RQ_ROUND_COMMIT|C.RQ_ROUND_VOTE,
		C.uint32_t(b.groups), C.uint32_t(b.slots), C.uint32_t(b.ld), mask, mask,
		i64(b.match), i64(b.commit), i64(b.leaderTerm), u32(b.logOff), i64(b.logTerm), nil,
		i64(b.voterTerm), i64(b.votedFor), i64(b.lastLogTerm), i64(b.candTerm), i64(b.candID),
		i64(b.commitOut), i64(b.voterTermOut), u16(b.grantMask), u16(b.stepdownMask),
		u8(b.elected), i64(b.maxTerm), u8(b.status))
	if rc != C.RQ_OK {
		return errors.New("rq_round64: " + C.GoString(C.rq_strerror(rc)))
	}
	return nil
}
