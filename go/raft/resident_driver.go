package raft

/*
#include <stdlib.h>
#include <string.h>
#include "rq.h"
*/
import "C"

import (
	"errors"
	"unsafe"
)

// Binding shape of the device-resident path: the calls a Go host needs to keep
// the structure-of-arrays image of its groups in HBM between rounds. This file
// is deliberately only the binding. The host logic above it -- which groups to
// upload, where their logs live in the arena, when to fall back to the wide
// layout, how to apply the gathered rows to the Node objects -- is implemented
// and tested in C++ (raft_amd/csrc/host/raft_host_resident.cpp, ResidentDriver);
// that driver is the reference implementation of the host, and a Go host follows
// it step by step:
//
//	first Step: flatten every group into tiles (64 groups per tile, slot rows of
//	    64 elements) in C memory, upload, rq_plan_round once;
//	every Step: rows of the groups the host touched -> groupsScatter;
//	    planLaunch; download the one-bit-per-group "changed" summary; gather
//	    the rows of the set bits -> groupsGather; apply them to the Nodes;
//	a value beyond int32, or arena offsets beyond 32 bits: lay the image out
//	    afresh (RQ_LAYOUT_I64, or a compacted arena).
//
// Every buffer C sees is C memory (rq_host_alloc / rq_device_alloc): no Go
// pointer crosses cgo, and the calls may arrive on any OS thread.
//
// Nothing in this package has been compiled: the image this code was written in
// has no Go toolchain.

// residentImage owns the device buffers a descriptor names.
type residentImage struct {
	eng  *rqEngine
	desc C.rq_round_desc
	plan *C.rq_plan
}

func rqError(rc C.int, where string) error {
	if rc == C.RQ_OK {
		return nil
	}
	return errors.New(where + ": " + C.GoString(C.rq_strerror(rc)))
}

// newRoundDesc returns a zeroed descriptor with the header fields set.
// C.sizeof_rq_round_desc is a C.size_t: the conversion is explicit.
func newRoundDesc(layout C.uint32_t, groups, slots uint32) C.rq_round_desc {
	var d C.rq_round_desc
	d.struct_size = C.uint32_t(C.sizeof_rq_round_desc)
	d.layout = layout
	d.what = C.RQ_ROUND_COMMIT | C.RQ_ROUND_VOTE
	d.G = C.uint32_t(groups)
	d.n = C.uint32_t(slots)
	d.ld = C.uint32_t(groups)
	return d
}

// setSegments tells a launch that the groups are ordered by cluster size:
// firstGroup has len(sizes)+1 entries (0 ... G, inner boundaries multiples of
// 64), sizes[i] is 1..9 for a segment of plain clusters of that size and 0 for a
// segment decided through its member masks.
func setSegments(d *C.rq_round_desc, firstGroup []uint32, sizes []uint8) error {
	if len(sizes) < 1 || len(sizes) > C.RQ_MAX_SEGMENTS || len(firstGroup) != len(sizes)+1 {
		return errors.New("segment table: 1..16 segments and one more boundary")
	}
	d.n_segments = C.uint32_t(len(sizes))
	for i, g := range firstGroup {
		d.seg_first_group[i] = C.uint32_t(g)
	}
	for i, n := range sizes {
		d.seg_n[i] = C.uint8_t(n)
	}
	return nil
}

func (e *rqEngine) deviceAlloc(bytes uint64) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	rc := C.rq_device_alloc(e.h, C.uint64_t(bytes), C.RQ_MEM_DEFAULT, &p)
	return p, rqError(rc, "rq_device_alloc")
}

func (e *rqEngine) hostAlloc(bytes uint64) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	rc := C.rq_host_alloc(e.h, C.uint64_t(bytes), &p)
	return p, rqError(rc, "rq_host_alloc")
}

func (e *rqEngine) toDevice(dst, src unsafe.Pointer, bytes uint64) error {
	return rqError(C.rq_copy_to_device(e.h, dst, src, C.uint64_t(bytes)), "rq_copy_to_device")
}

func (e *rqEngine) fromDevice(dst, src unsafe.Pointer, bytes uint64) error {
	return rqError(C.rq_copy_from_device(e.h, dst, src, C.uint64_t(bytes)), "rq_copy_from_device")
}

// prepare validates the descriptor once; launch then costs two arguments.
func (im *residentImage) prepare() error {
	return rqError(C.rq_plan_round(im.eng.h, &im.desc, &im.plan), "rq_plan_round")
}

// prepareRounds makes ONE prepared launch over several images (one per cluster
// size in its own compact tiles, or several short batches): rq_plan_rounds. The
// descriptors live in C memory for the duration of the call only.
func (e *rqEngine) prepareRounds(descs []C.rq_round_desc) (*C.rq_plan, error) {
	if len(descs) < 1 || len(descs) > C.RQ_MAX_SEGMENTS {
		return nil, errors.New("rq_plan_rounds: 1..16 images")
	}
	bytes := C.size_t(len(descs)) * C.size_t(C.sizeof_rq_round_desc)
	mem := (*C.rq_round_desc)(C.malloc(bytes))
	if mem == nil {
		return nil, errors.New("rq_plan_rounds: out of memory")
	}
	defer C.free(unsafe.Pointer(mem))
	copy(unsafe.Slice(mem, len(descs)), descs)
	var plan *C.rq_plan
	rc := C.rq_plan_rounds(e.h, mem, C.uint32_t(len(descs)), &plan)
	return plan, rqError(rc, "rq_plan_rounds")
}

// launch enqueues the round on the device's default stream and returns.
func (im *residentImage) launch() error {
	return rqError(C.rq_plan_launch(im.plan, nil), "rq_plan_launch")
}

func (im *residentImage) sync() error {
	return rqError(C.rq_stream_sync(im.eng.h, nil), "rq_stream_sync")
}

// scatter overwrites the listed groups with rows laid out as rq.h describes
// (4n+7 elements each); gids, rows, rowLogOff, rowLogTerms and rowStatus are
// device pointers.
func (im *residentImage) scatter(count uint32, gids, rows, rowLogOff, rowLogTerms, rowStatus unsafe.Pointer) error {
	rc := C.rq_groups_scatter_dev(im.eng.h, nil, &im.desc, C.uint32_t(count), (*C.uint32_t)(gids), rows,
		(*C.uint32_t)(rowLogOff), rowLogTerms, (*C.uint8_t)(rowStatus))
	return rqError(rc, "rq_groups_scatter_dev")
}

// gather packs the outputs of the listed groups (n+2 elements and one flag word each).
func (im *residentImage) gather(count uint32, gids, rowsOut, flagsOut unsafe.Pointer) error {
	rc := C.rq_groups_gather_dev(im.eng.h, nil, &im.desc, C.uint32_t(count), (*C.uint32_t)(gids), rowsOut,
		(*C.uint64_t)(flagsOut))
	return rqError(rc, "rq_groups_gather_dev")
}

func (im *residentImage) close() {
	if im.plan != nil {
		C.rq_plan_destroy(im.plan)
		im.plan = nil
	}
}
