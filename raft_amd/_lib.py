"""ctypes binding of librq.so (the C-ABI declared in include/rq.h).

The library holds only gfx950 device code plus its launch plumbing; there is
no CPU fallback here or anywhere in this package.  If the shared object is
missing the import fails loudly and tells the caller how to build it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RQ_LIBRARY: another build of the same library (A/B measurements); the default is the in-tree one
LIB_PATH = os.environ.get("RQ_LIBRARY") or os.path.join(_HERE, "librq.so")

i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p
u32 = C.c_uint32
u64 = C.c_uint64
cint = C.c_int



class RoundDesc(C.Structure):
    """rq_round_desc (include/rq.h); every pointer is a raw device address."""
    _fields_ = [("struct_size", u32), ("layout", u32), ("what", u32), ("variant", C.c_int32),
                ("G", u32), ("n", u32), ("ld", u32), ("log_term_bits", u32), ("summary_bits", u32),
                ("reserved", u32), ("log_terms", u64), ("tile_bytes", u64), ("out_tile_bytes", u64),
                ("old_mask", vp), ("new_mask", vp),
                ("match", vp), ("commit", vp), ("leader_term", vp), ("log_off", vp),
                ("log_term", vp), ("log_base", vp), ("log_term_base", vp), ("log_len", vp),
                ("voter_term", vp), ("voted_for", vp), ("voter_last_log_term", vp),
                ("cand_term", vp), ("cand_id", vp),
                ("commit_out", vp), ("voter_term_out", vp), ("grant_mask", vp), ("stepdown_mask", vp),
                ("elected", vp), ("max_term", vp), ("status_out", vp), ("elected_bits", vp),
                ("changed_bits", vp),
                ("commit_summary", vp), ("summary_base", vp), ("summary_escapes", vp),
                ("start_signal", vp), ("log_term_start", vp),
                ("n_segments", u32), ("seg_first_group", u32 * 17), ("seg_n", C.c_uint8 * 16)]


class AppendDesc(C.Structure):
    """rq_append_desc (include/rq.h)."""
    _fields_ = [("struct_size", u32), ("layout", u32), ("G", u32), ("n", u32), ("ld", u32),
                ("flog_term_bits", u32), ("flog_terms", u64),
                ("active_mask", vp),
                ("follower_term", vp), ("leader_term", vp), ("prev_log_index", vp), ("prev_log_term", vp),
                ("n_new", vp), ("flog_off", vp), ("flog_term", vp), ("flog_base", vp), ("flog_term_base", vp),
                ("follower_term_out", vp), ("success_mask", vp), ("stepdown_mask", vp),
                ("match_index_io", vp), ("next_index_io", vp), ("status_out", vp),
                ("tile_bytes", u64), ("out_tile_bytes", u64),
                ("leader_commit", vp), ("last_new_index", vp), ("follower_commit_io", vp),
                ("commit_round", vp)]


_COMMIT_IN = [i32p, i32p, i32p, u32p, i32p]          # match commit leader_term log_off log_term
_VOTE_IN = [i32p, i32p, i32p, i32p, i32p]            # voter_term voted_for last_log_term cand_term cand_id
_VOTE_OUT = [i32p, u16p, u16p, u8p, i32p]            # voter_term_out grant stepdown elected max_term
_MASKS = [u16p, u16p]

# name -> argtypes; every function returns int except where noted below.
PROTOTYPES = {
    "rq_version": [],
    "rq_engine_create": [cint, u32, u32, u64, C.POINTER(vp)],
    "rq_engine_destroy": [vp],
    "rq_last_hip_error": [vp],
    "rq_engine_device": [vp],
    "rq_commit_advance": [vp, u32, u32, u32] + _COMMIT_IN + [i32p, u8p],
    "rq_vote_tally": [vp, u32, u32, u32] + _VOTE_IN + _VOTE_OUT + [u8p],
    "rq_fused_round": [vp, u32, u32, u32] + _COMMIT_IN + _VOTE_IN + [i32p] + _VOTE_OUT + [u8p],
    "rq_commit_advance_joint": [vp, u32, u32, u32] + _MASKS + _COMMIT_IN + [i32p, u8p],
    "rq_vote_tally_joint": [vp, u32, u32, u32] + _MASKS + _VOTE_IN + _VOTE_OUT + [u8p],
    "rq_fused_round_joint": [vp, u32, u32, u32] + _MASKS + _COMMIT_IN + _VOTE_IN + [i32p] + _VOTE_OUT + [u8p],
    "rq_follower_clamp": [vp, u64, i32p, i32p, i32p, i32p],
    "rq_append_check": [vp, u32, u32, u32, u16p, i32p, i32p, i32p, i32p, i32p, u32p, i32p,
                        i32p, u16p, u16p, i32p, i32p, u8p],
    "rq_append_check_dev": [vp, vp, C.POINTER(AppendDesc)],
    "rq_upload": [vp, u32, u32, u32] + _COMMIT_IN + _VOTE_IN,
    "rq_step": [vp, u32],
    "rq_download": [vp, i32p] + _VOTE_OUT + [u8p],
    # device-pointer launches: descriptors of raw addresses so torch data_ptr() ints pass through
    "rq_round_dev": [vp, vp, C.POINTER(RoundDesc)],
    "rq_plan_round": [vp, C.POINTER(RoundDesc), C.POINTER(vp)],
    "rq_plan_rounds": [vp, C.POINTER(RoundDesc), u32, C.POINTER(vp)],
    "rq_groups_scatter_dev": [vp, vp, C.POINTER(RoundDesc), u32, vp, vp, vp, vp, vp],
    "rq_groups_gather_dev": [vp, vp, C.POINTER(RoundDesc), u32, vp, vp, vp],
    # wide layout on host buffers: what, G, m, ld, then 2 masks + 11 inputs + 7 outputs
    "rq_round64": [vp, cint, u32, u32, u32] + [vp] * 20,
    "rq_follower_clamp64": [vp, u64, vp, vp, vp, vp],
    "rq_append_check64": [vp, u32, u32, u32] + [vp] * 14,
    "rq_summary_apply_dev": [vp, vp, vp, u64, u64, u64, u32, u32, u32, u32, vp, vp],
    "rq_fused_round_noalloc": [cint, vp, u32, u32, u32] + _COMMIT_IN + _VOTE_IN + [i32p] + _VOTE_OUT + [u8p],
    "rq_plan_launch": [vp, vp],
    "rq_plan_destroy": [vp],
    "rq_stream_sync": [vp, vp],
    "rq_stream_wait_signal": [vp, vp, vp, u64],
    "rq_comm_unique_id": [u8p],
    "rq_comm_create": [cint, u8p, cint, cint, u32, C.POINTER(vp)],
    "rq_comm_destroy": [vp],
    "rq_comm_allgather": [vp, vp, u32, vp, vp, u64],
    "rq_comm_wait": [vp, vp, u32],
    "rq_comm_sync": [vp],
    "rq_comm_last_error": [vp, C.POINTER(cint), C.POINTER(cint)],
    "rq_device_alloc": [vp, u64, cint, C.POINTER(vp)],
    "rq_device_free": [vp, vp],
    "rq_host_alloc": [vp, u64, C.POINTER(vp)],
    "rq_host_free": [vp, vp],
    "rq_copy_to_device": [vp, vp, vp, u64],
    "rq_copy_from_device": [vp, vp, vp, u64],
}

_lib = None


class RqError(RuntimeError):
    def __init__(self, code, where, hip=0):
        self.code = code
        self.hip = hip
        msg = lib().rq_strerror(code).decode()
        if hip:
            msg += " [hipError %d]" % hip
        super().__init__("%s: %s (%d)" % (where, msg, code))


def lib():
    """Load librq.so once.  torch (if used by the process) must own the HIP
    runtime: it ships its own libamdhip64 with the same soname, so it is
    imported first and librq.so then binds to the already-loaded runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "raft_amd/librq.so is not built; run `make` (or __graft_entry__.build()) "
            "in the repository root. There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so first)
    except ImportError:
        pass
    l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, args in PROTOTYPES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = cint
    l.rq_strerror.argtypes = [cint]
    l.rq_strerror.restype = C.c_char_p
    _lib = l
    return l
