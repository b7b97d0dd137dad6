"""rq_fused_round_noalloc -- the host-buffer fused round staged through a fixed
array inside the library's device image -- through the C-ABI, bit for bit against
the CPU oracle (the checker) and against rq_fused_round (the entry point it is a
variant of) on the same inputs."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as o
from raft_amd import synth

pytestmark = pytest.mark.gpu

i32p, u32p, u16p, u8p = (C.POINTER(t) for t in (C.c_int32, C.c_uint32, C.c_uint16, C.c_uint8))


def _noalloc(l, b, stream=None, device=0):
    G, n, ld = int(b["G"]), int(b["n"]), int(b["ld"])
    ins = [np.ascontiguousarray(b[k]) for k in ("match", "commit", "leader_term", "log_off", "log_term",
                                                "voter_term", "voted_for", "last_log_term", "cand_term", "cand_id")]
    if ins[4].size == 0:
        ins[4] = np.zeros(1, np.int32)
    types = [i32p, i32p, i32p, u32p, i32p, i32p, i32p, i32p, i32p, i32p]
    out = dict(commit=np.full(G, -7, np.int32), voter_term=np.full((n, ld), -7, np.int32),
               grant_mask=np.zeros(G, np.uint16), stepdown_mask=np.zeros(G, np.uint16),
               elected=np.zeros(G, np.uint8), max_term=np.zeros(G, np.int32), status=np.zeros(G, np.uint8))
    otypes = [i32p, i32p, u16p, u16p, u8p, i32p, u8p]
    rc = l.rq_fused_round_noalloc(device, stream, G, n, ld,
                                  *[a.ctypes.data_as(t) for a, t in zip(ins, types)],
                                  *[a.ctypes.data_as(t) for a, t in zip(out.values(), otypes)])
    return rc, out


def _same(out, ref, G):
    assert np.array_equal(out["commit"], ref.commit)
    assert np.array_equal(out["voter_term"][:, :G], ref.voter_term[:, :G])
    for f in ("grant_mask", "stepdown_mask", "elected", "max_term", "status"):
        assert np.array_equal(out[f], getattr(ref, f)), f


def _check(out, engine, b):
    """First the oracle (what decides right and wrong), then the engine's own entry point."""
    _same(out, o.fused_round(*b.fused_args(), G=b.G), b.G)
    _same(out, engine.fused_round(*b.fused_args(), G=b.G), b.G)


def test_noalloc_round_equals_the_oracle_and_the_engine_round(engine):
    """Two calls in a row, the second smaller (it must not see the first one's
    data), then an adversarial batch, a padded stride and other cluster sizes."""
    from raft_amd import _lib
    l = _lib.lib()
    cases = [synth.gen_fused(30000, 5, seed=1), synth.gen_fused(1777, 5, seed=2),
             synth.gen_adversarial(9001, 7, seed=3, ld=9088), synth.gen_fused(1, 1, seed=4),
             synth.gen_adversarial(4000, 16, seed=5), synth.gen_fused(65, 3, seed=6, ld=68)]
    for b in cases:
        rc, out = _noalloc(l, b)
        assert rc == 0, rc
        _check(out, engine, b)


def test_noalloc_round_limits_and_errors(engine):
    from raft_amd import _lib
    l = _lib.lib()
    big = synth.gen_fused(200000, 5, seed=9)                  # ~29 MB of staging: beyond the 16 MiB array
    rc, _ = _noalloc(l, big)
    assert rc == -2                                           # RQ_E_RANGE, nothing ran
    b = synth.gen_fused(100, 3, seed=1)
    bad = synth.Batch(b); bad["log_off"] = b.log_off[::-1].copy()
    assert _noalloc(l, bad)[0] == -1                          # offsets not monotone
    assert _noalloc(l, b, device=-1)[0] == -1
    assert _noalloc(l, b, device=99)[0] in (-1, -4)           # no such device
    rc, out = _noalloc(l, b)                                  # still works after the refusals
    assert rc == 0
    _check(out, engine, b)
    # on a caller's stream
    import torch
    s = torch.cuda.Stream()
    rc, out = _noalloc(l, b, stream=s.cuda_stream)
    assert rc == 0
    _check(out, engine, b)
