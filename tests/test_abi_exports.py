"""CPU tests of the drop-in boundary: librq.so loads, exports every symbol
include/rq.h declares, and its argument checks work without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rq_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("rq_engine_create", "rq_engine_destroy", "rq_commit_advance", "rq_vote_tally",
                 "rq_fused_round", "rq_fused_round_joint", "rq_follower_clamp", "rq_upload",
                 "rq_step", "rq_download", "rq_fused_round_dev", "rq_strerror", "rq_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from raft_amd import _lib
    l = _lib.lib()
    for name in _declared():
        assert hasattr(l, name), name
    assert set(_lib.PROTOTYPES) | {"rq_strerror"} == set(_declared())


def test_version_and_strerror():
    from raft_amd import _lib
    l = _lib.lib()
    assert l.rq_version() == 0x000100
    assert l.rq_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert l.rq_strerror(code) not in (b"", b"unknown rq error")
    assert l.rq_strerror(-99) == b"unknown rq error"


def test_create_rejects_bad_arguments_and_fails_loudly_without_gpu():
    from raft_amd import _lib
    import torch
    l = _lib.lib()
    h = C.c_void_p()
    assert l.rq_engine_create(0, 0, 5, 16, C.byref(h)) == -1          # zero groups
    assert l.rq_engine_create(0, 16, 17, 16, C.byref(h)) == -1        # too many slots
    assert l.rq_engine_create(-1, 16, 5, 16, C.byref(h)) == -1
    assert l.rq_engine_create(0, 16, 5, 16, None) == -1
    if not torch.cuda.is_available():
        # no CPU fallback: creation must fail, not degrade
        assert l.rq_engine_create(0, 16, 5, 16, C.byref(h)) == -5
        assert not h.value
        from raft_amd import QuorumEngine, RqError
        with pytest.raises(RqError):
            QuorumEngine(16)
    assert l.rq_engine_destroy(None) == -1
    assert l.rq_step(None, 1) == -1


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "raft_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.lower() or fn == "__init__.py" and False, os.path.join(dp, fn)
