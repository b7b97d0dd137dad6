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
                 "rq_step", "rq_download", "rq_round_dev", "rq_plan_round", "rq_append_check_dev",
                 "rq_strerror", "rq_version"):
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
    assert l.rq_version() == 0x000300
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
                assert "oracle" not in src.lower(), os.path.join(dp, fn)


def test_descriptor_structs_match_the_header():
    """The ctypes mirrors of rq_round_desc / rq_append_desc have the C structs' sizes
    and field order (a C probe compiled against include/rq.h prints sizeof and every
    field's offset)."""
    import subprocess
    import tempfile
    from raft_amd import _lib
    lines = ['#include "rq.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for cname, cls in (("rq_round_desc", _lib.RoundDesc), ("rq_append_desc", _lib.AppendDesc)):
        lines.append('printf("%%zu\\n", sizeof(%s));' % cname)
        for fname, _ in cls._fields_:
            lines.append('printf("%%zu\\n", offsetof(%s, %s));' % (cname, fname))
    lines += ["return 0; }"]
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.c")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(td, "probe")], check=True)
        out = subprocess.run([os.path.join(td, "probe")], check=True, capture_output=True, text=True).stdout.split()
    vals = iter(int(x) for x in out)
    for cls in (_lib.RoundDesc, _lib.AppendDesc):
        assert next(vals) == C.sizeof(cls)
        for fname, _ in cls._fields_:
            assert next(vals) == getattr(cls, fname).offset, (cls.__name__, fname)


def test_device_entry_points_reject_bad_descriptors_without_a_gpu():
    from raft_amd import _lib
    l = _lib.lib()
    plan = C.c_void_p()
    d = _lib.RoundDesc()
    assert l.rq_round_dev(None, None, C.byref(d)) == -1
    assert l.rq_plan_round(None, C.byref(d), C.byref(plan)) == -1
    a = _lib.AppendDesc()
    assert l.rq_append_check_dev(None, None, C.byref(a)) == -1
    assert l.rq_summary_apply_dev(None, None, None, 0, 0, 0, 1, 1, 1, 8, None, None) == -1

