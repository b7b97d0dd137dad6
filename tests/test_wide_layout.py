"""Wide (int64) layout: the rq_round64* entry points and the oracle's
orc_round64.

CPU part: orc_round64 is pinned against the compact-layout oracle on inputs both
can represent, against the known-answer vectors, and against hand-derived
64-bit cases.  GPU part (marked gpu): the HIP kernels through the C-ABI against
orc_round64, bit-exact.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as o
from raft_amd import synth

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))
I64MIN, I64MAX = np.iinfo(np.int64).min, np.iinfo(np.int64).max
OUT_FIELDS = ("commit", "voter_term", "grant_mask", "stepdown_mask", "elected", "max_term", "status")
ARGS = ("match", "commit", "leader_term", "log_off", "log_term", "voter_term", "voted_for",
        "last_log_term", "cand_term", "cand_id")


def wide_kwargs(w):
    kw = {k: w[k] for k in ARGS}
    kw["G"] = int(w["G"])
    for k in ("old_mask", "new_mask", "log_base"):
        if k in w:
            kw[k] = w[k]
    return kw


def members_of(w):
    G, m = int(w["G"]), w["match"].shape[0]
    if "old_mask" not in w:
        return np.ones((m, G), bool)
    mem = (w["old_mask"] | w["new_mask"]).astype(np.int64)[:G]
    return ((mem[None, :] >> np.arange(m)[:, None]) & 1).astype(bool)


def assert_same(got, exp, w, what=3):
    G = int(w["G"])
    mem = members_of(w)
    for f in OUT_FIELDS:
        a = got[f] if isinstance(got, dict) else getattr(got, f)
        e = exp[f]
        if e is None:
            continue
        if f == "voter_term":
            assert np.array_equal(a[:, :G][mem], e[:, :G][mem]), f
        else:
            assert np.array_equal(a[:G], e[:G]), f


# ---------------------------------------------------------------------------
# CPU: the wide oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 9, 12, 16])
def test_oracle64_equals_compact_oracle(n):
    for gen in (synth.gen_fused, synth.gen_adversarial):
        b = gen(2000, n, seed=31 * n)
        r32 = o.fused_round(*b.fused_args(), G=b.G)
        w, _ = synth.widen(b)
        r64 = o.round64(3, **wide_kwargs(w))
        for f in OUT_FIELDS:
            assert np.array_equal(r32[f].astype(np.int64), r64[f].astype(np.int64)), (n, gen.__name__, f)
        # one decision at a time gives the same answers as both at once
        rc = o.round64(1, **wide_kwargs(w)); rv = o.round64(2, **wide_kwargs(w))
        assert np.array_equal(rc.commit, r64.commit) and np.array_equal(rv.elected, r64.elected)
        assert np.array_equal(rc.status | rv.status, r64.status)


def test_oracle64_joint_and_window_equal_compact_oracle():
    b = synth.gen_mixed(4000, seed=9)
    r32 = o.fused_round(*b.fused_args(), G=b.G, old_mask=b.old_mask, new_mask=b.new_mask)
    w, _ = synth.widen(b)
    assert_same(o.round64(3, **wide_kwargs(w)), {f: r32[f] for f in OUT_FIELDS}, w)
    bw = synth.window_logs(synth.gen_fused(3000, 5, seed=4), keep_committed=1)
    r32 = o.fused_round(*bw.fused_args(), G=bw.G, log_base=bw.log_base)
    w, _ = synth.widen(bw)
    assert_same(o.round64(3, **wide_kwargs(w)), {f: r32[f] for f in OUT_FIELDS}, w)


def shifted_expectation(b, r32, dshift, term_shift):
    """What the compact oracle's answers become when the batch is widened with shifts."""
    G = int(b["G"])
    exp = dict(commit=r32["commit"].astype(np.int64) + dshift[:G],
               voter_term=r32["voter_term"].astype(np.int64) + term_shift,
               max_term=r32["max_term"].astype(np.int64) + term_shift)
    for f in ("grant_mask", "stepdown_mask", "elected", "status"):
        exp[f] = r32[f]
    return exp


def test_oracle64_is_invariant_under_order_preserving_shifts():
    for b, kw in ((synth.gen_fused(3000, 5, seed=2), {}),
                  (synth.gen_mixed(3000, seed=3), None)):
        if kw is None:
            kw = dict(old_mask=b.old_mask, new_mask=b.new_mask)
        r32 = o.fused_round(*b.fused_args(), G=b.G, **kw)
        w, d = synth.widen(b, index_shift=1 << 40, term_shift=1 << 45, id_shift=1 << 50)
        assert w["match"].max() > 2 ** 32 and w["cand_term"].min() > 2 ** 32
        assert_same(o.round64(3, **wide_kwargs(w)), shifted_expectation(b, r32, d, 1 << 45), w)


def test_oracle64_vote_kats():
    ks = KATS["vote"]
    for shift in (0, 1 << 40):
        vt = np.array([[k["voter"]["cur"] + shift for k in ks]], np.int64)
        vf = np.array([[k["voter"]["voted_for"] for k in ks]], np.int64)
        ll = np.array([[k["voter"]["last_log_term"] + shift for k in ks]], np.int64)
        r = o.round64(2, voter_term=vt, voted_for=vf, last_log_term=ll,
                      cand_term=[k["req"]["term"] + shift for k in ks], cand_id=[k["req"]["cand"] for k in ks])
        for g, k in enumerate(ks):
            e = k["expect"]
            assert r.voter_term[0, g] == e["cur"] + shift, k["id"]
            assert r.max_term[g] == e["term_result"] + shift, k["id"]
            assert bool(r.grant_mask[g] & 1) == e["grant"], k["id"]


# hand-derived 64-bit cases: (match, commit, leader_term, log_terms, log_base) -> (commit', oob)
WIDE_COMMIT_KATS = [
    # quorum of 5 = third largest = 2^40+5; window starts at 2^40, slot 5 holds term 7
    ([2 ** 40 + 7, 2 ** 40 + 3, 2 ** 40 + 9, 2 ** 40 + 5, 0], 2 ** 40 + 4, 7, [1, 1, 2, 2, 3, 7, 7, 8, 8, 9], 2 ** 40, 2 ** 40 + 5, False),
    # same, leader term differs from log[N].term: no advance
    ([2 ** 40 + 7, 2 ** 40 + 3, 2 ** 40 + 9, 2 ** 40 + 5, 0], 2 ** 40 + 4, 8, [1, 1, 2, 2, 3, 7, 7, 8, 8, 9], 2 ** 40, 2 ** 40 + 4, False),
    # N below the window base: the entry is not on the device -> out of range, no advance
    ([2 ** 40 - 1] * 3, -5, 1, [1, 1], 2 ** 40, -5, True),
    # N = INT64_MAX with base 0: far outside a 2-entry log
    ([I64MAX, I64MAX, 0], 0, 1, [1, 1], 0, 0, True),
    # N = INT64_MAX inside a window ending exactly there (base = INT64_MAX - 1)
    ([I64MAX, I64MAX, 0], 0, 4, [3, 4], I64MAX - 1, I64MAX, False),
    # everything at INT64_MIN: N == commit, not greater -> unchanged, log never read
    ([I64MIN] * 5, I64MIN, 1, [], 0, I64MIN, False),
    # commit = INT64_MIN, N = -1: greater, but index -1 is outside any log
    ([-1, -1, -1], I64MIN, 1, [1], 0, I64MIN, True),
    # a value that aliases 5 in its low 32 bits must not be treated as 5
    ([2 ** 32 + 5] * 3, 0, 1, [1] * 8, 0, 0, True),
]


def run_wide_commit_kats(fn):
    for match, commit, lt, log_terms, base, expect, oob in WIDE_COMMIT_KATS:
        n = len(match)
        r = fn(1, match=np.array(match, np.int64).reshape(n, 1), commit=[commit], leader_term=[lt],
               log_off=[0, len(log_terms)], log_term=np.array(log_terms, np.int64), log_base=[base])
        assert int(r.commit[0]) == expect, (match, commit, base)
        assert bool(r.status[0] & 1) == oob, (match, commit, base)


WIDE_VOTE_KATS = [
    # (voter cur, votedFor, lastLogTerm) x (term, cand) -> (cur', grant, stepdown, empty-log status)
    ((2 ** 40, -1, 2 ** 40), (2 ** 40 + 1, 2 ** 35), (2 ** 40 + 1, True, True, False)),
    ((2 ** 40 + 2, -1, 1), (2 ** 40 + 1, 7), (2 ** 40 + 2, False, False, False)),     # stale candidate
    ((5, 2 ** 35, 1), (5, 2 ** 35), (5, True, False, False)),                         # voted for this 64-bit id
    ((5, 2 ** 35, 1), (5, 2 ** 35 + 2 ** 32), (5, False, False, False)),              # ids differ above bit 32
    ((5, -1, 2 ** 33), (6, 1), (6, False, True, False)),                              # voter log term ahead
    ((I64MIN, -1, I64MIN + 1), (I64MIN, 3), (I64MIN, False, False, False)),           # voter's log term above the term
    ((I64MIN + 1, -1, I64MIN + 1), (I64MIN + 1, 3), (I64MIN + 1, True, False, False)),  # extremes, equal terms
    ((I64MAX, -1, 0), (I64MAX, 3), (I64MAX, True, False, False)),
    ((I64MAX - 1, -1, 0), (I64MAX, 3), (I64MAX, True, True, False)),
    ((4, -1, I64MIN), (4, 3), (4, False, False, True)),                               # empty log reached
    ((4, -1, I64MIN), (3, 3), (4, False, False, False)),                              # rejected before the log is read
    ((4, -1, -(2 ** 31)), (4, 3), (4, True, False, False)),                           # the compact layout's "empty" is a value here
]


def run_wide_vote_kats(fn):
    for (cur, vf, llt), (term, cand), (cur2, grant, stepdown, empty) in WIDE_VOTE_KATS:
        r = fn(2, voter_term=np.array([[cur]], np.int64), voted_for=np.array([[vf]], np.int64),
               last_log_term=np.array([[llt]], np.int64), cand_term=[term], cand_id=[cand])
        case = (cur, vf, llt, term, cand)
        assert int(r.voter_term[0, 0]) == cur2, case
        assert int(r.max_term[0]) == cur2, case
        assert bool(r.grant_mask[0] & 1) == grant, case
        assert bool(r.stepdown_mask[0] & 1) == stepdown, case
        assert bool(r.status[0] & 2) == empty, case
        assert bool(r.elected[0]) == (grant and cur2 == term), case


def test_oracle64_wide_kats():
    run_wide_commit_kats(o.round64)
    run_wide_vote_kats(o.round64)


def test_oracle64_adversarial_extremes_against_bruteforce():
    """numpy restatement of the two rules for fixed-size groups, on int64 extremes."""
    n, G = 5, 4000
    w = synth.gen_adversarial64(G, n, seed=12)
    r = o.round64(3, **wide_kwargs(w))
    N = np.sort(w["match"][:, :G], axis=0)[(n - 1) // 2]
    lens = np.diff(w["log_off"].astype(np.int64))[:G]
    adv = N > w["commit"][:G]
    inb = adv & (N >= 0) & (N < lens)
    term = np.zeros(G, np.int64)
    term[inb] = w["log_term"][w["log_off"].astype(np.int64)[:G][inb] + N[inb]]
    exp_commit = np.where(inb & (term == w["leader_term"][:G]), N, w["commit"][:G])
    assert np.array_equal(r.commit, exp_commit)
    assert np.array_equal((r.status & 1).astype(bool), adv & ~inb)
    ct = w["cand_term"][None, :G]
    reach = ct >= w["voter_term"][:, :G]
    empty = w["last_log_term"][:, :G] == I64MIN
    free = (w["voted_for"][:, :G] == -1) | (w["voted_for"][:, :G] == w["cand_id"][None, :G])
    grant = reach & free & ~empty & (w["last_log_term"][:, :G] <= ct)
    assert np.array_equal(r.grant_mask, (grant * (1 << np.arange(n))[:, None]).sum(0).astype(np.uint16))
    assert np.array_equal(r.voter_term[:, :G], np.maximum(w["voter_term"][:, :G], ct))
    assert np.array_equal((r.status & 2).astype(bool), (reach & empty).any(0))


# ---------------------------------------------------------------------------
# GPU: rq_round64 (host buffers) and rq_round_dev with RQ_LAYOUT_I64 (device buffers)
# ---------------------------------------------------------------------------
def engine_round64(engine):
    def fn(what, **kw):
        return engine.round64(what, **kw)
    return fn


@pytest.mark.gpu
def test_gpu_wide_kats(engine):
    run_wide_commit_kats(engine_round64(engine))
    run_wide_vote_kats(engine_round64(engine))


@pytest.mark.gpu
@pytest.mark.parametrize("n", list(range(1, 10)) + [10, 13, 16])
def test_gpu_wide_every_cluster_size(engine, n):
    for G, ld in ((1, 1), (63, 63), (65, 68), (1000, 1000), (4099, 4352), (10000, 10000)):
        for gen in (synth.gen_fused, synth.gen_adversarial64):
            b = gen(G, n, seed=500 * n + G, ld=ld)
            w = synth.widen(b)[0] if gen is synth.gen_fused else b
            kw = wide_kwargs(w)
            exp = o.round64(3, **kw)
            assert_same(engine.round64(3, **kw), exp, w)
            c = engine.round64(1, **kw)
            v = engine.round64(2, **kw)
            assert np.array_equal(c.commit, exp.commit), (n, G)
            assert np.array_equal(v.elected, exp.elected) and np.array_equal(v.max_term, exp.max_term), (n, G)
            assert np.array_equal(c.status | v.status, exp.status), (n, G)


@pytest.mark.gpu
@pytest.mark.parametrize("bucketed", [True, False])
def test_gpu_wide_mixed_joint_shifted_window(engine, bucketed):
    b = synth.gen_mixed(20011, seed=21, bucketed=bucketed)
    w, d = synth.widen(b, index_shift=1 << 40, term_shift=1 << 45, id_shift=1 << 50)
    kw = wide_kwargs(w)
    exp = o.round64(3, **kw)
    assert_same(engine.round64(3, **kw), exp, w)
    # and the compact engine on the unshifted batch agrees through the shift map
    r32 = engine.fused_round(*b.fused_args(), G=b.G, old_mask=b.old_mask, new_mask=b.new_mask)
    got32 = {f: getattr(r32, f) for f in OUT_FIELDS}
    assert_same(exp, shifted_expectation(b, got32, d, 1 << 45), w)


@pytest.mark.gpu
def test_gpu_wide_errors(engine):
    from raft_amd._lib import RqError
    b = synth.gen_fused(100, 3, seed=1)
    w, _ = synth.widen(b)
    kw = wide_kwargs(w)
    with pytest.raises(RqError):          # one mask only
        engine.round64(3, old_mask=np.full(100, 7, np.uint16), **kw)
    with pytest.raises(RqError):          # empty configuration
        engine.round64(3, old_mask=np.zeros(100, np.uint16), new_mask=np.full(100, 7, np.uint16), **kw)
    with pytest.raises(RqError):          # no decision selected
        engine.round64(0, **kw)
    kw2 = dict(kw); kw2["log_off"] = kw["log_off"][::-1].copy()
    with pytest.raises(RqError):          # offsets not monotone
        engine.round64(1, **kw2)
    r = engine.round64(3, **dict(kw, G=0))
    assert r.commit.size == 0


@pytest.fixture(scope="module")
def dev():
    import torch
    from raft_amd.device import DeviceEngine
    assert torch.cuda.is_available()
    d = DeviceEngine(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 0x001, 0x100, 0x201, 0x400])
def test_gpu_wide_device_launch(dev, variant):
    import torch
    from raft_amd.device import DeviceBatch
    for b in (synth.gen_fused(70001, 5, seed=5), synth.gen_mixed(30011, seed=6),
              synth.gen_fused(4097, 12, seed=7, ld=4097 + 63)):
        w, d = synth.widen(b, index_shift=1 << 36, term_shift=1 << 33, id_shift=1 << 34)
        G = int(w["G"])
        lo16 = torch.zeros(G, dtype=torch.int16, device="cuda:0")
        esc = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        base = torch.from_numpy(np.ascontiguousarray(w["commit"][:G] - 3)).to("cuda:0")
        db = DeviceBatch(w, commit_lo16=lo16, summary_escapes=esc, summary_base=base)
        dev.round64(db, 3, variant)
        got = db.results()
        exp = o.round64(3, **wide_kwargs(w))
        assert_same(got, exp, w)
        bits = np.unpackbits(got["elected_bits"].view(np.uint8), bitorder="little")[:G]
        assert np.array_equal(bits, exp.elected)
        delta = exp.commit - (w["commit"][:G] - 3)
        fit = (delta >= 0) & (delta < 65535)
        assert np.array_equal(got["commit_lo16"], np.where(fit, delta, 65535).astype(np.uint16))
        assert got["summary_escapes"] == int((~fit).sum())
        # the single-decision launches
        db2 = DeviceBatch(w)
        dev.round64(db2, 1, variant); dev.round64(db2, 2, variant)
        assert_same(db2.results(), exp, w)


@pytest.mark.gpu
def test_gpu_wide_full_size_1m_by_5(dev):
    """1M groups x 5 slots in the wide layout: shifted out of int32 range, the
    decisions must be the compact kernel's decisions moved by the same shifts;
    a slice of it is also checked against the oracle directly."""
    import torch
    from raft_amd.device import DeviceBatch
    G = 1 << 20
    b = synth.gen_fused(G, 5)
    w, d = synth.widen(b, index_shift=1 << 41, term_shift=1 << 42, id_shift=1 << 43)
    db32, db64 = DeviceBatch(b), DeviceBatch(w)
    dev.fused_round(db32); dev.fused_round(db64)
    r32, r64 = db32.results(), db64.results()
    assert_same(r64, shifted_expectation(b, r32, d, 1 << 42), w)
    assert np.array_equal(r64["elected_bits"], r32["elected_bits"])
    assert 0.2 < r64["elected"].mean() < 0.8 and (r64["commit"] != w["commit"]).mean() > 0.2
    sl = slice(G - 50000, G)
    sub = {k: (w[k][..., sl] if k not in ("log_off", "log_term") else w[k]) for k in ARGS}
    off = w["log_off"].astype(np.int64)
    sub["log_off"] = (off[G - 50000:] - off[G - 50000]).astype(np.uint32)
    sub["log_term"] = w["log_term"][off[G - 50000]:]
    sub["log_base"] = w["log_base"][sl]
    exp = o.round64(3, **{k: np.ascontiguousarray(v) for k, v in sub.items()})
    assert np.array_equal(r64["commit"][sl], exp.commit)
    assert np.array_equal(r64["elected"][sl], exp.elected)
    assert np.array_equal(r64["voter_term"][:, sl], exp.voter_term)


# ---------------------------------------------------------------------------
# AppendEntries consistency batch and follower clamp in the wide layout
# ---------------------------------------------------------------------------
APPEND_PLANES = ("follower_term", "prev_log_index", "prev_log_term", "n_new", "match_index", "next_index")


def widen_append(b, term_shift=0):
    """int64 image of a gen_append batch; term_shift moves every term (follower,
    leader, prevLogTerm, log terms) up without changing any comparison."""
    w = synth.Batch(b)
    for k in APPEND_PLANES + ("leader_term", "flog_term"):
        w[k] = np.asarray(b[k]).astype(np.int64)
    for k in ("follower_term", "leader_term", "prev_log_term", "flog_term"):
        w[k] = w[k] + term_shift
    return w


def append_call(fn, w, full=True, **extra):
    kw = dict(n_new=w.n_new, match_index=w.match_index, next_index=w.next_index,
              active_mask=w.active_mask, G=w.G) if full else dict(G=w.G)
    kw.update(extra)
    return fn(w.follower_term, w.leader_term, w.prev_log_index, w.prev_log_term, w.flog_off, w.flog_term, **kw)


def assert_append_equal(got, exp, G, full, term_shift=0):
    assert np.array_equal(got["follower_term"][:, :G], exp["follower_term"][:, :G].astype(np.int64) + term_shift)
    for f in ("success_mask", "stepdown_mask", "status"):
        assert np.array_equal(got[f][:G], exp[f][:G]), f
    if full:
        assert np.array_equal(got["match_index"][:, :G], exp["match_index"][:, :G])
        assert np.array_equal(got["next_index"][:, :G], exp["next_index"][:, :G])


@pytest.mark.parametrize("n", [1, 3, 5, 9, 16])
def test_oracle64_append_check_and_clamp_equal_compact_oracle(n):
    for adv in (False, True):
        b = synth.gen_append(3000, n, seed=3 * n, adversarial=adv)
        exp = append_call(o.append_check, b)
        for shift in (0, 1 << 40):
            assert_append_equal(append_call(o.append_check64, widen_append(b, shift)), exp, b.G, True, shift)
    rng = np.random.default_rng(n)
    lc, ci, ln = (rng.integers(-5, 50, 1000) for _ in range(3))
    assert np.array_equal(o.follower_clamp64(lc, ci, ln), o.follower_clamp(lc, ci, ln))
    big = 1 << 45
    assert np.array_equal(o.follower_clamp64(lc + big, ci + big, ln + big), o.follower_clamp(lc, ci, ln).astype(np.int64) + big)


def wide_append_kats(fn):
    """prevLogIndex values that alias a valid slot in their low 32 bits must be out of range."""
    n, G = 3, 1
    off = np.arange(0, 2 * n + 1, 2, dtype=np.uint32)
    flog = np.tile(np.array([0, 1], np.int64) + (1 << 40), n)
    ft = np.full((n, G), (1 << 40) + 1, np.int64)
    pi = np.array([[1], [(1 << 32) + 1], [-1]], np.int64)
    pt = np.full((n, G), (1 << 40) + 1, np.int64)
    r = fn(ft, [(1 << 40) + 2], pi, pt, off, flog, n_new=np.full((n, G), 2, np.int64),
           match_index=np.zeros((n, G), np.int64), next_index=np.full((n, G), 5, np.int64))
    assert r["success_mask"][0] == 0b001 and r["stepdown_mask"][0] == 0b111 and r["status"][0] == 1
    assert r["follower_term"][:, 0].tolist() == [(1 << 40) + 2] * 3
    assert r["match_index"][:, 0].tolist() == [3, 0, 0] and r["next_index"][:, 0].tolist() == [4, 4, 4]


def test_oracle64_append_kats():
    wide_append_kats(o.append_check64)


@pytest.mark.gpu
def test_gpu_wide_append_kats(engine):
    wide_append_kats(lambda *a, **k: engine.append_check(*a, wide=True, **k))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 9, 16])
def test_gpu_wide_append_check(engine, n):
    for G, ld in ((1, 1), (65, 68), (5000, 5000), (30011, 30080)):
        for adv in (False, True):
            b = synth.gen_append(G, n, seed=7 * n + G, ld=ld, adversarial=adv)
            w = widen_append(b, 1 << 40)
            for full in (True, False):
                got = append_call(engine.append_check, w, full, wide=True)
                exp = append_call(o.append_check64, w, full)
                assert_append_equal(got, exp, G, full)


@pytest.mark.gpu
def test_gpu_wide_follower_clamp(engine):
    rng = np.random.default_rng(3)
    for count in (1, 3, 4, 5, 1023, 100003):
        lc, ci, ln = (rng.integers(-5, 50, count) + (1 << 44) for _ in range(3))
        assert np.array_equal(engine.follower_clamp(lc, ci, ln, wide=True), o.follower_clamp64(lc, ci, ln))
    ext = np.array([I64MIN, -1, 0, 1, I64MAX], np.int64)
    lc, ci, ln = (a.ravel() for a in np.meshgrid(ext, ext, ext))
    assert np.array_equal(engine.follower_clamp(lc, ci, ln, wide=True), o.follower_clamp64(lc, ci, ln))


@pytest.mark.gpu
def test_gpu_wide_plan_equals_direct_launch(dev):
    """rq_plan_round + rq_plan_launch against rq_round_dev in the wide layout, fixed and joint, every `what`."""
    from raft_amd.device import DeviceBatch
    from raft_amd._lib import RqError
    for b in (synth.gen_fused(50021, 5, seed=15), synth.gen_mixed(20003, seed=16)):
        w, _ = synth.widen(b, term_shift=1 << 37)
        exp = o.round64(3, **wide_kwargs(w))
        for what in (1, 2, 3):
            d1, d2 = DeviceBatch(w), DeviceBatch(w)
            for _ in range(2):                       # the second launch reuses the cached plan
                dev.round64(d1, what)
            dev.round_direct(d2, what)
            r1, r2 = d1.results(), d2.results()
            for f in OUT_FIELDS:
                assert np.array_equal(r1[f], r2[f]), (f, what)
            if what == 3:
                assert_same(r1, exp, w)
    import ctypes as C
    plan = C.c_void_p()
    d = dev.describe(d1, 3)
    for field, bad in (("what", 0), ("what", 4), ("match", None), ("layout", 6), ("struct_size", 8), ("n", 17),
                       ("log_term_bits", 12), ("reserved", 1)):
        e = type(d).from_buffer_copy(d)
        setattr(e, field, bad)
        assert dev._l.rq_plan_round(dev._h, C.byref(e), C.byref(plan)) == -1, field
        assert dev._l.rq_round_dev(dev._h, None, C.byref(e)) == -1, field
    assert dev._l.rq_plan_round(dev._h, C.byref(d), None) == -1
