"""examples/rq_cli: the C-ABI driven from a plain C program in its own process
(no Python, no torch, the system HIP runtime), fed with the committed
known-answer vectors and a seeded batch, checked against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o
from raft_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "examples", "rq_cli")
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "kats.json")))


def run_cli(mode, b):
    G, n = int(b["G"]), int(b["n"])
    parts = ["%d %d" % (G, n)]
    for k in ("match", "commit", "leader_term", "log_off", "log_term", "voter_term", "voted_for",
              "last_log_term", "cand_term", "cand_id"):
        a = np.asarray(b[k])
        a = a[:, :G] if a.ndim == 2 else a
        parts.append(" ".join(str(int(x)) for x in a.ravel()))
    if mode in ("segmented", "rounds"):
        parts.append(" ".join(str(int(x)) for x in b["old_mask"][:G]))
        parts.append(" ".join(str(int(x)) for x in b["new_mask"][:G]))
    p = subprocess.run([CLI, mode], input="\n".join(parts) + "\n", capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    rows = [np.array(line.split(), dtype=np.int64) for line in p.stdout.strip().split("\n")]
    assert len(rows) == n + 6
    if mode == "segmented":
        run_cli.segments = int(p.stderr.split("rq_cli:")[-1].split()[0])
    return dict(commit=rows[0], voter_term=np.stack(rows[1:1 + n]), grant_mask=rows[1 + n],
                stepdown_mask=rows[2 + n], elected=rows[3 + n], max_term=rows[4 + n], status=rows[5 + n])


def test_cli_is_built_and_refuses_bad_input():
    assert os.access(CLI, os.X_OK), "run `make`"
    p = subprocess.run([CLI, "host"], input="0 0\n", capture_output=True, text=True, timeout=30)
    assert p.returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["host", "device"])
def test_cli_known_answers_and_seeded_batch(mode):
    # the commit and tally vectors side by side in one 5-slot batch
    ck = [k for k in KATS["commit"] if len(k["match"]) == 5]
    tk = [k for k in KATS["tally"] if k["n"] == 5]
    G = min(len(ck), len(tk))
    log_terms = [k["log_terms"] for k in ck[:G]]
    b = synth.Batch(
        G=G, n=5, ld=G,
        match=np.array([k["match"] for k in ck[:G]], np.int32).T.copy(),
        commit=np.array([k["commit"] for k in ck[:G]], np.int32),
        leader_term=np.array([k["leader_term"] for k in ck[:G]], np.int32),
        log_off=np.concatenate([[0], np.cumsum([len(t) for t in log_terms])]).astype(np.uint32),
        log_term=np.concatenate(log_terms).astype(np.int32),
        voter_term=np.array([[v["cur"] for v in k["voters"]] for k in tk[:G]], np.int32).T.copy(),
        voted_for=np.array([[v["voted_for"] for v in k["voters"]] for k in tk[:G]], np.int32).T.copy(),
        last_log_term=np.array([[v["last_log_term"] for v in k["voters"]] for k in tk[:G]], np.int32).T.copy(),
        cand_term=np.array([k["term"] for k in tk[:G]], np.int32),
        cand_id=np.array([k["cand"] for k in tk[:G]], np.int32))
    r = run_cli(mode, b)
    assert r["commit"].tolist() == [k["expect"] for k in ck[:G]]
    assert [bool(s & 1) for s in r["status"]] == [k["oob"] for k in ck[:G]]
    assert [bool(x) for x in r["elected"]] == [k["expect"]["elected"] for k in tk[:G]]
    assert r["max_term"].tolist() == [k["expect"]["max_term"] for k in tk[:G]]
    assert [bin(int(m)).count("1") for m in r["grant_mask"]] == [k["expect"]["votes"] for k in tk[:G]]
    # a seeded adversarial batch against the oracle
    b = synth.gen_adversarial(3001, 5, seed=8)
    r = run_cli(mode, b)
    exp = o.fused_round(*b.fused_args())
    for f in ("commit", "voter_term", "grant_mask", "stepdown_mask", "elected", "max_term", "status"):
        assert np.array_equal(r[f], exp[f].astype(np.int64)), f


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["segmented", "rounds"])
def test_cli_groups_of_different_sizes_from_plain_c(mode):
    """Mixed 3-9 clusters + joint consensus in size order from a C program: `segmented` fills the
    size table of rq_round_desc itself, `rounds` cuts the state into two images for one
    rq_plan_rounds launch -- both against the oracle's member-mask round."""
    b = synth.gen_mixed(2500, seed=77, bucketed="size")
    r = run_cli(mode, b)
    exp = o.fused_round(*b.fused_args(), G=b.G, old_mask=b.old_mask, new_mask=b.new_mask)
    for f in ("commit", "grant_mask", "stepdown_mask", "elected", "max_term", "status"):
        assert np.array_equal(r[f], exp[f].astype(np.int64)), f
    mem = (((b.old_mask | b.new_mask).astype(np.int64)[None, :] >> np.arange(b.n)[:, None]) & 1).astype(bool)
    assert np.array_equal(r["voter_term"][mem], exp["voter_term"][:, :b.G].astype(np.int64)[mem])
    if mode == "segmented":
        assert 8 <= run_cli.segments <= 16


@pytest.mark.gpu
def test_plain_c_benchmark_runs():
    """examples/rq_bench: device memory, a plan and the timed loop through the C-ABI alone."""
    exe = os.path.join(ROOT, "examples", "rq_bench")
    shares = []
    for layout in ("tiles", "planes"):                         # the same drawn batch in both layouts
        p = subprocess.run([exe, "65537", "5", "4", "60", "12", "8", layout], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        f = p.stdout.split()
        assert [int(x) for x in f[:4]] == [65537, 5, 4, 60]
        assert float(f[4]) > 0 and float(f[7]) > 0
        shares.append(p.stderr.split("elected")[1].split("%")[0])
    assert shares[0] == shares[1]                              # identical decisions
    assert 10 < float(shares[0]) < 90                          # a plausible share of the groups elects


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [0, 1])
def test_cli_replicate_round_from_plain_c(extra):
    """`rq_cli replicate`: one AppendEntries batch with the follower commit rule and the
    leader's commit round in the same launch, the two descriptors (one pointing at the other)
    filled in by a C program -- against the oracle's append_check, append_commit and
    commit_advance run in that order."""
    G, nf = 2003, 4
    nr = nf + extra
    b = synth.with_follower_commit(synth.gen_append(G, nf, seed=31 + extra), seed=2)
    c = synth.gen_fused(G, nr, seed=32 + extra)
    match = np.concatenate([b.match_index, c.match[nf:nr]]).astype(np.int32)
    b["leader_term"] = c.leader_term[:G].copy()
    parts = ["%d %d %d" % (G, nf, extra)]
    for a in (b.follower_term, b.leader_term, b.prev_log_index, b.prev_log_term, b.n_new, b.flog_off, b.flog_term,
              match, b.next_index, b.follower_commit, b.last_new_index, c.commit[:G], c.log_off, c.log_term):
        parts.append(" ".join(str(int(x)) for x in np.asarray(a).ravel()))
    p = subprocess.run([CLI, "replicate"], input="\n".join(parts) + "\n", capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    rows = [np.array(line.split(), dtype=np.int64) for line in p.stdout.strip().split("\n")]
    assert len(rows) == nf + 3 + nr + nf + nf + 2
    it = iter(rows)
    take = lambda k: np.stack([next(it) for _ in range(k)])
    ft, ok, sd, st = take(nf), next(it), next(it), next(it)
    m, ni, fc, commit, rst = take(nr), take(nf), take(nf), next(it), next(it)
    exp = o.append_check(b.follower_term, b.leader_term, b.prev_log_index, b.prev_log_term, b.flog_off, b.flog_term,
                         n_new=b.n_new, match_index=b.match_index, next_index=b.next_index)
    exp_fc = o.append_commit(exp, c.commit[:G], b.last_new_index, b.follower_commit, n_new=b.n_new)
    m2 = match.copy(); m2[:nf] = exp["match_index"]
    exp_c, exp_rst = o.commit_advance(m2, c.commit, c.leader_term, c.log_off, c.log_term)
    assert np.array_equal(ft, exp["follower_term"]) and np.array_equal(ok, exp["success_mask"])
    assert np.array_equal(sd, exp["stepdown_mask"]) and np.array_equal(st, exp["status"])
    assert np.array_equal(m, m2) and np.array_equal(ni, exp["next_index"]) and np.array_equal(fc, exp_fc)
    assert np.array_equal(commit, exp_c) and np.array_equal(rst, exp_rst)
    assert (exp_c != c.commit[:G]).any() and (exp_fc != b.follower_commit).any()
