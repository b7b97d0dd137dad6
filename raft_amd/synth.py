"""Deterministic synthetic Raft-group batches (counter-based splitmix64).

Every value is a pure function of (seed, field id, flat index), so any slice
of a batch can be regenerated anywhere (host language, rank, test) without
shipping fixtures.  Layout matches the rq_* C-ABI: slot-major int32 planes
field[p, g], CSR leader logs (log_off uint32 [G+1], log_term int32 arena).
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
DEFAULT_SEED = 0x5EED0001
LOG_EMPTY = np.iinfo(np.int32).min

# field ids
F_LOGLEN, F_LOGT, F_MATCH, F_COMMIT, F_LTERM, F_CTERM, F_CID, F_VTERM, F_VFOR, F_VLLT, \
    F_NSLOT, F_JOINT, F_AUX, F_AUX2 = range(1, 15)


def splitmix64(x):
    """One splitmix64 output for each uint64 counter in x."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def draw(seed, field, idx):
    """uint64 draws for flat indices idx (array) of one field."""
    with np.errstate(over="ignore"):
        key = splitmix64(np.uint64(seed) ^ (np.uint64(field) * np.uint64(0xD1B54A32D192ED03)))
        return splitmix64(key ^ splitmix64(np.asarray(idx, dtype=np.uint64)))


def _below(seed, field, idx, k):
    """Integer draws in [0, k) (k scalar or array, k >= 1) as int64."""
    return (draw(seed, field, idx) % np.asarray(k, dtype=np.uint64)).astype(np.int64)


def _unit(seed, field, idx):
    """Uniform doubles in [0, 1)."""
    return (draw(seed, field, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def quorum_index(match, members=None):
    """ascending_sorted(match over slots)[(n-1)//2] per group (numpy; used only
    to SHAPE the synthetic distributions, e.g. 'about half the groups advance')."""
    m = np.sort(match.astype(np.int64), axis=0)
    return m[(match.shape[0] - 1) // 2]


class Batch(dict):
    """dict of numpy arrays with attribute access; keys follow the C-ABI names."""
    __getattr__ = dict.__getitem__

    def commit_args(self):
        return (self.match, self.commit, self.leader_term, self.log_off, self.log_term)

    def vote_args(self):
        return (self.voter_term, self.voted_for, self.last_log_term, self.cand_term, self.cand_id)

    def fused_args(self):
        return self.commit_args() + self.vote_args()


def _csr(lens):
    off = np.zeros(lens.size + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    if off[-1] > 0xFFFFFFFF:
        raise ValueError("log arena exceeds uint32 offsets")
    return off.astype(np.uint32)


def _monotone_log(seed, G, lens, off, tmax):
    """term_at(i) = 1 + floor(i * T_g / len_g): non-decreasing terms 1..T_g."""
    total = int(off[-1])
    grp = np.repeat(np.arange(G, dtype=np.int64), lens)
    pos = np.arange(total, dtype=np.int64) - off[:-1].astype(np.int64)[grp]
    T = 1 + _below(seed, F_LOGT, np.arange(G), tmax)
    return (1 + (pos * T[grp]) // lens[grp]).astype(np.int32), T


def _vote_side(seed, G, n, ld, base_term):
    """candTerm = t; voterTerm in {t-1: 60%, t: 30%, t+1: 10%}; votedFor in
    {-1: 50%, candId: 25%, other: 25%}; lastLogTerm ~ U[0, t+1]; slot 0 is the
    candidate itself (term t, voted for itself, log not ahead of t)."""
    g = np.arange(G, dtype=np.int64)
    t = base_term.astype(np.int64)
    cand_id = _below(seed, F_CID, g, n)
    vt = np.zeros((n, ld), np.int32)
    vf = np.zeros((n, ld), np.int32)
    ll = np.zeros((n, ld), np.int32)
    for p in range(n):
        idx = p * G + g
        u = _unit(seed, F_VTERM, idx)
        vt[p, :G] = t + np.where(u < 0.6, -1, np.where(u < 0.9, 0, 1))
        u = _unit(seed, F_VFOR, idx)
        other = (cand_id + 1 + _below(seed, F_AUX, idx, max(n - 1, 1))) % max(n, 1)
        vf[p, :G] = np.where(u < 0.5, -1, np.where(u < 0.75, cand_id, other))
        ll[p, :G] = _below(seed, F_VLLT, idx, t + 2)
    vt[0, :G] = t
    vf[0, :G] = cand_id
    ll[0, :G] = np.minimum(ll[0, :G], t)
    ct = np.zeros(ld, np.int32); ct[:G] = t
    ci = np.zeros(ld, np.int32); ci[:G] = cand_id
    return vt, vf, ll, ct[:G].copy(), ci[:G].copy()


def _commit_side(seed, G, n, ld, lens, off, log_term, members=None):
    """match[p] ~ U[0, len); about half the groups have N > commit and about
    70% of those pass the log[N].term == leaderTerm gate."""
    g = np.arange(G, dtype=np.int64)
    match = np.zeros((n, ld), np.int32)
    for p in range(n):
        match[p, :G] = _below(seed, F_MATCH, p * G + g, lens)
    if members is None:
        N = quorum_index(match[:, :G])
    else:
        N = members(match[:, :G])
    adv = _unit(seed, F_COMMIT, g) < 0.5
    below = _below(seed, F_AUX2, g, np.maximum(N, 1))           # in [0, N)
    commit = np.where(adv & (N > 0), below, N + _below(seed, F_AUX2, g, 3))
    Nc = np.clip(N, 0, np.maximum(lens - 1, 0))
    t_at = log_term[off[:-1].astype(np.int64) + Nc].astype(np.int64) if log_term.size else np.zeros(G, np.int64)
    gate = _unit(seed, F_LTERM, g) < 0.7
    leader_term = np.where(gate, t_at, t_at + 1)
    return match, commit.astype(np.int32), leader_term.astype(np.int32)


def gen_fused(G, n=5, seed=DEFAULT_SEED, ld=None, max_len=16, tmax=8):
    """1M x 5 style batch (the headline configuration at G = 1 << 20):
    log lengths ~ U[1, max_len]."""
    ld = G if ld is None else ld
    g = np.arange(G, dtype=np.int64)
    lens = 1 + _below(seed, F_LOGLEN, g, max_len)
    off = _csr(lens)
    log_term, T = _monotone_log(seed, G, lens, off, tmax)
    match, commit, leader_term = _commit_side(seed, G, n, ld, lens, off, log_term)
    vt, vf, ll, ct, ci = _vote_side(seed, G, n, ld, 2 + T)
    return Batch(G=G, n=n, ld=ld, match=match, commit=commit, leader_term=leader_term,
                 log_off=off, log_term=log_term, voter_term=vt, voted_for=vf,
                 last_log_term=ll, cand_term=ct, cand_id=ci, log_T=T)


def gen_commit_small(G=10000, n=3, seed=DEFAULT_SEED, log_len=32):
    """10k x 3 commit-only configuration: fixed 32-entry logs, monotone terms
    1 + floor(i*T/32) with T = leaderTerm in [1, 8], match in [0, 31]."""
    b = gen_fused(G, n, seed, max_len=1)  # vote side unused but kept complete
    g = np.arange(G, dtype=np.int64)
    lens = np.full(G, log_len, np.int64)
    off = _csr(lens)
    log_term, T = _monotone_log(seed, G, lens, off, 8)
    match, commit, _ = _commit_side(seed, G, n, G, lens, off, log_term)
    b.update(match=match, commit=commit, leader_term=T.astype(np.int32), log_off=off, log_term=log_term)
    return b


def _zipf_lens(seed, G, s=1.2, lo=1, hi=4096):
    """Zipf(s) lengths clipped to [lo, hi] by inverse-CDF on a finite support."""
    k = np.arange(lo, hi + 1, dtype=np.float64)
    cdf = np.cumsum(k ** (-s)); cdf /= cdf[-1]
    u = _unit(seed, F_LOGLEN, np.arange(G))
    return (lo + np.searchsorted(cdf, u, side="left")).astype(np.int64)


def masked_quorum_index(match, mask):
    """Quorum index over the member subset given by per-group bit masks."""
    m, G = match.shape
    member = ((mask[None, :].astype(np.int64) >> np.arange(m)[:, None]) & 1).astype(bool)
    big = np.iinfo(np.int64).max
    vals = np.where(member, match.astype(np.int64), big)   # non-members sort last
    s = np.sort(vals, axis=0)
    cnt = member.sum(axis=0)
    return s[np.maximum(cnt - 1, 0) // 2, np.arange(G)]


def gen_mixed(G, seed=DEFAULT_SEED, joint_frac=0.10, bucketed=True, n_lo=3, n_hi=9, logs_from=None):
    """Mixed 3-9 slot clusters, a fraction in joint consensus (old and new
    configurations overlapping in at least one slot, all within 16 slots),
    Zipf(1.2) log lengths clipped to [1, 4096].  bucketed=True orders groups by
    highest member slot so that waves are homogeneous (a host-side layout
    choice; results do not depend on it); bucketed="size" orders them by cluster
    size with the groups in joint consensus last, the order segment_table()
    describes to a segmented launch.  logs_from: a previous gen_mixed batch
    whose log arena (hundreds of MB at 1M groups) is reused unchanged."""
    m = 16
    g = np.arange(G, dtype=np.int64)
    n_old = n_lo + _below(seed, F_NSLOT, g, n_hi - n_lo + 1)
    is_joint = _unit(seed, F_JOINT, g) < joint_frac
    n_new = n_lo + _below(seed, F_NSLOT, G + g, n_hi - n_lo + 1)
    shift = 1 + _below(seed, F_AUX, 7 * G + g, np.maximum(n_old - 1, 1))   # 1..n_old-1: overlap >= 1
    n_new = np.minimum(n_new, m - shift)
    old_mask = (1 << n_old) - 1
    new_mask = np.where(is_joint, ((1 << n_new) - 1) << shift, old_mask)
    if bucketed:
        top = np.floor(np.log2((old_mask | new_mask).astype(np.float64))).astype(np.int64)
        if bucketed == "size":
            top = np.where(old_mask == new_mask, n_old, 32 + top)
        order = np.argsort(top, kind="stable")
        old_mask, new_mask = old_mask[order], new_mask[order]
    old_mask = old_mask.astype(np.uint16); new_mask = new_mask.astype(np.uint16)
    if logs_from is None:
        lens = _zipf_lens(seed, G)
        off = _csr(lens)
        log_term, T = _monotone_log(seed, G, lens, off, 8)
    else:                      # reuse another batch's (large) log arena, redraw everything else
        off, log_term, T = logs_from["log_off"], logs_from["log_term"], logs_from["log_T"]
        lens = np.diff(off.astype(np.int64))

    def members(match):
        return np.minimum(masked_quorum_index(match, old_mask), masked_quorum_index(match, new_mask))

    match, commit, leader_term = _commit_side(seed, G, m, G, lens, off, log_term, members)
    vt, vf, ll, ct, ci = _vote_side(seed, G, m, G, 2 + T)
    return Batch(G=G, n=m, ld=G, match=match, commit=commit, leader_term=leader_term,
                 log_off=off, log_term=log_term, voter_term=vt, voted_for=vf,
                 last_log_term=ll, cand_term=ct, cand_id=ci, old_mask=old_mask, new_mask=new_mask,
                 log_T=T)


def segment_table(old_mask, new_mask, max_segments=16):
    """(first_group, n): the segment table of rq_round_desc for an image whose groups are
    ordered by cluster size.  A 64-group tile whose groups all have old == new == the n
    lowest slots for one n in 1..9 belongs to a fixed-size segment of that n; every other
    tile (joint consensus, two sizes meeting, other member sets) to a segment of kind 0,
    decided through the masks.  Neighbouring tiles of one kind are merged.  Returns None
    when the order of the groups needs more than max_segments segments."""
    om = np.asarray(old_mask).astype(np.int64)
    nm = np.asarray(new_mask).astype(np.int64)
    G = om.size
    if G == 0:
        return None
    n = np.zeros(G, np.int64)
    for k in range(1, 10):
        n[(om == nm) & (om == (1 << k) - 1)] = k
    ntiles = (G + 63) // 64
    pad = np.full(ntiles * 64, -1, np.int64)
    pad[:G] = n
    tiles = pad.reshape(ntiles, 64)
    first = tiles[:, 0]
    same = ((tiles == first[:, None]) | (tiles == -1)).all(axis=1)
    kind = np.where(same, first, 0)
    starts = np.concatenate([[0], 1 + np.flatnonzero(kind[1:] != kind[:-1])])
    if starts.size > max_segments:
        return None
    first_group = [int(t) * 64 for t in starts] + [G]
    return first_group, [int(kind[t]) for t in starts]


def reorder(b, order):
    """The batch with its groups in the order `order` (a permutation of range(G)): every
    per-group and per-slot array permuted, the log arena rebuilt in the new order."""
    G = int(b["G"])
    order = np.asarray(order, dtype=np.int64)
    out = Batch(b)
    out["ld"] = G
    for k, a in b.items():
        if not isinstance(a, np.ndarray) or k in ("log_off", "log_term"):
            continue
        if a.ndim == 2:
            out[k] = np.ascontiguousarray(a[:, :G][:, order])
        elif a.ndim == 1 and a.size >= G:
            out[k] = np.ascontiguousarray(a[:G][order])
    off = b["log_off"].astype(np.int64)
    lens = np.diff(off)[order]
    new_off = _csr(lens)
    total = int(new_off[-1])
    src = np.repeat(off[:-1][order] - new_off[:-1].astype(np.int64), lens) + np.arange(total, dtype=np.int64)
    out["log_off"] = new_off
    out["log_term"] = b["log_term"][src] if total else b["log_term"][:0]
    return out


def split_by_size(b, max_images=16):
    """A mixed batch ordered by cluster size (gen_mixed(bucketed="size")) as separate
    images, one per run of equal kind: [(kind, lo, hi, sub-batch)].  kind n = 1..9: the
    groups lo..hi-1 are all plain clusters of n peers; the sub-batch has n slot planes
    and no member masks (a fixed-size round).  kind 0: everything else, with its masks
    and all the slot planes.  The sub-batches index the ORIGINAL log arena (log_off
    keeps its absolute offsets), so one resident arena serves every image.  Returns
    None when the order of the groups needs more than max_images runs."""
    G = int(b["G"])
    om = np.asarray(b["old_mask"]).astype(np.int64)
    nm = np.asarray(b["new_mask"]).astype(np.int64)
    kind = np.zeros(G, np.int64)
    for k in range(1, 10):
        kind[(om == nm) & (om == (1 << k) - 1)] = k
    starts = np.concatenate([[0], 1 + np.flatnonzero(kind[1:] != kind[:-1]), [G]])
    if starts.size - 1 > max_images:
        return None
    per_slot = ("match", "voter_term", "voted_for", "last_log_term")
    per_group = ("commit", "leader_term", "cand_term", "cand_id", "log_base", "log_T")
    out = []
    for lo, hi in zip(starts[:-1], starts[1:]):
        lo, hi = int(lo), int(hi)
        k = int(kind[lo])
        slots = k if k else int(b["n"])
        sub = Batch(G=hi - lo, n=slots, ld=hi - lo, log_off=b["log_off"][lo:hi + 1], log_term=b["log_term"])
        for f in per_slot:
            sub[f] = np.ascontiguousarray(np.asarray(b[f])[:slots, lo:hi])
        for f in per_group:
            if f in b:
                sub[f] = np.asarray(b[f])[lo:hi]
        if not k:
            sub["old_mask"], sub["new_mask"] = b["old_mask"][lo:hi], b["new_mask"][lo:hi]
        out.append((k, lo, hi, sub))
    return out


def raft_valid(b):
    """The batch with every log term above its leader's term lowered to that term: with
    the generators' non-decreasing logs that is the shape of every log Raft itself
    produces (terms never decrease, none above the leader's term)."""
    out = Batch(b)
    off = b["log_off"].astype(np.int64)
    G = off.size - 1
    lens = np.diff(off)
    total = int(off[-1])
    lt = np.repeat(b["leader_term"][:G], lens)
    terms = b["log_term"].copy()
    terms[:total] = np.minimum(terms[:total], lt.astype(terms.dtype))
    out["log_term"] = terms
    return out


def term_start_form(b):
    """(log_term_start[G], log_len[G]) of a batch: per group the index -- in matchIndex's
    index space, so the log window base is added -- of the first log entry whose term equals
    the leader's term, or one past the last entry if there is none; and the log's length."""
    off = b["log_off"].astype(np.int64)
    G = off.size - 1
    lens = np.diff(off)
    total = int(off[-1])
    lt = np.repeat(b["leader_term"][:G].astype(np.int64), lens)
    pos = np.arange(total, dtype=np.int64) - np.repeat(off[:-1], lens)
    cand = np.where(b["log_term"][:total].astype(np.int64) == lt, pos, np.repeat(lens, lens))
    start = lens.copy()
    nz = lens > 0
    if nz.any():
        start[nz] = np.minimum.reduceat(cand, off[:-1][nz])
    if "log_base" in b:
        start = start + b["log_base"][:G].astype(np.int64)
    return start.astype(b["match"].dtype), lens.astype(np.uint32)


def term_start_monotone(b):
    """term_start_form(raft_valid(b)) for a gen_fused / gen_mixed batch without touching the
    arena (hundreds of MB for the mixed workload): the generators' logs are
    term_at(i) = 1 + floor(i * T / len), so the first entry at or above the leader's
    term t is i = ceil((t - 1) * len / T), or there is none."""
    lens = np.diff(b["log_off"].astype(np.int64))
    G = lens.size
    T = b["log_T"].astype(np.int64)
    t = b["leader_term"][:G].astype(np.int64)
    first = np.where(t <= 1, 0, -((-(t - 1) * lens) // T))
    start = np.minimum(first, lens)
    return start.astype(b["match"].dtype), lens.astype(np.uint32)


def gen_adversarial(G, n, seed=DEFAULT_SEED, ld=None):
    """Edge-heavy batch for parity tests: empty leader logs, N outside the log,
    negative and extreme int32 values, empty voter logs, candidate id -1."""
    ld = G if ld is None else ld
    g = np.arange(G, dtype=np.int64)
    lens = _below(seed, F_LOGLEN, g, 6)                      # 0..5, empty logs included
    off = _csr(lens)
    total = int(off[-1])
    log_term = (_below(seed, F_LOGT, np.arange(total), 4) + 1).astype(np.int32)
    ext = np.array([np.iinfo(np.int32).min, -1, 0, 1, 2, 3, 5, np.iinfo(np.int32).max], np.int64)
    match = np.zeros((n, ld), np.int32)
    for p in range(n):
        idx = p * G + g
        small = _below(seed, F_MATCH, idx, 8) - 1               # -1..6: often >= len
        pick = _unit(seed, F_AUX, idx) < 0.05
        match[p, :G] = np.where(pick, ext[_below(seed, F_AUX2, idx, ext.size)], small)
    commit = (_below(seed, F_COMMIT, g, 7) - 2).astype(np.int32)
    leader_term = (_below(seed, F_LTERM, g, 4) + 1).astype(np.int32)
    vt = np.zeros((n, ld), np.int32); vf = np.zeros((n, ld), np.int32); ll = np.zeros((n, ld), np.int32)
    for p in range(n):
        idx = p * G + g
        vt[p, :G] = np.where(_unit(seed, F_VTERM, idx) < 0.03,
                             ext[_below(seed, F_AUX, 3 * n * G + idx, ext.size)],
                             _below(seed, F_VTERM, n * G + idx, 5))
        vf[p, :G] = _below(seed, F_VFOR, idx, n + 2) - 2       # -2..n-1, -1 is null
        u = _unit(seed, F_VLLT, idx)
        ll[p, :G] = np.where(u < 0.08, LOG_EMPTY, _below(seed, F_VLLT, n * G + idx, 6))
    cand_term = np.where(_unit(seed, F_CTERM, g) < 0.03,
                         ext[_below(seed, F_CTERM, G + g, ext.size)],
                         _below(seed, F_CTERM, 2 * G + g, 5)).astype(np.int32)
    cand_id = (_below(seed, F_CID, g, n + 1) - 1).astype(np.int32)   # -1 collides with null
    return Batch(G=G, n=n, ld=ld, match=match, commit=commit, leader_term=leader_term,
                 log_off=off, log_term=log_term, voter_term=vt, voted_for=vf,
                 last_log_term=ll, cand_term=cand_term, cand_id=cand_id)


def gen_append(G, n, seed=DEFAULT_SEED, ld=None, max_len=12, adversarial=False):
    """AppendEntries consistency batch: per (slot, group) follower logs in one
    CSR arena indexed by slot id s = p*ld + g, a leader term per group, and for
    every follower prevLogIndex / prevLogTerm / number of new entries plus the
    leader's matchIndex / nextIndex.  About 60% of the checks succeed; with
    adversarial=True indices outside the log, empty logs and stale leaders are
    frequent."""
    ld = G if ld is None else ld
    slots = n * ld
    sidx = np.arange(slots, dtype=np.int64)
    g_of = sidx % ld
    valid = g_of < G
    lens = np.where(valid, (0 if adversarial else 1) + _below(seed, F_LOGLEN, sidx, max_len), 0)
    off = _csr(lens)
    total = int(off[-1])
    grp = np.repeat(sidx, lens)
    pos = np.arange(total, dtype=np.int64) - off[:-1].astype(np.int64)[grp]
    T = 1 + _below(seed, F_LOGT, sidx, 6)
    flog = (1 + (pos * T[grp]) // np.maximum(lens[grp], 1)).astype(np.int32)
    lt = (3 + _below(seed, F_LTERM, np.arange(G), 6)).astype(np.int32)
    u = _unit(seed, F_VTERM, sidx)
    lt_s = np.where(valid, lt[np.minimum(g_of, G - 1)], 0).astype(np.int64)
    ft = lt_s + np.where(u < 0.5, -1, np.where(u < 0.9, 0, 1))
    span = np.maximum(lens, 1) + (3 if adversarial else 0)
    pi = _below(seed, F_MATCH, sidx, span) - (1 if adversarial else 0)
    inb = (pi >= 0) & (pi < lens)
    t_at = np.where(inb, flog[np.minimum(off[:-1].astype(np.int64) + np.clip(pi, 0, None), max(total - 1, 0))]
                    if total else 0, 0)
    pt = np.where(_unit(seed, F_AUX, sidx) < 0.75, t_at, t_at + 1)
    nn = _below(seed, F_AUX2, sidx, 4)
    mi = _below(seed, F_COMMIT, sidx, 8)
    ni = 1 + _below(seed, F_CID, sidx, 9)
    shape = (n, ld)
    act = ((1 << n) - 1) & ~(1 << _below(seed, F_CTERM, np.arange(G), n))     # the leader's own slot gets no RPC
    if adversarial:
        act = _below(seed, F_JOINT, np.arange(G), 1 << n)
    return Batch(G=G, n=n, ld=ld, follower_term=ft.astype(np.int32).reshape(shape), leader_term=lt,
                 prev_log_index=pi.astype(np.int32).reshape(shape), prev_log_term=pt.astype(np.int32).reshape(shape),
                 n_new=nn.astype(np.int32).reshape(shape), flog_off=off, flog_term=flog,
                 match_index=mi.astype(np.int32).reshape(shape), next_index=ni.astype(np.int32).reshape(shape),
                 active_mask=act.astype(np.uint16))


def with_follower_commit(b, seed=DEFAULT_SEED):
    """A copy of a gen_append batch with the inputs of the follower's commit rule
    (raft.go:172-176): the RPC's leaderCommit per group, every follower's commitIndex
    and the Index of the last new entry -- prevLogIndex + n_new for most slots, an
    arbitrary small value for a fifth of them (the reference takes whatever Index the
    entry carries, also one below commitIndex)."""
    G, n, ld = int(b["G"]), int(b["n"]), int(b["ld"])
    sidx = np.arange(n * ld, dtype=np.int64)
    dt = b["follower_term"].dtype
    w = Batch(b)
    w["leader_commit"] = _below(seed, F_COMMIT, 3 * n * ld + np.arange(G), 16).astype(dt)
    w["follower_commit"] = _below(seed, F_MATCH, 5 * n * ld + sidx, 12).astype(dt).reshape(n, ld)
    usual = b["prev_log_index"].astype(np.int64).reshape(-1) + b["n_new"].astype(np.int64).reshape(-1)
    odd = _below(seed, F_AUX, 7 * n * ld + sidx, 10)
    w["last_new_index"] = np.where(_unit(seed, F_AUX2, 9 * n * ld + sidx) < 0.2, odd, usual).astype(dt).reshape(n, ld)
    return w


def window_logs(b, keep_committed=0):
    """A copy of batch b whose log arena holds, for every group, only the entries
    above commitIndex - keep_committed (the uncommitted suffix plus a few entries
    a lazy compaction has not dropped yet); adds log_base.  Decisions are
    unchanged: the commit rule only ever reads log[N] with N > commitIndex."""
    G = int(b["G"])
    off = b["log_off"].astype(np.int64)
    lens = np.diff(off)
    base = np.clip(b["commit"][:G].astype(np.int64) + 1 - keep_committed, 0, lens)
    new_lens = lens - base
    new_off = _csr(new_lens)
    grp = np.repeat(np.arange(G, dtype=np.int64), new_lens)
    pos = np.arange(int(new_off[-1]), dtype=np.int64) - new_off[:-1].astype(np.int64)[grp]
    arena = b["log_term"][off[:-1][grp] + base[grp] + pos] if grp.size else np.zeros(0, np.int32)
    w = Batch(b)
    w["log_off"], w["log_term"], w["log_base"] = new_off, arena.astype(np.int32), base.astype(np.int32)
    return w


def window_flogs(b, keep_before=1):
    """A copy of AppendEntries batch b whose follower-log arena holds, for every slot,
    only the entries from prevLogIndex - keep_before on (the suffix around nextIndex:
    all a consistency check can ask for); adds flog_base [n, ld].  Decisions are
    unchanged: the check reads log[prevLogIndex] only, and an index outside the log
    stays outside the window."""
    n, ld = b["prev_log_index"].shape
    off = b["flog_off"].astype(np.int64)
    lens = np.diff(off)
    pi = b["prev_log_index"].astype(np.int64).ravel()
    base = np.clip(pi - keep_before, 0, lens)
    new_lens = lens - base
    new_off = _csr(new_lens)
    slot = np.repeat(np.arange(lens.size, dtype=np.int64), new_lens)
    pos = np.arange(int(new_off[-1]), dtype=np.int64) - new_off[:-1].astype(np.int64)[slot]
    arena = b["flog_term"][off[:-1][slot] + base[slot] + pos] if slot.size else np.zeros(0, b["flog_term"].dtype)
    w = Batch(b)
    w["flog_off"], w["flog_term"] = new_off, np.ascontiguousarray(arena)
    w["flog_base"] = base.astype(b["prev_log_index"].dtype).reshape(n, ld)
    return w


LOG_EMPTY64 = np.iinfo(np.int64).min
_WIDE_FIELDS = ("match", "commit", "leader_term", "log_term", "voter_term", "voted_for",
                "last_log_term", "cand_term", "cand_id", "log_base")


def widen(b, index_shift=0, term_shift=0, id_shift=0, seed=DEFAULT_SEED):
    """The wide-layout (int64) image of a compact batch.  With all shifts 0 it is
    the same values in 64 bits.  Non-zero shifts move the values out of int32
    range without changing any decision:
      index_shift: matchIndex, commitIndex and the log window base of group g
                   move up by index_shift + (a per-group draw below 2^20) -- the
                   arena then is a window starting at that base;
      term_shift:  every term (leader, log, voter, candidate, last-log) moves up
                   by term_shift, order and equality are preserved;
      id_shift:    every id except the null id -1 moves up by id_shift.
    Returns the batch and the per-group index shift (to map results back)."""
    G = int(b["G"])
    w = Batch(b)
    g = np.arange(G, dtype=np.int64)
    dshift = np.zeros(G, np.int64)
    if index_shift:
        dshift[:] = index_shift + _below(seed, F_AUX2, 7 * G + g, 1 << 20)
    for k in _WIDE_FIELDS:
        if k in b:
            w[k] = np.asarray(b[k]).astype(np.int64)
    if "log_base" not in w and index_shift:
        w["log_base"] = np.zeros(G, np.int64)
    if index_shift:
        w["match"][:, :G] += dshift[None, :]
        w["commit"][:G] += dshift
        w["log_base"][:G] += dshift
    empty = np.asarray(b["last_log_term"]) == LOG_EMPTY
    for k in ("leader_term", "log_term", "voter_term", "cand_term", "last_log_term"):
        w[k] = w[k] + term_shift
    w["last_log_term"] = np.where(empty, LOG_EMPTY64, w["last_log_term"])
    for k in ("voted_for", "cand_id"):
        w[k] = np.where(w[k] == -1, -1, w[k] + id_shift)
    return w, dshift


def gen_adversarial64(G, n, seed=DEFAULT_SEED, ld=None):
    """gen_adversarial in the wide layout, with the int32 extremes replaced by the
    int64 extremes and a sprinkling of values beyond +-2^40."""
    b = gen_adversarial(G, n, seed, ld)
    w, _ = widen(b)
    lo32, hi32 = np.iinfo(np.int32).min, np.iinfo(np.int32).max
    lo64, hi64 = np.iinfo(np.int64).min, np.iinfo(np.int64).max
    for k in ("match", "voter_term", "cand_term", "commit", "leader_term"):
        a = w[k]
        a = np.where(a == lo32, lo64, np.where(a == hi32, hi64, a))
        idx = np.arange(a.size, dtype=np.int64).reshape(a.shape)
        u = _unit(seed, F_JOINT, idx + 11 * a.size)
        big = (np.int64(1) << 40) + _below(seed, F_AUX, idx + 13 * a.size, 1000)
        a = np.where(u < 0.02, big, np.where(u < 0.04, -big, a))
        w[k] = a
    return w
