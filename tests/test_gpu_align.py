"""GPU parity of the pairwise aligner behind the star alignment (hite_amd/csrc/hite_align.hip), through the C ABI:
   * HIP == twin (oracle/hite_oracle_msa.c) per pair: cost, certificate, status, band, and the alignment byte for byte;
   * a CERTIFIED pair is the textbook global alignment (mismatch 1, gap 3) of oracle/hite_oracle_nw.c (cost and canonical path);
   * an uncertified pair is a valid alignment whose cost bounds the optimum from above."""
import numpy as np
import pytest

import oracle_lib as O
from test_align_oracle import make_pair, mutate, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import hite_amd

    c = hite_amd.Context(0)
    yield c
    c.close()


def pair_matrix(a, b, ops):
    """2 x cols alignment from the ops of one pair (insertion blocks left-justified, as the star layout does)"""
    rows = ([], [])
    nxt = 0
    for p in range(len(a)):
        q, gap = int(ops[p]) & 0x7FFF, int(ops[p]) >> 15
        for x in range(nxt, q):
            rows[0].append(ord("-"))
            rows[1].append(b[x])
        rows[0].append(a[p])
        if gap:
            rows[1].append(ord("-"))
            nxt = q
        else:
            rows[1].append(b[q])
            nxt = q + 1
    for x in range(nxt, len(b)):
        rows[0].append(ord("-"))
        rows[1].append(b[x])
    return np.array(rows, dtype=np.uint8)


def check_pairs(ctx, pairs, cap):
    """pairs: list of (a, b) uint8 arrays; runs them as two-row groups with exact_cap = cap -- three times: through the
    thread-per-pair kernels alone, with EVERY pair in the lane-parallel kernels (hite_align_lanes; 4 / 8 lanes per pair in the
    forward passes, a wavefront per pair in the traceback), and with the two kinds side by side"""
    res = []
    for lanes in (-2, 0, 400):
        ctx.align_lanes(lanes)
        try:
            res.append(check_pairs_once(ctx, pairs, cap))
        finally:
            ctx.align_lanes(-1)
    assert res[0] == res[1] == res[2]
    return res[0]


def check_pairs_once(ctx, pairs, cap):
    ctx.align_config(cap)
    prev = O.set_align_exact(cap)
    try:
        groups = [[bytes(a), bytes(b)] for a, b in pairs]
        got, info = ctx.star_msa(groups, info=True)
        n_cert = n_opt = n_drop = 0
        for (a, b), m, inf in zip(pairs, got, info):
            ops, ti = O.align_pair(a, b, cap)
            U, cert, status, kst, nw = (int(x) for x in inf[1])
            assert (U, cert, status, kst, nw) == (ti["U"], ti["cert"] if ti["status"] != 2 else 0, ti["status"], ti["kstar"], ti["nw"]), (inf[1], ti)
            if ops is None:
                n_drop += 1
                assert m is not None and m.shape[0] == 1 and bytes(m[0]) == bytes(a)
                continue
            exp = pair_matrix(a, b, ops)
            assert m is not None and m.shape == exp.shape and np.array_equal(m, exp)
            gap = (m[0] == ord("-")) | (m[1] == ord("-"))
            cost = 3 * int(gap.sum()) + int(((m[0] != m[1]) & ~gap).sum()) + int(((m[0] == m[1]) & ~np.isin(m[0], list(b"ACGT"))).sum())
            assert cost == U
            d = O.nw_distance(a, b)
            assert U >= d
            n_opt += U == d
            if cert:
                n_cert += 1
                nops, nd = O.nw_pair(a, b)
                assert U == nd and np.array_equal(m, pair_matrix(a, b, nops))
        return n_cert, n_opt, n_drop
    finally:
        O.set_align_exact(prev)
        ctx.align_config(8)


def test_align_families_all_modes(ctx):
    rng = np.random.default_rng(7001)
    pairs = [make_pair(rng, int(rng.integers(60, 900))) for _ in range(160)]
    for cap in (0, 8, 16, 32):
        n_cert, n_opt, n_drop = check_pairs(ctx, pairs, cap)
        assert n_drop == 0
        assert n_opt >= 0.98 * len(pairs)          # these families hardly ever need more than the narrow band
        if cap >= 16:
            assert n_cert >= 0.9 * len(pairs)


def test_align_long_indels_exact(ctx):
    """65-400 bp insertions / deletions: certified == textbook optimum; the fast mode may be worse, never certified wrongly"""
    rng = np.random.default_rng(7002)
    pairs = [make_pair(rng, int(rng.integers(500, 900)), big_indel=int(rng.integers(65, 401))) for _ in range(60)]
    n_cert, n_opt, n_drop = check_pairs(ctx, pairs, 32)
    assert n_drop == 0 and n_cert >= 0.8 * len(pairs) and n_opt >= n_cert
    check_pairs(ctx, pairs, 0)
    check_pairs(ctx, pairs, 16)


def test_align_ragged_and_edge(ctx):
    rng = np.random.default_rng(7003)
    pairs = []
    for m, n in [(1, 1), (1, 7), (7, 1), (2, 3), (16, 16), (17, 15), (40, 33), (33, 40), (130, 64), (64, 130), (200, 95), (95, 200),
                 (300, 120), (1000, 1000), (1500, 1490)]:
        for _ in range(3):
            pairs.append((rnd(rng, m), rnd(rng, n)))
    # identical sequences, all-N rows, N runs
    s = rnd(rng, 333)
    pairs.append((s, s.copy()))
    pairs.append((s, np.full(300, ord("N"), np.uint8)))
    t = s.copy(); t[100:140] = ord("N")
    pairs.append((s, t))
    pairs.append((t, s))
    for cap in (0, 16):
        n_cert, n_opt, n_drop = check_pairs(ctx, pairs, cap)
        assert n_drop >= 3       # (7, 1), (130, 64)-like rows shorter than half the centre are dropped


def test_align_long_pairs_lane_kernels(ctx):
    """windows of 2 000 - 11 000 columns -- what the automatic schedule hands to the lane-parallel kernels: several rounds of
    64 strips per wavefront in the traceback, long runs of up steps, pairs whose path leaves the slice (fall-back), all three
    kernel mixes of check_pairs, with and without wider bands"""
    rng = np.random.default_rng(7006)
    pairs = [make_pair(rng, L) for L in (1000, 1030, 2000, 2100, 3100, 5000, 8000, 11000)]
    pairs += [make_pair(rng, L, big_indel=bi) for L, bi in ((2500, 40), (4000, 70), (6000, 150), (3000, 300))]
    for cap in (8, 0, 16):
        check_pairs(ctx, pairs, cap)


def test_align_many_pairs_in_the_lane_kernels(ctx):
    """several hundred pairs in the lane-parallel class at once: wavefronts of 16 / 8 pairs whose pairs end in different strips
    (the whole-strip fast path of the forward kernels ends where the shortest pair of the wavefront does), lengths 40 - 2 500"""
    rng = np.random.default_rng(7007)
    pairs = [make_pair(rng, int(rng.integers(40, 700))) for _ in range(560)]
    pairs += [make_pair(rng, L, big_indel=bi) for L, bi in ((2500, 0), (1800, 50), (1200, 90), (900, 200))]
    for cap in (8, 0):
        check_pairs(ctx, pairs, cap)


def test_align_group_with_dropped_rows(ctx):
    """a row that cannot be aligned leaves the alignment; the other rows are unaffected (HIP == twin)"""
    rng = np.random.default_rng(7004)
    te = rnd(rng, 400)
    wins = [bytes(np.concatenate([rnd(rng, 50), mutate(rng, te, 0.05, 0.01), rnd(rng, 50)])) for _ in range(7)]
    wins.insert(3, bytes(rnd(rng, 90)))         # far shorter than half the centre
    wins.insert(6, bytes(rnd(rng, 120)))
    got = ctx.star_msa([wins, wins[:3]])
    exp, kept = O.star_msa(wins, rows=True)
    assert kept == len(wins) - 2 and got[0].shape == exp.shape and np.array_equal(got[0], exp)
    assert np.array_equal(got[1], O.star_msa(wins[:3]))
    sp = ctx.star_msa([wins], sparse=True)[0]
    keep = O.sparse_cols(exp).astype(bool)
    assert np.array_equal(sp, exp[:, keep])


def test_align_stats_counters(ctx):
    rng = np.random.default_rng(7005)
    pairs = [make_pair(rng, 300) for _ in range(20)]
    ctx.align_stats(reset=True)
    ctx.align_config(16)
    ctx.star_msa([[bytes(a), bytes(b)] for a, b in pairs])
    st = ctx.align_stats()
    ctx.align_config(8)
    assert st["pairs"] == 20 and st["dropped"] == 0 and st["exact_cap"] == 16      # ONE alignment per logical call (hite_star_msa_once)
    assert st["certified"] >= 18 and st["columns"] == sum(len(b) for _, b in pairs)
