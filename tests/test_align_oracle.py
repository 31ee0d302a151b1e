"""CPU tests of the pairwise-alignment definition (oracle/hite_oracle_nw.c: textbook full-matrix dynamic programme) and of
the twin of the product's banded bit-parallel aligner (oracle/hite_oracle_msa.c) against it."""
import numpy as np
import pytest

import oracle_lib as O

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rnd(rng, n):
    return ACGT[rng.integers(0, 4, n)]


def mutate(rng, s, sub, indel):
    out = []
    for c in s:
        r = rng.random()
        if r < indel / 2:
            continue
        if r < indel:
            out.append(c)
            out.append(ACGT[rng.integers(0, 4)])
            continue
        if rng.random() < sub:
            c = ACGT[(int(np.searchsorted(ACGT, c)) + rng.integers(1, 4)) % 4]
        out.append(c)
    return np.array(out, dtype=np.uint8)


def make_pair(rng, te_len, big_indel=0, flank_jitter=5):
    """two diverged copies of one element (0-15 % substitutions, 1 % indels each) in random flanks; optionally one long
    insertion / deletion in the second copy"""
    te = rnd(rng, te_len)
    c1 = mutate(rng, te, rng.uniform(0, 0.15), 0.01)
    c2 = mutate(rng, te, rng.uniform(0, 0.15), 0.01)
    if big_indel:
        p = int(rng.integers(60, len(c2) - 60))
        if rng.random() < 0.5:
            c2 = np.concatenate([c2[:p], rnd(rng, big_indel), c2[p:]])
        else:
            c2 = np.concatenate([c2[:p], c2[min(len(c2) - 30, p + big_indel):]])
    a = np.concatenate([rnd(rng, 50 + int(rng.integers(-flank_jitter, flank_jitter + 1))), c1, rnd(rng, 50)])
    b = np.concatenate([rnd(rng, 50), c2, rnd(rng, 50 + int(rng.integers(-flank_jitter, flank_jitter + 1)))])
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


# known answers worked out by hand for mismatch 1 / gap 3 (classic textbook pairs transcribed to the DNA alphabet)
KAT = [
    ("ACGT", "ACGT", 0),
    ("ACGT", "AGT", 3),              # one gap
    ("ACGT", "TGCA", 4),             # four mismatches beat any gapped alignment (>= 6)
    ("AAAA", "TTTTTT", 10),          # 4 mismatches + 2 gaps
    ("GATTACA", "GCATGCT", 4),       # Needleman-Wunsch's textbook pair, ungapped: 4 mismatches
    ("ACGTACGTAC", "ACGACGTTAC", 4), # ungapped (4 mismatches) beats delete + insert (6)
    ("A", "C", 1),
    ("ACGT", "ACNT", 1),             # N never matches
    ("ACNT", "ACNT", 1),             # ... not even N
    ("AAAAAAAAAA", "A", 27),
    ("ACGTTTTTTTTACGT", "ACGTACGT", 21),   # a 7-base deletion is cheaper (21) than misaligning the tail
]


@pytest.mark.parametrize("a,b,d", KAT)
def test_nw_known_answers(a, b, d):
    assert O.nw_distance(a, b) == d
    ops, dd = O.nw_pair(a, b)
    assert dd == d and O.ops_cost(a, b, ops) == d
    assert O.nw_distance(b, a) == d


def test_nw_canonical_tiebreak():
    # diagonal preferred, then up (gap in the row), then left (insertion):
    ops, d = O.nw_pair("AC", "A")       # C faces a gap after row position 1
    assert d == 3 and list(ops) == [0, 1 | 0x8000]
    ops, d = O.nw_pair("A", "CA")       # insertion of C before centre position 0
    assert d == 3 and list(ops) == [1]
    ops, d = O.nw_pair("AA", "A")       # co-optimal: canonical traceback takes the diagonal at the END
    assert d == 3 and list(ops) == [0 | 0x8000, 0]


def test_twin_equals_definition_when_certified():
    rng = np.random.default_rng(20250927)
    n_cert = 0
    for it in range(120):
        a, b = make_pair(rng, int(rng.integers(60, 700)))
        exp, d = O.nw_pair(a, b)
        for nw in (4, 8, 16):
            ops, r = O.bp_pair(a, b, nw)
            assert r["status"] in (0, 1)
            assert r["U"] >= d
            if r["status"] == 0:
                assert O.ops_cost(a, b, ops) == r["U"]
            if r["cert"]:
                assert r["U"] == d
                if r["status"] == 0:
                    assert (ops == exp).all()
                    n_cert += 1
    assert n_cert > 150


def test_twin_small_and_ragged():
    rng = np.random.default_rng(5)
    for m, n in [(1, 1), (1, 7), (7, 1), (2, 3), (40, 33), (33, 40), (130, 64), (64, 130), (200, 200)]:
        for it in range(6):
            a, b = rnd(rng, m), rnd(rng, n)
            exp, d = O.nw_pair(a, b)
            ops, info = O.align_pair(a, b, 16)
            if ops is None:
                assert info["status"] == 2 and m - 2 * (n - 1) > 2     # row shorter than half the centre
                continue
            assert O.ops_cost(a, b, ops) == info["U"] >= d
            if info["cert"]:
                assert (ops == exp).all()
            ops4, info4 = O.align_pair(a, b, 0)
            assert ops4 is not None and O.ops_cost(a, b, ops4) == info4["U"] >= d


def test_exact_mode_recovers_long_indels():
    """pairs with one 65-400 bp insertion / deletion: the path leaves the 64-row slice, the wide fall-back takes over; a
    certified result is the optimum, an uncertified one is never better than it"""
    rng = np.random.default_rng(99)
    worse_fast = fell_back = 0
    for it in range(40):
        a, b = make_pair(rng, int(rng.integers(500, 900)), big_indel=int(rng.integers(65, 401)))
        exp, d = O.nw_pair(a, b)
        ops, info = O.align_pair(a, b, 32)
        assert ops is not None
        assert O.ops_cost(a, b, ops) == info["U"]
        fell_back += bool(info["nw"] & 0x100)
        if info["cert"]:
            assert info["U"] == d and (ops == exp).all()
        else:
            assert info["U"] >= d
        opsf, inf = O.align_pair(a, b, 0)
        if opsf is not None:
            assert O.ops_cost(a, b, opsf) == inf["U"] >= d
            worse_fast += inf["U"] > d
            assert not (inf["cert"] and inf["U"] > d)
    assert fell_back >= 20     # indels of this size leave the slice: the reason the wide fall-back exists
    assert worse_fast <= 4


def test_star_msa_rows_are_pairwise_optimal():
    rng = np.random.default_rng(3)
    te = rnd(rng, 300)
    wins = [bytes(np.concatenate([rnd(rng, 50), mutate(rng, te, 0.08, 0.01), rnd(rng, 50)])) for _ in range(9)]
    m, kept = O.star_msa(wins, rows=True)
    assert kept == len(wins)
    centre = m[0]
    assert bytes(centre[centre != ord("-")]) == wins[0]
    for r in range(1, kept):
        row = m[r]
        assert bytes(row[row != ord("-")]) == wins[r]
        both = (centre != ord("-")) | (row != ord("-"))
        gap = both & ((centre == ord("-")) | (row == ord("-")))
        cost = int(((centre != row) & both & ~gap).sum()) + 3 * int(gap.sum())
        assert cost == O.nw_distance(wins[0], wins[r])


def test_uncertified_rows_of_pipeline_windows_equal_the_definition():
    """Measured, not proved (tools/uncertified_rows_vs_optimum.py: 3 246 of 3 246): on windows as the pipeline cuts them -- copies of a
    family with 50 flanking bases, up to 30 % apart -- the rows that NO band certifies still carry the optimal cost and the canonical
    optimal ops of the band-free definition.  Kept as a small regression sample: a change to the band's steering or to the schedule
    that makes uncertified rows worse shows up here."""
    import torch
    from hite_amd import synth

    w = synth.make_workload(genome_bp=4_000_000, n_tir=10, n_ltr=10, cands_per_family=10, seed=77, device=torch.device("cpu"), chrom_bp=2_000_000)
    genome = w["genome"].numpy()
    coff = np.asarray(w["contig_off"], dtype=np.int64)
    contigs = [genome[coff[i]:coff[i + 1]].tobytes() for i in range(len(coff) - 1)]
    n = len(w["cand_off"]) - 1
    pick = np.random.default_rng(3).permutation(n)[:8]
    tab = O.find_copies(contigs, [bytes(w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]]) for c in pick])
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    pairs = unc = 0
    for copies in tab:
        wins = []
        for (ci, s1, e1, minus, _a) in copies[:9]:
            lo, hi = s1 - 1 - 50, e1 + 50
            if lo < 0 or hi > len(contigs[ci]) or hi - lo < 100:
                continue
            s = contigs[ci][lo:hi]
            wins.append(s.translate(comp)[::-1] if minus else s)
        for row in wins[1:]:
            ops, info = O.align_pair(wins[0], row, 8)
            if ops is None:
                continue
            pairs += 1
            if not info["cert"]:
                unc += 1
                exp, d = O.nw_pair(wins[0], row)
                assert info["U"] == d and np.array_equal(ops, exp), (len(wins[0]), len(row), info, d)
    assert pairs >= 20 and unc >= 5, (pairs, unc)


def test_twin_equals_definition_on_padded_rows():
    """rows padded as hite_flank_region_align_clip pads them (the centre's own first / last bases in lower case, or '.'): the banded
    twin against the band-free definition, which reads a lower-case row byte as its base -- a certified pair is the definition's
    alignment, ops and cost; lower-case pads cost nothing on the centre's diagonal, '.' pads 1 each"""
    rng = np.random.default_rng(8026)
    n_cert = n_lower_free = 0
    for it in range(80):
        a, b = (bytes(x).decode() for x in make_pair(rng, int(rng.integers(120, 700))))
        pf, pb = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        cut = b[min(pf, len(b) // 3): len(b) - min(pb, len(b) // 3)]
        low = a[:pf].lower() + cut + (a[len(a) - pb:].lower() if pb else "")
        dot = "." * pf + cut + "." * pb
        exp_low, d_low = O.nw_pair(a, low)
        exp_dot, d_dot = O.nw_pair(a, dot)
        assert O.nw_pair(a, low.upper())[1] == d_low                      # a lower-case byte is its base, nothing else
        assert d_dot >= d_low
        n_lower_free += d_dot - d_low == pf + pb
        for row, exp, d in ((low, exp_low, d_low), (dot, exp_dot, d_dot)):
            ops, info = O.align_pair(a, row, 16)
            assert ops is not None and O.ops_cost(a, row, ops) == info["U"] >= d
            if info["cert"]:
                assert info["U"] == d and (ops == exp).all()
                n_cert += 1
    assert n_cert > 100 and n_lower_free > 40
